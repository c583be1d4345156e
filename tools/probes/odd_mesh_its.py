"""Per-solve PCG iteration counts (and final relative residuals) of the config-3 workload on an even and on odd meshes --
which solves of a load step cost the iterations.  python tools/probes/odd_mesh_its.py"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE

def run(nx, ny, steps=14):
    m = FE.Material(name='hill')
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * ny / nx)
    fe.assign([m]); fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.005 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    marks = []
    fe._step_hook = lambda il: marks.append(len(fe.solver_stats)); fe._max_load_steps = steps
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=50)
    print('%d x %d: niter per load step %s' % (nx, ny, list(fe.niter)))
    a = 0
    for il, b in enumerate(marks):
        print('   step %2d: ' % il + '  '.join('%d (%.0e)' % q for q in fe.solver_stats[a:b]))
        a = b
    fe._drop_engine()

import ast
for nx, ny in (ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ((128, 128), (128, 127))):
    run(nx, ny)
