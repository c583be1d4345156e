"""Multigrid on meshes with an odd number of elements (round 5: ceil-halving hierarchy with ghost elements): the config-3
workload on 1000 x 999, 999 x 999, 1023 x 1025 and 1000 x 1000 elements against 1024 x 1024 -- preconditioner in use, levels,
ms per load step, PCG iterations.  python tools/probes/odd_mesh_probe.py"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE

def run(nx, ny, steps=14):
    m = FE.Material(name='hill')
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * ny / nx)
    fe.assign([m]); fe.bcleft(0.); fe.bcbot(0.)
    if os.environ.get('PULL_X'):   # tension in x instead: the Dirichlet edge is the right one
        fe.bctop(0., 'force'); fe.bcright(0.005 * fe.lenx, 'disp')
    else:
        fe.bcright(0., 'force'); fe.bctop(0.005 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    eng = fe._ensure_engine()
    if os.environ.get('MG_OMEGA'):
        eng.set_precond(1, float(os.environ['MG_OMEGA']), int(os.environ.get('MG_NU', '2')))
    marks = {}
    def hook(il):
        if il == 8: eng.sync(); marks['t0'] = time.perf_counter(); marks['s0'] = len(fe.solver_stats)
        if il == steps: eng.sync(); marks['t1'] = time.perf_counter(); marks['s1'] = len(fe.solver_stats)
    fe._step_hook = hook; fe._max_load_steps = steps
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=50)
    its = [q[0] for q in fe.solver_stats[marks['s0']:marks['s1']]]
    pi = eng.precond_info()
    print('%5d x %5d: preconditioner %s (%d levels)  %.3f ms per load step (steps 8..%d)  %d solves, %d PCG iterations (max %d)  sgl_yy %.6f'
          % (nx, ny, 'multigrid' if pi[0] == 1 else 'Jacobi', pi[1], 1e3 * (marks['t1'] - marks['t0']) / (steps - 8), steps, len(its), sum(its), max(its), fe.sgl[-1][1]))
    fe._drop_engine()

import ast
for nx, ny in (ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ((1024, 1024), (1000, 1000), (1000, 999), (999, 999), (1023, 1025), (513, 511), (251, 127))):
    run(nx, ny)
