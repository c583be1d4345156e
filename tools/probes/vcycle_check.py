"""What one V-cycle does on an even and on odd meshes (plfx_precond_apply): symmetry <a, B b> = <b, B a>, and its answer to
r = K e for the linear field e of the homogeneous workload (error of the warm start), row by row.
python tools/probes/vcycle_check.py "((128,128),(128,127))" """
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE

def run(nx, ny, steps=9):
    m = FE.Material(name='hill')
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * ny / nx)
    fe.assign([m]); fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.005 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    fe._max_load_steps = steps
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=50)
    eng = fe._engine
    nd = fe.Ndof
    free = np.zeros(nd); free[np.asarray(fe.free_dofs())] = 1.
    rng = np.random.default_rng(1)
    a, b = free * rng.standard_normal(nd), free * rng.standard_normal(nd)
    Ba, Bb = eng.precond_apply(a), eng.precond_apply(b)
    print('%d x %d  (%d levels): symmetry <a,Bb> %.12e  <b,Ba> %.12e  rel diff %.2e' % (nx, ny, eng.precond_info()[1], a @ Bb, b @ Ba, abs(a @ Bb - b @ Ba) / abs(a @ Bb)))
    # the last increment du is the linear field of the homogeneous solution (free part)
    e = free * np.asarray(fe.du if hasattr(fe, 'du') else fe.u)
    r = free * eng.matvec(e)
    z = eng.precond_apply(r)
    d = (z - e).reshape(nx + 1, ny + 1, 2)
    ee = e.reshape(nx + 1, ny + 1, 2)
    print('   |B K e - e| / |e| = %.3e;  by row k (uy, max over columns, relative to max |e_y|), last 12 rows:' % (np.linalg.norm(z - e) / np.linalg.norm(e)))
    sc = np.max(np.abs(ee[:, :, 1]))
    print('   ' + ' '.join('%.1e' % v for v in np.max(np.abs(d[:, :, 1]), axis=0)[-12:] / sc))
    print('   first 6 rows: ' + ' '.join('%.1e' % v for v in np.max(np.abs(d[:, :, 1]), axis=0)[:6] / sc))
    print('   by column j (ux), last 6: ' + ' '.join('%.1e' % v for v in np.max(np.abs(d[:, :, 0]), axis=1)[-6:] / max(np.max(np.abs(ee[:, :, 0])), 1e-300)))
    fe._drop_engine()

import ast
for nx, ny in (ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ((128, 128), (128, 127), (127, 128))):
    run(nx, ny)
