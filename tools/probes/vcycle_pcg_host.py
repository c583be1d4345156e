"""PCG driven from the host with the library's operator (plfx_matvec) and V-cycle (plfx_precond_apply) on the error of a warm
start of the homogeneous workload (a linear field): residual history and WHERE the error stays, even vs odd meshes.
python tools/probes/vcycle_pcg_host.py "((128,128),(128,127))" """
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
    xs = free * np.asarray(fe.u)            # a linear field (free part)
    if os.environ.get('FIELD', 'x') == 'x':  # what a tangent update of the homogeneous workload changes: u_x = beta X, u_y = 0
        g = np.zeros((nx + 1, ny + 1, 2)); g[:, :, 0] = np.arange(nx + 1)[:, None] / nx
        xs = free * g.ravel()
    K = lambda v: free * eng.matvec(free * v)
    B = lambda v: free * eng.precond_apply(free * v)
    b = K(xs)
    x = np.zeros(nd); r = b.copy(); z = B(r); p = z.copy(); rz = r @ z
    nb = np.linalg.norm(b)
    print('%d x %d: ' % (nx, ny), end='')
    hist = []
    for it in range(1, 61):
        q = K(p); al = rz / (p @ q); x += al * p; r -= al * q
        hist.append(np.linalg.norm(r) / nb)
        if it in (2, 5, 10, 20):
            e = (x - xs).reshape(nx + 1, ny + 1, 2)
            sc = np.max(np.abs(xs))
            c_ = 0 if os.environ.get('FIELD', 'x') == 'x' else 1
            rows = np.max(np.abs(e[:, :, c_]), axis=0) / sc
            cols = np.max(np.abs(e[:, :, c_]), axis=1) / sc
            print('\n   it %2d relres %.1e  |e_y| by row (8 samples bottom..top): %s ; by column (left..right): %s'
                  % (it, hist[-1], ' '.join('%.0e' % v for v in rows[np.linspace(0, ny, 8).astype(int)]),
                     ' '.join('%.0e' % v for v in cols[np.linspace(0, nx, 8).astype(int)])), end='')
        if hist[-1] < 1e-10: break
        z = B(r); rzn = r @ z; p = z + rzn / rz * p; rz = rzn
    print('\n   iterations to 1e-10: %d;  relres: %s' % (it, ' '.join('%.0e' % v for v in hist[:24])))
    fe._drop_engine()

import ast
for nx, ny in (ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ((128, 128), (128, 127))):
    run(nx, ny)
