#!/usr/bin/env python3
"""Decision aid (scipy): smoothed prolongation ONLY between levels 0 and 1 (where the plastic bands live), Galerkin operators
with plain bilinear transfers below (their stencils stay 5 x 5), optionally stored in FP32 -- the cheapest hierarchy that
could keep the gain of smoothed aggregation.  python tools/probes/mg_proto_first_level.py gpurun_out/tang128.npz"""
import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.argv = ['x', sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tang128.npz']
src = open('tools/mg_proto.py').read()
src = src[:src.index("m0 = dirichlet_mask(n)")]
exec(src)
m0 = dirichlet_mask(n)
top = np.zeros((n + 1, n + 1, 2)); top[:, n, 1] = 1.
b = -(Kf @ top.ravel()) * m0
rng = np.random.default_rng(0)
b2 = rng.standard_normal(len(b)) * m0


def hier(nsmooth_levels, w, steps=1, fp32=False, kind='galerkin'):
    levels = []
    nx = n
    K = masked(Kf, dirichlet_mask(nx))
    D6 = D0
    while True:
        m = dirichlet_mask(nx)
        levels.append({'nx': nx, 'K': K, 'dinv': 1. / K.diagonal(), 'm': m})
        if nx % 2 or nx <= 2:
            break
        nc = nx // 2
        P = sp.diags(m) @ prolong(nc) @ sp.diags(dirichlet_mask(nc))
        if len(levels) <= nsmooth_levels:
            for _ in range(steps):
                P = P - w * sp.diags(levels[-1]['dinv']) @ (K @ P)
                P = sp.diags(m) @ P
        levels[-1]['P'] = P.tocsr()
        mc = dirichlet_mask(nc)
        Kc = (P.T @ K @ P).tocsr()
        Kc = (Kc + sp.diags(1. - mc)).tocsr()
        if fp32:
            Kc = Kc.astype(np.float32).astype(np.float64)
        K, nx = Kc, nc
    levels[-1]['lu'] = spla.splu(levels[-1]['K'].tocsc())
    return levels


def run(tag, lv):
    t = time.time()
    nnz1 = lv[1]['K'].nnz / lv[1]['K'].shape[0]
    print('%-58s its tension %3d random %3d   nnz/row level 1: %.0f  level 2: %.0f  (%.1fs)'
          % (tag, pcg(lv, b), pcg(lv, b2), nnz1, lv[2]['K'].nnz / lv[2]['K'].shape[0], time.time() - t), flush=True)


run('re-discretised (libplfx)', hierarchy('mean'))
run('Galerkin, bilinear P everywhere', hier(0, 0.))
for w in (0.3, 0.5):
    run('smoothed P (w=%.1f) level 0->1 only, Galerkin below' % w, hier(1, w))
    run('smoothed P (w=%.1f) levels 0->1->2, Galerkin below' % w, hier(2, w))
    run('smoothed P (w=%.1f) on all levels' % w, hier(99, w))
run('smoothed P (w=0.3, 2 steps) level 0->1 only', hier(1, 0.3, steps=2))
run('smoothed P (w=0.5) level 0->1 only, coarse operators in FP32', hier(1, 0.5, fp32=True))
