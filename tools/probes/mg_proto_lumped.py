#!/usr/bin/env python3
"""Decision aid (scipy): smoothed prolongation P = (I - w D^-1 K) P0 applied ON THE FLY (prolong + one operator pass, no
stored P) with Galerkin coarse operators LUMPED back to the 3 x 3 node stencil (9 blocks of 2 x 2: the block-ELL form the
library already has), on the first nlev levels.  Does the lumping keep the gain of the 5 x 5 / 7 x 7 Galerkin stencils?
python tools/probes/mg_proto_lumped.py gpurun_out/tang128.npz"""
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


def lump(Kc, nc):
    """collapse every block entry (I, J) whose node offset exceeds 1 onto the in-stencil neighbour in the same direction"""
    nn = nc + 1
    K = Kc.tocoo()
    ni, ci = K.row // 2, K.row % 2
    nj, cj = K.col // 2, K.col % 2
    ix, iy = ni // nn, ni % nn
    jx, jy = nj // nn, nj % nn
    dx = np.clip(jx - ix, -1, 1)
    dy = np.clip(jy - iy, -1, 1)
    nj2 = (ix + dx) * nn + (iy + dy)
    out = sp.coo_matrix((K.data, (K.row, nj2 * 2 + cj)), shape=K.shape).tocsr()
    return (0.5 * (out + out.T)).tocsr()     # keep it symmetric


def hier(nsm, w, lumped):
    levels = []
    nx = n
    K = masked(Kf, dirichlet_mask(nx))
    while True:
        m = dirichlet_mask(nx)
        levels.append({'nx': nx, 'K': K, 'dinv': 1. / K.diagonal(), 'm': m})
        if nx % 2 or nx <= 2:
            break
        nc = nx // 2
        P = sp.diags(m) @ prolong(nc) @ sp.diags(dirichlet_mask(nc))
        if len(levels) <= nsm:
            P = sp.diags(m) @ (P - w * sp.diags(levels[-1]['dinv']) @ (K @ P))
        levels[-1]['P'] = P.tocsr()
        mc = dirichlet_mask(nc)
        Kc = (P.T @ K @ P).tocsr()
        if lumped:
            Kc = lump(Kc, nc)
            Kc = masked(Kc, mc)
        else:
            Kc = (Kc + sp.diags(1. - mc)).tocsr()
        K, nx = Kc, nc
    levels[-1]['lu'] = spla.splu(levels[-1]['K'].tocsc())
    return levels


def run(tag, lv):
    t = time.time()
    try:
        a, c = pcg(lv, b), pcg(lv, b2)
    except Exception as e:  # noqa: BLE001
        a = c = -1
    print('%-64s its tension %3d random %3d   nnz/row level 1: %.0f level 2: %.0f  (%.1fs)'
          % (tag, a, c, lv[1]['K'].nnz / lv[1]['K'].shape[0], lv[2]['K'].nnz / lv[2]['K'].shape[0], time.time() - t), flush=True)


run('re-discretised (libplfx)', hierarchy('mean'))
for nsm in (1, 2, 3, 99):
    for w in (0.5,):
        run('smoothed P (w=%.1f) on %2d levels, Galerkin (wide stencils)' % (w, nsm), hier(nsm, w, False))
        run('smoothed P (w=%.1f) on %2d levels, Galerkin LUMPED to 3x3' % (w, nsm), hier(nsm, w, True))
run('bilinear P, Galerkin lumped (sanity)', hier(0, 0., True))
