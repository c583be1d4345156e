#!/usr/bin/env python3
"""Decision aid (scipy): fine-level smoother = column-block Gauss-Seidel (what a marching kernel could do for free: use the
NEW values of the column it has just left) instead of damped Jacobi; forward sweeps before, backward sweeps after the coarse
correction (symmetric).  Block width W: GS inside blocks of W columns, Jacobi across blocks.
python tools/probes/mg_proto_colgs.py gpurun_out/tang128.npz"""
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
lv = hierarchy('mean')
K0 = lv[0]['K'].tocsr()
nn = n + 1
cols = [np.arange(j * nn * 2, (j + 1) * nn * 2) for j in range(nn)]       # dofs of node column j
Krows = [K0[c] for c in cols]
dinv0 = lv[0]['dinv']


def gs_sweep(x, bb, om, W, backward):
    """column GS inside blocks of W columns (block b = columns [bW, bW+W)), Jacobi across blocks"""
    xo = x.copy()          # values seen across block boundaries
    xn = x.copy()
    order = range(nn - 1, -1, -1) if backward else range(nn)
    for j in order:
        blk = j // W
        lo, hi = blk * W, min(nn, blk * W + W)
        xs = xo.copy()
        sl = slice(lo * nn * 2, hi * nn * 2)
        xs[sl] = xn[sl]    # inside the block: the newest values
        r = bb[cols[j]] - Krows[j] @ xs
        xn[cols[j]] = xs[cols[j]] + om * dinv0[cols[j]] * r
    return xn


def make_vcycle(om0, W, nu=2):
    def vc(levels, l, bb, om=0.65, nu_=2):
        L = levels[l]
        if 'lu' in L:
            return L['lu'].solve(bb)
        if l == 0 and W > 0:
            x = np.zeros_like(bb)
            for _ in range(nu):
                x = gs_sweep(x, bb, om0, W, False)
            r = bb - L['K'] @ x
            x += L['P'] @ vc(levels, 1, L['P'].T @ r)
            for _ in range(nu):
                x = gs_sweep(x, bb, om0, W, True)
            return x
        x = np.zeros_like(bb)
        for _ in range(nu):
            x += om * L['dinv'] * (bb - L['K'] @ x)
        r = bb - L['K'] @ x
        x += L['P'] @ vc(levels, l + 1, L['P'].T @ r)
        for _ in range(nu):
            x += om * L['dinv'] * (bb - L['K'] @ x)
        return x
    return vc


print('Jacobi (built): its tension %d random %d' % (pcg(lv, b), pcg(lv, b2)), flush=True)
for W in (8, 129):
    for om in (0.65, 0.8, 1.0):
        vcycle = make_vcycle(om, W)
        t = time.time()
        try:
            print('column GS, block width %3d, omega %.2f: its tension %d random %d (%.0fs)' % (W, om, pcg(lv, b), pcg(lv, b2), time.time() - t), flush=True)
        except Exception as e:
            print('W', W, 'om', om, 'failed', e)
