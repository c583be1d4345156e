#!/usr/bin/env python3
"""Scipy prototype (decision aid, not product): V(2,2)-PCG iteration counts with a nodal 2 x 2 block-Jacobi smoother instead of
the point-Jacobi smoother that is built (plastic tangents are strongly anisotropic: the x / y coupling of a node is of the size
of its diagonal), and with a damped-Jacobi weight sweep for both.
    python tools/probes/mg_proto_blocksmoother.py gpurun_out/tang128.npz"""
import sys

import numpy as np
import scipy.sparse as sp

argv, sys.argv = sys.argv[1:], sys.argv[:1]
from mg_proto_coarsen4 import hierarchy, dirichlet_mask, assemble  # noqa: E402


def block_dinv(K, m):
    """inverse of the 2 x 2 nodal diagonal blocks (masked DOFs: identity rows), as a sparse block-diagonal matrix"""
    nd = K.shape[0]
    d = K.diagonal()
    i = np.arange(0, nd, 2)
    off = np.asarray(K[i, i + 1]).ravel()
    a, b, c = d[0::2], off, d[1::2]
    free = (m[0::2] != 0) & (m[1::2] != 0)
    b = np.where(free, b, 0.)
    det = a * c - b * b
    ia, ib, ic = c / det, -b / det, a / det
    rows = np.concatenate([i, i, i + 1, i + 1])
    cols = np.concatenate([i, i + 1, i, i + 1])
    vals = np.concatenate([ia, ib, ib, ic])
    return sp.coo_matrix((vals, (rows, cols)), shape=(nd, nd)).tocsr()


def vcycle(levels, l, b, om, nu, blk):
    L = levels[l]
    if 'lu' in L:
        return L['lu'].solve(b)
    S = L['bdinv'] if blk else sp.diags(L['dinv'])
    x = np.zeros_like(b)
    for _ in range(nu):
        x += om * (S @ (b - L['K'] @ x))
    r = b - L['K'] @ x
    x += L['P'] @ vcycle(levels, l + 1, L['P'].T @ r, om, nu, blk)
    for _ in range(nu):
        x += om * (S @ (b - L['K'] @ x))
    return x


def pcg(levels, b, om, nu, blk, rtol=1e-10, maxit=400):
    K = levels[0]['K']
    x = np.zeros_like(b)
    r = b.copy()
    zv = vcycle(levels, 0, r, om, nu, blk)
    p = zv.copy()
    rz = r @ zv
    bn = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = K @ p
        a = rz / (p @ q)
        x += a * p
        r -= a * q
        if np.linalg.norm(r) <= rtol * bn:
            return it
        zv = vcycle(levels, 0, r, om, nu, blk)
        rz2 = r @ zv
        p = zv + (rz2 / rz) * p
        rz = rz2
    return maxit


for f in argv:
    z = np.load(f)
    n = int(z['n'])
    D0 = z['D']
    m0 = dirichlet_mask(n)
    Kf = assemble(n, D0)
    top = np.zeros((n + 1, n + 1, 2))
    top[:, n, 1] = 1.
    b = -(Kf @ top.ravel()) * m0
    b2 = np.random.default_rng(0).standard_normal(len(b)) * m0
    lv = hierarchy(n, D0, [2] * 12)
    for L in lv:
        if 'lu' not in L:
            L['bdinv'] = block_dinv(L['K'], L['m'])
    print('==', f, 'n =', n)
    for blk in (False, True):
        for om in (0.5, 0.65, 0.8, 0.9):
            print('  %-22s omega %.2f  its tension %3d random %3d' % ('2x2 block Jacobi' if blk else 'point Jacobi (built)', om, pcg(lv, b, om, 2, blk), pcg(lv, b2, om, 2, blk)), flush=True)
