#!/usr/bin/env python3
"""Scipy prototype (decision aid, not product): PCG iteration counts of the V(2,2) preconditioner when some level pairs
coarsen by 4 instead of 2 (half the launch-latency-bound levels).  Coarse operators re-discretised from
the mean of the children's generators (what libplfx does); transfer = bilinear interpolation over the 4 x 4 cell.
    python tools/probes/mg_proto_coarsen4.py gpurun_out/tang128.npz [gpurun_out/tang256.npz]
Also runs the homogeneous elastic field of the same size (the headline workload's operator)."""
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def elem_K(D6):
    m = len(D6)
    D = np.zeros((m, 3, 3))
    D[:, 0, 0], D[:, 0, 1], D[:, 0, 2] = D6[:, 0], D6[:, 1], D6[:, 2]
    D[:, 1, 1], D[:, 1, 2], D[:, 2, 2] = D6[:, 3], D6[:, 4], D6[:, 5]
    D[:, 1, 0], D[:, 2, 0], D[:, 2, 1] = D[:, 0, 1], D[:, 0, 2], D[:, 1, 2]
    K = np.zeros((m, 8, 8))
    g = 1. / np.sqrt(3.)
    xs = np.array([-1., -1., 1., 1.])
    ys = np.array([-1., 1., -1., 1.])
    for xi in (-g, g):
        for eta in (-g, g):
            dNx = xs * (1. + ys * eta) / 4. * 2.
            dNy = ys * (1. + xs * xi) / 4. * 2.
            B = np.zeros((3, 8))
            B[0, 0::2] = dNx
            B[1, 1::2] = dNy
            B[2, 0::2] = dNy
            B[2, 1::2] = dNx
            K += np.einsum('ia,mij,jb->mab', B, D, B) * 0.25
    return K


def assemble(nx, D6):
    ny = nx
    j, k = np.divmod(np.arange(nx * ny), ny)
    n1 = j * (ny + 1) + k
    nodes = np.stack([n1, n1 + 1, n1 + ny + 1, n1 + ny + 2], axis=1)
    dofs = np.stack([2 * nodes, 2 * nodes + 1], axis=2).reshape(-1, 8)
    Ke = elem_K(D6)
    r = np.repeat(dofs, 8, axis=1).ravel()
    c = np.tile(dofs, (1, 8)).ravel()
    nd = 2 * (nx + 1) * (ny + 1)
    return sp.coo_matrix((Ke.ravel(), (r, c)), shape=(nd, nd)).tocsr()


def dirichlet_mask(nx):
    nn = nx + 1
    m = np.ones((nn, nn, 2))
    m[0, :, 0] = 0.
    m[:, 0, 1] = 0.
    m[:, nx, 1] = 0.
    return m.ravel()


def prolong(nc, f):
    """bilinear interpolation from (nc+1)^2 to (f nc+1)^2 nodes"""
    nf = f * nc
    P1 = sp.lil_matrix((nf + 1, nc + 1))
    for i in range(nf + 1):
        q, r = divmod(i, f)
        if r == 0:
            P1[i, q] = 1.
        else:
            P1[i, q] = 1. - r / f
            P1[i, q + 1] = r / f
    P1 = P1.tocsr()
    return sp.kron(sp.kron(P1, P1), sp.identity(2)).tocsr()


def masked(K, m):
    M = sp.diags(m)
    return (M @ K @ M + sp.diags(1. - m)).tocsr()


def hierarchy(n, D0, factors, galerkin=False):
    levels = []
    nx, D6 = n, D0
    K = masked(assemble(nx, D6), dirichlet_mask(nx))
    for f in list(factors) + [0]:
        m = dirichlet_mask(nx)
        levels.append({'nx': nx, 'K': K, 'dinv': 1. / K.diagonal(), 'm': m})
        if f == 0 or nx % f or nx // f < 2:
            break
        nc = nx // f
        P = sp.diags(m) @ prolong(nc, f) @ sp.diags(dirichlet_mask(nc))
        levels[-1]['P'] = P.tocsr()
        D6 = D6.reshape(nc, f, nc, f, 6).mean(axis=(1, 3)).reshape(-1, 6)
        if galerkin:
            mc = dirichlet_mask(nc)
            K = ((P.T @ K @ P) + sp.diags(1. - mc)).tocsr()
        else:
            K = masked(assemble(nc, D6), dirichlet_mask(nc))
        nx = nc
    levels[-1]['lu'] = spla.splu(levels[-1]['K'].tocsc())
    levels[-1].pop('P', None)
    return levels


def vcycle(levels, l, b, om, nus):
    L = levels[l]
    if 'lu' in L:
        return L['lu'].solve(b)
    nu = nus[min(l, len(nus) - 1)]
    x = np.zeros_like(b)
    for _ in range(nu):
        x += om * L['dinv'] * (b - L['K'] @ x)
    r = b - L['K'] @ x
    x += L['P'] @ vcycle(levels, l + 1, L['P'].T @ r, om, nus)
    for _ in range(nu):
        x += om * L['dinv'] * (b - L['K'] @ x)
    return x


def pcg(levels, b, om=0.65, nus=(2,), rtol=1e-10, maxit=400):
    K = levels[0]['K']
    x = np.zeros_like(b)
    r = b.copy()
    zv = vcycle(levels, 0, r, om, nus)
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
        zv = vcycle(levels, 0, r, om, nus)
        rz2 = r @ zv
        p = zv + (rz2 / rz) * p
        rz = rz2
    return maxit


def run(name, n, D0):
    m0 = dirichlet_mask(n)
    Kf = assemble(n, D0)
    top = np.zeros((n + 1, n + 1, 2))
    top[:, n, 1] = 1.
    b = -(Kf @ top.ravel()) * m0
    rng = np.random.default_rng(0)
    b2 = rng.standard_normal(len(b)) * m0
    print('== %s, n = %d' % (name, n))
    scheds = {
        'all x2 (built)': [2] * 12,
        'x2 x2 then x4...': [2, 2] + [4] * 6,
        'x2 then x4...': [2] + [4] * 6,
        'all x4': [4] * 6,
        'x2 x4 x2 x4': [2, 4, 2, 4, 2, 4],
    }
    for sname, fac in scheds.items():
        for nus in ((2,), (2, 3), (2, 4)):
            if sname == 'all x2 (built)' and nus != (2,):
                continue
            lv = hierarchy(n, D0, fac)
            sizes = [L['nx'] for L in lv]
            print('  %-18s nu=%-7s levels %-28s its tension %3d random %3d' % (sname, nus, sizes, pcg(lv, b, nus=nus), pcg(lv, b2, nus=nus)))


for f in sys.argv[1:]:
    z = np.load(f)
    n = int(z['n'])
    run('dumped tangent field ' + f, n, z['D'])
    E, nu = 200e3, 0.3
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    Dh = np.tile(np.array([lam + 2 * mu, lam, 0., lam + 2 * mu, 0., mu]), (n * n, 1))
    run('homogeneous elastic', n, Dh)
