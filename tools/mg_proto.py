#!/usr/bin/env python3
"""CPU prototype (scipy) of the multigrid preconditioner on a dumped tangent field (tools/dump_tangent.py): PCG
iteration counts with coarse operators (a) re-discretised from arithmetic-mean generators (what libplfx does) and
(b) Galerkin P^T K P.  Decision aid only - not part of the product or of the oracle.
`python tools/mg_proto.py gpurun_out/tang128.npz`"""
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

z = np.load(sys.argv[1])
n = int(z['n'])
Kf = sp.coo_matrix((z['val'], (z['row'], z['col']))).tocsr()
D0 = z['D']  # (Nel, 6): D11 D12 D16 D22 D26 D66 (plane strain rows 0, 1, 5 of the tangent)
print('n =', n, 'GPU PCG its of the last solves:', z['its'][-6:].tolist())


def elem_K(D6):
    """8x8 stiffness matrices of unit-thickness square Q4 elements from (m,6) plane-strain tangents; node order
    (j,k),(j,k+1),(j+1,k),(j+1,k+1), dofs (ux,uy) per node"""
    m = len(D6)
    D = np.zeros((m, 3, 3))
    D[:, 0, 0], D[:, 0, 1], D[:, 0, 2] = D6[:, 0], D6[:, 1], D6[:, 2]
    D[:, 1, 1], D[:, 1, 2], D[:, 2, 2] = D6[:, 3], D6[:, 4], D6[:, 5]
    D[:, 1, 0], D[:, 2, 0], D[:, 2, 1] = D[:, 0, 1], D[:, 0, 2], D[:, 1, 2]
    K = np.zeros((m, 8, 8))
    g = 1. / np.sqrt(3.)
    xs = np.array([-1., -1., 1., 1.])  # xi of the 4 nodes (x index j slow)
    ys = np.array([-1., 1., -1., 1.])
    for xi in (-g, g):
        for eta in (-g, g):
            dNx = xs * (1. + ys * eta) / 4. * 2.  # d/dx on a unit square: * 2/h with h = 1
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
    m[0, :, 0] = 0.   # left: ux
    m[:, 0, 1] = 0.   # bottom: uy
    m[:, nx, 1] = 0.  # top: uy
    return m.ravel()


def prolong(nc):
    """bilinear interpolation from (nc+1)^2 to (2nc+1)^2 nodes, both components"""
    nf = 2 * nc
    P1 = sp.lil_matrix((nf + 1, nc + 1))
    for i in range(nf + 1):
        if i % 2 == 0:
            P1[i, i // 2] = 1.
        else:
            P1[i, i // 2] = 0.5
            P1[i, i // 2 + 1] = 0.5
    P1 = P1.tocsr()
    return sp.kron(sp.kron(P1, P1), sp.identity(2)).tocsr()


A = assemble(n, D0)
d = abs(A - Kf)
print('check: assembled-from-D vs dumped K: max |diff| = %.3e (max |K| = %.3e)' % (d.max(), abs(Kf).max()))


def masked(K, m):
    M = sp.diags(m)
    return (M @ K @ M + sp.diags(1. - m)).tocsr()


def hierarchy(kind):
    levels = []
    nx, D6 = n, D0
    K = masked(Kf, dirichlet_mask(nx))
    while True:
        m = dirichlet_mask(nx)
        levels.append({'nx': nx, 'K': K, 'dinv': 1. / K.diagonal(), 'm': m})
        if nx % 2 or nx <= 2:
            break
        nc = nx // 2
        P = sp.diags(m) @ prolong(nc) @ sp.diags(dirichlet_mask(nc))
        levels[-1]['P'] = P.tocsr()
        if kind == 'galerkin':
            Kc = (P.T @ K @ P).tocsr()
            mc = dirichlet_mask(nc)
            Kc = (Kc + sp.diags(1. - mc)).tocsr()
        else:
            D6 = D6.reshape(nc, 2, nc, 2, 6).mean(axis=(1, 3)).reshape(-1, 6)
            Kc = masked(assemble(nc, D6), dirichlet_mask(nc))
        K, nx = Kc, nc
    levels[-1]['lu'] = spla.splu(levels[-1]['K'].tocsc())
    return levels


def vcycle(levels, l, b, om=0.65, nu=2):
    L = levels[l]
    if 'lu' in L:
        return L['lu'].solve(b)
    x = np.zeros_like(b)
    for _ in range(nu):
        x += om * L['dinv'] * (b - L['K'] @ x)
    r = b - L['K'] @ x
    x += L['P'] @ vcycle(levels, l + 1, L['P'].T @ r, om, nu)
    for _ in range(nu):
        x += om * L['dinv'] * (b - L['K'] @ x)
    return x


def pcg(levels, b, rtol=1e-10, maxit=500):
    K = levels[0]['K']
    x = np.zeros_like(b)
    r = b.copy()
    zv = vcycle(levels, 0, r)
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
        zv = vcycle(levels, 0, r)
        rz2 = r @ zv
        p = zv + (rz2 / rz) * p
        rz = rz2
    return maxit


m0 = dirichlet_mask(n)
top = np.zeros((n + 1, n + 1, 2))
top[:, n, 1] = 1.
b = -(Kf @ top.ravel()) * m0
rng = np.random.default_rng(0)
b2 = rng.standard_normal(len(b)) * m0
for kind in ('mean', 'galerkin'):
    lv = hierarchy(kind)
    print('%-9s levels %d: PCG its (tension rhs, cold start) %d, (random rhs) %d'
          % (kind, len(lv), pcg(lv, b), pcg(lv, b2)))


# ---- smoother variants on the re-discretised hierarchy
def block_dinv(K):
    """inverse of the 2x2 nodal diagonal blocks as a sparse block-diagonal matrix"""
    nd = K.shape[0]
    a = K.diagonal()[0::2]
    d_ = K.diagonal()[1::2]
    Kc = K.tocsr()
    bb = np.asarray(Kc[np.arange(0, nd, 2), np.arange(1, nd, 2)]).ravel()
    det = a * d_ - bb * bb
    ia, id_, ib = d_ / det, a / det, -bb / det
    r = np.concatenate([np.arange(0, nd, 2), np.arange(1, nd, 2), np.arange(0, nd, 2), np.arange(1, nd, 2)])
    c = np.concatenate([np.arange(0, nd, 2), np.arange(1, nd, 2), np.arange(1, nd, 2), np.arange(0, nd, 2)])
    return sp.coo_matrix((np.concatenate([ia, id_, ib, ib]), (r, c)), shape=(nd, nd)).tocsr()


def vcycle_b(levels, l, b, om, nu):
    L = levels[l]
    if 'lu' in L:
        return L['lu'].solve(b)
    x = np.zeros_like(b)
    for _ in range(nu):
        x += om * (L['Binv'] @ (b - L['K'] @ x))
    r = b - L['K'] @ x
    x += L['P'] @ vcycle_b(levels, l + 1, L['P'].T @ r, om, nu)
    for _ in range(nu):
        x += om * (L['Binv'] @ (b - L['K'] @ x))
    return x


lv = hierarchy('mean')
for L in lv:
    L['Binv'] = block_dinv(L['K'])
_v = vcycle
for om in (0.65, 0.8, 0.9, 1.0):
    try:
        vcycle = lambda levels, l, r, om=om: vcycle_b(levels, l, r, om, 2)  # noqa: E731
        print('block-Jacobi 2x2, omega %.2f: PCG its %d / %d' % (om, pcg(lv, b), pcg(lv, b2)))
    except Exception as e:  # noqa: BLE001
        print('omega', om, 'failed', e)
vcycle = _v
