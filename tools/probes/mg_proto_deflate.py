#!/usr/bin/env python3
"""Decision aid (scipy, CPU): does DEFLATION with Ritz vectors harvested from a previous solve cut the PCG iteration count
of the multigrid-preconditioned solve on a dumped elastic-plastic tangent field?  
python tools/probes/mg_proto_deflate.py gpurun_out/tang128.npz"""
import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.linalg as sl
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
K = lv[0]['K']


def pcg_store(b, rtol=1e-10, maxit=500, W=None, x0=None):
    """PCG (optionally deflated with W); returns its, x, stored Z, R"""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    if W is not None:
        KW = K @ W
        E = W.T @ KW
        Ei = np.linalg.inv(E)
        r = b - K @ x
        x = x + W @ (Ei @ (W.T @ r))
    r = b - K @ x
    Z, R = [], []
    z = vcycle(lv, 0, r)
    Z.append(z.copy()); R.append(r.copy())
    p = z - (W @ (Ei @ (KW.T @ z)) if W is not None else 0.)
    rz = r @ z
    bn = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = K @ p
        a = rz / (p @ q)
        x += a * p
        r -= a * q
        if np.linalg.norm(r) <= rtol * bn:
            return it, x, np.array(Z).T, np.array(R).T
        z = vcycle(lv, 0, r)
        Z.append(z.copy()); R.append(r.copy())
        rz2 = r @ z
        p = z - (W @ (Ei @ (KW.T @ z)) if W is not None else 0.) + (rz2 / rz) * p
        rz = rz2
    return maxit, x, np.array(Z).T, np.array(R).T


def ritz(Z, R, k, Wold=None):
    """k smallest Ritz pairs of K v = theta M v on span(Z); M z_j = r_j (M = the inverse of the V-cycle), so Z^T M Z = Z^T R"""
    sc = 1. / np.sqrt(np.abs(np.sum(Z * R, axis=0)))
    Zs, Rs = Z * sc, R * sc
    G = Zs.T @ (K @ Zs)
    H = Zs.T @ Rs
    H = 0.5 * (H + H.T)
    G = 0.5 * (G + G.T)
    # H is (numerically) the identity for exact Lanczos; regularise by dropping its null space
    w, U = np.linalg.eigh(H)
    keep = w > 1e-8 * w.max()
    T = U[:, keep] / np.sqrt(w[keep])
    th, Y = np.linalg.eigh(T.T @ G @ T)
    return th[:k], Zs @ (T @ Y[:, :k])


t = time.time()
it0, x, Z, R = pcg_store(b)
print('plain MG-PCG: tension rhs %d its (%.1fs)' % (it0, time.time() - t), flush=True)
it0b, xb, Zb, Rb = pcg_store(b2)
print('plain MG-PCG: random rhs %d its' % it0b, flush=True)
for k in (4, 8, 16, 32, 64):
    th, W = ritz(Z, R, k)
    it1, _, _, _ = pcg_store(b2, W=W)
    it2, _, _, _ = pcg_store(b * (1 + 0.0) + 0.05 * np.linalg.norm(b) / np.linalg.norm(b2) * b2, W=W)
    # exact lowest eigenvectors of D^-1 K for comparison are too expensive at this size; also harvest from the random solve
    th2, W2 = ritz(Zb, Rb, k)
    it3, _, _, _ = pcg_store(b, W=W2)
    print('k=%2d: smallest Ritz (M metric) %.3g..%.3g | deflated: random rhs %d (plain %d), perturbed tension rhs %d, tension rhs w/ vectors from random solve %d (plain %d)'
          % (k, th[0], th[-1], it1, it0b, it2, it3, it0), flush=True)
