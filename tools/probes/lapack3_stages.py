#!/usr/bin/env python3
"""How pylabfea_amd/csrc/plfx_lapack3.hpp was pinned (build container only; numpy's bundled OpenBLAS is the oracle here).

The replay of dgeev for symmetric 3 x 3 matrices is compared with the library numpy itself loads, stage by stage and BITWISE:
its dgehrd / dorghr / dlahqr / dtrevc3 are called through ctypes (64-bit-integer symbols scipy_*_64_) and held against
candidate arithmetic forms of the BLAS kernels underneath (plain IEEE in source order / fused multiply-add), evaluated exactly
with rational arithmetic.  Findings (OpenBLAS 0.3.29, Haswell kernels, this container):
  dgehd2 -> dlarf('Right'): w = fma(H(r,3), v2, H(r,2)) (dgemv_n tail), H += fma(w, -tau v, H) (dger = daxpy tail)
            dlarf('Left') : w = H(2,c) + H(3,c) v2 NOT fused (dgemv_t), H(3,c) = fma(v2, -tau w, H(3,c))
  dorg2r  : Q(3,3) = fma(v2, -tau v2, 1)
  dlahqr  : plain Fortran in source order, except dnrm2 inside dlarfg (x87 extended precision) and drot:
            x' = fma(c, x, s y), y' = fma(c, y, -(s x))
  dtrevc3 : blocked back-transformation = one dgemm: fused accumulation over j ascending from the first product
With these, 300 of 300 Hessenberg forms, 2000 of 2000 eigenvalue triples and 99.9 % of 20 000 eigenvector matrices are
bit-identical to numpy's; the library's own routine is then checked against np.linalg.eig by tests/test_lapack3.py.

    python tools/probes/lapack3_stages.py            # dgehrd hypotheses (the decisive stage), then the end-to-end census
"""
import ctypes as C
import glob
import itertools
import math
import os
import sys
from fractions import Fraction as F

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = C.CDLL(glob.glob(os.path.join(os.path.dirname(np.__file__), '..', 'numpy.libs', 'libscipy_openblas*'))[0])
i64 = C.c_int64


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def mat(s):
    return np.array([[s[0], s[5], s[4]], [s[5], s[1], s[3]], [s[4], s[3], s[2]]])


def dgehrd(A):
    A = np.asfortranarray(A.copy())
    n, ilo, ihi, lda, lw, info = i64(3), i64(1), i64(3), i64(3), i64(256), i64(0)
    tau, work = np.zeros(3), np.zeros(256)
    L.scipy_dgehrd_64_(C.byref(n), C.byref(ilo), C.byref(ihi), P(A), C.byref(lda), P(tau), P(work), C.byref(lw), C.byref(info))
    return A, tau


def fma(a, b, c):
    return float(F(a) * F(b) + F(c))


def lapy2(x, y):
    w, z = max(abs(x), abs(y)), min(abs(x), abs(y))
    return w if z == 0 else w * math.sqrt(1. + (z / w) ** 2)


def my_dgehd2(A, f_gemv, f_ger, f_gemvt, f_ger2):
    H = [[float(A[i][j]) for j in range(3)] for i in range(3)]
    alpha, x = H[1][0], H[2][0]
    beta = -math.copysign(lapy2(alpha, abs(x)), alpha)
    tau = (beta - alpha) / beta
    v2 = x * (1. / (alpha - beta))
    for r in range(3):
        w = fma(H[r][2], v2, H[r][1]) if f_gemv else H[r][1] + H[r][2] * v2
        if f_ger:
            H[r][1], H[r][2] = fma(w, -tau, H[r][1]), fma(w, -tau * v2, H[r][2])
        else:
            H[r][1], H[r][2] = H[r][1] + w * (-tau), H[r][2] + w * (-tau * v2)
    for c in (1, 2):
        w = fma(H[2][c], v2, H[1][c]) if f_gemvt else H[1][c] + H[2][c] * v2
        t = -tau * w
        H[1][c] = H[1][c] + t
        H[2][c] = fma(v2, t, H[2][c]) if f_ger2 else H[2][c] + v2 * t
    H[1][0], H[2][0] = beta, 0.
    return np.array(H)


def main():
    rng = np.random.default_rng(5)
    tests = [mat(rng.normal(size=6) * 100) for _ in range(300)]
    print('dgehd2: (fused dgemv_n, fused dger, fused dgemv_t, fused second dger) -> Hessenberg forms bit-identical to the library')
    for hyp in itertools.product((0, 1), repeat=4):
        ok = sum(np.array_equal(np.triu(dgehrd(A)[0], -1), my_dgehd2(A, *hyp)) for A in tests)
        print('   ', hyp, ok, 'of', len(tests))
    # end to end: the library's own routine (host entry of the C-ABI) against np.linalg.eig
    from pylabfea_amd import _lib
    for name, gen in (('random full', lambda n: rng.normal(size=(n, 6)) * 100),
                      ('nearly diagonal', lambda n: rng.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 1e-6, 1e-6, 1e-6])),
                      ('small integers', lambda n: np.round(rng.normal(size=(n, 6)) * 3))):
        S = gen(20000)
        w, V = _lib.eig3_host(S)
        same_w = same_v = 0
        for i, s in enumerate(S):
            rw, rv = np.linalg.eig(mat(s))
            if np.iscomplexobj(rw):
                continue
            same_w += np.array_equal(w[i], rw)
            sg = np.sign(np.sum(V[i] * rv, axis=0))
            same_v += np.array_equal(V[i] * sg, rv)
        print('%-16s eigenvalues bit-identical %5d, eigenvectors (up to sign) %5d of %d' % (name, same_w, same_v, len(S)))


if __name__ == '__main__':
    main()
