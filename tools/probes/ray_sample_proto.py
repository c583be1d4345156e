"""Prototype (numpy) of the sampled-ray form of ML_full_yf's ray search (round 5).

f(t) = b + sum_k d_k exp(-gamma |t D - v_k|^2) along a ray x = t su is sampled at N equally spaced points with ONE
pass over the support vectors (geometric recurrence in the sample index), converted to Chebyshev coefficients by a fixed
N x N matrix, and every later evaluation (2 % marching bracket, brentq iterates) runs on the polynomial.
Checks: sampling error, interpolation error against the a-priori bound, for the fixtures' SVCs.
"""
import sys
import numpy as np
from fractions import Fraction
from math import factorial, sqrt

def cheb_matrix(N):
    """exact inverse of V[j,i] = T_i(u_j), u_j = -1 + 2 j/(N-1), rounded to double"""
    u = [Fraction(-1) + Fraction(2 * j, N - 1) for j in range(N)]
    V = [[None] * N for _ in range(N)]
    for j in range(N):
        T0, T1 = Fraction(1), u[j]
        for i in range(N):
            V[j][i] = T0
            T0, T1 = T1, 2 * u[j] * T1 - T0
    # Gauss-Jordan on fractions
    A = [row[:] + [Fraction(int(i == j)) for i in range(N)] for j, row in enumerate(V)]
    for c in range(N):
        p = next(r for r in range(c, N) if A[r][c] != 0)
        A[c], A[p] = A[p], A[c]
        iv = 1 / A[c][c]
        A[c] = [x * iv for x in A[c]]
        for r in range(N):
            if r != c and A[r][c] != 0:
                f = A[r][c]
                A[r] = [x - f * y for x, y in zip(A[r], A[c])]
    return np.array([[float(A[i][N + j]) for j in range(N)] for i in range(N)])

def clenshaw(c, u):
    b1 = np.zeros_like(u); b2 = np.zeros_like(u)
    for k in range(len(c) - 1, 0, -1):
        b1, b2 = 2 * u * b1 - b2 + c[k], b1
    return u * b1 - b2 + c[0]

def run(name, N=16, nray=200, seed=0):
    z = np.load('tests/golden/%s.npz' % name)
    sv, dual = z['par_sv'], z['par_dual']
    gam, b, sc = float(z['par_gamma']), float(z['par_intercept']), float(z['par_scale_seq'])
    dev_only = bool(z['par_dev_only']); sy = float(z['par_sy'])
    rng = np.random.default_rng(seed)
    M = cheb_matrix(N)
    Sd = np.abs(dual).sum()
    K = 1.0865
    worst_s = worst_p = worst_b = 0.
    for r in range(nray):
        s = rng.standard_normal(6)
        s[:3] -= s[:3].mean() * rng.uniform(0, 1)
        sd = s.copy(); sd[:3] -= sd[:3].mean()
        seq = sqrt(0.5 * ((s[0]-s[1])**2 + (s[1]-s[2])**2 + (s[2]-s[0])**2 + 6 * (s[3]**2 + s[4]**2 + s[5]**2)))
        su = s / seq
        D = (su.copy())
        if dev_only:
            D[:3] -= D[:3].mean()
        D = D / sc
        DD = D @ D
        ck = sv @ D
        vv = (sv * sv).sum(1)
        x0 = sy * (0.5 if su[0] * su[1] < -1e-5 else 1.0)
        lo, hi = (0.47 * sy, 1.35 * sy) if x0 < sy else (0.72 * sy, 1.30 * sy)
        dl = (hi - lo) / (N - 1)
        def f(t):
            t = np.atleast_1d(t)[:, None]
            return b + (dual * np.exp(-gam * (t * t * DD - 2 * t * ck + vv))).sum(1)
        # recurrence sampling
        w = dual * np.exp(-gam * (lo * lo * DD - 2 * lo * ck + vv))
        rho = np.exp(2 * gam * dl * (ck - DD * lo))
        fs = np.empty(N)
        for j in range(N):
            fs[j] = b + np.exp(-gam * DD * dl * dl * j * j) * w.sum()
            w = w * rho
        tj = lo + dl * np.arange(N)
        worst_s = max(worst_s, np.abs(fs - f(tj)).max())
        c = M @ fs
        tt = np.linspace(lo, hi, 2001)
        u = (2 * tt - (lo + hi)) / (hi - lo)
        e = np.abs(clenshaw(c, u) - f(tt)).max()
        worst_p = max(worst_p, e)
        bound = Sd * K * sqrt(factorial(N)) / (4 * N) * (sqrt(2 * gam * DD) * dl) ** N
        worst_b = max(worst_b, bound)
    print('%-16s nsv %4d gamma %.3g scale %.4g sum|dual| %.4g  N %2d: sampling err %.2e  poly err %.2e  a-priori bound %.2e  |M|inf %.3g'
          % (name, len(dual), gam, sc, Sd, N, worst_s, worst_p, worst_b, np.abs(M).sum(1).max()))

if __name__ == '__main__':
    for name in ('svc_hill', 'svc_gossbarlat', 'svc_shear', 'svc_j2train'):
        for N in (12, 16):
            try:
                run(name, N)
            except Exception as ex:
                print(name, 'skipped:', ex)
