"""Host-side tensor helpers of the façade (mirror of pylabfea/basic.py for the names the hot path
and its harness use).  These operate on the handful of homogenised (N,6) records a solve produces
(``sgl/egl/epgl``); everything per element / per node runs in libplfx on the GPU.

Reference: /root/reference/src/pylabfea/basic.py:26 (yf_tolerance), :30 (sig_eq_j2), :304 (sig_dev),
:328 (eps_eq).
"""
import numpy as np

yf_tolerance = 5.e-3
"""Tolerance: plastic yielding if yield function > yf_tolerance (basic.py:26)"""


def _as2d(a, name):
    a = np.asarray(a, dtype=float)
    sh = a.shape
    if sh == (3,) or sh == (6,):
        return a[None, :], True
    if a.ndim == 2 and sh[1] in (3, 6):
        return a, False
    raise TypeError('%s: unknown format of tensor, shape=%s' % (name, sh))


def sig_eq_j2(sig):
    """J2 equivalent stress of principal (3,)/(N,3) or Voigt (6,)/(N,6) stresses (basic.py:30-65).

    The reference diagonalises Voigt input first; the J2 invariant is evaluated here directly from the
    Voigt components, which is the same number without the eigen-solve."""
    s, single = _as2d(sig, 'sig_eq_j2')
    d12 = s[:, 0] - s[:, 1]
    d23 = s[:, 1] - s[:, 2]
    d31 = s[:, 2] - s[:, 0]
    sj2 = 0.5 * (np.square(d12) + np.square(d23) + np.square(d31))
    if s.shape[1] == 6:
        sj2 = sj2 + 3. * (np.square(s[:, 3]) + np.square(s[:, 4]) + np.square(s[:, 5]))
    seq = np.sqrt(sj2)
    return seq[0] if single else seq


def sig_dev(sig):
    """Deviatoric part of a (principal or Voigt) stress (basic.py:304-325)."""
    s, single = _as2d(sig, 'sig_dev')
    sd = np.array(s)
    sd[:, 0:3] -= (np.sum(s[:, 0:3], axis=1) / 3.)[:, None]
    return sd[0] if single else sd


def eps_eq(eps):
    """Equivalent strain of principal or Voigt (engineering shear) strains (basic.py:328-360)."""
    e, single = _as2d(eps, 'eps_eq')
    n = np.sum(e[:, 0:3] * e[:, 0:3], axis=1)
    if e.shape[1] == 6:
        n = n + 0.5 * np.sum(e[:, 3:6] * e[:, 3:6], axis=1)
    eeq = np.sqrt(2. * n / 3.)
    return eeq[0] if single else eeq
