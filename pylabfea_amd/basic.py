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


def _voigt_to_tensors(sig, name):
    """(6,), (N,6), (3,3) or (N,3,3) -> (N,3,3) symmetric tensors + flag 'single input'"""
    a = np.asarray(sig, dtype=float)
    if a.shape == (3, 3):
        return a[None], True
    if a.ndim == 3 and a.shape[1:] == (3, 3):
        return np.array(a), False
    if a.shape == (6,):
        a, single = a[None], True
    elif a.ndim == 2 and a.shape[1] == 6:
        single = False
    else:
        raise TypeError('Unknown format of stress in %s: nsc=%d, sh=%s' % (name, len(a), a.shape))
    t = np.empty((len(a), 3, 3))
    t[:, 0, 0], t[:, 1, 1], t[:, 2, 2] = a[:, 0], a[:, 1], a[:, 2]
    t[:, 1, 2] = t[:, 2, 1] = a[:, 3]
    t[:, 0, 2] = t[:, 2, 0] = a[:, 4]
    t[:, 0, 1] = t[:, 1, 0] = a[:, 5]
    return t, single


def sig_princ(sig):
    """Principal stresses and eigenvector matrices in the reference's *axis-tracking* order (basic.py:107-179): the
    eigenpairs of the general solver ``np.linalg.eig`` are not sorted but grouped by the Cartesian axis on which each
    row of the eigenvector matrix has its largest component (rows with axis 0 first, then 1, then 2, stable), and the
    eigenvector matrix is flipped to a positive determinant.  Same LAPACK call as the reference, so the order agrees
    with it also where the rule is ambiguous; the device-side closed form (`sig_princ_dev`) is used on the hot path."""
    t, single = _voigt_to_tensors(sig, 'sig_princ')
    sp, ev = np.linalg.eig(t)                         # batched dgeev, as one call per tensor in the reference
    iev = np.argmax(np.abs(ev), axis=2)               # (N,3): dominant axis of every row of ev
    order = np.argsort(iev, axis=1, kind='stable')    # rows with axis 0 first, then 1, then 2
    spa = np.take_along_axis(sp, order, axis=1)
    eva = np.take_along_axis(ev, order[:, :, None], axis=1)
    eva = np.where((np.linalg.det(eva) < 0.)[:, None, None], -eva, eva)
    if single:
        return spa[0], eva[0]
    return spa, eva


_A_VEC = np.array([1., -0.5, -0.5]) / np.sqrt(1.5)   # unit vectors spanning the deviatoric plane (basic.py:21-24)
_B_VEC = np.array([0., 1., -1.]) / np.sqrt(2.)


def sig_polar_ang(sig):
    """Polar angle of (principal or Voigt) stresses in the deviatoric plane, in [-pi, pi] (basic.py:68-104)."""
    s, single = _as2d(sig, 'sig_polar_ang')
    sp = sig_princ(s)[0] if s.shape[1] == 6 else s
    dev = sp - (np.sum(sp, axis=1) / 3.)[:, None]
    vn = np.linalg.norm(dev, axis=1)
    vn[vn < 1.e-4] = 1.
    dn = dev / vn[:, None]
    theta = np.arctan2(dn @ _B_VEC, dn @ _A_VEC)
    return theta[0] if single else theta


class Stress(object):
    """A Voigt stress with its tensor, principal values (axis-tracking order), hydrostatic and deviatoric parts
    (basic.py:366-484); ``seq(mat)`` is the material's equivalent stress (evaluated on the GPU through
    ``Material.calc_seq``), J2 without a material."""

    def __init__(self, sv):
        self.v = self.voigt = np.array(sv, dtype=float)
        self.t = self.tens = _voigt_to_tensors(self.v, 'Stress')[0][0]
        self.princ, self.evec = sig_princ(self.tens)
        self.p = self.princ
        self.h = self.hydrostatic = np.sum(self.p) / 3.
        self.d = self.dev = self.v - np.array([self.h, self.h, self.h, 0., 0., 0.])

    def seq(self, mat=None):
        return sig_eq_j2(self.p) if mat is None else mat.calc_seq(self.v)

    def theta(self):
        return sig_polar_ang(self.p)

    def seq_j2(self):
        return sig_eq_j2(self.p)


class Strain(object):
    """A Voigt strain (engineering shear) with tensor, principal values and equivalent strain (basic.py:487-545)."""

    def __init__(self, sv):
        self.v = self.voigt = np.array(sv, dtype=float)
        self.t = self.tens = _voigt_to_tensors(self.v, 'Strain')[0][0]
        self.princ, self.evec = np.linalg.eig(self.tens)
        self.p = self.princ

    def eeq(self):
        return eps_eq(self.v)

    def inv(self):
        """element-wise inverse of the Voigt components, zeros kept"""
        out = np.zeros(6)
        nz = np.abs(self.voigt) > 1.e-9
        out[nz] = 1. / self.voigt[nz]
        return out
