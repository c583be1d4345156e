"""ctypes loader for the CPU oracle (oracle/libplfx_oracle.so).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg —
never by anything under ``pylabfea_amd/``.  The functions mirror the reference signatures
(pyLabFEA v4.4.2 ``Material.response/calc_seq/calc_fgrad/calc_yf/ML_full_yf``,
``Element.calc_Bmat/calc_Kel``) on batches.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get('PLFO_LIB', os.path.join(HERE, 'libplfx_oracle.so'))   # PLFO_LIB: the sanitizer build (make -C oracle asan)

ELASTIC, HILL6, PRINC3, SVC6, TRESCA, BARLAT, SVC3, SVC_WH = 0, 1, 2, 3, 4, 5, 6, 7


class _Mat(C.Structure):
    _fields_ = [('kind', C.c_int), ('sdim', C.c_int), ('E', C.c_double), ('nu', C.c_double),
                ('sy', C.c_double), ('khard', C.c_double), ('hill', C.c_double * 6),
                ('dp', C.c_double * 3), ('nsv', C.c_int), ('ndof', C.c_int),
                ('dev_only', C.c_int), ('gamma', C.c_double), ('intercept', C.c_double),
                ('scale_seq', C.c_double), ('sv', C.c_void_p), ('dual', C.c_void_p),
                ('barlat', C.c_double * 18), ('barlat_exp', C.c_double), ('scale_wh', C.c_double)]


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.plfo_brentq.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


class Material(object):
    """Parameter record of one material (analytic Hill-6p/J2, elastic, or SVC)."""

    def __init__(self, kind=HILL6, E=0., nu=0., sy=0., khard=0., hill=None, drucker=0., sdim=6,
                 sv=None, dual=None, gamma=0., intercept=0., scale_seq=1., dev_only=False,
                 barlat=None, barlat_exp=0., scale_wh=1.):
        m = _Mat()
        m.scale_wh = scale_wh
        m.kind = kind
        m.sdim = sdim
        m.E, m.nu, m.sy, m.khard = E, nu, sy, khard
        h = np.ones(6) if hill is None else np.asarray(hill, dtype=float)
        for i in range(6):
            m.hill[i] = h[i] if i < len(h) else 1.
        for i in range(3):
            m.dp[i] = drucker
        if barlat is not None:
            for i in range(18):
                m.barlat[i] = barlat[i]
            m.barlat_exp = barlat_exp
        self._sv = self._dual = None
        if sv is not None:
            self._sv = _c(sv)
            self._dual = _c(dual)
            m.nsv, m.ndof = self._sv.shape
            m.sv = self._sv.ctypes.data
            m.dual = self._dual.ctypes.data
            m.gamma, m.intercept, m.scale_seq = gamma, intercept, scale_seq
            m.dev_only = int(dev_only)
        self.c = m

    @classmethod
    def from_golden(cls, z, prefix='par_'):
        """Build from the ``par_*`` entries written by oracle/gen_golden.py."""
        if prefix + 'sv' in z and z[prefix + 'sv'].shape[1] == 15:   # work-hardening features
            return cls(kind=SVC_WH, E=float(z[prefix + 'E']), nu=float(z[prefix + 'nu']), sy=float(z[prefix + 'sy']),
                       khard=float(z[prefix + 'khard']), hill=z[prefix + 'hill'], sv=z[prefix + 'sv'], dual=z[prefix + 'dual'],
                       gamma=float(z[prefix + 'gamma']), intercept=float(z[prefix + 'intercept']),
                       scale_seq=float(z[prefix + 'scale_seq']), dev_only=bool(z[prefix + 'dev_only']),
                       scale_wh=float(z[prefix + 'scale_wh']))
        if prefix + 'sv' in z:
            return cls(kind=SVC6 if int(z[prefix + 'sdim']) == 6 else SVC3, E=float(z[prefix + 'E']), nu=float(z[prefix + 'nu']),
                       sy=float(z[prefix + 'sy']), khard=float(z[prefix + 'khard']),
                       hill=z[prefix + 'hill'], sv=z[prefix + 'sv'], dual=z[prefix + 'dual'],
                       gamma=float(z[prefix + 'gamma']), intercept=float(z[prefix + 'intercept']),
                       scale_seq=float(z[prefix + 'scale_seq']), dev_only=bool(z[prefix + 'dev_only']))
        dp = z[prefix + 'dp']
        sdim = int(z[prefix + 'sdim'])
        return cls(kind=HILL6 if sdim == 6 else PRINC3, E=float(z[prefix + 'E']), nu=float(z[prefix + 'nu']),
                   sy=float(z[prefix + 'sy']), khard=float(z[prefix + 'khard']),
                   hill=z[prefix + 'hill'], drucker=float(dp[0]), sdim=int(z[prefix + 'sdim']))


def _mat_array(mats):
    arr = (_Mat * len(mats))()
    for i, m in enumerate(mats):
        arr[i] = m.c
    return arr


def sig_princ(sig):
    sig = _c(sig).reshape(-1, 6)
    out = np.empty((len(sig), 3))
    for i in range(len(sig)):
        lib().plfo_sig_princ(_p(sig[i]), _p(out[i]))
    return out


def calc_seq(mat, sig):
    sig = _c(sig).reshape(-1, 6)
    out = np.empty(len(sig))
    lib().plfo_seq_batch(C.byref(mat.c), len(sig), _p(sig), _p(out))
    return out


def calc_fgrad(mat, sig):
    sig = _c(sig).reshape(-1, 6)
    out = np.empty_like(sig)
    lib().plfo_fgrad_batch(C.byref(mat.c), len(sig), _p(sig), _p(out))
    return out


def calc_yf(mat, sig, epl=None):
    sig = _c(sig).reshape(-1, 6)
    epl = np.zeros_like(sig) if epl is None else _c(epl).reshape(-1, 6)
    out = np.empty(len(sig))
    lib().plfo_yf_batch(C.byref(mat.c), len(sig), _p(sig), _p(epl), _p(out))
    return out


def ML_full_yf(mat, sig, epl=None):
    sig = _c(sig).reshape(-1, 6)
    epl = np.zeros_like(sig) if epl is None else _c(epl).reshape(-1, 6)
    out = np.empty(len(sig))
    st = np.zeros(len(sig), dtype=np.int32)
    lib().plfo_full_yf_batch(C.byref(mat.c), len(sig), _p(sig), _p(epl), _p(out), _p(st))
    return out, st


def ML_full_yf_ld(mat, sig, epl, ld):
    """ML_full_yf with a loading direction (material.py:454-462), as calc_scf calls it (model.py:1049-1053)."""
    sig = _c(sig).reshape(-1, 6)
    epl = np.zeros_like(sig) if epl is None else _c(epl).reshape(-1, 6)
    ld = _c(ld).reshape(6)
    out = np.empty(len(sig))
    lib().plfo_full_yf_ld_batch(C.byref(mat.c), len(sig), _p(sig), _p(epl), _p(ld), _p(out))
    return out


def response(mats, CVs, sig, epl, deps, mat_id=None, nthreads=0):
    """Batched Material.response.  mats: list of Material, CVs: (nmat,36) element CV."""
    if isinstance(mats, Material):
        mats = [mats]
    sig = _c(sig).reshape(-1, 6)
    n = len(sig)
    epl = _c(epl).reshape(-1, 6)
    deps = _c(deps).reshape(-1, 6)
    CVs = _c(CVs).reshape(len(mats), 36)
    mid = np.zeros(n, dtype=np.int32) if mat_id is None else _c(mat_id, np.int32)
    fy = np.zeros(n)
    so = np.zeros((n, 6))
    dp = np.zeros((n, 6))
    ct = np.zeros((n, 36))
    ns = np.zeros(n, dtype=np.int32)
    lib().plfo_response_batch(_mat_array(mats), n, _p(mid), _p(sig), _p(epl), _p(deps), _p(CVs),
                              _p(fy), _p(so), _p(dp), _p(ct), _p(ns), int(nthreads))
    return fy, so, dp, ct, ns


def calc_Bmat(lx, ly, x, y, planestress, CV, E, nu):
    B = np.zeros(48)
    lib().plfo_calc_Bmat(C.c_double(lx), C.c_double(ly), C.c_double(x), C.c_double(y),
                         int(planestress), _p(_c(CV).reshape(36)), C.c_double(E), C.c_double(nu), _p(B))
    return B.reshape(6, 8)


def calc_Kel(lx, ly, thick, planestress, CV, E, nu, D):
    K = np.zeros(64)
    lib().plfo_calc_Kel(C.c_double(lx), C.c_double(ly), C.c_double(thick), int(planestress),
                        _p(_c(CV).reshape(36)), C.c_double(E), C.c_double(nu),
                        _p(_c(D).reshape(36)), _p(K))
    return K.reshape(8, 8)


def strain(lx, ly, planestress, CV, E, nu, ue):
    e = np.zeros(6)
    lib().plfo_strain(C.c_double(lx), C.c_double(ly), int(planestress), _p(_c(CV).reshape(36)),
                      C.c_double(E), C.c_double(nu), _p(_c(ue)), _p(e))
    return e


def kel_batch(lxy, mat_id, thick, planestress, CVs, Es, nus, D):
    lxy = _c(lxy).reshape(-1, 2)
    nel = len(lxy)
    mid = _c(mat_id, np.int32)
    CVs = _c(CVs).reshape(-1, 36)
    D = _c(D).reshape(nel, 36)
    K = np.empty((nel, 64))
    lib().plfo_kel_batch(nel, _p(lxy), _p(mid), C.c_double(thick), int(planestress), _p(CVs),
                         _p(_c(Es)), _p(_c(nus)), _p(D), _p(K))
    return K.reshape(nel, 8, 8)


def strain_batch(conn, lxy, mat_id, planestress, CVs, Es, nus, u):
    conn = _c(conn, np.int32).reshape(-1, 4)
    nel = len(conn)
    eps = np.empty((nel, 6))
    lib().plfo_strain_batch(nel, _p(conn), _p(_c(lxy).reshape(nel, 2)), _p(_c(mat_id, np.int32)),
                            int(planestress), _p(_c(CVs).reshape(-1, 36)), _p(_c(Es)), _p(_c(nus)),
                            _p(_c(u)), _p(eps))
    return eps


def pcg_csr(K, b, free, x0, rtol=1.e-10, maxit=200000, nthreads=0):
    """Jacobi-PCG on a scipy CSR matrix restricted to the free DOFs (plfo_pcg_csr); returns (x, iterations, relres)"""
    n = K.shape[0]
    indptr = _c(K.indptr, np.int32)
    indices = _c(K.indices, np.int32)
    data = _c(K.data)
    x = np.array(x0, dtype=np.float64, copy=True)
    fm = _c(free, np.uint8)
    rel = C.c_double(0.)
    f = lib().plfo_pcg_csr
    f.restype = C.c_int
    its = f(int(n), _p(indptr), _p(indices), _p(data), _p(_c(b)), _p(fm), _p(x), C.c_double(rtol), int(maxit),
            int(nthreads), C.byref(rel))
    return x, int(its), float(rel.value)


# ---- work-hardening-aware SVC materials (kind SVC_WH): Material.khard is explicit state
def fgrad_wh(mat, sig, epl=None):
    """calc_fgrad(sig, epl) point by point: (gradient (N,6), raw hardening value (N,))"""
    sig = _c(sig).reshape(-1, 6)
    n = len(sig)
    e = None if epl is None else _c(epl).reshape(-1, 6)
    a = np.empty((n, 6))
    kh = np.empty(n)
    lib().plfo_fgrad_wh_batch(C.byref(mat.c), n, _p(sig), None if e is None else _p(e), _p(a), _p(kh))
    return a, kh


def full_yf_wh(mat, sig, epl=None, khard=None):
    sig = _c(sig).reshape(-1, 6)
    n = len(sig)
    e = None if epl is None else _c(epl).reshape(-1, 6)
    k = None if khard is None else _c(np.broadcast_to(np.asarray(khard, dtype=float), (n,)).copy())
    out = np.empty(n)
    lib().plfo_full_yf_wh_batch(C.byref(mat.c), n, _p(sig), None if e is None else _p(e), None if k is None else _p(k), _p(out))
    return out


def response_wh(mats, CVs, sig, epl, deps, khard_in=None, mat_id=None, sequential=False, nthreads=0):
    """response() with the hardening modulus as state.  sequential=False: per point, khard_in / khard_out are (N,).
    sequential=True: ONE material object per material carried through the points in index order (the reference's loop over
    the elements); khard_in / khard_out are (nmat,)."""
    if isinstance(mats, Material):
        mats = [mats]
    sig = _c(sig).reshape(-1, 6)
    n = len(sig)
    epl = _c(epl).reshape(-1, 6)
    deps = _c(deps).reshape(-1, 6)
    CVs = _c(CVs).reshape(len(mats), 36)
    mid = np.zeros(n, dtype=np.int32) if mat_id is None else _c(mat_id, np.int32)
    nk = len(mats) if sequential else n
    kin = None if khard_in is None else _c(np.broadcast_to(np.asarray(khard_in, dtype=float), (nk,)).copy())
    kout = np.zeros(nk)
    fy = np.zeros(n)
    so = np.zeros((n, 6))
    dp = np.zeros((n, 6))
    ct = np.zeros((n, 36))
    ns = np.zeros(n, dtype=np.int32)
    kpt = np.zeros(n)
    lib().plfo_response_wh_batch(_mat_array(mats), n, _p(mid), _p(sig), _p(epl), _p(deps), _p(CVs),
                                 None if kin is None else _p(kin), _p(fy), _p(so), _p(dp), _p(ct), _p(ns), _p(kout),
                                 int(bool(sequential)), int(nthreads), _p(kpt))
    if sequential:
        return fy, so, dp, ct, ns, kout, kpt
    return fy, so, dp, ct, ns, kout


def yf_wh(mat, sig, epl=None):
    sig = _c(sig).reshape(-1, 6)
    e = None if epl is None else _c(epl).reshape(-1, 6)
    out = np.empty(len(sig))
    lib().plfo_yf_wh_batch(C.byref(mat.c), len(sig), _p(sig), None if e is None else _p(e), _p(out))
    return out
