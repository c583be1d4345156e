"""ctypes binding of libplfx.so (the C-ABI declared in include/plfx.h).

There is deliberately NO CPU fallback: if the HIP library is missing or no MI355X is visible,
every entry point raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C pylabfea_amd/csrc``).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PLFX_LIB', os.path.join(_HERE, 'libplfx.so'))  # PLFX_LIB: kernel-variant experiments

# yield-function kinds (include/plfx.h)
ELASTIC, HILL6, PRINC3, SVC6, TRESCA, BARLAT, SVC3, SVC_WH = 0, 1, 2, 3, 4, 5, 6, 7

# state ids of plfx_state_get/_set
ST_SIG, ST_EPS, ST_EPL, ST_RES_SIG, ST_RES_DEPL, ST_ELSTIFF, ST_U, ST_F, ST_DU, ST_FYN, ST_MAXSTEPS, ST_KHARD = range(12)

# timing families
T_SWEEP, T_SPMV, T_CGUPD, T_ASSEMBLE, T_VCYCLE, T_SMOOTH, T_SWEEP_HEAVY, T_COMM = range(8)


class PlfxError(RuntimeError):
    pass


class CMaterial(C.Structure):
    _fields_ = [('kind', C.c_int32), ('sdim', C.c_int32), ('CV', C.c_double * 36),
                ('E', C.c_double), ('nu', C.c_double), ('sy', C.c_double), ('khard', C.c_double),
                ('hill', C.c_double * 6), ('drucker', C.c_double), ('nsv', C.c_int32),
                ('nfeat', C.c_int32), ('dev_only', C.c_int32), ('_pad', C.c_int32),
                ('gamma', C.c_double), ('intercept', C.c_double), ('scale_seq', C.c_double),
                ('sv', C.c_void_p), ('dual', C.c_void_p), ('barlat', C.c_double * 18),
                ('barlat_exp', C.c_double), ('barlat_normal', C.c_int32), ('_pad2', C.c_int32),
                ('scale_wh', C.c_double)]


# every symbol include/plfx.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    'plfx_create', 'plfx_destroy', 'plfx_last_error', 'plfx_version', 'plfx_device_count', 'plfx_device_info',
    'plfx_stream', 'plfx_sync', 'plfx_set_materials', 'plfx_seq_batch', 'plfx_fgrad_batch',
    'plfx_yf_batch', 'plfx_full_yf_batch', 'plfx_response_batch', 'plfx_set_mesh', 'plfx_get_bmat',
    'plfx_get_kel', 'plfx_state_get', 'plfx_state_set', 'plfx_state_reset', 'plfx_gather',
    'plfx_assemble', 'plfx_get_csr', 'plfx_apply_bc', 'plfx_solve', 'plfx_sweep', 'plfx_scf_stats',
    'plfx_update_state', 'plfx_global_sums', 'plfx_comm_unique_id', 'plfx_comm_init',
    'plfx_timing_get', 'plfx_timing_reset', 'plfx_timing_enable', 'plfx_set_grid', 'plfx_set_precond',
    'plfx_precond_info', 'plfx_set_operator', 'plfx_operator_info', 'plfx_gen_structured', 'plfx_reuse_info', 'plfx_timing_select', 'plfx_finish_fetch', 'plfx_sweep_info', 'plfx_matvec', 'plfx_set_bc_plan', 'plfx_apply_bc_plan',
    'plfx_set_finish_set', 'plfx_finish_step', 'plfx_scf_all', 'plfx_comm_info', 'plfx_comm_init_callback',
    'plfx_set_bc_sources',
    'plfx_load_step', 'plfx_set_strip', 'plfx_strip_info', 'plfx_allreduce_host',
    'plfx_response_batch_kh', 'plfx_fgrad_batch_wh', 'plfx_timing_sample', 'plfx_solve_fallbacks', 'plfx_comm_selftest',
    'plfx_indefinite_info', 'plfx_pattern_selftest', 'plfx_precond_bench', 'plfx_set_wh_mode', 'plfx_wh_info', 'plfx_wh_carry', 'plfx_set_mesh_structured',
    'plfx_svc_info', 'plfx_sqmr_info', 'plfx_fgrad_seq_batch', 'plfx_precond_apply', 'plfx_predict_info',
    'plfx_set_response_maxit', 'plfx_sig_princ_host', 'plfx_eig3_host',
]

_lib = None


def load():
    """dlopen libplfx.so; raises PlfxError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PlfxError('libplfx.so not found at %s - build it first (__graft_entry__.build()); '
                        'pylabfea_amd has no CPU fallback' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.plfx_last_error.restype = C.c_char_p
    lib.plfx_version.restype = C.c_char_p
    lib.plfx_stream.restype = C.c_void_p
    lib.plfx_destroy.restype = None
    _lib = lib
    return lib


class CStep(C.Structure):
    """plfx_step of include/plfx.h: one load step of Model.solve"""
    _fields_ = [('il', C.c_int32), ('nonlin', C.c_int32), ('has_nodeset', C.c_int32), ('warm', C.c_int32),
                ('maxit', C.c_int32), ('defer_slot', C.c_int32), ('rtol', C.c_double),
                ('bcl0', C.c_double * 2), ('bcb0', C.c_double * 2),
                ('max_dbcr', C.c_double * 2), ('max_dbct', C.c_double * 2), ('max_dbcn', C.c_double * 2),
                ('bcr', C.c_double * 2), ('bct', C.c_double * 2), ('bcn', C.c_double * 2),
                ('bcr0', C.c_double * 2), ('bct0', C.c_double * 2), ('bcn0', C.c_double * 2),
                ('sld', C.c_double * 6),
                ('dbcr', C.c_double * 2), ('dbct', C.c_double * 2), ('dbcn', C.c_double * 2),
                ('scale_bc', C.c_double),
                ('nit', C.c_int32), ('nconv', C.c_int32), ('nsweeps', C.c_int32), ('nsolves', C.c_int32),
                ('soft_fail', C.c_int32), ('inconsistent_entry', C.c_int32),
                ('its', C.c_int32 * 40), ('relres', C.c_double * 40)]


def device_count():
    """GPUs visible to this process (plfx_device_count; 0 without a GPU)"""
    return int(load().plfx_device_count())


def _dp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def gen_structured(NX, NY):
    """(conn[Nel,4], noleft, noright, nobot, notop) of the reference's structured grid from the library's own index
    generator (plfx_gen_structured; host-only, works without a GPU)"""
    lib = load()
    conn = np.empty((NX * NY, 4), dtype=np.int32)
    le, ri = np.empty(NY + 1, dtype=np.int32), np.empty(NY + 1, dtype=np.int32)
    bo, to = np.empty(NX + 1, dtype=np.int32), np.empty(NX + 1, dtype=np.int32)
    rc = lib.plfx_gen_structured(int(NX), int(NY), _dp(conn), _dp(le), _dp(ri), _dp(bo), _dp(to))
    if rc:
        raise PlfxError('plfx_gen_structured(%d, %d): error %d' % (NX, NY, rc))
    return conn, le, ri, bo, to


def sig_princ_host(sig):
    """basic.sig_princ's principal stresses (reference order) of (N,6) Voigt stresses from the library's host-side LAPACK replay"""
    lib = load()
    s = _f64(sig).reshape(-1, 6)
    sp = np.empty((len(s), 3))
    rc = lib.plfx_sig_princ_host(len(s), _dp(s), _dp(sp))
    if rc < 0:
        raise PlfxError('plfx_sig_princ_host: error %d' % rc)
    return sp


def eig3_host(sig):
    """(w[N,3], V[N,3,3]): eigenvalues in LAPACK dgeev's order and unit eigenvectors (columns) of the symmetric stress tensors"""
    lib = load()
    s = _f64(sig).reshape(-1, 6)
    w, V = np.empty((len(s), 3)), np.empty((len(s), 3, 3))
    rc = lib.plfx_eig3_host(len(s), _dp(s), _dp(w), _dp(V))
    if rc < 0:
        raise PlfxError('plfx_eig3_host: error %d' % rc)
    return w, V


def pack_material(kind, CV, E=0., nu=0., sy=0., khard=0., hill=None, drucker=0., svc=None, barlat=None,
                  barlat_exp=0., barlat_normal=False):
    """Build a plfx_material record.  svc = dict(sv, dual, gamma, intercept, scale_seq, dev_only).
    Returns (struct, keepalive) - keepalive holds the arrays the struct points to."""
    m = CMaterial()
    m.kind = int(kind)
    m.sdim = 3 if kind in (PRINC3, SVC3) else 6
    if barlat is not None:
        for i in range(18):
            m.barlat[i] = float(barlat[i])
        m.barlat_exp = float(barlat_exp)
        m.barlat_normal = int(bool(barlat_normal))
    cv = _f64(CV).reshape(36)
    for i in range(36):
        m.CV[i] = cv[i]
    m.E, m.nu = float(E), float(nu)
    m.sy = 0. if sy is None else float(sy)
    m.khard = 0. if khard is None else float(khard)
    h = np.ones(6) if hill is None else np.asarray(hill, dtype=float)
    for i in range(6):
        m.hill[i] = h[i] if i < len(h) else 1.
    m.drucker = float(drucker or 0.)
    keep = []
    if svc is not None:
        sv = _f64(svc['sv'])
        dual = _f64(svc['dual']).reshape(-1)
        keep = [sv, dual]
        m.nsv, m.nfeat = sv.shape
        m.dev_only = int(bool(svc.get('dev_only', False)))
        m.gamma = float(svc['gamma'])
        m.intercept = float(svc['intercept'])
        m.scale_seq = float(svc['scale_seq'])
        m.scale_wh = float(svc.get('scale_wh', 1.) or 1.)
        m.sv = sv.ctypes.data
        m.dual = dual.ctypes.data
    return m, keep


class Context(object):
    """One libplfx context = one GPU + one HIP stream.  Thin, typed wrapper over the C-ABI."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.plfx_create(int(device), C.byref(self.h))
        if rc != 0:
            msg = self.lib.plfx_last_error(self.h).decode() if self.h else 'plfx_create failed'
            if self.h:
                self.lib.plfx_destroy(self.h)
                self.h = C.c_void_p()
            raise PlfxError('libplfx: %s' % msg)
        self.nmat = 0
        self.nel = 0
        self.nel_owned = 0
        self.ndof = 0
        self._keep = []

    def close(self):
        if getattr(self, 'h', None):
            self.lib.plfx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, soft=False):
        if rc < 0 or (rc > 0 and not soft):
            raise PlfxError('libplfx error %d: %s' % (rc, self.lib.plfx_last_error(self.h).decode()))
        return rc

    # -- info
    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        hbm = C.c_int64()
        self._chk(self.lib.plfx_device_info(self.h, name, 256, C.byref(cus), C.byref(hbm)))
        return name.value.decode(), cus.value, hbm.value

    def stream(self):
        return self.lib.plfx_stream(self.h)

    def sync(self):
        self._chk(self.lib.plfx_sync(self.h))

    # -- materials
    def set_materials(self, records):
        """records: list of (CMaterial, keepalive)"""
        arr = (CMaterial * len(records))()
        self._keep = []
        for i, (m, keep) in enumerate(records):
            arr[i] = m
            self._keep.append(keep)
        self._chk(self.lib.plfx_set_materials(self.h, len(records), arr))
        self.nmat = len(records)

    # -- batched point evaluation (host AoS arrays)
    def seq(self, mat, sig):
        sig = _f64(sig).reshape(-1, 6)
        out = np.empty(len(sig))
        self._chk(self.lib.plfx_seq_batch(self.h, int(mat), len(sig), _dp(sig), _dp(out)))
        return out

    def fgrad(self, mat, sig):
        sig = _f64(sig).reshape(-1, 6)
        out = np.empty_like(sig)
        self._chk(self.lib.plfx_fgrad_batch(self.h, int(mat), len(sig), _dp(sig), _dp(out)))
        return out

    def fgrad_seq(self, mat, sig, seq):
        """calc_fgrad(sig, seq=seq) of an analytic Hill material: Voigt deviator over 2 seq (material.py:834-847)"""
        sig = _f64(sig).reshape(-1, 6)
        seq = _f64(seq).reshape(-1)
        if len(seq) != len(sig):
            raise ValueError('fgrad_seq: one equivalent stress per stress expected')
        out = np.empty_like(sig)
        self._chk(self.lib.plfx_fgrad_seq_batch(self.h, int(mat), len(sig), _dp(sig), _dp(seq), _dp(out)))
        return out

    def fgrad_wh(self, mat, sig, epl=None):
        """calc_fgrad(sig, epl) of a work-hardening SVC material: (gradient (N,6), raw hardening value per point (N,))"""
        sig = _f64(sig).reshape(-1, 6)
        epl = np.zeros_like(sig) if epl is None else _f64(epl).reshape(-1, 6)
        out = np.empty_like(sig)
        kh = np.empty(len(sig))
        self._chk(self.lib.plfx_fgrad_batch_wh(self.h, int(mat), len(sig), _dp(sig), _dp(epl), _dp(out), _dp(kh)))
        return out, kh

    def yf(self, mat, sig, epl=None):
        sig = _f64(sig).reshape(-1, 6)
        epl = np.zeros_like(sig) if epl is None else _f64(epl).reshape(-1, 6)
        out = np.empty(len(sig))
        self._chk(self.lib.plfx_yf_batch(self.h, int(mat), len(sig), _dp(sig), _dp(epl), _dp(out)))
        return out

    def full_yf(self, mat, sig, epl=None, ld=None):
        sig = _f64(sig).reshape(-1, 6)
        epl = np.zeros_like(sig) if epl is None else _f64(epl).reshape(-1, 6)
        ldp = None if ld is None else _f64(ld).reshape(6)
        out = np.empty(len(sig))
        st = np.zeros(len(sig), dtype=np.int32)
        self._chk(self.lib.plfx_full_yf_batch(self.h, int(mat), len(sig), _dp(sig), _dp(epl), _dp(ldp),
                                              _dp(out), _dp(st)))
        return out, st

    def set_response_maxit(self, maxit=50):
        self._chk(self.lib.plfx_set_response_maxit(self.h, int(maxit)))

    def response(self, sig, epl, deps, mat_id=None, khard_in=None, return_khard=False, maxit=50):
        """khard_in / return_khard: entry / exit value of Material.khard per point (work-hardening SVC materials);
        maxit: Material.response's argument (sub-steps of a sub-divided increment)"""
        if maxit != 50:
            self.set_response_maxit(maxit)
            try:
                return self.response(sig, epl, deps, mat_id, khard_in, return_khard)
            finally:
                self.set_response_maxit(50)
        sig = _f64(sig).reshape(-1, 6)
        n = len(sig)
        epl = _f64(epl).reshape(-1, 6)
        deps = _f64(deps).reshape(-1, 6)
        mid = None if mat_id is None else _i32(mat_id)
        fy = np.empty(n)
        so = np.empty((n, 6))
        dp = np.empty((n, 6))
        ct = np.empty((n, 36))
        ns = np.empty(n, dtype=np.int32)
        if khard_in is not None or return_khard:
            kin = None if khard_in is None else _f64(np.broadcast_to(np.asarray(khard_in, dtype=float), (n,)).copy())
            kout = np.empty(n)
            self._chk(self.lib.plfx_response_batch_kh(self.h, n, _dp(mid), _dp(sig), _dp(epl), _dp(deps), _dp(kin),
                                                      _dp(fy), _dp(so), _dp(dp), _dp(ct), _dp(ns), _dp(kout)))
            if return_khard:
                return fy, so, dp, ct, ns, kout
            return fy, so, dp, ct, ns
        self._chk(self.lib.plfx_response_batch(self.h, n, _dp(mid), _dp(sig), _dp(epl), _dp(deps),
                                               _dp(fy), _dp(so), _dp(dp), _dp(ct), _dp(ns)))
        return fy, so, dp, ct, ns

    # -- mesh / state
    def set_mesh(self, conn, mat_id, lxy, nnode, thick, planestress, el_begin=0, el_end=None):
        conn = _i32(conn).reshape(-1, 4)
        nel = len(conn)
        mat_id = _i32(mat_id)
        lxy = _f64(lxy).reshape(nel, 2)
        if el_end is None:
            el_end = nel
        self._chk(self.lib.plfx_set_mesh(self.h, nel, int(nnode), _dp(conn), _dp(mat_id), _dp(lxy),
                                         C.c_double(thick), int(bool(planestress)), int(el_begin),
                                         int(el_end)))
        self.nel = nel
        self.nel_owned = el_end - el_begin
        self.ndof = 2 * int(nnode)

    def set_mesh_structured(self, NX, NY, dx_col, dy, thick, planestress, mat_col=None, mat_el=None, el_begin=0, el_end=None):
        """Model.mesh's structured grid by description (plfx_set_mesh_structured): the library writes the index arrays"""
        nel = int(NX) * int(NY)
        dx = _f64(dx_col).reshape(int(NX))
        mc = None if mat_col is None else _i32(mat_col).reshape(int(NX))
        me = None if mat_el is None else _i32(mat_el).reshape(nel)
        if el_end is None:
            el_end = nel
        self._chk(self.lib.plfx_set_mesh_structured(self.h, int(NX), int(NY), _dp(mc), _dp(me), _dp(dx), C.c_double(dy),
                                                    C.c_double(thick), int(bool(planestress)), int(el_begin), int(el_end)))
        self.nel = nel
        self.nel_owned = el_end - el_begin
        self.ndof = 2 * (int(NX) + 1) * (int(NY) + 1)

    def set_grid(self, nx, ny):
        self._chk(self.lib.plfx_set_grid(self.h, int(nx), int(ny)))

    def sweep_info(self):
        """(sweeps, element tangents rewritten by them) since the context was created"""
        a, b = C.c_int64(), C.c_int64()
        self._chk(self.lib.plfx_sweep_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sqmr_info(self):
        """solves with an indefinite tangent stiffness that SQMR completed on its own (plfx_indefinite_info counts all of them)"""
        a = C.c_int64()
        self._chk(self.lib.plfx_sqmr_info(self.h, C.byref(a)))
        return a.value

    def svc_info(self):
        """(bit mask of the 6-feature SVC materials on the 16-lanes-per-element kernels, mask of those run one thread per
        element, sweep launches of either form) since the context was created"""
        a, b, r, t = C.c_int(), C.c_int(), C.c_int64(), C.c_int64()
        self._chk(self.lib.plfx_svc_info(self.h, C.byref(a), C.byref(b), C.byref(r), C.byref(t)))
        return a.value, b.value, r.value, t.value

    def reuse_info(self):
        """(assemblies, BC applications, solves) answered from unchanged inputs since the context was created"""
        a, b, s = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.plfx_reuse_info(self.h, C.byref(a), C.byref(b), C.byref(s)))
        return a.value, b.value, s.value

    def predict_info(self):
        """(applied, skipped, rejected): warm-started solves answered by the interpolation of the last two solutions /
        not tried further (alpha < 0.01) / tried and iterated from the plain warm start instead"""
        a, b, r = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.lib.plfx_predict_info(self.h, C.byref(a), C.byref(b), C.byref(r)))
        return a.value, b.value, r.value

    def set_precond(self, kind, omega=0., nu=0):
        self._chk(self.lib.plfx_set_precond(self.h, int(kind), C.c_double(omega), int(nu)))

    def precond_info(self):
        k = C.c_int()
        lv = C.c_int()
        self._chk(self.lib.plfx_precond_info(self.h, C.byref(k), C.byref(lv)))
        return k.value, lv.value

    def precond_apply(self, r):
        """one V-cycle applied to a host vector (tests)"""
        r = _f64(r).reshape(-1)
        z = np.empty_like(r)
        self._chk(self.lib.plfx_precond_apply(self.h, _dp(r), _dp(z)))
        return z

    def precond_bench(self, reps=100):
        """(microseconds per V-cycle, microseconds of it below the fine level) from `reps` back-to-back applications"""
        a, b = C.c_double(), C.c_double()
        self._chk(self.lib.plfx_precond_bench(self.h, int(reps), C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_wh_mode(self, sequential):
        """work-hardening SVC: hardening modulus handed from element to element like the reference (True) or per point"""
        self._chk(self.lib.plfx_set_wh_mode(self.h, 1 if sequential else 0))

    def wh_info(self):
        """(sequential carry in use, sweeps run in that mode, passes they took)"""
        a, b, c_, u = C.c_int(), C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.lib.plfx_wh_info(self.h, C.byref(a), C.byref(b), C.byref(c_), C.byref(u)))
        self.wh_unresolved = u.value
        return bool(a.value), b.value, c_.value

    def wh_carry(self, mat, value=None):
        """the hardening modulus material `mat` holds now (sequential carry); value != None sets it first"""
        g = C.c_double()
        sv = None if value is None else C.byref(C.c_double(float(value)))
        self._chk(self.lib.plfx_wh_carry(self.h, int(mat), sv, C.byref(g)))
        return g.value

    def solve_fallbacks(self):
        """solves completed by Jacobi-PCG after multigrid-PCG broke down or stalled"""
        n = C.c_int64()
        self._chk(self.lib.plfx_solve_fallbacks(self.h, C.byref(n)))
        return n.value

    def indefinite_info(self):
        """dict: solves with an indefinite tangent stiffness, how they were completed (MINRES with the surrogate V-cycle /
        GMRES), surrogate hierarchies built, elements replaced in the last one (plfx_indefinite_info)"""
        v = [C.c_int64() for _ in range(5)]
        self._chk(self.lib.plfx_indefinite_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(('solves', 'by_minres_surrogate', 'by_gmres', 'surrogates_built', 'elements_shifted'),
                        [x.value for x in v]))

    def set_operator(self, kind):
        self._chk(self.lib.plfx_set_operator(self.h, int(kind)))

    def matvec(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (self.ndof,):
            raise ValueError('matvec: x must have shape (ndof,)')
        y = np.empty(self.ndof)
        self._chk(self.lib.plfx_matvec(self.h, _dp(x), _dp(y)))
        return y

    def operator_info(self):
        m = C.c_int()
        lv = C.c_int()
        self._chk(self.lib.plfx_operator_info(self.h, C.byref(m), C.byref(lv)))
        return m.value, lv.value

    def get_bmat(self, e):
        B = np.empty((4, 6, 8))
        self._chk(self.lib.plfx_get_bmat(self.h, int(e), _dp(B)))
        return B

    def get_kel(self, e):
        K = np.empty((8, 8))
        self._chk(self.lib.plfx_get_kel(self.h, int(e), _dp(K)))
        return K

    def state_get(self, which):
        if which in (ST_U, ST_F, ST_DU):
            out = np.empty(self.ndof)
        elif which in (ST_FYN, ST_MAXSTEPS, ST_KHARD):
            out = np.empty(self.nel_owned)
        elif which == ST_ELSTIFF:
            out = np.empty((self.nel_owned, 36))
        else:
            out = np.empty((self.nel_owned, 6))
        self._chk(self.lib.plfx_state_get(self.h, int(which), _dp(out)))
        return out

    def state_set(self, which, arr):
        arr = _f64(arr)
        self._chk(self.lib.plfx_state_set(self.h, int(which), _dp(arr)))

    def state_reset(self):
        self._chk(self.lib.plfx_state_reset(self.h))

    def gather(self, which, idx):
        idx = _i32(idx)
        out = np.empty(len(idx))
        self._chk(self.lib.plfx_gather(self.h, int(which), len(idx), _dp(idx), _dp(out)))
        return out

    # -- assembly / solve
    def assemble(self):
        self._chk(self.lib.plfx_assemble(self.h))

    def get_csr(self):
        """Assembled global matrix as scipy.sparse.csr_matrix."""
        import scipy.sparse as sp
        nnz = C.c_int64()
        self._chk(self.lib.plfx_get_csr(self.h, C.byref(nnz), None, None, None))
        rowptr = np.empty(self.ndof + 1, dtype=np.int32)
        col = np.empty(nnz.value, dtype=np.int32)
        val = np.empty(nnz.value)
        self._chk(self.lib.plfx_get_csr(self.h, C.byref(nnz), _dp(rowptr), _dp(col), _dp(val)))
        return sp.csr_matrix((val, col, rowptr), shape=(self.ndof, self.ndof))

    def apply_bc(self, idx, du_presc, w, fext=None):
        idx = _i32(idx)
        du_presc = _f64(du_presc)
        w = _f64(w)
        fx = None if fext is None else _f64(fext)
        self._chk(self.lib.plfx_apply_bc(self.h, len(idx), _dp(idx), _dp(du_presc), _dp(w), _dp(fx)))

    def set_bc_plan(self, seg_len, idx):
        seg_len = _i32(seg_len)
        idx = _i32(idx)
        self._chk(self.lib.plfx_set_bc_plan(self.h, len(seg_len), _dp(seg_len), _dp(idx)))
        self._plan_nseg = len(seg_len)

    def apply_bc_plan(self, seg_val, fext=None):
        seg_val = _f64(seg_val)
        if len(seg_val) != self._plan_nseg:
            raise ValueError('apply_bc_plan: one value per registered segment expected')
        fx = None if fext is None else _f64(fext)
        bad = C.c_int(-1)
        self._chk(self.lib.plfx_apply_bc_plan(self.h, _dp(seg_val), _dp(fx), C.byref(bad)))
        return bad.value

    def set_bc_sources(self, src, k, fsrc, fk, flen, fidx, fshare):
        src, k, fsrc, fk, flen, fidx = (_i32(a) for a in (src, k, fsrc, fk, flen, fidx))
        fshare = _f64(fshare)
        self._chk(self.lib.plfx_set_bc_sources(self.h, len(src), _dp(src), _dp(k), len(fsrc), _dp(fsrc), _dp(fk),
                                               _dp(flen), _dp(fidx), _dp(fshare)))

    def load_step(self, step):
        """one load step (plfx_load_step); returns the data of finish_step"""
        uu, ff, sums = self._fin
        self._chk(self.lib.plfx_load_step(self.h, C.byref(step), _dp(uu), _dp(ff), _dp(sums)))
        return uu, ff, sums.reshape(3, 6)

    @staticmethod
    def lib_has_mailbox():
        """the pinned host mailbox is on unless PLFX_MAILBOX=0 (deferred end-of-step results need it)"""
        return os.environ.get('PLFX_MAILBOX', '1') != '0'

    def finish_fetch(self, slot):
        """end-of-step data of a load step that ran with step.defer_slot = slot + 1"""
        uu, ff, sums = self._fin
        self._chk(self.lib.plfx_finish_fetch(self.h, int(slot), _dp(uu), _dp(ff), _dp(sums)))
        return uu, ff, sums.reshape(3, 6)

    def set_finish_set(self, idx):
        idx = _i32(idx)
        self._chk(self.lib.plfx_set_finish_set(self.h, len(idx), _dp(idx)))
        self._fin = (np.empty(len(idx)), np.empty(len(idx)), np.empty(18))

    def finish_step(self):
        uu, ff, sums = self._fin
        self._chk(self.lib.plfx_finish_step(self.h, _dp(uu), _dp(ff), _dp(sums)))
        return uu, ff, sums.reshape(3, 6)

    def solve(self, rtol=1e-12, maxit=100000, warm=False):
        it = C.c_int()
        rr = C.c_double()
        rc = self._chk(self.lib.plfx_solve(self.h, C.c_double(rtol), int(maxit), int(bool(warm)),
                                           C.byref(it), C.byref(rr)), soft=True)
        return it.value, rr.value, rc == 0

    def sweep(self, nit):
        ch = C.c_int()
        cv = C.c_int()
        self._chk(self.lib.plfx_sweep(self.h, int(nit), C.byref(ch), C.byref(cv)))
        return bool(ch.value), bool(cv.value)

    def scf_stats(self, sld):
        """returns (count, min, sum) of the calc_scf list entries"""
        sld = _f64(sld).reshape(6)
        s = C.c_double()
        mn = C.c_double()
        cnt = C.c_int64()
        self._chk(self.lib.plfx_scf_stats(self.h, _dp(sld), C.byref(s), None, C.byref(mn), C.byref(cnt),
                                          C.c_double(0.), 0))
        return cnt.value, mn.value, s.value

    def scf_all(self, sld):
        """(count, min, sum, centred sum of squares) of the calc_scf list in one call"""
        sld = _f64(sld).reshape(6)
        s = C.c_double()
        s2 = C.c_double()
        mn = C.c_double()
        cnt = C.c_int64()
        self._chk(self.lib.plfx_scf_all(self.h, _dp(sld), C.byref(cnt), C.byref(mn), C.byref(s), C.byref(s2)))
        return cnt.value, mn.value, s.value, s2.value

    def scf_sumsq(self, mean):
        s2 = C.c_double()
        self._chk(self.lib.plfx_scf_stats(self.h, None, None, C.byref(s2), None, None, C.c_double(mean), 1))
        return s2.value

    def update_state(self):
        self._chk(self.lib.plfx_update_state(self.h))

    def global_sums(self):
        out = np.empty(18)
        self._chk(self.lib.plfx_global_sums(self.h, _dp(out)))
        return out.reshape(3, 6)

    # -- multi-GPU
    def comm_info(self):
        r, n, d = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.plfx_comm_info(self.h, C.byref(r), C.byref(n), C.byref(d)))
        return r.value, n.value, bool(d.value)

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._chk(self.lib.plfx_comm_unique_id(buf))
        return bytes(buf.raw)

    def comm_init(self, uid, rank, nranks):
        buf = C.create_string_buffer(bytes(uid), 128)
        self._chk(self.lib.plfx_comm_init(self.h, buf, int(rank), int(nranks)))

    def set_strip(self, own_col0, own_col1, global_col0, global_nx, coarse_level):
        """this context holds one x-strip of a larger structured grid (plfx_set_strip)"""
        self._chk(self.lib.plfx_set_strip(self.h, int(own_col0), int(own_col1), int(global_col0), int(global_nx),
                                          int(coarse_level)))

    def strip_info(self):
        """(active, halo, coarse_level, levels of the replicated coarse hierarchy, halo refreshes, coarse gathers,
        all-reduces of partial sums, exchanges of the halo elements' stiffness generators)"""
        a, h, l, cl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        nh, nc, npart, ng = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.lib.plfx_strip_info(self.h, C.byref(a), C.byref(h), C.byref(l), C.byref(cl), C.byref(nh),
                                           C.byref(nc), C.byref(npart), C.byref(ng)))
        return bool(a.value), h.value, l.value, cl.value, nh.value, nc.value, npart.value, ng.value

    def comm_selftest(self):
        """send / receive to self + all-reduce on scratch buffers through the RCCL communicator (raises on a mismatch)"""
        self._chk(self.lib.plfx_comm_selftest(self.h))

    def allreduce_host(self, values, op=0):
        """all-reduce <= 32 host doubles over the context's communicator (op 0 sum, 3 min)"""
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._chk(self.lib.plfx_allreduce_host(self.h, a.ctypes.data_as(C.c_void_p), int(a.size), int(op)))
        return a

    _ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)

    def comm_init_callback(self, rank, nranks, fn):
        """host-staged collectives (tests on one GPU, hosts without RCCL): ``fn(array, op)`` all-reduces the NumPy
        array in place (op 0 = sum, 3 = min) across the ranks"""
        def trampoline(user, buf, count, dtype, op):
            try:
                ct = C.c_int32 if dtype == 1 else C.c_double
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ct)), shape=(count,))
                fn(arr, op)
                return 0
            except Exception:  # noqa: BLE001 - reported as an error code across the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._ar_cb = self._ALLREDUCE_FN(trampoline)   # keep alive as long as the context
        self._chk(self.lib.plfx_comm_init_callback(self.h, int(rank), int(nranks), self._ar_cb, None))

    # -- instrumentation
    def timing_enable(self, on=True):
        self._chk(self.lib.plfx_timing_enable(self.h, int(bool(on))))

    def timing_select(self, families=None):
        """time only the given families (T_* constants); None = all"""
        mask = 0xFF if families is None else sum(1 << int(f) for f in set(families))
        self._chk(self.lib.plfx_timing_select(self.h, C.c_uint(mask)))

    def timing_sample(self, every=1):
        """time only every n-th launch of the selected families"""
        self._chk(self.lib.plfx_timing_sample(self.h, int(every)))

    def timing_reset(self):
        self._chk(self.lib.plfx_timing_reset(self.h))

    def timing_get(self, which):
        ms = C.c_double()
        n = C.c_int64()
        self._chk(self.lib.plfx_timing_get(self.h, int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value
