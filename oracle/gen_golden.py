#!/usr/bin/env python3
"""Golden-vector generator — TEST INFRASTRUCTURE, not product code.

Imports the *unmodified* reference (pyLabFEA v4.4.2, /root/reference/src) in the
build container and dumps inputs/outputs of the hot-path functions as small
``.npz`` fixtures under ``tests/golden/``.  The reference never travels to the GPU
box; only these numeric fixtures do.

Run (build container only):

    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src \
        python oracle/gen_golden.py [--only material,element,mesh,solve,svc]

Reference entry points exercised (file:line relative to /root/reference/src/pylabfea):
  material.py:207 response, :348 calc_yf, :414 ML_full_yf, :576 calc_seq,
  :678 calc_seqB, :704 calc_fgrad, :974 get_sflow, :1009 epl_dot, :1057 C_tan,
  :2401 elasticity, :2466 plasticity, :3062 calc_properties
  model.py:262 Element.__init__, :365 calc_Kel, :439 calc_Bmat, :758 mesh,
  :954 setupK, :979 solve, :1473 calc_global
  basic.py:107 sig_princ, :304 sig_dev, :328 eps_eq
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

os.environ.setdefault('MPLBACKEND', 'Agg')
import pylabfea as FE  # noqa: E402  (the reference)

assert FE.__version__ == '4.4.2'


# ----------------------------------------------------------------------------
# material definitions shared by several fixtures
# ----------------------------------------------------------------------------
MATERIALS = {
    # name: (elasticity kwargs, plasticity kwargs)
    'j2': (dict(E=200.e3, nu=0.3), dict(sy=150., khard=500., sdim=6)),
    'j2_k0': (dict(E=200.e3, nu=0.3), dict(sy=150., khard=0., sdim=6)),
    'hill6': (dict(E=200.e3, nu=0.3),
              dict(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)),
    'hill6_dp': (dict(E=200.e3, nu=0.3),
                 dict(sy=120., hill=[1.2, 0.9, 1.1, 1.3, 0.7, 1.0], khard=250., sdim=6,
                      drucker=0.15)),
    'hill6_rv': (dict(E=200.e3, nu=0.3),
                 dict(sy=50., rv=[1.2, 1.0, 0.8, 1.0, 1.0, 1.0], sdim=6)),
    'workhard': (dict(E=300.e3, nu=0.3), dict(sy=150., khard=2000.)),
    'cubic': (dict(C11=170.e3, C12=124.e3, C44=75.e3), dict(sy=80., khard=300., sdim=6,
                                                             hill=[1.1, 0.9, 1.0, 1.2, 1.0, 0.9])),
    # sdim=3: flow rule on principal stresses (tests/test_basic.py:106, :120)
    'hill3': (dict(E=300.e3, nu=0.3), dict(sy=150., hill=[0.7, 1., 1.4], khard=100., sdim=3)),
    'j2s3': (dict(E=300.e3, nu=0.3), dict(sy=150., khard=500., sdim=3)),
}


def make_material(name):
    el, pl = MATERIALS[name]
    m = FE.Material(name=name)
    m.elasticity(**el)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.plasticity(**{k: (list(v) if isinstance(v, list) else v) for k, v in pl.items()})
    return m


def mat_params(m):
    """Flat parameter record consumed by both the oracle and the product tests."""
    d0 = m.lhs if m.lhs is not None else np.ones(3) * m.drucker
    hill = np.ones(6)
    hill[:len(m.hill)] = m.hill
    return dict(CV=np.array(m.CV, dtype=float), E=float(m.E), nu=float(m.nu),
                sy=float(m.sy), khard=float(m.khard), hill=hill,
                dp=np.array(d0, dtype=float), sdim=int(m.sdim),
                hill_6p=bool(m.hill_6p), hill_3p=bool(m.hill_3p))


def element_CV(m, planestress):
    """CV exactly as Model.Element.__init__ builds it (model.py:272-303)."""
    fe = FE.Model(dim=2, planestress=planestress)
    fe.geom([1.], LY=1.)
    fe.assign([m])
    fe.mesh(NX=1, NY=1)
    return np.array(fe.element[0].CV, dtype=float)


# ----------------------------------------------------------------------------
# 1. material-level fixtures
# ----------------------------------------------------------------------------
def rand_unit6(rng, n):
    v = rng.normal(size=(n, 6))
    return v / np.linalg.norm(v, axis=1)[:, None]


def gen_response_inputs(m, CV, rng, n, twod):
    """Seeded inputs covering the four branches of Material.response."""
    sig = np.zeros((n, 6))
    epl = np.zeros((n, 6))
    deps = np.zeros((n, 6))
    for i in range(n):
        kind = i % 8
        d = rng.normal(size=6)
        if twod:
            d[3] = d[4] = 0.
        d /= np.linalg.norm(d)
        # plastic pre-strain (deviatoric-ish), half of the samples
        if i % 2 == 1:
            e = rng.normal(size=6) * 2.e-3
            if twod:
                e[3] = e[4] = 0.
            e[0:3] -= e[0:3].mean()
            epl[i] = e
        sflow = m.get_sflow(epl[i])
        seq_d = m.calc_seq(d)
        if abs(seq_d) < 1e-3:
            d[0] += 1.
            seq_d = m.calc_seq(d)
        if kind == 0:      # deep elastic, small step
            sig[i] = d / seq_d * sflow * rng.uniform(0.0, 0.5)
            deps[i] = rand_unit6(rng, 1)[0] * rng.uniform(1e-6, 2e-4)
        elif kind == 1:    # inside, step crosses surface (split branch)
            sig[i] = d / seq_d * sflow * rng.uniform(0.3, 0.9)
            deps[i] = d * rng.uniform(4e-4, 4e-3)
        elif kind == 2:    # on surface, small outward step (1-step branch)
            sig[i] = d / seq_d * sflow * rng.uniform(0.9995, 1.0005)
            deps[i] = (d + 0.3 * rand_unit6(rng, 1)[0]) * rng.uniform(1e-6, 5e-5)
        elif kind == 3:    # on surface, large step (50 sub-steps + scale-back)
            sig[i] = d / seq_d * sflow * rng.uniform(0.999, 1.003)
            deps[i] = (d + 0.5 * rand_unit6(rng, 1)[0]) * rng.uniform(3e-4, 5e-3)
        elif kind == 4:    # on surface, unloading / neutral
            sig[i] = d / seq_d * sflow * rng.uniform(0.99, 1.0)
            deps[i] = -d * rng.uniform(1e-5, 1e-3) + 0.2 * rand_unit6(rng, 1)[0] * 1e-4
        elif kind == 5:    # slightly inside (fy0 in (-0.15*?,0)) large step
            sig[i] = d / seq_d * (sflow - rng.uniform(0.0, 0.14))
            deps[i] = rand_unit6(rng, 1)[0] * rng.uniform(1e-4, 3e-3)
        elif kind == 6:    # zero initial state, big step (first plastic increment)
            sig[i] = 0.
            epl[i] = 0.
            deps[i] = d * rng.uniform(5e-4, 6e-3)
        else:              # generic random
            sig[i] = d / seq_d * sflow * rng.uniform(0.0, 1.02)
            deps[i] = rand_unit6(rng, 1)[0] * 10 ** rng.uniform(-6, -2.3)
        if twod:
            sig[i, 3] = sig[i, 4] = 0.
            deps[i, 3] = deps[i, 4] = 0.
    return sig, epl, deps


def run_response(m, sig, epl, deps, CV):
    n = len(sig)
    fy = np.zeros(n)
    so = np.zeros((n, 6))
    dp = np.zeros((n, 6))
    ct = np.zeros((n, 36))
    ns = np.zeros(n, dtype=np.int32)
    for i in range(n):
        m.msg['nsteps'] = -1
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            f, s, d, c = m.response(sig[i], epl[i], deps[i], CV)
        fy[i] = f
        so[i] = s
        dp[i] = d
        ct[i] = np.asarray(c).reshape(36)
        ns[i] = m.msg['nsteps']
    return fy, so, dp, ct, ns


def gen_material():
    t0 = time.time()
    for name in MATERIALS:
        m = make_material(name)
        rng = np.random.default_rng(abs(hash(name)) % 1000 if False else sum(map(ord, name)))
        rec = {('par_' + k): v for k, v in mat_params(m).items()}
        # --- batch evaluations on random Voigt stresses
        N = 1000
        sig = rng.normal(size=(N, 6)) * np.array([120., 120., 120., 60., 60., 60.])
        sig[:50, 3:] = 0.           # axis-aligned
        sig[50:60] *= 1.e-3          # tiny
        s3 = (m.sdim == 3)
        if s3:
            sig[:, 3:5] = 0.         # plane states: the axis-tracking order of sig_princ is well defined
        epl = rng.normal(size=(N, 6)) * 3.e-3
        epl[:, 0:3] -= epl[:, 0:3].mean(axis=1)[:, None]
        epl[::3] = 0.
        rec['b_sig'] = sig
        rec['b_epl'] = epl
        rec['b_seq'] = m.calc_seq(sig)
        if s3:
            rec['b_sp'] = FE.sig_princ(sig)[0]
            rec['b_fgrad'] = m.calc_fgrad(rec['b_sp'])      # (N,3): gradient w.r.t. principal stresses
        else:
            rec['b_fgrad'] = m.calc_fgrad(sig)
        rec['b_yf'] = np.array([m.calc_yf(sig[i], epl=epl[i]) for i in range(N)])
        rec['b_sflow'] = np.array([m.get_sflow(epl[i]) for i in range(N)])
        rec['b_peeq'] = FE.eps_eq(epl)
        rec['b_sdev'] = FE.sig_dev(sig)
        # C_tan / epl_dot on a subset
        nsub = 100
        CV = np.array(m.CV)
        ctan = np.zeros((nsub, 36))
        pdot = np.zeros((nsub, 6))
        dd = rng.normal(size=(nsub, 6)) * 1.e-3
        if s3:
            dd[:, 3:5] = 0.
        for i in range(nsub):
            ctan[i] = m.C_tan(sig[i], CV, epl=epl[i]).reshape(36)
            pdot[i] = m.epl_dot(sig[i], epl[i], CV, dd[i])
        rec['b_deps'] = dd
        rec['b_ctan'] = ctan
        rec['b_pdot'] = pdot
        # --- response, plane strain (material CV), plane stress CV, 3-d inputs
        for tag, CVr, twod in (('pe', element_CV(m, False), True),
                               ('ps', element_CV(m, True), True),
                               ('3d', np.array(m.CV), False)):
            if s3 and tag == '3d':
                continue
            n = 240
            s, e, d = gen_response_inputs(m, CVr, rng, n, twod)
            fy, so, dp, ct, ns = run_response(m, s, e, d, CVr)
            rec['r%s_CV' % tag] = CVr
            rec['r%s_sig' % tag] = s
            rec['r%s_epl' % tag] = e
            rec['r%s_deps' % tag] = d
            rec['r%s_fy' % tag] = fy
            rec['r%s_sig_out' % tag] = so
            rec['r%s_depl' % tag] = dp
            rec['r%s_ct' % tag] = ct
            rec['r%s_nsteps' % tag] = ns
        np.savez_compressed(os.path.join(OUT, 'material_%s.npz' % name), **rec)
        nb = {t: np.bincount(np.minimum(rec['r%s_nsteps' % t], 49) // 49)
              for t in ('pe', 'ps', '3d') if 'r%s_nsteps' % t in rec}
        print('material', name, 'done', '%.1fs' % (time.time() - t0), nb)
    # equivalent stresses that have no flow rule in the reference (calc_fgrad raises, material.py:822-825)
    rng = np.random.default_rng(99)
    sig = rng.normal(size=(500, 6)) * np.array([120., 120., 120., 60., 60., 60.])
    sig[:40, 3:] = 0.
    rec = {'sig': sig}
    mt = FE.Material(name='tresca')
    mt.elasticity(E=200.e3, nu=0.3)
    mt.plasticity(sy=100., tresca=True, sdim=6)
    rec['tresca_seq'] = mt.calc_seq(sig)
    bar = [0.81766, -0.36431, 0.86993, 1.07052, 0.85640, 1.22269, 0.47370, 0.49740, 0.52740,
           0.27328, 0.56924, 0.66140, 0.40440, 1.30700, 0.98900, 0.49500, 1.27500, 0.53400]
    mb = FE.Material(name='barlat')
    mb.elasticity(E=151220., nu=0.3)
    mb.plasticity(sy=46.76, barlat=bar, barlat_exp=8, sdim=6)     # settings of examples/train_goss_barlat.py:36-41
    rec['barlat_par'] = np.array(bar)
    rec['barlat_exp'] = np.array(8.)
    rec['barlat_seq'] = mb.calc_seq(sig)
    np.savez_compressed(os.path.join(OUT, 'seq_extra.npz'), **rec)
    print('tresca/barlat done')
    # SURVEY anchor values
    m = make_material('j2')
    f, s, d, c = m.response(np.array([10, 80, 30, 0, 0, 5.]), np.zeros(6),
                            np.array([-2e-4, 8e-4, 0, 0, 0, 2e-4]), m.CV)
    assert abs(f - (-0.17118857254814657)) < 1e-12


# ----------------------------------------------------------------------------
# 2. element fixtures: B matrices, Kel, dense K
# ----------------------------------------------------------------------------
def gen_element():
    rec = {}
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m2 = FE.Material()
    m2.elasticity(C11=170.e3, C12=124.e3, C44=75.e3)
    k = 0
    for mat in (m, m2):
        for ps in (False, True):
            for (lx, ly, lz) in ((1., 1., 1.), (0.25, 0.5, 2.), (4. / 3., 0.7, 1.)):
                fe = FE.Model(dim=2, planestress=ps)
                fe.geom([lx], LY=ly, LZ=lz)
                fe.assign([mat])
                fe.mesh(NX=1, NY=1)
                el = fe.element[0]
                rec['e%d_par' % k] = np.array([lx, ly, lz, float(ps), mat.E, mat.nu])
                rec['e%d_CV' % k] = np.array(el.CV)
                rec['e%d_B' % k] = np.array(el.Bmat)          # (4,6,8)
                rec['e%d_Kel' % k] = np.array(el.Kel)
                rec['e%d_gp' % k] = np.array([el.gpx, el.gpy])
                # Kel with a non-trivial symmetric tangent
                rng = np.random.default_rng(k)
                a = rng.normal(size=(6, 6))
                D = np.array(el.CV) + 1.e3 * (a + a.T)
                el.elstiff = D
                el.calc_Kel()
                rec['e%d_D' % k] = D
                rec['e%d_KelD' % k] = np.array(el.Kel)
                k += 1
    rec['n'] = k
    # dense K + pattern on small meshes
    for (nx, ny, ps) in ((3, 2, False), (8, 8, False), (5, 4, True)):
        fe = FE.Model(dim=2, planestress=ps)
        fe.geom([2., 1.], LY=1.5)
        fe.assign([m, m2])
        fe.mesh(NX=nx, NY=ny)
        K = fe.setupK()
        rec['K_%dx%d' % (nx, ny)] = K
        rec['Kconn_%dx%d' % (nx, ny)] = np.array([el.nodes for el in fe.element], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, 'element.npz'), **rec)
    print('element done')


# ----------------------------------------------------------------------------
# 3. mesh integer products
# ----------------------------------------------------------------------------
def mesh_record(fe):
    mat_index = {id(mm): i for i, mm in enumerate(fe.mat)}
    return dict(
        npos=np.array(fe.npos), conn=np.array([el.nodes for el in fe.element], dtype=np.int64),
        mat_id=np.array([mat_index[id(el.Mat)] for el in fe.element], dtype=np.int64),
        lxy=np.array([[el.Lelx, el.Lely] for el in fe.element]),
        noleft=np.array(fe.noleft, dtype=np.int64), noright=np.array(fe.noright, dtype=np.int64),
        nobot=np.array(fe.nobot, dtype=np.int64), notop=np.array(fe.notop, dtype=np.int64),
        noinner=np.array(fe.noinner, dtype=np.int64),
        dims=np.array([fe.NnodeX, fe.NnodeY, fe.Nnode, fe.Nel, fe.Ndof], dtype=np.int64))


def gen_mesh():
    rec = {}
    ma = FE.Material(num=1)
    ma.elasticity(E=100.e3, nu=0.35)
    mb = FE.Material(num=2)
    mb.elasticity(E=300.e3, nu=0.3)
    cases = []
    for n in (4, 18, 32):
        fe = FE.Model(dim=2)
        fe.geom([4.], LY=4.)
        fe.assign([ma])
        fe.mesh(NX=n, NY=n)
        cases.append(('sq%d' % n, fe))
    fe = FE.Model(dim=2, planestress=True)
    fe.geom([2, 1, 2, 1, 2], LY=4.)
    fe.assign([ma, mb, ma, mb, ma])
    fe.mesh(NX=16, NY=4)
    cases.append(('lam16x4', fe))
    fe = FE.Model(dim=2)
    fe.geom([2, 1, 2, 1, 2], LY=4.)
    fe.assign([ma, mb, ma, mb, ma])
    fe.mesh(NX=13, NY=3)          # non-proportional: largest section absorbs the remainder
    cases.append(('lam13x3', fe))
    fe = FE.Model(dim=2)
    fe.geom([2., 2.], LY=4.)
    fe.assign([ma, mb])
    fe.mesh(NX=4, NY=4)
    cases.append(('lam4x4', fe))
    NX = NY = 18
    el = np.ones((NX, NY))
    el[6:12, 6:12] = 2
    fe = FE.Model(dim=2)
    fe.geom(sect=2, LX=4., LY=4.)
    fe.assign([ma, mb])
    fe.mesh(elmts=el, NX=NX, NY=NY)
    cases.append(('incl18', fe))
    rec['inclusion_elmts'] = el
    for name, fe in cases:
        for k, v in mesh_record(fe).items():
            rec['%s_%s' % (name, k)] = v
    rec['names'] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, 'mesh.npz'), **rec)
    print('mesh done')


# ----------------------------------------------------------------------------
# 4. model-level traces of Model.solve
# ----------------------------------------------------------------------------
class ResponseRecorder(object):
    """Wraps Material.response to log the number of calls per solve."""

    def __init__(self, mat):
        self.mat = mat
        self.n = 0
        self.orig = mat.response

        def wrapped(sig, epl, deps, CV, maxit=50):
            self.n += 1
            return self.orig(sig, epl, deps, CV, maxit)
        mat.response = wrapped


class SolveTracer(object):
    """Captures every linear solve of Model.solve by patching numpy.linalg.solve and reading the
    locals of the calling frame (calc_BC's ``ind`` list, the right-hand side, the BC increments and
    the load-step / iteration counters live there, model.py:1290-1335)."""

    def __init__(self):
        self.rows = []
        self.ind = None
        self.du = []

    def __enter__(self):
        self._orig = np.linalg.solve

        def traced(a, b):
            x = self._orig(a, b)
            fr = sys._getframe(1)
            if fr.f_code.co_name == 'solve' and 'ind' in fr.f_locals:
                L = fr.f_locals
                if self.ind is None:
                    self.ind = np.array(L['ind'], dtype=np.int64)
                dbcr, dbct = L.get('dbcr'), L.get('dbct')
                self.rows.append([L.get('il', -1), L.get('nit', -1), dbcr[0], dbcr[1], dbct[0], dbct[1],
                                  float(np.linalg.norm(b)), float(np.sum(b)), float(np.linalg.norm(x)),
                                  float(L.get('scale_bc', np.nan)) if 'scale_bc' in L else np.nan])
            return x
        np.linalg.solve = traced
        return self

    def __exit__(self, *a):
        np.linalg.solve = self._orig

    def store(self, rec, prefix):
        rec[prefix + '_trace'] = np.array(self.rows, dtype=float)
        rec[prefix + '_ind'] = self.ind


def solve_record(fe, prefix, rec, tsolve=None):
    rec[prefix + '_u'] = np.array(fe.u)
    rec[prefix + '_f'] = np.array(fe.f)
    rec[prefix + '_sgl'] = np.array(fe.sgl)
    rec[prefix + '_egl'] = np.array(fe.egl)
    rec[prefix + '_epgl'] = np.array(fe.epgl)
    rec[prefix + '_sig'] = np.array([el.sig for el in fe.element])
    rec[prefix + '_eps'] = np.array([el.eps for el in fe.element])
    rec[prefix + '_epl'] = np.array([el.epl for el in fe.element])
    rec[prefix + '_elstiff'] = np.array([np.asarray(el.elstiff).reshape(36) for el in fe.element])
    rec[prefix + '_nsteps'] = np.array(fe.nsteps)
    rec[prefix + '_niter'] = np.array(fe.niter, dtype=np.int64)
    rec[prefix + '_co_nconv'] = np.array(fe.co_nconv, dtype=np.int64)
    g = fe.glob
    rec[prefix + '_globbc'] = np.array([g['ebc1'], g['ebc2'], g['sbc1'], g['sbc2'],
                                        g['ebc12'], g['sbc12'], g['ebc21'], g['sbc21']], dtype=float)
    rec[prefix + '_glob'] = np.array([g['sig'], g['eps'], g['epl']])
    if tsolve is not None:
        rec[prefix + '_tsolve'] = np.array(tsolve)


def tension_model(mat, n, eps, planestress=False, L=4.):
    fe = FE.Model(dim=2, planestress=planestress)
    fe.geom([L], LY=L)
    fe.assign([mat])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(eps * fe.leny, 'disp')
    fe.mesh(NX=n, NY=n)
    return fe


def inclusion_elmts(n):
    el = np.ones((n, n))
    a, b = int(n / 3), 2 * int(n / 3)
    el[a:b, a:b] = 2
    return el


def gen_solve():
    rec = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        # config 1: 32x32 elastic (BASELINE.json configs[0])
        m = FE.Material()
        m.elasticity(E=200.e3, nu=0.3)
        fe = tension_model(m, 32, 0.001)
        t = time.time()
        with SolveTracer() as tr:
            fe.solve()
        solve_record(fe, 'el32', rec, time.time() - t)
        tr.store(rec, 'el32')
        print('el32', rec['el32_tsolve'], fe.glob['sig'][1])

        # 8x8 and 12x12 J2 / Hill uniaxial tension (homogeneous)
        for name, n, eps, ms in (('j2_8', 8, 0.002, None), ('hill6_8', 8, 0.002, None),
                                 ('hill6_12', 12, 0.003, 8)):
            mat = make_material(name.split('_')[0])
            rr = ResponseRecorder(mat)
            fe = tension_model(mat, n, eps)
            t = time.time()
            with SolveTracer() as tr:
                fe.solve(min_step=ms)
            dt = time.time() - t
            solve_record(fe, name, rec, dt)
            tr.store(rec, name)
            rec[name + '_ncalls'] = np.array(rr.n)
            print(name, '%.1fs' % dt, fe.nsteps, fe.niter, rr.n, fe.sgl[-1][1])

        # soft elastic inclusion in plastic matrix (heterogeneous, exercises divergence)
        for name, n, eps, ms in (('incl_j2_9', 9, 0.002, None), ('incl_hill6_12', 12, 0.0015, 6)):
            mat = make_material(name.split('_')[1])
            soft = FE.Material(num=2)
            soft.elasticity(E=1.e3, nu=0.27)
            fe = FE.Model(dim=2, planestress=False)
            fe.geom(sect=2, LX=4., LY=4.)
            fe.assign([mat, soft])
            fe.bcleft(0.)
            fe.bcbot(0.)
            fe.bcright(0., 'force')
            fe.bctop(eps * fe.leny, 'disp')
            fe.mesh(elmts=inclusion_elmts(n), NX=n, NY=n)
            rr = ResponseRecorder(mat)
            t = time.time()
            with SolveTracer() as tr:
                fe.solve(min_step=ms)
            dt = time.time() - t
            solve_record(fe, name, rec, dt)
            tr.store(rec, name)
            rec[name + '_ncalls'] = np.array(rr.n)
            print(name, '%.1fs' % dt, fe.nsteps, fe.niter, rr.n)

        # tests/test_basic.py: laminate elastic 16x4 plane stress (test_model a)
        mat1 = FE.Material()
        mat1.elasticity(E=100.e3, nu=0.35)
        mat2 = FE.Material()
        mat2.elasticity(E=300.e3, nu=0.3)
        fe = FE.Model(dim=2, planestress=True)
        fe.geom([2, 1, 2, 1, 2], LY=4.)
        fe.assign([mat1, mat2, mat1, mat2, mat1])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.1 * fe.leny, 'disp')
        fe.mesh(NX=16, NY=4)
        with SolveTracer() as tr:
            fe.solve()
        solve_record(fe, 'lam16x4', rec)
        tr.store(rec, 'lam16x4')

        # tests/test_basic.py:test_bcnode 18x18 inclusion, free sides, corner node fixed
        NX = NY = 18
        el = inclusion_elmts(18)
        ma = FE.Material(num=1)
        ma.elasticity(E=100.e3, nu=0.27)
        mb = FE.Material(num=2)
        mb.elasticity(E=3.e3, nu=0.3)
        fe = FE.Model(dim=2, planestress=False)
        fe.geom(sect=2, LX=4., LY=4.)
        fe.assign([ma, mb])
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bcleft(0., 'force')
        fe.bctop(0.01 * fe.leny, 'disp')
        fe.mesh(elmts=el, NX=NX, NY=NY)
        hh = [no in fe.nobot for no in fe.noleft]
        noc = np.nonzero(hh)[0]
        fe.bcnode(noc, 0., 'disp', 'x')
        with SolveTracer() as tr:
            fe.solve()
        solve_record(fe, 'bcnode18', rec)
        tr.store(rec, 'bcnode18')
        rec['bcnode18_noc'] = np.array(noc)

        # calc_properties harness (2x2 plane stress, 4 load cases) for the sdim=6 materials
        for name, eps, ms in (('workhard', 0.1, None), ('hill6', 0.05, None), ('j2', 0.01, 5)):
            mat = make_material(name)
            t = time.time()
            mat.calc_properties(eps=eps, sigeps=True, min_step=ms)
            for lc in ('stx', 'sty', 'et2', 'ect'):
                p = 'prop_%s_%s' % (name, lc)
                rec[p + '_sig'] = np.array(mat.sigeps[lc]['sig'])
                rec[p + '_eps'] = np.array(mat.sigeps[lc]['eps'])
                rec[p + '_epl'] = np.array(mat.sigeps[lc]['epl'])
                rec[p + '_ys'] = np.array([mat.prop[lc]['ys'], mat.propJ2[lc]['ys']])
                rec[p + '_seq'] = np.array(mat.prop[lc]['seq'])
                rec[p + '_seqJ2'] = np.array(mat.propJ2[lc]['seq'])
                rec[p + '_peeq'] = np.array(mat.propJ2[lc]['peeq'])
            rec['prop_%s_args' % name] = np.array([eps, -1 if ms is None else ms], dtype=float)
            print('calc_properties', name, '%.1fs' % (time.time() - t))

        # tests/test_basic.py:test_model (b): 4x4 laminate elastic + J2 with sdim=3, plane strain
        mat1 = FE.Material()
        mat1.elasticity(E=100.e3, nu=0.35)
        mat2 = FE.Material()
        mat2.elasticity(E=300.e3, nu=0.3)
        mat2.plasticity(sy=150., khard=500., sdim=3)
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([2, 2], LY=4.)
        fe.assign([mat1, mat2])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.1 * fe.leny, 'disp')
        fe.mesh(NX=4, NY=4)
        t = time.time()
        fe.solve()
        fe.calc_global()
        solve_record(fe, 'lam4x4_j2s3', rec, time.time() - t)
        print('lam4x4_j2s3', fe.glob['epl'][1], fe.nsteps, fe.niter)
        # tests/test_basic.py:test_plasticity: Hill-3p, sdim=3, calc_properties(eps=0.05)
        mat = make_material('hill3')
        mat.calc_properties(eps=0.05, sigeps=True)
        for lc in ('stx', 'sty', 'et2', 'ect'):
            p = 'prop_hill3_%s' % lc
            rec[p + '_sig'] = np.array(mat.sigeps[lc]['sig'])
            rec[p + '_eps'] = np.array(mat.sigeps[lc]['eps'])
            rec[p + '_epl'] = np.array(mat.sigeps[lc]['epl'])
            rec[p + '_ys'] = np.array([mat.prop[lc]['ys'], mat.propJ2[lc]['ys']])
            rec[p + '_seq'] = np.array(mat.prop[lc]['seq'])
            rec[p + '_seqJ2'] = np.array(mat.propJ2[lc]['seq'])
            rec[p + '_peeq'] = np.array(mat.propJ2[lc]['peeq'])
        rec['prop_hill3_args'] = np.array([0.05, -1.])
        print('calc_properties hill3', mat.propJ2['stx']['ys'], mat.propJ2['sty']['seq'][-1])

        # re-entrant solve (checkpoint/resume semantics): two successive loads
        mat = make_material('j2')
        fe = tension_model(mat, 6, 0.0012)
        fe.solve()
        fe.bctop(0.002 * fe.leny, 'disp')
        fe.solve()
        solve_record(fe, 'resume_j2_6', rec)

    np.savez_compressed(os.path.join(OUT, 'solve.npz'), **rec)
    print('solve done')


# ----------------------------------------------------------------------------
# 5. SVC yield function fixtures
# ----------------------------------------------------------------------------
def svc_params(mat):
    svm = mat.svm_yf
    return dict(sv=np.array(svm.support_vectors_), dual=np.array(svm.dual_coef_[0, :]),
                intercept=np.array(svm.intercept_[0]), gamma=np.array(mat.gam_yf, dtype=float),
                scale_seq=np.array(mat.scale_seq, dtype=float),
                dev_only=np.array(bool(mat.dev_only)), CV=np.array(mat.CV), E=np.array(mat.E),
                nu=np.array(mat.nu), sy=np.array(mat.sy), khard=np.array(float(mat.khard)),
                hill=np.array(mat.hill), sdim=np.array(mat.sdim), Ndof=np.array(mat.Ndof))


def gen_svc3(name, ml, rec):
    """sdim=3 ML material: features (seq/scale - 1, polar angle/pi); plane stress states only."""
    rng = np.random.default_rng(11)
    N = 300
    sy = ml.sy
    sig = np.zeros((N, 6))
    sig[:, [0, 1, 2, 5]] = rng.normal(size=(N, 4))
    sig[:60, 5] = 0.
    sig /= np.linalg.norm(sig, axis=1)[:, None]
    sig *= (sy * rng.uniform(0.2, 1.5, size=N))[:, None]
    rec['b_sig'] = sig
    rec['b_yf'] = ml.calc_yf(sig)
    rec['b_seq'] = ml.calc_seq(sig)
    sp = FE.sig_princ(sig)[0]
    rec['b_sp'] = sp
    rec['b_fgrad'] = ml.calc_fgrad(sp)               # (N,3)
    nf = 100
    rec['b_full_yf'] = np.array([ml.ML_full_yf(sig[i], verb=False) for i in range(nf)])
    for tag, ps in (('pe', False), ('ps', True)):
        CVr = element_CV(ml, ps)
        n = 64
        s, e, d = gen_response_inputs(ml, CVr, rng, n, True)
        e[:] = 0.
        fy, so, dp, ct, ns = run_response(ml, s, e, d, CVr)
        print('svc3', name, tag, np.bincount(ns))
        rec['r%s_CV' % tag] = CVr
        rec['r%s_sig' % tag] = s
        rec['r%s_epl' % tag] = e
        rec['r%s_deps' % tag] = d
        rec['r%s_fy' % tag] = fy
        rec['r%s_sig_out' % tag] = so
        rec['r%s_depl' % tag] = dp
        rec['r%s_ct' % tag] = ct
        rec['r%s_nsteps' % tag] = ns
    t = time.time()
    ml.calc_properties(eps=0.01, sigeps=True, min_step=12)
    for lc in ('stx', 'sty', 'et2', 'ect'):
        rec['prop_%s_sig' % lc] = np.array(ml.sigeps[lc]['sig'])
        rec['prop_%s_epl' % lc] = np.array(ml.sigeps[lc]['epl'])
        rec['prop_%s_ys' % lc] = np.array([ml.prop[lc]['ys'], ml.propJ2[lc]['ys']])
    print('svc3 calc_properties %.1fs' % (time.time() - t), ml.propJ2['stx']['ys'], ml.propJ2['sty']['seq'][-1],
          ml.propJ2['ect']['peeq'][-1])
    np.savez_compressed(os.path.join(OUT, 'svc_%s.npz' % name), **rec)


def gen_svc():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cases = {}
        # config 4 material: examples/train_hill.py:25-49
        mat_h = FE.Material(name='Hill-reference', num=1)
        mat_h.elasticity(E=200.e3, nu=0.3)
        mat_h.plasticity(sy=50., rv=[1.2, 1.0, 0.8, 1.0, 1.0, 1.0], sdim=6)
        ml = FE.Material('ML-Hill-p1', num=2)
        t = time.time()
        ml.train_SVC(C=2.0, gamma=1.0, mat_ref=mat_h, Nseq=25, Nlc=300, Fe=0.1, Ce=0.99,
                     gridsearch=False)
        print('train_hill SVC: %d SVs, %.1fs' % (len(ml.svm_yf.support_vectors_), time.time() - t))
        cases['hill'] = (ml, mat_h)
        # tests/test_ml.py:37-52 (test_ml_shear) material
        mat_s = FE.Material(name='Hill-shear')
        mat_s.elasticity(E=200.e3, nu=0.3)
        mat_s.plasticity(sy=150., hill=[1.4, 1., 0.7, 1.2, .8, 1.], sdim=6)
        mls = FE.Material('Hill-ML')
        mls.train_SVC(C=2, gamma=0.5, mat_ref=mat_s, Nseq=4, Nlc=300, Fe=0.7, Ce=0.95)
        mls.dev_only = False
        print('test_ml_shear SVC: %d SVs' % len(mls.svm_yf.support_vectors_))
        cases['shear'] = (mls, mat_s)

        # tests/test_ml.py:71-92 (test_ml_training): SVC trained on a J2 reference, sdim=6
        mat_J2 = FE.Material(name='J2-reference')
        mat_J2.elasticity(E=200000., nu=0.3)
        mat_J2.plasticity(sy=60., sdim=6)
        ml2 = FE.Material('ML-J2_C15_G25')
        ml2.dev_only = False
        ml2.train_SVC(C=15., gamma=2.5, mat_ref=mat_J2, Nlc=150, Nseq=25, Fe=0.1, Ce=0.99)
        print('test_ml_training SVC: %d SVs' % len(ml2.svm_yf.support_vectors_))
        cases['j2train'] = (ml2, mat_J2)
        # tests/test_ml.py:10-27 (test_ml_plasticity): sdim=3 SVC on (seq, polar angle) features
        mat_h3 = FE.Material(name='anisotropic Hill')
        mat_h3.elasticity(E=200.e3, nu=0.3)
        mat_h3.plasticity(sy=150., hill=[0.7, 1., 1.4], drucker=0., khard=0., sdim=3)
        ml3 = FE.Material(name='ML flow rule')
        ml3.elasticity(E=200.e3, nu=0.3)
        ml3.plasticity(sy=150., sdim=3)
        x_train, y_train = ml3.create_sig_data(36, mat_ref=mat_h3, extend=True)
        ml3.setup_yf_SVM_3D(x_train, y_train, C=10, gamma=4., fs=0.3)
        print('test_ml_plasticity SVC (sdim=3): %d SVs' % len(ml3.svm_yf.support_vectors_))
        cases['hill3d'] = (ml3, mat_h3)

        for name, (ml, ref) in cases.items():
            rec = {('par_' + k): v for k, v in svc_params(ml).items()}
            if ml.sdim == 3:
                gen_svc3(name, ml, rec)
                continue
            rng = np.random.default_rng(7 + len(name))
            N = 400
            sy = ml.sy
            u = rand_unit6(rng, N)
            sig = u * (sy * rng.uniform(0.2, 1.6, size=N))[:, None]
            sig[:40, 3:5] = 0.
            rec['b_sig'] = sig
            rec['b_yf'] = ml.calc_yf(sig)
            rec['b_fgrad'] = ml.calc_fgrad(sig)
            rec['b_seq'] = ml.calc_seq(sig)
            nf = 120
            fyf = np.zeros(nf)
            for i in range(nf):
                fyf[i] = ml.ML_full_yf(sig[i], verb=False)
            rec['b_full_yf'] = fyf
            # response through element CVs
            for tag, ps in (('pe', False), ('ps', True)):
                CVr = element_CV(ml, ps)
                n = 64
                s, e, d = gen_response_inputs(ml, CVr, rng, n, True)
                e[:] = 0.
                t = time.time()
                fy, so, dp, ct, ns = run_response(ml, s, e, d, CVr)
                print('svc', name, tag, 'response %.1fs' % (time.time() - t), np.bincount(ns))
                rec['r%s_CV' % tag] = CVr
                rec['r%s_sig' % tag] = s
                rec['r%s_epl' % tag] = e
                rec['r%s_deps' % tag] = d
                rec['r%s_fy' % tag] = fy
                rec['r%s_sig_out' % tag] = so
                rec['r%s_depl' % tag] = dp
                rec['r%s_ct' % tag] = ct
                rec['r%s_nsteps' % tag] = ns
            if name == 'j2train':
                t = time.time()
                ml.calc_properties(verb=False, eps=0.01, sigeps=True)
                for lc in ('stx', 'sty', 'et2', 'ect'):
                    rec['prop_%s_sig' % lc] = np.array(ml.sigeps[lc]['sig'])
                    rec['prop_%s_epl' % lc] = np.array(ml.sigeps[lc]['epl'])
                    rec['prop_%s_ys' % lc] = np.array([ml.prop[lc]['ys'], ml.propJ2[lc]['ys']])
                print('j2train calc_properties %.1fs' % (time.time() - t), ml.propJ2['et2']['ys'],
                      ml.propJ2['ect']['peeq'][-1])
            np.savez_compressed(os.path.join(OUT, 'svc_%s.npz' % name), **rec)

        # tests/test_ml.py:test_ml_shear model (6x3 plane stress simple shear)
        mls = cases['shear'][0]
        fem = FE.Model(dim=2, planestress=True)
        fem.geom([2], LY=2.)
        fem.assign([mls])
        fem.bcbot(0., bctype='disp', bcdir='y')
        fem.bcbot(0., bctype='disp', bcdir='x')
        fem.bcleft(0., bctype='force')
        fem.bcright(0., bctype='force')
        fem.bctop(0.006 * fem.leny, bctype='disp', bcdir='x')
        fem.bctop(0., bctype='disp', bcdir='y')
        fem.mesh(NX=6, NY=3)
        t = time.time()
        fem.solve()
        fem.calc_global()
        rec = {}
        solve_record(fem, 'shear6x3', rec, time.time() - t)
        print('shear6x3 %.1fs' % rec['shear6x3_tsolve'], fem.glob['sig'][5], fem.element[3].epl[5])
        np.savez_compressed(os.path.join(OUT, 'svc_solve.npz'), **rec)
    print('svc done')


# ----------------------------------------------------------------------------
# 6. SVC parameter files in the reference's wire format (Material.export_MLparam, material.py:2130-2271)
# ----------------------------------------------------------------------------
def gen_mlparam():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out = os.path.join(OUT, 'mlparam')
        os.makedirs(out, exist_ok=True)
        rec = {}
        for tag, dev_only in (('J2', False), ('J2dev', True)):
            mat_J2 = FE.Material(name='J2-reference')
            mat_J2.elasticity(E=200000., nu=0.3)
            mat_J2.plasticity(sy=60., sdim=6)
            ml = FE.Material('ML-%s_C15_G25' % tag)
            ml.dev_only = dev_only
            ml.train_SVC(C=15., gamma=2.5, mat_ref=mat_J2, Nlc=100, Nseq=12, Fe=0.1, Ce=0.99)
            ml.export_MLparam('oracle/gen_golden.py', file='abq_' + ml.name, path=out)
            rng = np.random.default_rng(3)
            sig = rand_unit6(rng, 200) * (ml.sy * rng.uniform(0.3, 1.5, size=200))[:, None]
            rec[tag + '_sig'] = sig
            rec[tag + '_yf'] = ml.calc_yf(sig)
            rec[tag + '_fgrad'] = ml.calc_fgrad(sig)
            for k, v in svc_params(ml).items():
                rec[tag + '_par_' + k] = v
            rec[tag + '_C'] = np.array(float(ml.C_yf))
            print('mlparam', tag, len(ml.svm_yf.support_vectors_), 'SVs')
        np.savez_compressed(os.path.join(OUT, 'mlparam.npz'), **rec)
    print('mlparam done')


# ----------------------------------------------------------------------------
# 7. basic.py helpers of SURVEY row 16 (sig_princ axis tracking, polar angle, Stress / Strain)
# ----------------------------------------------------------------------------
def gen_basic():
    rng = np.random.default_rng(0)
    s = rng.normal(size=(500, 6)) * 100
    s[:100, 3:5] = 0
    s[100:150, 3:] = 0
    sp, ev = FE.sig_princ(s)
    e = rng.normal(size=6) * 1e-3
    e[4] = 0
    a = FE.Stress(s[7])
    np.savez_compressed(os.path.join(OUT, 'basic.npz'), sig=s, princ=sp, evec=ev, polar=FE.sig_polar_ang(s),
                        polar_p=FE.sig_polar_ang(sp), eps=e, eeq=FE.Strain(e).eeq(), einv=FE.Strain(e).inv(),
                        seq7=a.seq(), h7=a.h, d7=a.d, theta7=a.theta())
    print('basic done')


# ----------------------------------------------------------------------------
# 8. ML_full_yf with a loading direction (material.py:454-462), the form calc_scf uses (model.py:1049-1053)
# ----------------------------------------------------------------------------
def gen_fullyf_ld():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        mat_h = FE.Material(name='Hill-reference', num=1)
        mat_h.elasticity(E=200.e3, nu=0.3)
        mat_h.plasticity(sy=50., rv=[1.2, 1.0, 0.8, 1.0, 1.0, 1.0], sdim=6)
        ml = FE.Material('ML-Hill-p1', num=2)
        ml.train_SVC(C=2.0, gamma=1.0, mat_ref=mat_h, Nseq=25, Nlc=300, Fe=0.1, Ce=0.99, gridsearch=False)
        z = np.load(os.path.join(OUT, 'svc_hill.npz'))
        assert np.array_equal(ml.svm_yf.support_vectors_, z['par_sv']), 'training is not reproducible'
        rng = np.random.default_rng(21)
        N = 90
        sig = rand_unit6(rng, N) * (ml.sy * rng.uniform(0.05, 0.8, size=N))[:, None]
        sig[:20, 3:5] = 0.
        lds = np.array([[0., 1., 0., 0., 0., 0.], [1., 0., 0., 0., 0., 0.], [0., 1., 0., 0., 0., 1.],
                        [1., -1., 0., 0., 0., 0.], [0., 0., 0., 0., 0., 0.]])
        out = np.zeros((len(lds), N))
        for a, ld in enumerate(lds):
            for i in range(N):
                out[a, i] = ml.ML_full_yf(sig[i], np.zeros(6), ld=np.array(ld), verb=False)
        np.savez_compressed(os.path.join(OUT, 'svc_fullyf_ld.npz'), sig=sig, ld=lds, full_yf=out)
    print('fullyf_ld done')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='material,element,mesh,solve,svc,mlparam,basic,fullyf_ld,scaled_input,configs,wh')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    todo = args.only.split(',')
    if 'material' in todo:
        gen_material()
    if 'element' in todo:
        gen_element()
    if 'mesh' in todo:
        gen_mesh()
    if 'solve' in todo:
        gen_solve()
    if 'svc' in todo:
        gen_svc()
    if 'mlparam' in todo:
        gen_mlparam()
    if 'basic' in todo:
        gen_basic()
    if 'fullyf_ld' in todo:
        gen_fullyf_ld()
    if 'scaled_input' in todo:
        gen_scaled_input()
    if 'configs' in todo:
        gen_configs()
    if 'wh' in todo:
        gen_wh()


# ----------------------------------------------------------------------------
# 9. create_scaled_input (material.py:2301-2346, stress features) and the Barlat maps (material.py:2578-2591)
# ----------------------------------------------------------------------------
def gen_scaled_input():
    rng = np.random.default_rng(11)
    sig = rand_unit6(rng, 64) * rng.uniform(10., 120., size=64)[:, None]
    rec = {'sig': sig}
    for tag, sdim, dev_only, ndof in (('full6', 6, False, 6), ('dev6', 6, True, 6), ('cyl3', 3, False, 2)):
        m = FE.Material(name='features-' + tag)
        m.elasticity(E=200000., nu=0.3)
        m.plasticity(sy=60., sdim=sdim)
        m.scale_seq, m.Ndof, m.dev_only = 60., ndof, dev_only
        rec[tag] = m.create_scaled_input(sig)
        rec[tag + '_single'] = m.create_scaled_input(sig[5])
    rec['cyl3_princ'] = m.create_scaled_input(sig[:, 0:3])
    bar = [0.81766, -0.36431, 0.86993, 1.07052, 0.85640, 1.22269, 0.47370, 0.49740, 0.52740,
           0.27328, 0.56924, 0.66140, 0.40440, 1.30700, 0.98900, 0.49500, 1.27500, 0.53400]
    mb = FE.Material(name='barlat')
    mb.elasticity(E=151220., nu=0.3)
    mb.plasticity(sy=46.76, barlat=bar, barlat_exp=8, sdim=6)
    rec['barlat_par'] = np.array(bar)
    rec['Bar_m1'], rec['Bar_m2'] = mb.Bar_m1, mb.Bar_m2
    rec['seqB'] = np.array([mb.calc_seqB(v) for v in sig])
    np.savez_compressed(os.path.join(OUT, 'scaled_input.npz'), **rec)
    print('scaled_input done')


# ----------------------------------------------------------------------------
# 10. the schedules BASELINE.json's configs are quoted on, on meshes the reference can hold
#     (VERDICT r1 item 1 / 6): config 2 (J2, eps 0.004, min_step 20), config 3 (Hill-6p, eps 0.005, min_step 50),
#     config 4 (train_hill SVC, eps 0.001, min_step 10), config 5 (laminate J2 + Goss-Barlat-trained SVC, eps 0.003,
#     min_step 20).  Homogeneous tension is mesh-size independent, so the full-size GPU runs must reproduce these traces.
# ----------------------------------------------------------------------------
GOSS_BARLAT = [0.81766901, -0.36431565, 0.31238124, 0.84321164, -0.01812166, 0.8320893, 0.35952332,
               0.08127502, 1.29314957, 1.0956107, 0.90916744, 0.27655112, 1.090482, 1.18282173,
               -0.01897814, 0.90539357, 1.88256105, 0.0127306]


def train_goss_barlat():
    """examples/train_goss_barlat.py:36-41, 44-49, 70-83 without the plotting."""
    from scipy.optimize import fsolve

    def find_yloc(x, sig, mat):
        return mat.calc_seq(sig * x[:, None]) - mat.sy
    mat_GB = FE.Material(name='Yld2004-18p_from_Goss')
    mat_GB.elasticity(E=151220., nu=0.3)
    mat_GB.plasticity(sy=46.76, barlat=GOSS_BARLAT[0:18], barlat_exp=8)
    sunit = FE.load_cases(number_3d=100, number_6d=200)
    x1 = fsolve(find_yloc, np.ones(len(sunit)) * mat_GB.sy, args=(sunit, mat_GB), xtol=1.e-5)
    sig = sunit * x1[:, None]
    data_GS = FE.Data(sig, mat_name="Goss-Barlat", wh_data=False)
    ml = FE.Material('ML-Goss-Barlat_C3.0_G1.5', num=1)
    ml.from_data(data_GS.mat_data)
    ml.elasticity(C11=mat_GB.C11, C12=mat_GB.C12, C44=mat_GB.C44)
    ml.train_SVC(C=3.0, gamma=1.5, Ce=0.99, Fe=0.1, Nseq=25, gridsearch=False)
    return ml, mat_GB, sig


def svc6_point_record(name, ml, rec):
    """calc_yf / calc_fgrad / calc_seq / ML_full_yf / response vectors of a 6-feature SVC material (as in gen_svc)."""
    rng = np.random.default_rng(7 + len(name))
    N = 400
    u = rand_unit6(rng, N)
    sig = u * (ml.sy * rng.uniform(0.2, 1.6, size=N))[:, None]
    sig[:40, 3:5] = 0.
    rec['b_sig'] = sig
    rec['b_yf'] = ml.calc_yf(sig)
    rec['b_fgrad'] = ml.calc_fgrad(sig)
    rec['b_seq'] = ml.calc_seq(sig)
    nf = 120
    rec['b_full_yf'] = np.array([ml.ML_full_yf(sig[i], verb=False) for i in range(nf)])
    for tag, ps in (('pe', False), ('ps', True)):
        CVr = element_CV(ml, ps)
        s, e, d = gen_response_inputs(ml, CVr, rng, 64, True)
        e[:] = 0.
        fy, so, dp, ct, ns = run_response(ml, s, e, d, CVr)
        print('svc', name, tag, np.bincount(ns))
        for k, v in (('CV', CVr), ('sig', s), ('epl', e), ('deps', d), ('fy', fy), ('sig_out', so), ('depl', dp),
                     ('ct', ct), ('nsteps', ns)):
            rec['r%s_%s' % (tag, k)] = v


def gen_configs():
    rec = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        # configs 2 and 3 on 8x8
        for name, mname, eps, ms in (('cfg2_j2_8', 'j2', 0.004, 20), ('cfg3_hill6_8', 'hill6', 0.005, 50)):
            mat = make_material(mname)
            rr = ResponseRecorder(mat)
            fe = tension_model(mat, 8, eps)
            t = time.time()
            with SolveTracer() as tr:
                fe.solve(min_step=ms)
            dt = time.time() - t
            solve_record(fe, name, rec, dt)
            tr.store(rec, name)
            rec[name + '_ncalls'] = np.array(rr.n)
            print(name, '%.1fs' % dt, fe.nsteps, fe.niter, rr.n, fe.sgl[-1][1], '%.1f updates/s' % (rr.n / dt))
        # config 4: train_hill SVC on 4x4
        mat_h = FE.Material(name='Hill-reference', num=1)
        mat_h.elasticity(E=200.e3, nu=0.3)
        mat_h.plasticity(sy=50., rv=[1.2, 1.0, 0.8, 1.0, 1.0, 1.0], sdim=6)
        ml = FE.Material('ML-Hill-p1', num=2)
        ml.train_SVC(C=2.0, gamma=1.0, mat_ref=mat_h, Nseq=25, Nlc=300, Fe=0.1, Ce=0.99, gridsearch=False)
        old = np.load(os.path.join(OUT, 'svc_hill.npz'))
        same = (old['par_sv'].shape == ml.svm_yf.support_vectors_.shape
                and np.array_equal(old['par_sv'], ml.svm_yf.support_vectors_)
                and np.array_equal(old['par_dual'], ml.svm_yf.dual_coef_[0, :]))
        print('train_hill SVC retrained: %d SVs, identical to tests/golden/svc_hill.npz: %s'
              % (len(ml.svm_yf.support_vectors_), same))
        rec['cfg4_same_as_svc_hill'] = np.array(same)
        if not same:
            for k, v in svc_params(ml).items():
                rec['cfg4_par_' + k] = v
        rr = ResponseRecorder(ml)
        fe = tension_model(ml, 4, 0.001)
        t = time.time()
        with SolveTracer() as tr:
            fe.solve(min_step=10)
        dt = time.time() - t
        solve_record(fe, 'cfg4_svc_4', rec, dt)
        tr.store(rec, 'cfg4_svc_4')
        rec['cfg4_svc_4_ncalls'] = np.array(rr.n)
        print('cfg4_svc_4 %.1fs' % dt, fe.nsteps, fe.niter, rr.n, fe.sgl[-1][1], '%.2f updates/s' % (rr.n / dt))
        np.savez_compressed(os.path.join(OUT, 'solve_configs.npz'), **rec)

        # config 5 materials: J2 + SVC trained on Barlat Yld2004-18p (Goss texture)
        mlb, mat_GB, sig_train = train_goss_barlat()
        print('Goss-Barlat SVC: %d SVs' % len(mlb.svm_yf.support_vectors_))
        recb = {('par_' + k): v for k, v in svc_params(mlb).items()}
        recb['barlat_par'] = np.array(GOSS_BARLAT)
        recb['yield_stresses_barlat'] = sig_train          # 300 yield points of the Barlat reference material
        recb['yf_at_barlat_yield'] = mlb.calc_yf(sig_train)
        svc6_point_record('gossbarlat', mlb, recb)
        np.savez_compressed(os.path.join(OUT, 'svc_gossbarlat.npz'), **recb)
        matj = make_material('j2')
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([2, 1, 2, 1, 2], LY=8.)
        fe.assign([matj, mlb, matj, mlb, matj])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.003 * fe.leny, 'disp')
        fe.mesh(NX=8, NY=4)
        rr1, rr2 = ResponseRecorder(matj), ResponseRecorder(mlb)
        t = time.time()
        with SolveTracer() as tr:
            fe.solve(min_step=20)
        dt = time.time() - t
        solve_record(fe, 'cfg5_lam_8x4', rec, dt)
        tr.store(rec, 'cfg5_lam_8x4')
        rec['cfg5_lam_8x4_ncalls'] = np.array([rr1.n, rr2.n])
        print('cfg5_lam_8x4 %.1fs' % dt, fe.nsteps, fe.niter, rr1.n, rr2.n, fe.sgl[-1][1])
    np.savez_compressed(os.path.join(OUT, 'solve_configs.npz'), **rec)
    print('configs done')


# ----------------------------------------------------------------------------
# 11. SVC with work-hardening features: examples/train_hardening.py:23-79, 171-206 on a reduced data set
#     (40 load cases, 3e-3 plastic strain increments, Nseq = 8 -> ~1000 support vectors x 15 features)
# ----------------------------------------------------------------------------
def train_hardening(Nlc=40, epl_max=0.03, depl=3.e-3, Nseq=8, khard=1000.0):
    from scipy.optimize import fsolve
    mat_h = FE.Material(name='Hill-reference', num=1)
    mat_h.elasticity(E=200.e3, nu=0.3)
    mat_h.plasticity(sy=50., rv=[1.2, 1.0, 0.8, 1.0, 1.0, 1.0], khard=khard, sdim=6)
    nl3d = int(Nlc / 3)
    sunit = FE.load_cases(number_3d=nl3d, number_6d=Nlc - nl3d)
    x1 = fsolve(mat_h.find_yloc, np.ones(Nlc) * mat_h.sy, args=(sunit,), xtol=1.e-5)
    sig_ideal = sunit * x1[:, None]
    SV = np.linalg.inv(mat_h.CV)
    lc_data = dict()
    for i, st in enumerate(sig_ideal):   # create_data of the example
        epl = np.zeros(6)
        peeq = 0.0
        sig_list, epl_list, etot_list = [], [], []
        seq = FE.sig_eq_j2(st)
        su = st / seq
        ind = np.zeros(6, dtype=int)
        for j, v in enumerate(su):
            ind[j] = 1 if v > 0.0 else (2 if v < 0.0 else 0)
        key = f'Us_A{ind[0]}B{ind[1]}C{ind[2]}D{ind[3]}E{ind[4]}F{ind[5]}_HI{i:03d}_NNNNN_Tx_NN'
        dsig = seq / 5
        for j in range(6):
            sg = su * j * dsig
            sig_list.append(sg)
            epl_list.append(np.array(epl))
            etot_list.append(np.dot(SV, sg))
        while peeq < epl_max:
            peeq = FE.eps_eq(epl) + depl
            sg = su * (seq + peeq * khard)
            epl += mat_h.calc_fgrad(sig=sg, epl=epl) * depl
            sig_list.append(sg)
            epl_list.append(np.array(epl))
            etot_list.append(epl + np.dot(SV, sg))
        sig_, epl_, etot_ = np.array(sig_list), np.array(epl_list), np.array(etot_list)
        lc_data[key] = {"Stress": sig_, "Eq_Stress": FE.sig_eq_j2(sig_), "Strain_Plastic": epl_,
                        "Eq_Strain_Plastic": FE.eps_eq(epl_), "Shifted_Strain_Plastic": None,
                        "Strain_Total": etot_, "Eq_Strain_Total": FE.eps_eq(etot_)}
    dd = FE.Data(lc_data, mat_name='ML_Hill_hardening', epl_start=0.0, epl_crit=0.0, epl_max=epl_max, depl=depl,
                 wh_data=True)
    ml = FE.Material(name='ML_Hill_hardening_C2.0_G1.5', num=2)
    ml.from_data(dd.mat_data)
    ml.train_SVC(C=2.0, gamma=1.5, Ce=0.99, Fe=0.1, Nseq=Nseq, gridsearch=False)
    return ml, mat_h


def gen_wh():
    import contextlib
    import io
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with contextlib.redirect_stdout(io.StringIO()):
            ml, mat_h = train_hardening()
        assert ml.whdat and ml.Ndof == 15 and ml.ind_wh == 6 and ml.sdim == 6
        print('work-hardening SVC: %d SVs x %d features, scale_wh %.5f, scale_seq %.4f'
              % (len(ml.svm_yf.support_vectors_), ml.Ndof, ml.scale_wh, ml.scale_seq))
        rec = {('par_' + k): v for k, v in svc_params(ml).items()}
        rec['par_scale_wh'] = np.array(float(ml.scale_wh))
        rec['par_ind_wh'] = np.array(int(ml.ind_wh))
        rec['par_epc'] = np.array(float(ml.epc))
        rng = np.random.default_rng(21)
        N = 240
        u = rand_unit6(rng, N)
        sig = u * (ml.sy * rng.uniform(0.3, 1.6, size=N))[:, None]
        e = rng.normal(size=(N, 6))
        e[:, :3] -= e[:, :3].mean(axis=1)[:, None]
        e *= (rng.uniform(0., 0.03, size=N) / FE.eps_eq(e))[:, None]
        e[:40] = 0.
        rec['b_sig'], rec['b_epl'] = sig, e
        rec['b_x'] = ml.create_scaled_input(sig, e, 0., 0., 0.)
        rec['b_yf'] = ml.calc_yf(sig, epl=e)
        fg, kh = np.zeros((N, 6)), np.zeros(N)
        for i in range(N):
            fg[i] = ml.calc_fgrad(sig[i], epl=e[i])
            kh[i] = ml.khard
        rec['b_fgrad'], rec['b_khard'] = fg, kh
        ml.khard = 0.
        fgb = ml.calc_fgrad(sig[:50], epl=e[:50])       # batched call: khard = mean over the points, clipped
        rec['b_fgrad_batch50'], rec['b_khard_batch50'] = fgb, np.array(ml.khard)
        nf = 80
        kfix = rng.uniform(0., 900., size=nf)
        fyf = np.zeros(nf)
        for i in range(nf):
            ml.khard = kfix[i]
            fyf[i] = ml.ML_full_yf(sig[i], epl=e[i], verb=False)
        rec['b_full_yf'], rec['b_full_yf_khard'] = fyf, kfix
        for tag, ps in (('pe', False), ('ps', True)):
            CVr = element_CV(ml, ps)
            n = 72
            s, ep, d = gen_response_inputs(ml, CVr, rng, n, True)
            ep = rng.normal(size=(n, 6))
            ep[:, :3] -= ep[:, :3].mean(axis=1)[:, None]
            ep[:, 3:5] = 0.
            ep *= (rng.uniform(0., 0.02, size=n) / FE.eps_eq(ep))[:, None]
            ep[:24] = 0.
            kin = rng.uniform(0., 900., size=n)
            kin[:12] = 0.
            out = [[], [], [], [], [], []]
            for i in range(n):
                ml.khard = kin[i]
                fy, so, dp, ct = ml.response(s[i], ep[i], d[i], CVr)
                for q, v in zip(out, (fy, so, dp, np.asarray(ct).reshape(36), ml.msg['nsteps'], ml.khard)):
                    q.append(v)
            print('wh response', tag, np.bincount(np.array(out[4])))
            rec['r%s_CV' % tag], rec['r%s_sig' % tag], rec['r%s_epl' % tag], rec['r%s_deps' % tag] = CVr, s, ep, d
            rec['r%s_khard_in' % tag] = kin
            for k, q in zip(('fy', 'sig_out', 'depl', 'ct', 'nsteps', 'khard_out'), out):
                rec['r%s_%s' % (tag, k)] = np.array(q)
        # model level: 4x4 plane-strain tension; Material.khard starts at 0 and is carried through the element loop
        ml.khard = 0.
        fe = tension_model(ml, 4, 0.004)
        t = time.time()
        with SolveTracer() as tr:
            fe.solve(min_step=8)
        solve_record(fe, 'wh4', rec, time.time() - t)
        tr.store(rec, 'wh4')
        rec['wh4_khard_final'] = np.array(float(ml.khard))
        print('wh4 %.1fs' % rec['wh4_tsolve'], fe.nsteps, fe.niter, fe.sgl[-1][1], 'khard', ml.khard)
    np.savez_compressed(os.path.join(OUT, 'svc_workhard.npz'), **rec)
    print('wh done')


if __name__ == '__main__':
    main()
