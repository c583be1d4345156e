"""The native Barlat Yld2004-18p normal (EXTENSION: the reference has an equivalent stress for Barlat materials,
material.py:678-702, but no flow rule -- calc_fgrad raises, material.py:822-825; BASELINE.json's north star asks for
"J2, Hill, Barlat and SVC-surrogate yield functions with their normals").  Opt-in per material
(``Material.enable_barlat_normal()`` / ``plfx_material.barlat_normal``); without it the façade raises like the reference.

What pins it, in the absence of a reference implementation:
  * the equivalent stress itself is the reference's (fixtures ``scaled_input.npz:seqB``, ``seq_extra.npz``);
  * the gradient is checked against central finite differences of that equivalent stress, and against Euler's theorem
    (seq is homogeneous of degree one: a . sigma = seq) and volume preservation (a is deviatoric);
  * with all 18 coefficients 1 and exponent 2 Yld2004-18p IS von Mises: equivalent stress, normal, ``response`` and a whole
    ``Model.solve`` must then reproduce the reference's J2 golden vectors / traces;
  * GPU against the oracle (independent C restatement of the same formulas) for the Goss coefficients of
    examples/train_goss_barlat.py:36-41, point level and model level."""
import os

import numpy as np
import pytest

from oracle import oracle as O

GOSS = [0.81766901, -0.36431565, 0.31238124, 0.84321164, -0.01812166, 0.8320893, 0.35952332,
        0.08127502, 1.29314957, 1.0956107, 0.90916744, 0.27655112, 1.090482, 1.18282173,
        -0.01897814, 0.90539357, 1.88256105, 0.0127306]


def oracle_barlat(par, a, sy, khard=0., E=151220., nu=0.3):
    return O.Material(kind=O.BARLAT, E=E, nu=nu, sy=sy, khard=khard, barlat=par, barlat_exp=a)


def rand_stress(n, scale, seed=3):
    rng = np.random.default_rng(seed)
    s = rng.normal(size=(n, 6))
    s /= np.linalg.norm(s, axis=1)[:, None]
    return s * (scale * rng.uniform(0.3, 1.5, size=n))[:, None]


def fd_grad(seq_fn, sig, h):
    g = np.zeros_like(sig)
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        g[:, k] = (seq_fn(sig + d) - seq_fn(sig - d)) / (2. * h)
    return g


# ---------------------------------------------------------------------------------------------------- CPU: the oracle
def test_oracle_seq_is_the_references(golden_dir):
    z = np.load(os.path.join(golden_dir, 'scaled_input.npz'))
    m = oracle_barlat(z['barlat_par'], 8., 46.76)
    assert np.max(np.abs(O.calc_seq(m, z['sig']) - z['seqB'])) < 1e-11 * np.max(z['seqB'])


@pytest.mark.parametrize('par,a', [(GOSS, 8.), (GOSS, 6.), (np.ones(18), 2.), (np.ones(18), 8.)])
def test_oracle_gradient_fd_euler_deviatoric(par, a):
    m = oracle_barlat(par, a, 46.76)
    sig = rand_stress(200, 46.76)
    g = O.calc_fgrad(m, sig)
    fd = fd_grad(lambda s: O.calc_seq(m, s), sig, 1e-4)
    assert np.max(np.abs(g - fd)) < 2e-7 * np.max(np.abs(fd))
    assert np.max(np.abs(np.sum(g * sig, axis=1) - O.calc_seq(m, sig))) < 1e-10 * 46.76     # Euler: a . sigma = seq
    assert np.max(np.abs(g[:, :3].sum(axis=1))) < 1e-12                                      # deviatoric normal


def test_oracle_isotropic_exponent2_is_von_mises(golden_dir):
    """all coefficients 1, a = 2: equivalent stress, normal and response equal the reference's J2 material"""
    z = np.load(os.path.join(golden_dir, 'material_j2.npz'))
    mb = oracle_barlat(np.ones(18), 2., float(z['par_sy']), khard=float(z['par_khard']), E=float(z['par_E']), nu=float(z['par_nu']))
    assert np.max(np.abs(O.calc_seq(mb, z['b_sig']) - z['b_seq'])) < 1e-11 * np.max(z['b_seq'])
    assert np.max(np.abs(O.calc_fgrad(mb, z['b_sig']) - z['b_fgrad'])) < 1e-11
    for tag in ('pe', 'ps', '3d'):
        CV = z['r%s_CV' % tag]
        fy, so, dp, ct, ns = O.response(mb, CV, z['r%s_sig' % tag], z['r%s_epl' % tag], z['r%s_deps' % tag])
        sy = float(z['par_sy'])
        assert np.array_equal(ns, z['r%s_nsteps' % tag])
        assert np.max(np.abs(so - z['r%s_sig_out' % tag])) < 1e-8 * sy
        assert np.max(np.abs(dp - z['r%s_depl' % tag])) < 1e-11
        assert np.max(np.abs(ct - z['r%s_ct' % tag])) < 1e-6 * CV[0, 0]


def test_facade_raises_without_the_extension():
    import pylabfea_amd as FE
    m = FE.Material(name='Yld2004-18p')
    m.elasticity(E=151220., nu=0.3)
    m.plasticity(sy=46.76, barlat=GOSS, barlat_exp=8)
    with pytest.raises(ValueError):
        m.calc_fgrad(np.array([40., 0., 0., 0., 0., 5.]))     # like the reference (material.py:822-825)
    with pytest.raises(AttributeError):
        FE.Material().enable_barlat_normal()


# ---------------------------------------------------------------------------------------------------- GPU
def facade_barlat(par, a, sy, khard=0., E=151220., nu=0.3):
    import pylabfea_amd as FE
    m = FE.Material(name='Yld2004-18p')
    m.elasticity(E=E, nu=nu)
    m.plasticity(sy=sy, khard=khard, barlat=list(par), barlat_exp=a)
    return m.enable_barlat_normal()


@pytest.mark.gpu
@pytest.mark.parametrize('par,a', [(GOSS, 8.), (np.ones(18), 2.)])
def test_gpu_gradient_vs_oracle_and_fd(par, a):
    m = facade_barlat(par, a, 46.76)
    om = oracle_barlat(par, a, 46.76)
    sig = rand_stress(500, 46.76, seed=9)
    g = m.calc_fgrad(sig)
    assert np.max(np.abs(g - O.calc_fgrad(om, sig))) < 1e-10
    fd = fd_grad(lambda s: m.calc_seq(s), sig, 1e-4)
    assert np.max(np.abs(g - fd)) < 2e-7 * np.max(np.abs(fd))
    assert np.max(np.abs(m.calc_seq(sig) - O.calc_seq(om, sig))) < 1e-11 * 46.76


@pytest.mark.gpu
def test_gpu_isotropic_exponent2_is_the_references_j2(golden_dir):
    """Yld2004-18p with unit coefficients and a = 2 against the REFERENCE's J2 vectors and its 8x8 tension trace"""
    from test_gpu_model import check_fields, tension_model
    z = np.load(os.path.join(golden_dir, 'material_j2.npz'))
    sy, kh = float(z['par_sy']), float(z['par_khard'])
    m = facade_barlat(np.ones(18), 2., sy, khard=kh, E=float(z['par_E']), nu=float(z['par_nu']))
    assert np.max(np.abs(m.calc_fgrad(z['b_sig']) - z['b_fgrad'])) < 1e-11
    for tag in ('pe', 'ps', '3d'):
        CV = z['r%s_CV' % tag]
        ns_ref = z['r%s_nsteps' % tag]
        out = [m.response(z['r%s_sig' % tag][i], z['r%s_epl' % tag][i], z['r%s_deps' % tag][i], CV) for i in range(40)]
        so = np.array([o[1] for o in out])
        ct = np.array([o[3] for o in out])
        assert np.max(np.abs(so - z['r%s_sig_out' % tag][:40])) < 1e-8 * sy
        assert np.max(np.abs(ct - z['r%s_ct' % tag][:40].reshape(-1, 6, 6))) < 1e-6 * CV[0, 0]
        assert len(ns_ref) >= 40
    g = np.load(os.path.join(golden_dir, 'solve.npz'))
    fe = tension_model(m, 8, 0.002)
    fe.solve()
    check_fields(fe, g, 'j2_8')


@pytest.mark.gpu
def test_gpu_goss_barlat_response_and_model_vs_oracle():
    """Goss-texture coefficients (examples/train_goss_barlat.py:36-41): batched response against the oracle, consistency
    of the return mapping, and an 16x16 tension test against the oracle's sparse direct solve"""
    import pylabfea_amd as FE
    from pylabfea_amd import _lib
    from oracle.solve_ref import RefSolver
    sy, kh = 46.76, 200.
    m = facade_barlat(GOSS, 8., sy, khard=kh)
    om = oracle_barlat(GOSS, 8., sy, khard=kh)
    CV = np.array(m.CV)
    rng = np.random.default_rng(5)
    n = 400
    sig = rand_stress(n, sy, seed=6)
    sig *= (sy * rng.uniform(0.5, 1.0, size=n) / np.maximum(O.calc_seq(om, sig), 1e-9))[:, None]   # inside / on the locus
    epl = np.zeros((n, 6))
    deps = rng.normal(size=(n, 6)) * 3e-4
    ctx = _lib.Context(0)
    ctx.set_materials([m._record(CV)])
    fy, so, dp, ct, ns = ctx.response(sig, epl, deps)
    fy2, so2, dp2, ct2, ns2 = O.response(om, CV, sig, epl, deps)
    assert np.array_equal(ns, ns2) and len(set(ns.tolist())) > 1          # elastic, one-step and sub-stepped points
    assert np.max(np.abs(so - so2)) < 1e-8 * sy
    assert np.max(np.abs(dp - dp2)) < 1e-10
    assert np.max(np.abs(ct - ct2)) < 1e-6 * CV[0, 0]
    plastic = np.linalg.norm(dp, axis=1) > 0
    yf_end = m.calc_yf(so[plastic], epl=(epl + dp)[plastic])
    assert np.max(yf_end) < 5.1e-3 * (sy + kh * 0.01)                     # back on the (hardened) yield locus

    def build():
        mm = facade_barlat(GOSS, 8., sy, khard=kh)
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([4.], LY=4.)
        fe.assign([mm])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.002 * fe.leny, 'disp')
        fe.mesh(NX=16, NY=16)
        return fe
    fe = build()
    fe.solve(min_step=6)
    ref = RefSolver(build()).solve(min_step=6)
    assert fe.nsteps == ref.nsteps and list(fe.niter) == list(ref.niter)
    assert np.max(fe._state('epl')) > 0.
    s = np.max(np.abs(ref.sig))
    assert np.max(np.abs(fe.u - ref.u)) < 1e-6 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 1e-6 * s
    assert np.max(np.abs(fe.sgl - ref.sgl)) < 1e-6 * s
