"""Host-side tensor helpers of SURVEY row 16 against values of the reference's basic.py (tests/golden/basic.npz,
oracle/gen_golden.py:gen_basic): the axis-tracking principal stresses are the same LAPACK call + re-ordering rule,
so they must agree bit for bit."""
import os

import numpy as np
import pytest


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'basic.npz'))


def test_sig_princ_axis_tracking(g):
    import pylabfea_amd as FE
    sp, ev = FE.sig_princ(g['sig'])
    assert np.array_equal(sp, g['princ']) and np.array_equal(ev, g['evec'])
    sp1, ev1 = FE.sig_princ(g['sig'][3])
    assert np.array_equal(sp1, g['princ'][3]) and np.array_equal(ev1, g['evec'][3])
    # the reference's documented examples
    assert np.allclose(FE.sig_princ(np.array([1., 5., 3., 0., 0., 0.]))[0], [1., 5., 3.])
    assert np.allclose(FE.sig_princ(np.array([0., 0., 0., 0., 0., 10.]))[0], [10., -10., 0.])
    with pytest.raises(TypeError):
        FE.sig_princ(np.zeros(5))


def test_polar_angle_stress_strain(g):
    import pylabfea_amd as FE
    assert np.max(np.abs(FE.sig_polar_ang(g['sig']) - g['polar'])) < 1e-14
    assert np.max(np.abs(FE.sig_polar_ang(g['princ']) - g['polar_p'])) < 1e-14
    s = FE.Stress(g['sig'][7])
    assert abs(s.seq() - float(g['seq7'])) < 1e-12 and abs(s.h - float(g['h7'])) < 1e-12
    assert np.allclose(s.d, g['d7'], rtol=0, atol=1e-12) and abs(s.theta() - float(g['theta7'])) < 1e-14
    assert abs(s.seq_j2() - float(g['seq7'])) < 1e-12
    e = FE.Strain(g['eps'])
    assert abs(e.eeq() - float(g['eeq'])) < 1e-18 and np.array_equal(e.inv(), g['einv'])


# ---------------------------------------------------------------------------------------------------- general 3-d states
@pytest.fixture(scope='module')
def pg(golden_dir):
    return np.load(os.path.join(golden_dir, 'princ_general.npz'))


def test_sig_princ_general_states_bit_identical(pg):
    """400 stresses incl. out-of-plane shear, nearly diagonal and equal-normal-stress states (oracle/gen_princ_general.py)"""
    import pylabfea_amd as FE
    sp, ev = FE.sig_princ(pg['sig'])
    assert np.array_equal(sp, pg['princ']) and np.array_equal(ev, pg['evec'])


def hill3_material(pg):
    import pylabfea_amd as FE
    E, nu, sy, khard, dr = pg['par']
    m = FE.Material()
    m.elasticity(E=float(E), nu=float(nu))
    m.plasticity(sy=float(sy), hill=list(pg['hill']), khard=float(khard), drucker=float(dr), sdim=3)
    return m


@pytest.mark.gpu
def test_gpu_principal_stress_material_on_general_states(pg):
    """3-parameter Hill on principal stresses (sdim = 3) for stress states WITH out-of-plane shear, where the order of the
    principal stresses -- which the Hill form depends on -- follows LAPACK (basic.py:153-175, material.py:667-670): equal to
    the reference for every state, batch and single calls"""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = hill3_material(pg)
    sig = pg['sig']
    sc = np.max(np.abs(pg['seq']))
    assert np.max(np.abs(m.calc_seq(sig) - pg['seq'])) < 1e-11 * sc
    one = np.array([m.calc_seq(sig[i]) for i in range(0, len(sig), 7)])
    assert np.max(np.abs(one - pg['seq_single'])) < 1e-11 * sc
    assert np.max(np.abs(m.calc_yf(sig, epl=pg['epl']) - pg['yf'])) < 1e-11 * sc
    # had the device ordered these states by its own (plane-state) rule, a third of them would differ by per cent
    import pylabfea_amd as FE
    mt = FE.Material()
    mt.elasticity(E=200.e3, nu=0.3)
    mt.plasticity(sy=100., tresca=True, sdim=3)
    assert np.max(np.abs(mt.calc_seq(sig) - pg['tresca_seq'])) < 1e-10 * np.max(pg['tresca_seq'])   # order-independent form


@pytest.mark.gpu
def test_gpu_two_feature_svc_on_general_states(pg, golden_dir):
    """the 2-feature SVC of sdim = 3 (J2 stress, polar angle of the principal stresses; fixture svc_hill3d.npz) on the same
    general states: features from the reference's create_scaled_input, decision function by its formula"""
    import warnings
    import pylabfea_amd as FE
    z = np.load(os.path.join(golden_dir, 'svc_hill3d.npz'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = FE.Material(name='ML-hill3d')
        m.elasticity(CV=z['par_CV'])
        m.plasticity(sy=float(z['par_sy']), sdim=3)
        m.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
    assert np.max(np.abs(m.create_scaled_input(pg['sig']) - pg['ml3_x'])) < 1e-13
    f = m.calc_yf(pg['sig'])
    assert np.max(np.abs(f - pg['ml3_yf'])) < 1e-9 * max(1., np.max(np.abs(pg['ml3_yf'])))


@pytest.mark.gpu
def test_gpu_calc_fgrad_with_a_voigt_stress_of_a_principal_stress_material(pg):
    """calc_fgrad(sig (6,)) of the analytic sdim = 3 material: equivalent stress in sig_princ's (LAPACK) order over the
    deviator of the VOIGT normals, no shear rows (material.py:834-847) -- not the principal-space normal of epl_dot / C_tan;
    the `seq` argument of the reference replaces the equivalent stress, for sdim = 6 as well"""
    import warnings
    import pylabfea_amd as FE
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = hill3_material(pg)
        m6 = FE.Material()
        m6.elasticity(E=200.e3, nu=0.3)
        m6.plasticity(sy=100., hill=list(pg['hill6']), khard=100., drucker=0.05, sdim=6)
    sig = pg['sig']
    a = np.array([m.calc_fgrad(sig[i]) for i in range(len(sig))])
    assert a.shape == (len(sig), 6) and np.all(a[:, 3:] == 0.)
    assert np.max(np.abs(a - pg['fgrad6'])) < 1e-12
    # the principal-space normal (what the flow rule uses) is something else on states with shear
    ap = m.calc_fgrad(pg['princ'])
    assert ap.shape == (len(sig), 3) and np.max(np.abs(ap[100:] - a[100:, :3])) > 1e-2
    # plane states without shear: both coincide
    k = np.flatnonzero((sig[:, 3] == 0.) & (sig[:, 4] == 0.))
    assert len(k) == 40
    q = pg['fgrad6_seq_in']
    b = np.array([m.calc_fgrad(sig[i], seq=q[i]) for i in range(0, len(sig), 3)])
    assert np.max(np.abs(b - pg['fgrad6_seq'][::3])) < 1e-12
    b6 = np.array([m6.calc_fgrad(sig[i], seq=q[i]) for i in range(0, len(sig), 3)])
    assert np.max(np.abs(b6 - pg['fgrad6_seq_sdim6'][::3])) < 1e-12
    assert np.max(np.abs(m6.calc_fgrad(sig, seq=q) - pg['fgrad6_seq_sdim6'])) < 1e-12   # (N,6) batch, sdim = 6


@pytest.mark.gpu
def test_gpu_ml_full_yf_of_the_two_feature_svc_on_general_states(pg, golden_dir):
    """ML_full_yf of the sdim = 3 SVC on states with out-of-plane shear: the ray search runs on the LAPACK-ordered principal
    state (Material._princ_rows); handing the device the Voigt state as it is gets 44 of these 160 wrong by up to 205 MPa"""
    import warnings
    import pylabfea_amd as FE
    z = np.load(os.path.join(golden_dir, 'svc_hill3d.npz'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = FE.Material(name='ML-hill3d')
        m.elasticity(CV=z['par_CV'])
        m.plasticity(sy=float(z['par_sy']), sdim=3)
        m.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
        s = pg['ml3_full_sig']
        assert np.max(np.abs(m.calc_yf(s) - pg['ml3_full_yf_check'])) < 1e-9
        f = np.array([m.ML_full_yf(s[i], verb=False) for i in range(len(s))])
        fb = m.ML_full_yf(s, verb=False)
    # brentq's last iterate: xtol = 1e-5 (material.py:497-499) on a distance of ~ 100 MPa
    assert np.max(np.abs(f - pg['ml3_full_yf'])) < 2e-5
    assert np.array_equal(f, fb)


@pytest.mark.gpu
def test_gpu_response_on_states_with_out_of_plane_shear(pg):
    """Material.response of a principal-stress (sdim = 3) material re-orders the principal stresses of the
    current stress in every sub-step (material.py:250-340 through basic.py:153-175: np.linalg.eig's order + the row-argmax
    rule).  On states with out-of-plane shear that order is LAPACK's: the device replays dgeev for 3 x 3 symmetric matrices
    (csrc/plfx_lapack3.hpp).  All 160 response rows of the reference fixture, nsteps exact, 1e-9 sy; no warning any more."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = hill3_material(pg)
    CV, s, e, d = pg['r_CV'], pg['r_sig'], pg['r_epl'], pg['r_deps']
    sy = float(pg['par'][2])
    assert np.sum((s[:, 3] != 0.) | (s[:, 4] != 0.)) > 100 and np.sum(pg['r_nsteps'] == 49) > 50
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        for i in range(len(s)):
            f, so, dp, ct = m.response(s[i], e[i], d[i], CV)
            assert m.msg['nsteps'] == pg['r_nsteps'][i], i
            assert np.max(np.abs(so - pg['r_sig_out'][i])) < 1e-9 * sy, i
            assert np.max(np.abs(dp - pg['r_depl'][i])) < 1e-12, i
            assert abs(f - pg['r_fy'][i]) < 1e-9 * sy, i
            assert np.max(np.abs(ct.reshape(36) - pg['r_ct'][i])) < 1e-8 * CV[0, 0], i
    # the batch entry gives the same numbers in one launch
    fy, so, dp, ct, ns = m.response_batch(s, e, d, CV)
    assert np.array_equal(ns, pg['r_nsteps']) and np.max(np.abs(so - pg['r_sig_out'])) < 1e-9 * sy


@pytest.mark.gpu
@pytest.mark.parametrize('maxit', [20, 7])
def test_gpu_response_with_maxit(pg, maxit):
    """Material.response(..., maxit) (material.py:207, 288-291: the sub-division count of an increment whose trial step ends
    outside the yield locus; msg['nsteps'] = maxit - 1 then) against reference vectors for maxit = 20 and 7, sdim = 3 and 6."""
    import warnings
    import pylabfea_amd as FE
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m3 = hill3_material(pg)
        m6 = FE.Material()
        m6.elasticity(E=200.e3, nu=0.3)
        m6.plasticity(sy=100., hill=list(pg['hill6']), khard=100., drucker=0.05, sdim=6)
    CV, s, e, d = pg['r_CV'], pg['r_sig'], pg['r_epl'], pg['r_deps']
    sy = float(pg['par'][2])
    for mat, tag in ((m3, ''), (m6, '6')):
        k = 'rm%d%s_' % (maxit, tag)
        assert np.sum(pg[k + 'nsteps'] == maxit - 1) > 50
        for i in range(0, len(s), 2):
            f, so, dp, ct = mat.response(s[i], e[i], d[i], CV, maxit=maxit)
            assert mat.msg['nsteps'] == pg[k + 'nsteps'][i], (tag, i)
            assert np.max(np.abs(so - pg[k + 'sig_out'][i])) < 1e-9 * sy, (tag, i)
            assert np.max(np.abs(dp - pg[k + 'depl'][i])) < 1e-12, (tag, i)
            assert np.max(np.abs(ct.reshape(36) - pg[k + 'ct'][i])) < 1e-8 * CV[0, 0], (tag, i)
        # ... and the default is untouched by the call before
        f, so, dp, ct = mat.response(s[0], e[0], d[0], CV)
        assert mat.msg['nsteps'] == pg['r_nsteps'][0] if tag == '' else True
    with pytest.raises(ValueError):
        m6.response(s[0], e[0], d[0], CV, maxit=0)


def test_fixture_fgrad_of_a_voigt_stress_is_the_mixed_form(pg):
    """the reference's calc_fgrad(sig (6,)) for sdim = 3 = Hill form on the Voigt normal deviator over 2 seq, seq in the
    LAPACK order (fixture self-consistency: what the new C-ABI entry plfx_fgrad_seq_batch is specified to compute)"""
    sig, seq = pg['sig'], pg['seq']
    h0, h1, h2 = pg['hill']
    d3 = pg['par'][4] / 3.
    sd = sig[:, :3] - sig[:, :3].mean(axis=1)[:, None]
    a = np.stack([((h0 + h2) * sd[:, 0] - h0 * sd[:, 1] - h2 * sd[:, 2]) / (2. * seq) + d3,
                  ((h1 + h0) * sd[:, 1] - h0 * sd[:, 0] - h1 * sd[:, 2]) / (2. * seq) + d3,
                  ((h2 + h1) * sd[:, 2] - h2 * sd[:, 0] - h1 * sd[:, 1]) / (2. * seq) + d3], axis=1)
    assert np.max(np.abs(a - pg['fgrad6'][:, :3])) < 1e-13 and np.all(pg['fgrad6'][:, 3:] == 0.)


@pytest.mark.gpu
def test_gpu_epl_dot_and_c_tan_point_functions(pg):
    """Material.epl_dot / C_tan (material.py:1009-1086) of the sdim = 3 and an sdim = 6 Hill material on 160 general states:
    for sdim = 3 the flow normal is the gradient w.r.t. the PRINCIPAL stresses (one LAPACK-ordered reduction per call), not
    calc_fgrad's form for a Voigt stress"""
    import warnings
    import pylabfea_amd as FE
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = hill3_material(pg)
        m6 = FE.Material()
        m6.elasticity(E=200.e3, nu=0.3)
        m6.plasticity(sy=100., hill=list(pg['hill6']), khard=100., drucker=0.05, sdim=6)
    CV, s, e, d = pg['r_CV'], pg['r_sig'], pg['r_epl'], pg['r_deps']
    for mat, kp, kc in ((m, 'ed_pdot', 'ed_ctan'), (m6, 'ed_pdot6', 'ed_ctan6')):
        npl = 0
        for i in range(len(s)):
            if np.any(np.isnan(pg[kp][i])) or np.any(np.isnan(pg[kc][i])):
                continue          # zero stress: the reference divides 0 / 0 in calc_fgrad
            p = mat.epl_dot(s[i], e[i], CV, d[i])
            assert np.max(np.abs(p - pg[kp][i])) < 1e-13 + 1e-9 * np.max(np.abs(pg[kp][i]))
            npl += bool(np.any(p != 0.))
            if i % 4 == 0 and np.linalg.norm(s[i]) > 1.:
                ct = mat.C_tan(s[i], CV, epl=e[i])
                assert np.max(np.abs(ct.reshape(36) - pg[kc][i])) < 1e-9 * CV[0, 0]
        assert npl > 70
