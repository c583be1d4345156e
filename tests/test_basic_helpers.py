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
