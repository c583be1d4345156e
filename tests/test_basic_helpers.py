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
