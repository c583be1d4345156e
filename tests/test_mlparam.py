"""SVC parameter files in the reference's wire format (Material.export_MLparam, material.py:2130-2271; read by the
Abaqus UMAT examples/UMAT/ml_umat.f:129-151).  Fixtures tests/golden/mlparam/* were written by the reference's own
export_MLparam (oracle/gen_golden.py:gen_mlparam); tests/golden/mlparam.npz holds the trained parameters and
calc_yf / calc_fgrad of the same materials."""
import filecmp
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'mlparam.npz'))


@pytest.mark.parametrize('tag', ['J2', 'J2dev'])
def test_from_mlparam_reads_reference_file(gold, golden_dir, tag):
    import pylabfea_amd as FE
    m = FE.Material(name='loaded')
    m.from_MLparam('abq_ML-%s_C15_G25' % tag, path=os.path.join(golden_dir, 'mlparam'))
    assert m.ML_yf and m.sdim == 6 and m.Ndof == 6
    assert np.array_equal(m.svc['sv'], gold[tag + '_par_sv'])
    assert np.array_equal(m.svc['dual'], gold[tag + '_par_dual'])
    assert m.svc['intercept'] == float(gold[tag + '_par_intercept'])
    assert m.gam_yf == float(gold[tag + '_par_gamma'])
    assert m.scale_seq == float(gold[tag + '_par_scale_seq']) == m.sy == float(gold[tag + '_par_sy'])
    assert m.dev_only == bool(gold[tag + '_par_dev_only'])
    assert m.C_yf == float(gold[tag + '_C'])
    assert np.allclose(m.CV, gold[tag + '_par_CV'], rtol=1e-15, atol=0.)
    assert abs(m.E - float(gold[tag + '_par_E'])) < 1e-9 * m.E and abs(m.nu - float(gold[tag + '_par_nu'])) < 1e-12


@pytest.mark.parametrize('tag', ['J2', 'J2dev'])
def test_export_mlparam_is_byte_identical(gold, golden_dir, tmp_path, tag):
    """read the reference's file, write it again: same bytes in the CSV, same data description in the JSON"""
    import pylabfea_amd as FE
    src = os.path.join(golden_dir, 'mlparam')
    m = FE.Material(name='ML-%s_C15_G25' % tag)
    m.from_MLparam('abq_ML-%s_C15_G25' % tag, path=src)
    m.export_MLparam('tests/test_mlparam.py', path=str(tmp_path))
    f = 'abq_ML-%s_C15_G25-svm.csv' % tag
    assert filecmp.cmp(os.path.join(src, f), str(tmp_path / f), shallow=False)
    a = json.load(open(os.path.join(src, f.replace('.csv', '_meta.json'))))
    b = json.load(open(str(tmp_path / f.replace('.csv', '_meta.json'))))
    assert a['Data']['Format'] == b['Data']['Format'] and a['Data']['Names'] == b['Data']['Names']
    assert a['Model']['Names'] == b['Model']['Names'] and a['Model']['Parameters'] == b['Model']['Parameters']


def test_mlparam_errors(tmp_path):
    import pylabfea_amd as FE
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=60., sdim=6)
    with pytest.raises(AttributeError):
        m.export_MLparam('x', path=str(tmp_path))   # no ML flow rule (material.py:2166-2167)
    props = np.zeros(160)
    props[0], props[1] = 3, 21                       # 21 features = stress + work hardening + texture descriptors
    np.savetxt(str(tmp_path / 'wh-svm.csv'), props.reshape(20, 8), delimiter=', ')
    with pytest.raises(NotImplementedError):
        FE.Material().from_MLparam('wh', path=str(tmp_path))
    props[0] = 500                                   # more vectors than the file holds
    props[1] = 6
    np.savetxt(str(tmp_path / 'bad-svm.csv'), props.reshape(20, 8), delimiter=', ')
    with pytest.raises(ValueError):
        FE.Material().from_MLparam('bad', path=str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['J2', 'J2dev'])
def test_loaded_material_on_gpu(gold, golden_dir, tag):
    """yield function and gradient of the loaded material against the reference's values"""
    import pylabfea_amd as FE
    m = FE.Material(name='loaded')
    m.from_MLparam('abq_ML-%s_C15_G25' % tag, path=os.path.join(golden_dir, 'mlparam'))
    sig = gold[tag + '_sig']
    yf = m.calc_yf(sig)
    assert np.max(np.abs(yf - gold[tag + '_yf'])) < 1e-10 * np.max(np.abs(gold[tag + '_yf']))
    a = m.calc_fgrad(sig)
    assert np.max(np.abs(a - gold[tag + '_fgrad'])) < 1e-10 * np.max(np.abs(gold[tag + '_fgrad']))


# ---------------------------------------------------------------------------------------------------------------------
# parameter files SHIPPED with the reference (examples/UMAT/models/*.csv, written by pyLabFEA 4.3: "v4.0 layout" -- slot 16
# holds Nset instead of the deviatoric-feature flag).  Data files, copied unchanged to tests/golden/mlparam/legacy_v4.3/.
LEGACY = {'J2': dict(nsv=255, sy=60., E=200000., uniaxial=60.), 'Goss-Barlat': dict(nsv=235, sy=46.78191221811742, E=151220., uniaxial=None)}


def legacy_decision(path, sig):
    """decision function straight from the file's numbers (UMAT rule ml_umat.f:415-440, deviatoric features)"""
    p = np.loadtxt(path, delimiter=',').ravel()
    nsv = int(p[0])
    dual, sv = p[29:29 + nsv], p[29 + nsv:29 + 7 * nsv].reshape(nsv, 6)
    s = np.array(sig, dtype=float)
    s[:, :3] -= s[:, :3].mean(axis=1)[:, None]
    x = s / p[8]
    d2 = ((x[:, None, :] - sv[None, :, :]) ** 2).sum(axis=2)
    return (dual[None, :] * np.exp(-p[6] * d2)).sum(axis=1) + p[5]


@pytest.mark.parametrize('tag', sorted(LEGACY))
def test_from_mlparam_reads_shipped_v40_files(golden_dir, tag):
    import pylabfea_amd as FE
    src = os.path.join(golden_dir, 'mlparam', 'legacy_v4.3')
    m = FE.Material(name='shipped').from_MLparam('abq_ML-%s_C15_G25' % tag, path=src)
    L = LEGACY[tag]
    assert m.mlparam_layout == 'v4.0' and m.ML_yf and m.sdim == 6 and m.Ndof == 6
    assert len(m.svc['dual']) == L['nsv'] and abs(np.sum(m.svc['dual'])) < 1e-9      # SVC dual constraint: sum = 0
    assert m.dev_only                                   # recovered from the trace-free support vectors
    assert abs(m.sy - L['sy']) < 1e-12 * L['sy'] and m.gam_yf == 2.5 and m.C_yf == 15.0
    assert abs(m.E - L['E']) < 1e-6 * L['E'] and abs(m.nu - 0.3) < 1e-9
    # a current-layout file (slot 16 = flag) is still recognised as such
    cur = FE.Material().from_MLparam('abq_ML-J2dev_C15_G25', path=os.path.join(golden_dir, 'mlparam'))
    assert cur.mlparam_layout == 'v4.4' and cur.dev_only


@pytest.mark.gpu
@pytest.mark.parametrize('tag', sorted(LEGACY))
def test_shipped_v40_material_on_gpu(golden_dir, tag):
    """yield function of the loaded shipped material on the device against the file's own numbers; the J2-trained one
    yields at its nominal 60 MPa in uniaxial tension and is pressure independent"""
    import pylabfea_amd as FE
    src = os.path.join(golden_dir, 'mlparam', 'legacy_v4.3')
    m = FE.Material(name='shipped').from_MLparam('abq_ML-%s_C15_G25' % tag, path=src)
    rng = np.random.default_rng(4)
    sig = rng.normal(size=(300, 6)) * m.sy * rng.uniform(0.2, 1.4, size=(300, 1))
    ref = legacy_decision(os.path.join(src, 'abq_ML-%s_C15_G25-svm.csv' % tag), sig)
    yf = m.calc_yf(sig)
    assert np.max(np.abs(yf - ref)) < 1e-10 * np.max(np.abs(ref))
    if LEGACY[tag]['uniaxial']:
        s = np.zeros((2, 6))
        s[0, 0], s[1, 0] = 0.98 * 60., 1.02 * 60.
        f = m.calc_yf(s)
        assert f[0] < 0. < f[1]
        assert abs(m.calc_yf(s + np.array([500., 500., 500., 0., 0., 0.]))[0] - f[0]) < 1e-9


def test_workhardening_parameter_file_round_trip(golden_dir, tmp_path):
    """15-feature SVC (6 stress + 9 work-hardening features): export_MLparam -> from_MLparam keeps every
    number (slot 9 = scale_wh, slot 7 = epc, 15 numbers per support vector)"""
    import pylabfea_amd as FE
    z = np.load(os.path.join(golden_dir, 'svc_workhard.npz'))
    m = FE.Material(name='ML-wh')
    m.elasticity(CV=z['par_CV'])
    m.plasticity(sy=float(z['par_sy']), sdim=6)
    m.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']),
              scale_wh=float(z['par_scale_wh']), C=2.0)
    m.export_MLparam('tests/test_mlparam.py', path=str(tmp_path))
    r = FE.Material(name='loaded').from_MLparam('abq_ML-wh', path=str(tmp_path))
    assert r.whdat and r.Ndof == 15 and r.ind_wh == 6 and r.scale_wh == float(z['par_scale_wh'])
    assert np.array_equal(r.svc['sv'], z['par_sv']) and np.array_equal(r.svc['dual'], z['par_dual'])
    assert r.mlparam_layout == 'v4.4' and r.C_yf == 2.0
