"""Host-side feature helpers of the façade against the reference (fixture tests/golden/scaled_input.npz written by
oracle/gen_golden.py:gen_scaled_input from Material.create_scaled_input, material.py:2301-2346, and the Barlat maps of
Material.plasticity, material.py:2578-2591)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'scaled_input.npz'))


@pytest.mark.parametrize('tag,sdim,dev_only,ndof', [('full6', 6, False, 6), ('dev6', 6, True, 6), ('cyl3', 3, False, 2)])
def test_create_scaled_input(gold, tag, sdim, dev_only, ndof):
    import pylabfea_amd as FE
    m = FE.Material(name='features')
    m.elasticity(E=200000., nu=0.3)
    m.plasticity(sy=60., sdim=sdim)
    m.scale_seq, m.Ndof, m.dev_only = 60., ndof, dev_only
    x = m.create_scaled_input(gold['sig'])
    assert x.shape == gold[tag].shape
    assert np.max(np.abs(x - gold[tag])) < 1e-13
    assert np.max(np.abs(m.create_scaled_input(gold['sig'][5]) - gold[tag + '_single'])) < 1e-13
    if sdim == 3:  # principal stresses as input
        assert np.max(np.abs(m.create_scaled_input(gold['sig'][:, 0:3]) - gold['cyl3_princ'])) < 1e-13
    with pytest.raises(NotImplementedError):
        m.create_scaled_input(gold['sig'], tex=np.zeros(3))


def test_barlat_maps(gold):
    import pylabfea_amd as FE
    mb = FE.Material(name='barlat')
    mb.elasticity(E=151220., nu=0.3)
    mb.plasticity(sy=46.76, barlat=list(gold['barlat_par']), barlat_exp=8, sdim=6)
    assert np.array_equal(mb.Bar_m1, gold['Bar_m1'])
    assert np.array_equal(mb.Bar_m2, gold['Bar_m2'])
    m = FE.Material(name='j2')
    m.elasticity(E=200000., nu=0.3)
    m.plasticity(sy=60., sdim=6)
    with pytest.raises(AttributeError):
        m.calc_seqB(np.ones(6))
