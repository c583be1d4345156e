"""CPU oracle (oracle/plfx_oracle.c) pinned against golden vectors dumped from the imported
reference (oracle/gen_golden.py -> tests/golden/*.npz).  Tolerances are relative FP64 round-off."""
import os

import numpy as np
import pytest

from oracle import oracle as O

MATS = ['j2', 'j2_k0', 'hill6', 'hill6_dp', 'hill6_rv', 'workhard', 'cubic']


def rel(a, b, floor=1e-300):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.max(np.abs(a - b) / (np.abs(b) + floor))


@pytest.fixture(scope='module', params=MATS)
def gold(request, golden_dir):
    z = np.load(os.path.join(golden_dir, 'material_%s.npz' % request.param))
    return z, O.Material.from_golden(z)


def test_seq_fgrad_yf(gold):
    z, m = gold
    assert rel(O.calc_seq(m, z['b_sig']), z['b_seq'], 1e-9) < 1e-13
    assert np.max(np.abs(O.calc_fgrad(m, z['b_sig']) - z['b_fgrad'])) < 1e-12
    assert np.max(np.abs(O.calc_yf(m, z['b_sig'], z['b_epl']) - z['b_yf'])) < 1e-10


@pytest.mark.parametrize('tag', ['pe', 'ps', '3d'])
def test_response(gold, tag):
    z, m = gold
    CV = z['r%s_CV' % tag]
    fy, so, dp, ct, ns = O.response(m, CV, z['r%s_sig' % tag], z['r%s_epl' % tag], z['r%s_deps' % tag])
    assert np.array_equal(ns, z['r%s_nsteps' % tag])
    sc = float(m.c.sy)
    assert np.max(np.abs(fy - z['r%s_fy' % tag])) < 1e-8 * sc
    assert np.max(np.abs(so - z['r%s_sig_out' % tag])) < 1e-9 * sc
    assert np.max(np.abs(dp - z['r%s_depl' % tag])) < 1e-12
    assert np.max(np.abs(ct - z['r%s_ct' % tag])) < 1e-7 * CV[0, 0]


def test_element(golden_dir):
    z = np.load(os.path.join(golden_dir, 'element.npz'))
    for k in range(int(z['n'])):
        lx, ly, lz, ps, E, nu = z['e%d_par' % k]
        CV = z['e%d_CV' % k]
        gp = z['e%d_gp' % k]
        for g in range(4):
            B = O.calc_Bmat(lx, ly, gp[0, g], gp[1, g], ps > 0, CV, E, nu)
            assert np.max(np.abs(B - z['e%d_B' % k][g])) < 1e-15
        K = O.calc_Kel(lx, ly, lz, ps > 0, CV, E, nu, CV)
        assert rel(K, z['e%d_Kel' % k], 1e-6) < 1e-12
        K = O.calc_Kel(lx, ly, lz, ps > 0, CV, E, nu, z['e%d_D' % k])
        assert rel(K, z['e%d_KelD' % k], 1e-6) < 1e-12


@pytest.mark.parametrize('name', ['hill', 'shear', 'j2train', 'gossbarlat'])
def test_svc(golden_dir, name):
    z = np.load(os.path.join(golden_dir, 'svc_%s.npz' % name))
    m = O.Material.from_golden(z)
    sig = z['b_sig']
    assert np.max(np.abs(O.calc_yf(m, sig) - z['b_yf'])) < 1e-10
    assert np.max(np.abs(O.calc_fgrad(m, sig) - z['b_fgrad'])) < 1e-12
    nf = len(z['b_full_yf'])
    fyf, st = O.ML_full_yf(m, sig[:nf])
    # brentq stops within xtol=1e-5 (stress units); identical iterates up to exp() round-off
    assert np.max(np.abs(fyf - z['b_full_yf'])) < 1e-7
    for tag in ('pe', 'ps'):
        CV = z['r%s_CV' % tag]
        fy, so, dp, ct, ns = O.response(m, CV, z['r%s_sig' % tag], z['r%s_epl' % tag], z['r%s_deps' % tag])
        assert np.array_equal(ns, z['r%s_nsteps' % tag])
        sc = float(m.c.sy)
        assert np.max(np.abs(fy - z['r%s_fy' % tag])) < 1e-6 * sc
        assert np.max(np.abs(so - z['r%s_sig_out' % tag])) < 1e-6 * sc
        assert np.max(np.abs(dp - z['r%s_depl' % tag])) < 1e-9
        assert np.max(np.abs(ct - z['r%s_ct' % tag])) < 1e-5 * CV[0, 0]


def test_svc_full_yf_with_loading_direction(golden_dir):
    """ML_full_yf(sig, epl, ld=...) (material.py:454-462) as calc_scf calls it (model.py:1049-1053), five directions
    incl. the inconsistent all-zero one (:456-461)"""
    z = np.load(os.path.join(golden_dir, 'svc_hill.npz'))
    g = np.load(os.path.join(golden_dir, 'svc_fullyf_ld.npz'))
    m = O.Material.from_golden(z)
    for a, ld in enumerate(g['ld']):
        out = O.ML_full_yf_ld(m, g['sig'], None, ld)
        assert np.max(np.abs(out - g['full_yf'][a])) < 1e-7, a
    # ld = None path through the same routine
    assert np.max(np.abs(O.ML_full_yf(m, z['b_sig'][:20])[0] - z['b_full_yf'][:20])) < 1e-7


@pytest.mark.parametrize('name', ['hill3', 'j2s3'])
def test_sdim3(golden_dir, name):
    """sdim=3 flow rule (principal stresses in the reference's axis-tracking order) on plane states."""
    z = np.load(os.path.join(golden_dir, 'material_%s.npz' % name))
    m = O.Material.from_golden(z)
    assert m.c.kind == O.PRINC3
    sig = z['b_sig']
    assert np.max(np.abs(O.sig_princ(sig) - z['b_sp'])) < 1e-10
    assert np.max(np.abs(O.calc_seq(m, sig) - z['b_seq'])) < 1e-10
    a = O.calc_fgrad(m, sig)                       # Voigt-embedded: [d/dsp, 0, 0, 0]
    ok = z['b_seq'] > 1e-6
    assert np.max(np.abs(a[ok, :3] - z['b_fgrad'][ok])) < 1e-10 and np.all(a[:, 3:] == 0.)
    assert np.max(np.abs(O.calc_yf(m, sig, z['b_epl']) - z['b_yf'])) < 1e-10
    for tag in ('pe', 'ps'):
        CV = z['r%s_CV' % tag]
        fy, so, dp, ct, ns = O.response(m, CV, z['r%s_sig' % tag], z['r%s_epl' % tag], z['r%s_deps' % tag])
        assert np.array_equal(ns, z['r%s_nsteps' % tag])
        sc = float(m.c.sy)
        assert np.max(np.abs(so - z['r%s_sig_out' % tag])) < 1e-9 * sc
        assert np.max(np.abs(dp - z['r%s_depl' % tag])) < 1e-12
        assert np.max(np.abs(ct - z['r%s_ct' % tag])) < 1e-7 * CV[0, 0]


def test_tresca_barlat(golden_dir):
    z = np.load(os.path.join(golden_dir, 'seq_extra.npz'))
    mt = O.Material(kind=O.TRESCA, sy=100.)
    assert np.max(np.abs(O.calc_seq(mt, z['sig']) - z['tresca_seq'])) < 1e-10
    mb = O.Material(kind=O.BARLAT, sy=46.76, barlat=z['barlat_par'], barlat_exp=float(z['barlat_exp']))
    assert np.max(np.abs(O.calc_seq(mb, z['sig']) - z['barlat_seq']) / z['barlat_seq']) < 1e-11


def test_svc_sdim3(golden_dir):
    """2-feature SVC of sdim=3 ML materials (tests/test_ml.py:test_ml_plasticity settings)."""
    z = np.load(os.path.join(golden_dir, 'svc_hill3d.npz'))
    m = O.Material.from_golden(z)
    assert m.c.kind == O.SVC3 and m.c.ndof == 2
    sig = z['b_sig']
    assert np.max(np.abs(O.calc_yf(m, sig) - z['b_yf'])) < 1e-10
    assert np.max(np.abs(O.calc_seq(m, sig) - z['b_seq'])) < 1e-10
    a = O.calc_fgrad(m, sig)
    assert np.max(np.abs(a[:, :3] - z['b_fgrad'])) < 1e-10 and np.all(a[:, 3:] == 0.)
    nf = len(z['b_full_yf'])
    fyf, st = O.ML_full_yf(m, sig[:nf])
    assert np.max(np.abs(fyf - z['b_full_yf'])) < 1e-7
    for tag in ('pe', 'ps'):
        CV = z['r%s_CV' % tag]
        fy, so, dp, ct, ns = O.response(m, CV, z['r%s_sig' % tag], z['r%s_epl' % tag], z['r%s_deps' % tag])
        assert np.array_equal(ns, z['r%s_nsteps' % tag])
        sc = float(m.c.sy)
        assert np.max(np.abs(so - z['r%s_sig_out' % tag])) < 1e-6 * sc
        assert np.max(np.abs(dp - z['r%s_depl' % tag])) < 1e-9
        assert np.max(np.abs(ct - z['r%s_ct' % tag])) < 1e-5 * CV[0, 0]
