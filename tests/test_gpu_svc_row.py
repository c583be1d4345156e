"""GPU parity of the 16-lanes-per-point SVC kernels (round 5: YfSvcRow -- sampled-ray form of the ML_full_yf ray search,
material.py:414-516, inside Material.response, material.py:207-346) at the POINT level, through plfx_response_batch /
plfx_full_yf_batch, which run the code path of the model sweeps for every 6-feature SVC whose tables fit the LDS.

 * against the pinned CPU oracle (iterate-exact brentq on FP64 sums) on seeded points of all four branches;
 * against the thread-per-point kernels of rounds 1-4 (PLFX_RESPONSE_ROW=0: brentq replay on the support-vector sums) on a
   large seeded batch, counting branch flips (none allowed outside 1e-5 of a threshold);
 * with a flow stress far from the trained yield locus, where the march leaves the sampled interval and the search falls back
   to the support-vector sums and to brentq (the paths the usual inputs never take);
 * two different SVC phases and a J2 phase in one model against the oracle's sparse direct solve, both SVCs on the row kernels.
Tolerance: 1e-6 of the yield stress (north star; brentq's own xtol = 1e-5 MPa is 2e-7 of it)."""
import os
import warnings

import numpy as np
import pytest

from test_gpu_model import FE, make_material, svc_material

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from pylabfea_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def load_svc(ctx, z, tag='pe', sy=None):
    from pylabfea_amd import _lib
    svc = dict(sv=z['par_sv'], dual=z['par_dual'], gamma=float(z['par_gamma']), intercept=float(z['par_intercept']),
               scale_seq=float(z['par_scale_seq']), dev_only=bool(z['par_dev_only']))
    CV = z['r%s_CV' % tag]
    rec = _lib.pack_material(_lib.SVC6, CV, E=float(z['par_E']), nu=float(z['par_nu']), sy=float(z['par_sy']) if sy is None else sy,
                             khard=float(z['par_khard']), hill=z['par_hill'], svc=svc)
    ctx.set_materials([rec])
    return CV


def seeded_points(z, n, seed, scale=(0.5, 1.05), amp=3e-4):
    """stresses around the trained yield locus (seq = sy x U(scale)), strain increments N(0, amp): elastic, split,
    one-step and 50-sub-step calls"""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    sy = float(z['par_sy'])
    j2 = O.Material(kind=O.HILL6, sy=sy)
    d = rng.normal(size=(n, 6))
    d[:, 3:5] = 0.
    sig = d / O.calc_seq(j2, d)[:, None] * sy * rng.uniform(scale[0], scale[1], size=n)[:, None]
    deps = rng.normal(size=(n, 6)) * amp
    deps[:, 3:5] = 0.
    return sig, np.zeros((n, 6)), deps


@pytest.mark.parametrize('name', ['hill', 'gossbarlat', 'j2train'])
def test_row_response_vs_oracle(ctx, golden_dir, name):
    from oracle import oracle as O
    z = np.load(os.path.join(golden_dir, 'svc_%s.npz' % name))
    CV = load_svc(ctx, z)
    assert ctx.svc_info()[0] == 1 and ctx.svc_info()[1] == 0      # material 0 on the row kernels
    sy = float(z['par_sy'])
    sig, epl, deps = seeded_points(z, 600, 1)
    fy, so, dp, ct, ns = ctx.response(sig, epl, deps)
    om = O.Material.from_golden(z)
    fy2, so2, dp2, ct2, ns2 = O.response(om, CV, sig, epl, deps)
    assert len(set(ns2)) == 2 and np.sum(ns2 == 49) > 50          # both phases of the sweep are exercised
    assert np.array_equal(ns, ns2)
    assert np.max(np.abs(so - so2)) < 1e-6 * sy
    assert np.max(np.abs(fy - fy2)) < 1e-6 * sy
    assert np.max(np.abs(dp - dp2)) < 1e-9
    assert np.max(np.abs(ct - ct2)) < 1e-5 * CV[0, 0]
    f, st = ctx.full_yf(0, sig)
    f2, st2 = O.ML_full_yf(om, sig)
    assert np.array_equal(st, st2) and np.max(np.abs(f - f2)) < 1e-6 * sy


def test_row_vs_thread_kernels_large_batch_no_branch_flips(ctx, golden_dir, monkeypatch):
    """60 000 seeded points: the row kernels against the thread-per-point kernels (brentq replay on FP64 sums)"""
    z = np.load(os.path.join(golden_dir, 'svc_hill.npz'))
    load_svc(ctx, z)
    sy = float(z['par_sy'])
    sig, epl, deps = seeded_points(z, 60000, 2, scale=(0.6, 1.03), amp=1.2e-4)
    a = ctx.response(sig, epl, deps)
    monkeypatch.setenv('PLFX_RESPONSE_ROW', '0')
    b = ctx.response(sig, epl, deps)
    monkeypatch.delenv('PLFX_RESPONSE_ROW')
    assert np.sum(b[4] == 49) > 1000 and np.sum(b[4] == 0) > 10000
    flips = a[4] != b[4]
    # a call may change branch only where the value its branch test reads sits within 1e-5 sy of the threshold
    # (material.py:253: fy1 < toler; :292: fy1 > toler -- toler = 5e-3 sflow)
    toler = 5e-3 * sy
    near = np.abs(np.abs(b[0]) - toler) < 1e-5 * sy
    assert np.sum(flips & ~near) == 0, (int(flips.sum()), int((flips & ~near).sum()))
    assert flips.mean() < 1e-4
    ok = ~flips
    assert np.max(np.abs(a[1][ok] - b[1][ok])) < 1e-6 * sy
    assert np.max(np.abs(a[0][ok] - b[0][ok])) < 1e-6 * sy
    assert np.max(np.abs(a[2][ok] - b[2][ok])) < 1e-9


@pytest.mark.parametrize('factor', [2.2, 0.45, 1.27])
def test_row_search_outside_the_sampled_interval(ctx, golden_dir, factor):
    """a flow stress `factor` times the trained one: the march starts far from the yield locus, leaves the interval the ray
    was sampled on ([0.72, 1.30] sflow, [0.47, 1.35] sflow on rays with s1 s2 < 0) and the bracket is not covered by the
    polynomial -- support-vector sums and brentq instead; results are the oracle's all the same"""
    from oracle import oracle as O
    z = np.load(os.path.join(golden_dir, 'svc_hill.npz'))
    sy = float(z['par_sy']) * factor
    CV = load_svc(ctx, z, sy=sy)
    om = O.Material.from_golden(z)
    om.c.sy = sy
    sig, epl, deps = seeded_points(z, 160, 3, scale=(0.4, 1.2))
    f, st = ctx.full_yf(0, sig)
    f2, st2 = O.ML_full_yf(om, sig)
    assert np.array_equal(st, st2)
    assert np.max(np.abs(f - f2)) < 1e-6 * sy
    fy, so, dp, ct, ns = ctx.response(sig, epl, deps)
    fy2, so2, dp2, ct2, ns2 = O.response(om, CV, sig, epl, deps)
    assert np.array_equal(ns, ns2)
    assert np.max(np.abs(so - so2)) < 1e-6 * sy
    assert np.max(np.abs(fy - fy2)) < 1e-6 * sy


def split_vectors(z, times):
    """the same decision function with `times` as many support vectors: every vector listed `times` times with 1 / times of its
    dual coefficient (sum_k d_k K(x, v_k) is unchanged)"""
    sv = np.repeat(z['par_sv'], times, axis=0)
    dual = np.repeat(z['par_dual'], times) / times
    return sv, dual


def test_svc_with_more_vectors_than_the_lds_holds(ctx, golden_dir):
    """4755 support vectors (3 x 1585; the LDS of a CU holds the tables of ~2200): the row kernels read the tables from device
    memory instead (k_*_row<false>) -- same code, same results; reference: material.py:398-405 evaluates any trained svm_yf"""
    from pylabfea_amd import _lib
    from oracle import oracle as O
    z = np.load(os.path.join(golden_dir, 'svc_hill.npz'))
    sv, dual = split_vectors(z, 3)
    svc = dict(sv=sv, dual=dual, gamma=float(z['par_gamma']), intercept=float(z['par_intercept']),
               scale_seq=float(z['par_scale_seq']), dev_only=bool(z['par_dev_only']))
    CV, sy = z['rpe_CV'], float(z['par_sy'])
    rec = _lib.pack_material(_lib.SVC6, CV, E=float(z['par_E']), nu=float(z['par_nu']), sy=sy, khard=float(z['par_khard']),
                             hill=z['par_hill'], svc=svc)
    ctx.set_materials([rec])
    assert ctx.svc_info()[:2] == (1, 0)                        # on the row kernels all the same
    sig, epl, deps = seeded_points(z, 400, 5)
    fy, so, dp, ct, ns = ctx.response(sig, epl, deps)
    om = O.Material(kind=O.SVC6, E=float(z['par_E']), nu=float(z['par_nu']), sy=sy, khard=float(z['par_khard']), hill=z['par_hill'],
                    sv=sv, dual=dual, gamma=svc['gamma'], intercept=svc['intercept'], scale_seq=svc['scale_seq'], dev_only=svc['dev_only'])
    fy2, so2, dp2, ct2, ns2 = O.response(om, CV, sig, epl, deps)
    assert np.sum(ns2 == 49) > 30 and np.array_equal(ns, ns2)
    assert np.max(np.abs(so - so2)) < 1e-6 * sy and np.max(np.abs(fy - fy2)) < 1e-6 * sy
    f, st = ctx.full_yf(0, sig)
    f2, st2 = O.ML_full_yf(om, sig)
    assert np.array_equal(st, st2) and np.max(np.abs(f - f2)) < 1e-6 * sy
    # ... and equal to the 1585-vector form of the same function on the LDS path
    load_svc(ctx, z)
    fy3, so3 = ctx.response(sig, epl, deps)[:2]
    assert np.max(np.abs(so - so3)) < 1e-7 * sy


def three_phase_model(golden_dir, NX, NY, big=False):
    """laminate of two DIFFERENT trained SVCs (Hill reference material of examples/train_hill.py, 1585 vectors; Barlat /
    Goss texture of examples/train_goss_barlat.py, 1418 vectors) around a J2 core"""
    ma = svc_material(golden_dir, 'hill')
    mb = make_material('j2')
    mc = svc_material(golden_dir, 'gossbarlat')
    if big:   # the Goss-Barlat SVC with 4254 support vectors (each listed three times): more than the LDS holds
        z = np.load(os.path.join(golden_dir, 'svc_gossbarlat.npz'))
        sv, dual = split_vectors(z, 3)
        mc.set_svc(sv, dual, float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']), dev_only=bool(z['par_dev_only']))
    ma.num, mb.num, mc.num = 1, 2, 3
    fe = FE().Model(dim=2, planestress=False)
    fe.geom([2, 1, 2], LY=4.)
    fe.assign([ma, mb, mc])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.0014 * fe.leny, 'disp')
    fe.mesh(NX=NX, NY=NY)
    return fe


@pytest.mark.parametrize('big', [False, True])
def test_two_svc_phases_in_one_model_vs_oracle(golden_dir, big):
    """two distinct 6-feature SVC phases (big: one of them with 4254 support vectors, its tables in device memory) and a J2
    phase: every SVC element on the row kernels, no thread-per-element SVC launch (plfx_svc_info)"""
    from oracle.solve_ref import RefSolver
    fe = three_phase_model(golden_dir, 20, 8, big)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=6)
        ref = RefSolver(three_phase_model(golden_dir, 20, 8, big)).solve(min_step=6)
    row, thread, nrow, nthread = fe._engine.svc_info()
    assert row == 0b101 and thread == 0 and nthread == 0 and nrow >= 2 * fe.n_sweeps   # both SVCs on the row kernels, no thread launch
    assert fe.nsteps == ref.nsteps and list(fe.niter) == list(ref.niter)
    mid = fe._mat_id
    epl = fe._state('epl')
    assert np.max(np.abs(epl[mid == 0])) > 0. and np.max(np.abs(epl[mid == 2])) > 0.         # both SVC phases yield
    s = np.max(np.abs(ref.sig))
    assert np.max(np.abs(fe.u - ref.u)) < 2e-6 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 2e-6 * s
    assert np.max(np.abs(epl - ref.epl)) < 2e-6 * np.max(np.abs(ref.eps))
    assert np.max(np.abs(fe.sgl - ref.sgl)) < 2e-6 * s
