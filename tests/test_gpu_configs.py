"""GPU parity on the schedules BASELINE.json's configs are quoted on.

``tests/golden/solve_configs.npz`` (oracle/gen_golden.py:gen_configs) holds traces of the UNMODIFIED reference for
  config 2: 8x8 J2 (sy=150, khard=500), eps=0.004, min_step=20
  config 3: 8x8 Hill-6p (sy=100, hill=[0.7,1,1.4,1,1.2,0.8], khard=100), eps=0.005, min_step=50
  config 4: 4x4 SVC of examples/train_hill.py (1585 support vectors), eps=0.001, min_step=10
  config 5: 8x4 laminate [2,1,2,1,2] (LY=8) of J2 and the SVC trained on Barlat Yld2004-18p / Goss texture
            (examples/train_goss_barlat.py), eps=0.003, min_step=20
The small meshes are compared field by field; the full-size runs (256^2, 1024^2, 512^2) of the homogeneous configs must
reproduce the small-mesh traces (uniform solution: mesh-size independent), which puts the regime the bench times
(load steps >= 10, scale_bc = 1, unchanged-input reuse of assemblies / solves) under the reference.
Bar: identical load-step / K-iteration / non-convergence counts, 1e-6 relative on fields (north star)."""
import os
import warnings

import numpy as np
import pytest

from test_gpu_model import FE, check_fields, close, make_material, svc_material, tension_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cfg(golden_dir):
    return np.load(os.path.join(golden_dir, 'solve_configs.npz'))


def homogeneous_check(fe, g, p, rtol=1e-6):
    """full-size run against the small-mesh reference trace of the same homogeneous problem"""
    assert fe.nsteps == int(g[p + '_nsteps'])
    assert list(fe.niter) == list(g[p + '_niter'])
    assert list(fe.co_nconv) == list(g[p + '_co_nconv'])
    assert close(fe.sgl, g[p + '_sgl'], rtol=rtol)
    assert close(fe.egl, g[p + '_egl'], rtol=rtol)
    assert close(fe.epgl, g[p + '_epgl'], scale=np.max(np.abs(g[p + '_egl'])), rtol=rtol)
    sig, epl, eps = fe._state('sig'), fe._state('epl'), fe._state('eps')
    assert np.max(np.abs(sig - sig[0])) < rtol * np.max(np.abs(sig))           # every element carries the same state
    assert np.max(np.abs(epl - epl[0])) < rtol * np.max(np.abs(eps))
    assert close(sig[0], g[p + '_sig'][0], rtol=rtol)
    assert close(epl[0], g[p + '_epl'][0], scale=np.max(np.abs(g[p + '_eps'])), rtol=rtol)
    assert close(fe._state('elstiff')[0], g[p + '_elstiff'][0], rtol=10 * rtol)
    gb = g[p + '_globbc']
    mine = np.array([fe.glob[k] for k in ('ebc1', 'ebc2', 'sbc1', 'sbc2')])
    assert close(mine[:2], gb[:2], scale=np.max(np.abs(g[p + '_egl'])), rtol=rtol)
    assert close(mine[2:], gb[2:4], scale=np.max(np.abs(g[p + '_sgl'])), rtol=rtol)
    free = fe.free_dofs()
    assert np.max(np.abs(fe.f[free])) < rtol * np.max(np.abs(fe.f))             # equilibrium


# ------------------------------------------------------------------ config 2: J2, eps 0.004, min_step 20
def test_config2_schedule_8x8(cfg):
    fe = tension_model(make_material('j2'), 8, 0.004)
    fe.solve(min_step=20)
    check_fields(fe, cfg, 'cfg2_j2_8')


def test_config2_full_size_256(cfg):
    """BASELINE.json configs[1]: 256x256 Q4, isotropic J2 plasticity, 20 load increments — exact workload."""
    fe = tension_model(make_material('j2'), 256, 0.004)
    fe.solve(min_step=20)
    homogeneous_check(fe, cfg, 'cfg2_j2_8')
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1


# ------------------------------------------------------------------ config 3: Hill-6p, eps 0.005, min_step 50
def test_config3_schedule_8x8(cfg):
    fe = tension_model(make_material('hill6'), 8, 0.005)
    fe.solve(min_step=50)
    check_fields(fe, cfg, 'cfg3_hill6_8')
    # the regime bench.py times: from load step 10 on the increments are equal and the library answers repeated solves /
    # assemblies from the previous ones (plfx_reuse_info) -- pinned here against the reference's trace
    ra, rb, rs = fe._engine.reuse_info()
    assert ra > 0 and rs > 0


def test_config3_full_size_1024(cfg):
    """BASELINE.json configs[2] = the bench workload: 1024x1024 Q4, Hill-48, eps=0.005, 50 increments."""
    fe = tension_model(make_material('hill6'), 1024, 0.005)
    fe.solve(min_step=50)
    homogeneous_check(fe, cfg, 'cfg3_hill6_8')
    ra, rb, rs = fe._engine.reuse_info()
    assert ra >= 20 and rs >= 20     # the il >= 10 reuse path ran (BENCH_r01: 20 of 40 assemblies, 25 of 60 solves per 10 steps)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1


def test_config3_reuse_off_is_the_same_trace(cfg):
    """PLFX_REUSE=0 (every assembly / solve recomputed, like the reference does) gives the same trace at 128^2."""
    os.environ['PLFX_REUSE'] = '0'
    try:
        fe = tension_model(make_material('hill6'), 128, 0.005)
        fe.solve(min_step=50)
    finally:
        del os.environ['PLFX_REUSE']
    homogeneous_check(fe, cfg, 'cfg3_hill6_8')
    assert fe._engine.reuse_info() == (0, 0, 0)


# ------------------------------------------------------------------ config 4: train_hill SVC, eps 0.001, min_step 10
def test_config4_schedule_4x4(cfg, golden_dir):
    assert bool(cfg['cfg4_same_as_svc_hill'])        # the trace was made with the SVC stored in svc_hill.npz
    fe = tension_model(svc_material(golden_dir, 'hill'), 4, 0.001)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
    check_fields(fe, cfg, 'cfg4_svc_4', rtol=2e-6)   # SVC: bounded by brentq's xtol = 1e-5 MPa in ML_full_yf


def test_config4_full_size_512(cfg, golden_dir):
    """BASELINE.json configs[3]: 512x512 mesh, SVC yield function with 1585 support vectors -- exact workload.
    The first ten load steps (elastic, scaled by calc_scf) are uniform and must reproduce the reference's 4x4 trace.  The
    eleventh takes the whole remaining load in one step and does not converge in the reference either (16 stiffness
    iterations); there the field is NOT uniform -- the trained SVC couples the in-plane stress to small shear components
    (1.8 % non-uniformity on the reference's 4x4 mesh) -- so that step is mesh dependent (tools/probes/cfg4_probe.py:
    1.4e-3 at 16^2 ... 1.6e-3 at 256^2, independent of the PCG tolerance) and is pinned by the oracle test below instead;
    here: identical iteration / non-convergence counts, the homogenised stress within 5e-3, equilibrium."""
    g, p = cfg, 'cfg4_svc_4'
    fe = tension_model(svc_material(golden_dir, 'hill'), 512, 0.001)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
    assert fe.nsteps == int(g[p + '_nsteps']) == 11
    assert list(fe.niter) == list(g[p + '_niter'])
    assert list(fe.co_nconv) == list(g[p + '_co_nconv'])
    s = np.max(np.abs(g[p + '_sgl']))
    assert np.max(np.abs(fe.sgl[:11] - g[p + '_sgl'][:11])) < 1e-6 * s       # ten uniform steps: strict
    assert np.max(np.abs(fe.egl[:11] - g[p + '_egl'][:11])) < 1e-6 * np.max(np.abs(g[p + '_egl']))
    assert np.max(np.abs(fe.sgl[11] - g[p + '_sgl'][11])) < 5e-3 * s         # the non-converged step: mesh dependent
    assert np.max(np.abs(fe.egl[11][1] - g[p + '_egl'][11][1])) < 1e-9       # prescribed global strain
    free = fe.free_dofs()
    assert np.max(np.abs(fe.f[free])) < 1e-6 * np.max(np.abs(fe.f))
    assert np.max(fe._state('max_steps')) == 49       # the 50-sub-step corrector ran (wave-per-element SVC kernels)


def test_config4_schedule_32x32_vs_oracle(golden_dir):
    """config 4's schedule on 32x32 elements (multigrid, matrix-free operator, wave-per-element SVC kernels) against the
    pinned oracle's sparse direct solve, including the non-converged last step"""
    from oracle.solve_ref import RefSolver
    fe = tension_model(svc_material(golden_dir, 'hill'), 32, 0.001)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
        ref = RefSolver(tension_model(svc_material(golden_dir, 'hill'), 32, 0.001)).solve(min_step=10)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1
    assert fe.nsteps == ref.nsteps and list(fe.niter) == list(ref.niter) and list(fe.co_nconv) == list(ref.co_nconv)
    s = np.max(np.abs(ref.sig))
    assert np.max(np.abs(fe.u - ref.u)) < 5e-6 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 5e-6 * s
    assert np.max(np.abs(fe._state('epl') - ref.epl)) < 5e-6 * np.max(np.abs(ref.eps))
    assert np.max(np.abs(fe.sgl - ref.sgl)) < 5e-6 * s


@pytest.fixture(scope='module')
def mid(golden_dir):
    """tests/golden/mid_configs.npz: the PINNED ORACLE's solutions of configs 4 and 5 on mid-size meshes (oracle/gen_mid_configs.py,
    run in the build container: 1.5 and 13 minutes of host time, which the GPU suite cannot afford per run)"""
    return np.load(os.path.join(golden_dir, 'mid_configs.npz'))


def test_config4_schedule_64x64_vs_oracle_fixture(mid, golden_dir):
    """between the 32 x 32 oracle comparison above and the 512 x 512 run (whose last, non-converged load step
    is only held to 5e-3 of the reference's 4 x 4 trace): config 4 on 64 x 64 elements, ALL 11 load steps, field by field at 5e-6
    against the oracle's sparse direct solve of the same mesh (fixture)."""
    p = 'cfg4_64'
    fe = tension_model(svc_material(golden_dir, 'hill'), 64, 0.001)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1
    assert fe.nsteps == int(mid[p + '_nsteps']) == 11
    assert list(fe.niter) == list(mid[p + '_niter']) and list(fe.co_nconv) == list(mid[p + '_co_nconv'])
    s = np.max(np.abs(mid[p + '_sig']))
    assert np.max(np.abs(fe.u - mid[p + '_u'])) < 5e-6 * np.max(np.abs(mid[p + '_u']))
    assert np.max(np.abs(fe._state('sig') - mid[p + '_sig'])) < 5e-6 * s
    assert np.max(np.abs(fe._state('epl') - mid[p + '_epl'])) < 5e-6 * np.max(np.abs(mid[p + '_eps']))
    assert np.max(np.abs(np.asarray(fe.sgl) - mid[p + '_sgl'])) < 5e-6 * s
    assert np.max(fe._state('max_steps')) == 49


@pytest.fixture(scope='module')
def mid128(golden_dir):
    return np.load(os.path.join(golden_dir, 'mid_configs_128.npz'))


def test_config4_schedule_128x128_vs_oracle_fixture(mid128, golden_dir):
    """One octave above the 64 x 64 comparison: config 4 on 128 x 128 elements, all 11 load steps, against the oracle's sparse
    direct solve (fixture: 21 minutes of the oracle on 8 cores, oracle/gen_mid_configs.py --larger 4), same bars."""
    mid, p = mid128, 'cfg4_128'
    fe = tension_model(svc_material(golden_dir, 'hill'), 128, 0.001)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1
    assert fe.nsteps == int(mid[p + '_nsteps']) == 11
    assert list(fe.niter) == list(mid[p + '_niter']) and list(fe.co_nconv) == list(mid[p + '_co_nconv'])
    s = np.max(np.abs(mid[p + '_sig']))
    err = {'u': np.max(np.abs(fe.u - mid[p + '_u'])) / np.max(np.abs(mid[p + '_u'])),
           'sig': np.max(np.abs(fe._state('sig') - mid[p + '_sig'])) / s,
           'epl': np.max(np.abs(fe._state('epl') - mid[p + '_epl'])) / np.max(np.abs(mid[p + '_eps'])),
           'sgl': np.max(np.abs(np.asarray(fe.sgl) - mid[p + '_sgl'])) / s}
    print('config 4 on 128 x 128 against the oracle fixture:', {k: float('%.2e' % v) for k, v in err.items()})
    assert max(err.values()) < 5e-6, err
    assert np.max(fe._state('max_steps')) == 49


# ------------------------------------------------------------------ config 5: J2 + Goss-Barlat-trained SVC laminate
def laminate_cfg5(golden_dir, NX, NY):
    ma = make_material('j2')
    mb = svc_material(golden_dir, 'gossbarlat')
    ma.num, mb.num = 1, 2
    fe = FE().Model(dim=2, planestress=False)
    fe.geom([2, 1, 2, 1, 2], LY=8.)
    fe.assign([ma, mb, ma, mb, ma])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.003 * fe.leny, 'disp')
    fe.mesh(NX=NX, NY=NY)
    return fe


def test_config5_real_materials_8x4(cfg, golden_dir):
    """BASELINE config 5's materials (J2 + SVC trained on Barlat Yld2004-18p, examples/train_goss_barlat.py:36-41, 70-83)
    on the laminate geometry, 8x4 elements, against the reference's trace."""
    fe = laminate_cfg5(golden_dir, 8, 4)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=20)
    check_fields(fe, cfg, 'cfg5_lam_8x4', rtol=2e-6)
    assert np.max(fe._state('epl')[fe._mat_id == 1]) > 0. and np.max(fe._state('epl')[fe._mat_id == 0]) > 0.


def test_config5_real_materials_64x32_vs_oracle(golden_dir):
    """the same laminate on 64x32 elements (multigrid + matrix-free operator, wave-per-element SVC kernels, material jumps)
    against the pinned oracle's sparse direct solve, first 6 load steps of the 20-increment schedule"""
    from oracle.solve_ref import RefSolver
    fe = laminate_cfg5(golden_dir, 64, 32)
    fe._max_load_steps = 6
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=20)
        ref = RefSolver(laminate_cfg5(golden_dir, 64, 32)).solve(min_step=20, max_load_steps=6)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1
    assert fe.nsteps == ref.nsteps and list(fe.niter) == list(ref.niter)
    s = np.max(np.abs(ref.sig))
    assert np.max(np.abs(fe.u - ref.u)) < 2e-6 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 2e-6 * s
    assert np.max(np.abs(fe._state('epl') - ref.epl)) < 2e-6 * np.max(np.abs(ref.eps))
    assert np.max(np.abs(fe.sgl - ref.sgl)) < 2e-6 * s


def test_config5_real_materials_32x16_vs_oracle_all_load_steps(golden_dir):
    """... and ALL 20 load steps on 32x16 elements (the SVC phase only starts yielding in load step 5, the 64x32
    comparison above ends after step 6): every plastic step of the SVC columns, the indefinite tangents of the last steps and
    the solves they need, field by field against the oracle's sparse direct solve"""
    from oracle.solve_ref import RefSolver
    fe = laminate_cfg5(golden_dir, 32, 16)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=20)
        ref = RefSolver(laminate_cfg5(golden_dir, 32, 16)).solve(min_step=20)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1
    assert fe._engine.svc_info()[0] == 0b10 and fe._engine.svc_info()[3] == 0          # the SVC phase on the row kernels
    assert fe.nsteps == ref.nsteps == 20
    # K-iteration counts: equal while the fields are smooth; the last load steps end when the largest tangent change of any
    # element falls below 1e-3 (model.py:1346-1355), a test on a maximum that round-off moves by an iteration
    assert list(fe.niter[:12]) == list(ref.niter[:12])
    assert np.max(np.abs(np.asarray(fe.niter) - np.asarray(ref.niter))) <= 2
    s = np.max(np.abs(ref.sig))
    assert np.max(np.abs(np.asarray(fe.sgl) - np.asarray(ref.sgl))) < 5e-6 * s
    assert np.max(np.abs(fe.u - ref.u)) < 2e-5 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 2e-5 * s
    assert np.max(np.abs(fe._state('epl') - ref.epl)) < 2e-5 * np.max(np.abs(ref.eps))
    assert np.max(np.abs(ref.epl[fe._mat_id == 1])) > 1e-3                            # deep in the plastic regime of the SVC phase


def test_config5_real_materials_128x64_vs_oracle_fixture_all_load_steps(mid, golden_dir):
    """config 5's laminate on 128 x 64 elements through ALL 20 load steps (every plastic step of the SVC
    columns, the indefinite tangents of the last steps and the GMRES solves they need) against the oracle's sparse direct solve
    of the same mesh (fixture; 13 minutes of host time).  Bars as in the 32 x 16 comparison above."""
    p = 'cfg5_128x64'
    fe = laminate_cfg5(golden_dir, 128, 64)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=20)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1
    assert fe.nsteps == int(mid[p + '_nsteps']) == 20
    rn = mid[p + '_niter']
    # K-iteration counts: equal through the elastic steps and the onset of yielding; afterwards a load step ends when the largest
    # tangent change of any element falls below 1e-3 (model.py:1346-1355) -- a test on a maximum over 8192 elements that the
    # difference between a 1e-10 iterative solve and the oracle's direct solve moves by an iteration (measured: step 11, 14 vs 15)
    assert list(fe.niter[:10]) == list(rn[:10])
    assert np.max(np.abs(np.asarray(fe.niter) - rn)) <= 2
    s = np.max(np.abs(mid[p + '_sig']))
    d_sgl = np.max(np.abs(np.asarray(fe.sgl) - mid[p + '_sgl'])) / s
    d_u = np.max(np.abs(fe.u - mid[p + '_u'])) / np.max(np.abs(mid[p + '_u']))
    d_sig = np.max(np.abs(fe._state('sig') - mid[p + '_sig'])) / s
    d_epl = np.max(np.abs(fe._state('epl') - mid[p + '_epl'])) / np.max(np.abs(mid[p + '_eps']))
    print('config 5 on 128 x 64 vs the oracle fixture: sgl %.1e u %.1e sig %.1e epl %.1e; niter %s vs %s; fall-back solves %d'
          % (d_sgl, d_u, d_sig, d_epl, list(fe.niter), list(rn), fe._engine.solve_fallbacks()))
    assert d_sgl < 5e-6 and d_u < 2e-5 and d_sig < 2e-5 and d_epl < 2e-5
    assert np.max(np.abs(mid[p + '_epl'][fe._mat_id == 1])) > 1e-3


def test_config5_real_materials_256x128_vs_oracle_fixture_all_load_steps(mid128, golden_dir):
    """One octave above the 128 x 64 comparison: config 5's laminate on 256 x 128 elements through all 20 load steps against the
    oracle's sparse direct solve (fixture: 40 minutes of the oracle on 8 cores, oracle/gen_mid_configs.py --larger 5).
    Global stress history and displacements at the bars of the smaller meshes (measured: sgl 2e-9, u 8e-7).  The element fields
    agree in the mean (RMS 1e-7 .. 7e-7) -- but NOT everywhere at 2e-5: in the last load steps the SVC tangents are indefinite
    (material.py:317-338), four of them end after 12-13 stiffness iterations here and after 14-15 in the oracle (the end test is a
    maximum over all elements of a non-smooth quantity, model.py:1346-1355), and a cluster of about 15 elements of one SVC
    section (element columns 171-179) carries the difference: up to 1e-4 of the largest strain (tools/probes/cfg5_256_diag.py).
    The bars say that: mean at 5e-6, at most 0.2 % of the elements beyond 2e-5, none beyond 5e-4.  That the cluster is the 1e-10
    of the iterative solves carried through those steps, and nothing else, is shown by the second run: with the solves at 1e-12
    (Model.cg_rtol) every element field agrees with the direct solve to 1e-6 (measured: sig 9.9e-7, epl 8.3e-7; bar 5e-6)."""
    mid, p = mid128, 'cfg5_256x128'
    if p + '_nsteps' not in mid.files:
        pytest.skip('fixture not generated (oracle/gen_mid_configs.py --larger 5)')
    fe = laminate_cfg5(golden_dir, 256, 128)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=20)
    assert fe._engine.precond_info()[0] == 1 and fe._engine.operator_info()[0] == 1
    assert fe.nsteps == int(mid[p + '_nsteps']) == 20
    rn = mid[p + '_niter']
    assert list(fe.niter[:10]) == list(rn[:10])      # (see the 128 x 64 test for why the later counts may differ by an iteration)
    assert np.max(np.abs(np.asarray(fe.niter) - rn)) <= 2
    s = np.max(np.abs(mid[p + '_sig']))
    d_sgl = np.max(np.abs(np.asarray(fe.sgl) - mid[p + '_sgl'])) / s
    d_u = np.max(np.abs(fe.u - mid[p + '_u'])) / np.max(np.abs(mid[p + '_u']))
    e_sig = np.abs(fe._state('sig') - mid[p + '_sig']) / s
    e_epl = np.abs(fe._state('epl') - mid[p + '_epl']) / np.max(np.abs(mid[p + '_eps']))
    beyond = int(np.sum(np.maximum(e_sig.max(axis=1), e_epl.max(axis=1)) > 2e-5))
    print('config 5 on 256 x 128 vs the oracle fixture: sgl %.1e u %.1e sig max %.1e rms %.1e epl max %.1e rms %.1e, %d of %d elements beyond '
          '2e-5; niter %s vs %s; fall-back solves %d'
          % (d_sgl, d_u, e_sig.max(), np.sqrt(np.mean(e_sig ** 2)), e_epl.max(), np.sqrt(np.mean(e_epl ** 2)), beyond, len(e_sig),
             list(fe.niter), [int(v) for v in rn], fe._engine.solve_fallbacks()))
    assert d_sgl < 5e-6 and d_u < 2e-5
    assert np.sqrt(np.mean(e_sig ** 2)) < 5e-6 and np.sqrt(np.mean(e_epl ** 2)) < 5e-6
    assert beyond <= 0.002 * len(e_sig) and max(e_sig.max(), e_epl.max()) < 5e-4
    assert np.max(np.abs(mid[p + '_epl'][fe._mat_id == 1])) > 1e-3
    ft = laminate_cfg5(golden_dir, 256, 128)
    ft.cg_rtol = 1e-12
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ft.solve(min_step=20)
    t_sig = np.max(np.abs(ft._state('sig') - mid[p + '_sig'])) / s
    t_epl = np.max(np.abs(ft._state('epl') - mid[p + '_epl'])) / np.max(np.abs(mid[p + '_eps']))
    t_u = np.max(np.abs(ft.u - mid[p + '_u'])) / np.max(np.abs(mid[p + '_u']))
    print('   with the solves at 1e-12: u %.1e sig max %.1e epl max %.1e; niter %s' % (t_u, t_sig, t_epl, list(ft.niter)))
    assert ft.nsteps == 20 and np.max(np.abs(np.asarray(ft.niter) - rn)) <= 2
    assert t_u < 5e-6 and t_sig < 5e-6 and t_epl < 5e-6


def test_config5_sgl_does_not_depend_on_the_mesh(golden_dir):
    """Size-independent property of BASELINE config 5 (laminate [2,1,2,1,2] along y, J2 + the SVC trained on Barlat
    Yld2004-18p, eps = 0.003, min_step = 20): the fields are uniform along y and piecewise constant per section, so the
    global stress history must not depend on the mesh as long as the section boundaries fall on element edges.  (This is the
    property that exposed the indefinite tangents of DESIGN.md section 5: 2048 x 2048 elements gave 139.06 instead of 144.13.)"""
    import warnings
    import pylabfea_amd as FE
    z = np.load(os.path.join(golden_dir, 'svc_gossbarlat.npz'))

    def run(NX, NY):
        ma = FE.Material(num=1)
        ma.elasticity(E=200.e3, nu=0.3)
        ma.plasticity(sy=150., khard=500., sdim=6)
        mb = FE.Material(name='ML-Goss-Barlat', num=2)
        mb.elasticity(CV=z['par_CV'])
        mb.plasticity(sy=float(z['par_sy']), sdim=6)
        mb.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
        fe = FE.Model(dim=2)
        fe.geom([2, 1, 2, 1, 2], LY=8.)
        fe.assign([ma, mb, ma, mb, ma])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.003 * fe.leny, 'disp')
        fe.mesh(NX=NX, NY=NY)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=20)
        return fe
    a, b = run(256, 32), run(512, 128)
    assert a.nsteps == b.nsteps == 20
    sa, sb = np.array(a.sgl), np.array(b.sgl)
    assert np.max(np.abs(sa - sb)) < 5e-5 * np.max(np.abs(sa))     # (measured 2e-5, tools/probes/cfg5_margins.py)
    # the value every mesh from 512 x 64 to 2048 x 2048 gives is the REFERENCE's: its own trace of this schedule on 8 x 4 elements
    # (fixture cfg5_lam_8x4_sgl: 144.1353) -- the laminate solution does not depend on the mesh
    rs = np.load(os.path.join(golden_dir, 'solve_configs.npz'))['cfg5_lam_8x4_sgl']
    assert abs(sa[-1][1] - rs[-1][1]) < 1e-4 * rs[-1][1] and np.max(np.abs(sa - rs)) < 2e-4 * np.max(np.abs(rs))


def test_config5_full_size_2048_real_materials(golden_dir):
    """BASELINE config 5 at its stated size with its real materials, THE WHOLE SCHEDULE: 2048 x 2048 laminate [2,1,2,1,2] of
    J2 and the SVC trained on Barlat Yld2004-18p / Goss (examples/train_goss_barlat.py:36-41, 70-83; laminate sections by
    model.py:826-830 -> element columns [512, 256, 512, 256, 512]), eps = 0.003, min_step = 20 -- all 20 load steps (5 elastic
    ones, then the SVC phase yields: its 1 M elements run the 50-sub-step corrector on the wave-per-element kernels).  Pinned
    through the size-independent property of the laminate (uniform along y, piecewise constant per section): the global
    stress / strain history equals the one of the 512 x 64 mesh, whose materials are pinned against the reference's trace
    (8 x 4) and the oracle (64 x 32) above -- and it is held directly to the REFERENCE's own 8 x 4 trace of this schedule
    (fixture cfg5_lam_8x4_*, written by oracle/gen_golden.py with the unmodified reference)."""
    big = laminate_cfg5(golden_dir, 2048, 2048)
    small = laminate_cfg5(golden_dir, 512, 64)
    for fe in (small, big):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=20)
    assert big.Nel == 2048 * 2048 and big.Ndof == 8396802
    assert np.array_equal(np.bincount(big._mat_id.reshape(2048, 2048)[:, 0]), [512, 256, 512, 256, 512])   # sections A B A B A
    svc_el = np.isin(big._mat_id, (1, 3))
    eng = big._engine
    assert eng.precond_info()[0] == 1 and eng.operator_info()[0] == 1          # multigrid + matrix-free operator
    assert big.nsteps == small.nsteps == 20
    sb, ss = np.array(big.sgl), np.array(small.sgl)
    eb, es = np.array(big.egl), np.array(small.egl)
    pb, ps = np.array(big.epgl), np.array(small.epgl)
    rel = np.array([s[1] for s in big.solver_stats])
    nfb = eng.solve_fallbacks()
    info = eng.indefinite_info()
    d_sgl = np.max(np.abs(sb - ss), axis=1) / np.max(np.abs(ss))
    print('config 5 at 2048^2, 20 load steps: %d sweeps, %d solves (%d completed by the fall-back solver, %s), %d PCG iterations\n'
          '  sgl_yy %s\n  niter 2048^2 %s\n  niter 512x64 %s\n  rel. difference of sgl per load step %s'
          % (big.n_sweeps, len(rel), nfb, info, sum(s[0] for s in big.solver_stats), np.round(sb[:, 1], 3).tolist(),
             list(big.niter), list(small.niter), ['%.1e' % v for v in d_sgl]))
    # K-iteration counts: a load step ends when no element's tangent moved by more than 1e-3 (model.py:1346-1355); where the
    # field is round-off-uniform the maximum over 4 M elements stays above that threshold longer than the maximum over 32 k
    # (profiles/r03b_config5_niter_vs_mesh.txt), so the counts are size independent only in the elastic steps and at first yield
    assert list(big.niter[:6]) == list(small.niter[:6])
    assert not np.any(big.co_nconv) and not np.any(small.co_nconv)   # (none in the reference's trace either)
    # the whole history against the 512 x 64 mesh (measured margin 2e-5 over 12 steps, tools/probes/cfg5_margins.py)
    assert np.max(np.abs(sb[:13] - ss[:13])) < 5e-5 * np.max(np.abs(ss))
    assert np.max(np.abs(sb - ss)) < 1e-4 * np.max(np.abs(ss))
    assert np.max(np.abs(eb - es)) < 1e-4 * np.max(np.abs(es))
    assert np.max(np.abs(pb - ps)) < 1e-4 * np.max(np.abs(es))
    # ... and against the reference's own trace of this schedule on 8 x 4 elements (the laminate solution is mesh independent to
    # ~1e-4: the trained SVC couples tension to small shear components, which the free edge sees)
    g = np.load(os.path.join(golden_dir, 'solve_configs.npz'))
    rs = g['cfg5_lam_8x4_sgl']
    assert int(g['cfg5_lam_8x4_nsteps']) == 20
    assert np.max(np.abs(sb - rs)) < 2e-4 * np.max(np.abs(rs))
    assert abs(sb[-1][1] - rs[-1][1]) < 1e-4 * rs[-1][1]                       # 144.13 (the reference: 144.1353)
    assert sb[5][1] > 134. and sb[-1][1] > sb[5][1]                             # the SVC phase has yielded (134.67 at step 5)
    ms = big._state('max_steps')
    assert np.sum(ms[svc_el] == 49) > 1000000                                   # ... on the 50-sub-step corrector
    # every linear solve reached the tolerance; the ones PCG could not finish (indefinite tangents, material.py:317-338)
    # were completed by the fall-back solver and are reported
    assert np.all(rel <= 1.0000001 * big.cg_rtol), rel.max()
    assert 0 < nfb <= 0.4 * len(rel)                                            # (48 of 254 in profiles/r02j_config5_full_solve.txt)
    assert 0 < info['solves'] <= nfb and info['by_gmres'] == info['solves'] and info['by_minres_surrogate'] == 0
