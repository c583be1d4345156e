"""Edge cases and error behaviour of the C-ABI on the GPU (empty batches, bad arguments, call order,
degenerate inputs), and of the façade (shapes, un-meshed model, unsupported options)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from pylabfea_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def j2_record(kind=None):
    from pylabfea_amd import _lib
    CV = np.zeros((6, 6))
    CV[:3, :3] = 115384.6153846154
    CV[[0, 1, 2], [0, 1, 2]] = 269230.7692307692
    CV[[3, 4, 5], [3, 4, 5]] = 76923.07692307692
    return _lib.pack_material(_lib.HILL6 if kind is None else kind, CV, E=200.e3, nu=0.3, sy=150., khard=500.), CV


def test_call_order_and_bad_arguments(ctx):
    from pylabfea_amd import _lib
    with pytest.raises(_lib.PlfxError):           # batch call before set_materials
        ctx.seq(0, np.zeros((1, 6)))
    rec, CV = j2_record()
    ctx.set_materials([rec])
    with pytest.raises(_lib.PlfxError):           # material index out of range
        ctx.seq(3, np.zeros((1, 6)))
    with pytest.raises(_lib.PlfxError):           # mat_id out of range
        ctx.response(np.zeros((2, 6)), np.zeros((2, 6)), np.zeros((2, 6)), mat_id=[0, 5])
    with pytest.raises(_lib.PlfxError):           # assemble before set_mesh
        ctx.assemble()
    bad = _lib.pack_material(99, CV)
    with pytest.raises(_lib.PlfxError):           # unknown kind
        ctx.set_materials([bad])
    asym = np.array(CV)
    asym[0, 1] += 1.
    with pytest.raises(_lib.PlfxError):           # non-symmetric CV
        ctx.set_materials([_lib.pack_material(_lib.HILL6, asym, E=1., nu=0.3, sy=1.)])
    ctx.set_materials([rec])
    conn = np.array([[0, 1, 2, 7]])               # node id out of range
    with pytest.raises(_lib.PlfxError):
        ctx.set_mesh(conn, [0], [[1., 1.]], 4, 1., False)
    conn = np.array([[0, 1, 2, 3]])
    ctx.set_mesh(conn, [0], [[1., 1.]], 4, 1., False)
    with pytest.raises(_lib.PlfxError):           # solve before apply_bc
        ctx.solve()
    with pytest.raises(_lib.PlfxError):           # grid that does not match the mesh
        ctx.set_grid(2, 2)
    ctx.assemble()
    with pytest.raises(_lib.PlfxError):           # prescribed DOF out of range
        ctx.apply_bc([99], [0.], [0.])


def test_empty_and_degenerate_batches(ctx):
    rec, CV = j2_record()
    ctx.set_materials([rec])
    assert ctx.seq(0, np.zeros((0, 6))).shape == (0,)
    fy, so, dp, ct, ns = ctx.response(np.zeros((0, 6)), np.zeros((0, 6)), np.zeros((0, 6)))
    assert fy.shape == (0,) and ct.shape == (0, 36)
    # zero stress, zero increment: elastic step, tangent = CV, nothing changes
    fy, so, dp, ct, ns = ctx.response(np.zeros((3, 6)), np.zeros((3, 6)), np.zeros((3, 6)))
    assert np.all(so == 0.) and np.all(dp == 0.) and np.all(ns == 0)
    assert np.allclose(ct[0].reshape(6, 6), CV) and np.allclose(fy, -150.)
    # huge increments from the stress-free state (5 % strain in one step): finite, and equal to the oracle
    from oracle import oracle as O
    deps = np.zeros((4, 6))
    deps[0, 1] = 0.05
    deps[1, 5] = 0.08
    deps[2, :3] = [0.05, -0.02, 0.01]
    deps[3, [0, 5]] = [1e-9, -1e-9]           # tiny
    fy, so, dp, ct, ns = ctx.response(np.zeros((4, 6)), np.zeros((4, 6)), deps)
    om = O.Material(kind=O.HILL6, E=200.e3, nu=0.3, sy=150., khard=500.)
    fy2, so2, dp2, ct2, ns2 = O.response(om, CV, np.zeros((4, 6)), np.zeros((4, 6)), deps)
    assert np.array_equal(ns, ns2) and np.all(np.isfinite(so)) and np.all(np.isfinite(ct))
    assert np.max(np.abs(so - so2)) < 1e-9 * 150. and np.max(np.abs(dp - dp2)) < 1e-12
    assert np.max(np.abs(ct - ct2)) < 1e-7 * CV[0, 0]


def test_single_element_model_and_force_bc():
    """1x1 mesh (no multigrid hierarchy -> Jacobi-PCG), force-controlled loading on the right edge."""
    import pylabfea_amd as FE
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    fe = FE.Model(dim=2, planestress=True)
    fe.geom([2.], LY=2.)
    fe.assign([m])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(100., 'force')      # total force 100 on an edge of area 2 -> sig_xx = 50
    fe.bctop(0., 'force')
    fe.mesh(NX=1, NY=1)
    fe.solve()
    assert fe._engine.precond_info()[0] == 0
    assert abs(fe.element[0].sig[0] - 50.) < 1e-9 and abs(fe.element[0].sig[1]) < 1e-9
    assert abs(fe.element[0].eps[0] - 50. / 200.e3) < 1e-14
    assert abs(fe.glob['sbc1'] - 50.) < 1e-9


def test_large_mesh_with_odd_dimensions_gets_a_multigrid_hierarchy():
    """401 x 399 elements (Model.mesh accepts any NX, NY, model.py:758-952): no exact halving; since round 5 every level covers
    the grid with cells of one size and a wider last column / row (three children), smoothed with an area-scaled diagonal
    (DESIGN 10.7).  The preconditioner only changes the iteration count: results equal the Jacobi-PCG / assembled-operator path
    of the same library, with a fraction of its iterations -- and at most 1.5x those of the even neighbour 400 x 400."""
    import warnings
    import pylabfea_amd as FE

    def run(precond, operator, nx=401, ny=399):
        m = FE.Material()
        m.elasticity(E=200.e3, nu=0.3)
        m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
        fe = FE.Model(dim=2, planestress=False)
        fe.precond, fe.operator = precond, operator
        fe.geom([4.], LY=4. * ny / nx)
        fe.assign([m])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.004 * fe.leny, 'disp')
        fe.mesh(NX=nx, NY=ny)
        fe._max_load_steps = 4
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=6)
        return fe, sum(q[0] for q in fe.solver_stats)

    a, ita = run(None, None)
    assert a._engine.precond_info()[0] == 1 and a._engine.precond_info()[1] >= 4 and a._engine.operator_info()[0] == 1
    b, itb = run(0, 0)
    assert b._engine.precond_info()[0] == 0
    assert a.nsteps == b.nsteps and list(a.niter) == list(b.niter) and np.max(np.abs(a._state('epl'))) > 0.
    assert ita < 0.1 * itb
    assert np.max(np.abs(np.asarray(a.sgl) - np.asarray(b.sgl))) < 1e-7 * np.max(np.abs(b.sgl))
    assert np.max(np.abs(a.u - b.u)) < 1e-7 * np.max(np.abs(b.u))
    assert np.max(np.abs(a._state('sig') - b._state('sig'))) < 1e-6 * np.max(np.abs(b._state('sig')))   # (both solves stop at rtol 1e-10)
    e, ite = run(None, None, 400, 400)
    assert e._engine.precond_info()[0] == 1 and list(e.niter) == list(a.niter)
    assert ita <= 1.5 * ite + 4


def test_vcycle_is_symmetric_and_keeps_uniform_fields_uniform():
    """plfx_precond_apply (one V-cycle on a host vector) on a mesh with an odd number of rows: the cycle is a symmetric operator
    (<a, B b> = <b, B a>), and it maps a residual that is uniform along y onto a correction that is uniform along y -- the
    invariant subspace the homogeneous workload's solves live in; the area-scaled smoothing diagonal of the levels with a row of
    another height keeps it (with the true diagonal the answer varies by tens of per cent from row to row)."""
    import warnings
    import pylabfea_amd as FE
    nx, ny = 96, 95
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * ny / nx)
    fe.assign([m])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.005 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    fe._max_load_steps = 9
    os.environ['PLFX_MG_ODD'] = '1'
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=50)
    finally:
        del os.environ['PLFX_MG_ODD']
    eng = fe._engine
    assert eng.precond_info()[0] == 1 and np.max(np.abs(fe._state('epl'))) > 0.
    free = np.zeros(fe.Ndof)
    free[np.asarray(fe.free_dofs())] = 1.
    rng = np.random.default_rng(3)
    a, b = free * rng.standard_normal(fe.Ndof), free * rng.standard_normal(fe.Ndof)
    s1, s2 = a @ eng.precond_apply(b), b @ eng.precond_apply(a)
    assert abs(s1 - s2) < 1e-12 * max(abs(s1), abs(s2))
    g = np.zeros((nx + 1, ny + 1, 2))
    g[:, :, 0] = np.arange(nx + 1)[:, None] / nx          # u_x = X / L, u_y = 0: what a tangent update changes
    e = free * g.ravel()
    r = free * eng.matvec(e)
    z = (free * eng.precond_apply(r)).reshape(nx + 1, ny + 1, 2)
    zx = z[:, 1:ny, 0]                                    # interior rows (the two edge rows carry the shear coupling of the edges)
    spread = np.max(np.abs(zx - zx[:, [ny // 2]]), axis=1)
    assert np.max(spread) < 1e-8 * np.max(np.abs(zx))     # (measured: 1.3e-10)


def test_large_non_proportional_laminate_gets_a_multigrid_hierarchy():
    """320 x 256 elements in five sections of thickness 3:1:2:1:2 -- nes = round(NX LS / lenx) leaves the sections with
    dx = 3/106, 1/36, 2/71 ... (model.py:826-847), a mesh that ran Jacobi-PCG on the assembled operator until round 5.  Now:
    exact matrix-free operator with per-column widths (KOp::colr) under the V-cycle of the uniform grid.
    Results equal the block-ELL operator (exact class shapes) + Jacobi-PCG path, with a fraction of its iterations."""
    import warnings
    import pylabfea_amd as FE

    def run(precond, operator):
        ma = FE.Material(num=1)
        ma.elasticity(E=200.e3, nu=0.3)
        ma.plasticity(sy=150., khard=500., sdim=6)
        mb = FE.Material(num=2)
        mb.elasticity(E=120.e3, nu=0.33)
        mb.plasticity(sy=90., hill=[0.8, 1.1, 1.3, 1., 0.9, 1.2], khard=300., sdim=6)
        fe = FE.Model(dim=2, planestress=True)
        fe.precond, fe.operator = precond, operator
        fe.geom([3, 1, 2, 1, 2], LY=9. * 256 / 320)
        fe.assign([ma, mb, ma, mb, ma])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.004 * fe.leny, 'disp')
        fe.mesh(NX=320, NY=256)
        fe._max_load_steps = 9
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=20)
        return fe, sum(q[0] for q in fe.solver_stats)

    a, ita = run(None, None)
    dx = a._grid['dx_col']
    assert 1.005 < np.max(dx) / np.min(dx) < 1.1 and len(np.unique(np.round(dx, 12))) == 3
    assert a._engine.precond_info()[0] == 1 and a._engine.precond_info()[1] >= 5 and a._engine.operator_info()[0] == 1
    b, itb = run(0, 0)
    assert b._engine.precond_info()[0] == 0 and b._engine.operator_info()[0] == 0
    assert a.nsteps == b.nsteps and np.max(np.abs(a._state('epl'))) > 0.
    assert ita < 0.1 * itb
    assert np.max(np.abs(np.asarray(a.sgl) - np.asarray(b.sgl))) < 1e-7 * np.max(np.abs(b.sgl))
    assert np.max(np.abs(a.u - b.u)) < 1e-7 * np.max(np.abs(b.u))
    assert np.max(np.abs(a._state('sig') - b._state('sig'))) < 1e-6 * np.max(np.abs(b._state('sig')))


def test_two_solution_initial_guess(monkeypatch):
    """a warm-started multigrid solve on a mesh of >= 16384 nodes is answered by x + alpha d (d = the
    difference of the last two solutions, 0 <= alpha <= 1 the residual-minimal step) when -- and only when -- that vector
    satisfies the tolerance as it is; every solve that iterates starts from x exactly as with PLFX_PREDICT=0.  Same load
    steps / K-iterations, fields equal to solver tolerance, fewer PCG iterations with it; meshes below the size gate never
    try it."""
    import warnings
    import pylabfea_amd as FE

    def run(on, n=256):
        monkeypatch.setenv('PLFX_PREDICT', '1' if on else '0')
        m = FE.Material()
        m.elasticity(E=200.e3, nu=0.3)
        m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([4.], LY=4.)
        fe.assign([m])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.005 * fe.leny, 'disp')
        fe.mesh(NX=n, NY=n)
        fe._max_load_steps = 16
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=50)
        return fe, [q[0] for q in fe.solver_stats], fe._engine.predict_info()
    a, ita, pa = run(False)
    b, itb, pb = run(True)
    s, _, ps = run(True, 64)
    monkeypatch.delenv('PLFX_PREDICT')
    assert pa == (0, 0, 0) and pb[0] >= 5 and ps == (0, 0, 0)
    assert a.nsteps == b.nsteps and list(a.niter) == list(b.niter) and len(ita) == len(itb)
    assert sum(itb) < sum(ita)
    # an accepted start costs no iteration; a rejected one leaves the plain warm start
    assert sum(1 for q in itb if q == 0) >= sum(1 for q in ita if q == 0) + pb[0] - 1
    assert np.max(np.abs(np.asarray(a.sgl) - np.asarray(b.sgl))) < 1e-8 * np.max(np.abs(a.sgl))
    assert np.max(np.abs(a.u - b.u)) < 1e-8 * np.max(np.abs(a.u))


def _predict_on_off(monkeypatch, build, **kw):
    import warnings
    out = []
    for on in (False, True):
        monkeypatch.setenv('PLFX_PREDICT', '1' if on else '0')
        fe = build()
        for k, v in kw.items():
            if k.startswith('_'):
                setattr(fe, k, v)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(**{k: v for k, v in kw.items() if not k.startswith('_')})
        out.append((fe, [q[0] for q in fe.solver_stats], fe._engine.predict_info()))
    monkeypatch.delenv('PLFX_PREDICT')
    return out


def test_interpolated_start_through_the_onset_of_yielding_1024(monkeypatch):
    """the bench mesh (1024 x 1024 Hill) through the onset of yielding (load steps 1..12, first yield in
    step 5), interpolated start on against off: same load steps / K-iterations, u to 1e-9 relative.  Solves whose
    interpolated start fails the tolerance test iterate from the plain warm start (the acceptance rule has no tuned constant).
    The stresses are derivatives of u over one element (h = L / 1024): two solutions of the same systems to the same residual
    tolerance 1e-10 differ by ~1e-10 in u and by up to ~1e3 x that in its gradient (measured 4.5e-10 / 2.5e-8) -- the bar on
    sig is 1e-7, an order below the parity bar of the fields (1e-6)."""
    from test_gpu_model import make_material, tension_model
    (a, ita, pa), (b, itb, pb) = _predict_on_off(monkeypatch, lambda: tension_model(make_material('hill6'), 1024, 0.005),
                                                 min_step=50, _max_load_steps=12)
    assert pa == (0, 0, 0) and pb[0] > 0
    assert a.nsteps == b.nsteps and list(a.niter) == list(b.niter) and len(ita) == len(itb)
    assert np.max(np.abs(a._state('epl'))) > 0.
    du = np.max(np.abs(a.u - b.u)) / np.max(np.abs(a.u))
    ds = np.max(np.abs(a._state('sig') - b._state('sig'))) / np.max(np.abs(a._state('sig')))
    print('1024^2 steps 1..12, interpolated start on vs off: u %.2e sig %.2e; PCG iterations %d vs %d; accepted/skipped/rejected %s'
          % (du, ds, sum(itb), sum(ita), pb))
    assert du < 1e-9 and ds < 1e-7
    assert sum(itb) <= sum(ita)


def test_interpolated_start_config5_512x64(monkeypatch, golden_dir):
    """... and on config 5's laminate (J2 + SVC trained on Barlat / Goss) at 512 x 64 through all 20 load steps: the long and
    the indefinite solves of the plastic SVC phase are exactly the ones of PLFX_PREDICT=0 unless a start was accepted before them."""
    from test_gpu_configs import laminate_cfg5
    (a, ita, pa), (b, itb, pb) = _predict_on_off(monkeypatch, lambda: laminate_cfg5(golden_dir, 512, 64), min_step=20)
    assert pa == (0, 0, 0)
    assert a.nsteps == b.nsteps == 20 and list(a.niter[:8]) == list(b.niter[:8])
    du = np.max(np.abs(a.u - b.u)) / np.max(np.abs(a.u))
    ds = np.max(np.abs(a._state('sig') - b._state('sig'))) / np.max(np.abs(a._state('sig')))
    dg = np.max(np.abs(np.asarray(a.sgl) - np.asarray(b.sgl))) / np.max(np.abs(a.sgl))
    print('config 5 at 512x64, interpolated start on vs off: u %.2e sig %.2e sgl %.2e; PCG iterations %d vs %d; accepted/skipped/rejected %s; niter %s vs %s'
          % (du, ds, dg, sum(itb), sum(ita), pb, list(b.niter), list(a.niter)))
    assert du < 1e-9 and ds < 1e-7 and dg < 1e-9


def test_facade_errors():
    import pylabfea_amd as FE
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    with pytest.raises(AttributeError):
        m.response(np.zeros(6), np.zeros(6), np.zeros(6), m.CV)     # elastic material has no flow rule
    m.plasticity(sy=100., sdim=6)
    with pytest.raises(ValueError):
        m.response(np.zeros((2, 6)), np.zeros(6), np.zeros(6), m.CV)
    with pytest.raises(ValueError):
        m.calc_fgrad(np.zeros(6), epl=np.zeros(3))
    with pytest.raises(TypeError):
        m.calc_seq(np.zeros(5))
    with pytest.raises(NotImplementedError):
        m.plasticity(sy=100., lhs=[0.1, 0.1, 0.1])
    with pytest.raises(AttributeError):
        m.ML_full_yf(np.ones(6))
    fe = FE.Model(dim=2)
    fe.geom([1.], LY=1.)
    fe.assign([m])
    with pytest.raises(NotImplementedError):
        fe.mesh(NX=2, NY=2, SF=2)
    with pytest.raises(NotImplementedError):
        FE.Model(dim=1)


@pytest.mark.gpu
def test_structured_mesh_description_equals_explicit_index_arrays():
    """plfx_set_mesh_structured (the library writes Model.mesh's index arrays from the grid description, model.py:893, :935-948)
    against plfx_set_mesh with the arrays the facade materialises: bit-identical solves -- homogeneous Hill and a laminate whose
    sections have different element widths (block-ELL operator there)."""
    import warnings
    import pylabfea_amd as FE

    def run(kind, explicit):
        a = FE.Material(num=1)
        a.elasticity(E=200.e3, nu=0.3)
        a.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
        fe = FE.Model(dim=2, planestress=False)
        if kind == 'hill':
            fe.geom([4.], LY=4.)
            fe.assign([a])
        else:
            b = FE.Material(num=2)
            b.elasticity(E=100.e3, nu=0.35)
            fe.geom([2., 1., 2.5], LY=4.)
            fe.assign([a, b, a])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.003 * fe.leny, 'disp')
        fe.mesh(NX=16, NY=8)
        fe._explicit_mesh = explicit
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve()
        return fe

    for kind in ('hill', 'laminate'):
        s, e = run(kind, False), run(kind, True)
        assert s.nsteps == e.nsteps and list(s.niter) == list(e.niter)
        assert np.array_equal(s.u, e.u) and np.array_equal(s.f, e.f) and np.array_equal(s._state('sig'), e._state('sig'))
        assert s._engine.precond_info() == e._engine.precond_info() and s._engine.operator_info() == e._engine.operator_info()


def test_setup_pass_column_walk_variants_are_bit_identical():
    """k_grid_setup<SRC, COLS>: one column per thread on small levels, a four-column walk from 2^19 nodes.  Both
    forms (and the eight-column one kept as a knob) must give the same bits -- diagonal, operator snapshot, level-1 generators --
    on every kind of level: plain, odd-size (area-scaled diagonal) and per-column widths.  PLFX_SETUP_COLS is read once per process:
    tools/probes/lib_ab.py's child runs the three solves under each setting; its line per case carries a digest of u, sig, epl,
    sgl, egl and the PCG iteration count of every solve."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, 'tools', 'probes', 'lib_ab.py')
    res = {}
    for cols in ('1', '4', '8'):
        env = dict(os.environ, PLFX_SETUP_COLS=cols)
        out = subprocess.run([sys.executable, script, '--child'], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        res[cols] = [ln for ln in out.stdout.splitlines() if ln.count('|') == 2]
        assert len(res[cols]) == 3
    assert res['1'] == res['4'] == res['8']
    # the predictor solve that waits for its first test before it speculates changes the order of launches only
    out = subprocess.run([sys.executable, script, '--child'], env=dict(os.environ, PLFX_WAIT_FIRST='0'), capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert [ln for ln in out.stdout.splitlines() if ln.count('|') == 2] == res['1']
