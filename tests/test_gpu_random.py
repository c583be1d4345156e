"""Randomised (seeded) parity: random orthotropic elastic matrices and Hill coefficients, GPU vs CPU oracle;
assembled matrix on a plastic state vs the oracle's element matrices; non-uniform laminate (several element
classes, Jacobi-PCG path) vs the oracle's sparse direct solve."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_material(rng):
    from oracle import oracle as O
    from pylabfea_amd import _lib
    E = rng.uniform(50e3, 400e3)
    nu = rng.uniform(0.15, 0.4)
    hh = E / ((1. + nu) * (1. - 2. * nu))
    CV = np.zeros((6, 6))
    CV[:3, :3] = nu * hh
    CV[[0, 1, 2], [0, 1, 2]] = (1. - nu) * hh
    CV[[3, 4, 5], [3, 4, 5]] = (0.5 - nu) * hh
    # orthotropic perturbation, kept symmetric positive definite
    P = rng.uniform(-0.1, 0.1, size=(3, 3)) * hh * 0.2
    CV[:3, :3] += P + P.T
    CV[[3, 4, 5], [3, 4, 5]] *= rng.uniform(0.8, 1.2, size=3)
    sy = rng.uniform(40., 400.)
    khard = rng.choice([0., rng.uniform(10., 5000.)])
    hill = rng.uniform(0.6, 1.5, size=6)
    # perfect plasticity + a Drucker pressure term makes the reference's scale-back iteration unstable for
    # percent-size strain steps (round-off is amplified through the 50 sub-steps): not a meaningful parity case
    drucker = rng.choice([0., rng.uniform(0., 0.2)]) if khard > 0. else 0.
    rec = _lib.pack_material(_lib.HILL6, CV, E=E, nu=nu, sy=sy, khard=khard, hill=hill, drucker=drucker)
    om = O.Material(kind=O.HILL6, E=E, nu=nu, sy=sy, khard=khard, hill=hill, drucker=drucker)
    return rec, om, CV, sy


def test_random_materials_response():
    from oracle import oracle as O
    from pylabfea_amd import _lib
    ctx = _lib.Context(0)
    rng = np.random.default_rng(2024)
    nflip = ntot = 0
    for trial in range(12):
        rec, om, CV, sy = random_material(rng)
        ctx.set_materials([rec])
        n = 3000
        d = rng.normal(size=(n, 6))
        seq = O.calc_seq(om, d)
        sig = d / seq[:, None] * sy * rng.uniform(0.0, 1.03, size=n)[:, None]
        deps = rng.normal(size=(n, 6)) * 10 ** rng.uniform(-5.5, -2.5, size=n)[:, None]
        epl = rng.normal(size=(n, 6)) * 2e-3 * (rng.uniform(size=n) < 0.5)[:, None]
        fy, so, dp, ct, ns = ctx.response(sig, epl, deps)
        fy2, so2, dp2, ct2, ns2 = O.response(om, CV, sig, epl, deps)
        ok = ns == ns2
        nflip += int(np.sum(~ok))
        ntot += n
        assert np.max(np.abs(so[ok] - so2[ok])) < 1e-8 * sy
        assert np.max(np.abs(dp[ok] - dp2[ok])) < 1e-10
        assert np.max(np.abs(ct[ok] - ct2[ok])) < 1e-6 * CV[0, 0]
        assert np.max(np.abs(ctx.seq(0, sig) - O.calc_seq(om, sig))) < 1e-9 * sy
    assert nflip <= 2e-4 * ntot      # inputs within round-off of a branch threshold may flip
    ctx.close()


def test_assembled_matrix_plastic_state():
    """K on a plastic state (random symmetric tangents) vs sum of the oracle's element matrices."""
    import scipy.sparse as sp
    import pylabfea_amd as FE
    from oracle import oracle as O
    from pylabfea_amd import _lib
    mat = FE.Material()
    mat.elasticity(E=200.e3, nu=0.3)
    mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    for ps in (False, True):
        fe = FE.Model(dim=2, planestress=ps)
        fe.geom([3.], LY=2.)
        fe.assign([mat])
        fe.mesh(NX=24, NY=16)
        eng = fe._ensure_engine()
        rng = np.random.default_rng(3)
        CV = fe._element_CV(mat)
        A = rng.normal(size=(fe.Nel, 6, 6)) * 5.e3
        D = CV[None] + A + A.transpose(0, 2, 1)
        eng.state_set(_lib.ST_ELSTIFF, D.reshape(fe.Nel, 36))
        eng.assemble()
        K = eng.get_csr()
        Kel = O.kel_batch(fe._lxy, fe._mat_id, fe.thick, ps, CV.reshape(1, 36), [mat.E], [mat.nu], D.reshape(-1, 36))
        dofs = np.stack((2 * fe._conn, 2 * fe._conn + 1), axis=2).reshape(fe.Nel, 8)
        Kref = sp.coo_matrix((Kel.ravel(), (np.repeat(dofs, 8, axis=1).ravel(), np.tile(dofs, (1, 8)).ravel())),
                             shape=K.shape).tocsr()
        diff = (K - Kref)
        assert abs(diff).max() < 1e-11 * abs(Kref).max()
        assert abs(K - K.T).max() < 1e-11 * abs(Kref).max()


@pytest.mark.parametrize('NX,NY,mg', [(13, 6, 0), (52, 8, 1), (26, 12, 1)])
def test_nonuniform_laminate_vs_oracle(NX, NY, mg):
    """Laminate with non-proportional sections (three element classes: dx = LS[i] / nes[i] differs from section to section,
    model.py:826-847) and two plastic materials against the oracle's sparse direct solve.  13 x 6: odd NX, no hierarchy
    (Jacobi-PCG); 52 x 8 (column widths within 1.17 of each other) and 26 x 12 (1.33): the exact matrix-free operator with
    per-column widths under the V-cycle of the uniform grid (round 5)"""
    import pylabfea_amd as FE
    from oracle.solve_ref import RefSolver

    def build():
        ma = FE.Material(num=1)
        ma.elasticity(E=200.e3, nu=0.3)
        ma.plasticity(sy=150., khard=500., sdim=6)
        mb = FE.Material(num=2)
        mb.elasticity(E=120.e3, nu=0.33)
        mb.plasticity(sy=90., hill=[0.8, 1.1, 1.3, 1., 0.9, 1.2], khard=300., sdim=6)
        fe = FE.Model(dim=2, planestress=True)
        fe.geom([2, 1, 2, 1, 2], LY=4.)
        fe.assign([ma, mb, ma, mb, ma])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.004 * fe.leny, 'disp')
        fe.mesh(NX=NX, NY=NY)
        return fe
    fe = build()
    dx = fe._grid['dx_col']
    assert np.max(dx) / np.min(dx) > 1.1
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=5)
    assert fe._engine.precond_info()[0] == mg and fe._engine.operator_info()[0] == 1   # (matrix-free with per-column widths)
    ref = RefSolver(build()).solve(min_step=5)
    assert fe.nsteps == ref.nsteps and list(fe.niter) == list(ref.niter)
    s = np.max(np.abs(ref.sig))
    assert np.max(np.abs(fe.u - ref.u)) < 1e-6 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 1e-6 * s
    assert np.max(np.abs(fe._state('epl') - ref.epl)) < 1e-6 * np.max(np.abs(ref.eps))
    assert np.max(np.abs(fe.sgl - ref.sgl)) < 1e-6 * s


def test_composite_j2_svc_laminate_vs_oracle(golden_dir):
    """BASELINE config 5 in small: laminate [2,1,2,1,2] of a J2 phase and an SVC phase (the 1585-vector SVC of config 4)
    on a uniform grid -> multigrid + matrix-free operator, analytic thread-per-element kernels and wave-per-element SVC
    kernels in one sweep, material jumps across the coarse levels; against the oracle's sparse direct solve."""
    import pylabfea_amd as FE
    from oracle.solve_ref import RefSolver
    z = np.load(os.path.join(golden_dir, 'svc_hill.npz'))

    def build():
        ma = FE.Material(num=1)
        ma.elasticity(E=200.e3, nu=0.3)
        ma.plasticity(sy=150., khard=500., sdim=6)
        mb = FE.Material(name='ML-Hill', num=2)
        mb.elasticity(CV=z['par_CV'])
        mb.plasticity(sy=float(z['par_sy']), sdim=6)
        mb.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']),
                   float(z['par_scale_seq']))
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([2, 1, 2, 1, 2], LY=4.)
        fe.assign([ma, mb, ma, mb, ma])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.0012 * fe.leny, 'disp')
        fe.mesh(NX=16, NY=8)
        return fe
    fe = build()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=4)
    eng = fe._engine
    assert eng.precond_info()[0] == 1 and eng.operator_info()[0] == 1
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = RefSolver(build()).solve(min_step=4)
    assert fe.nsteps == ref.nsteps and list(fe.niter) == list(ref.niter)
    assert np.max(fe._state('epl')[fe._mat_id == 1]) > 0.   # the SVC phase yields
    assert np.max(fe._state('max_steps')) == 49             # and runs the 50-sub-step corrector
    s = np.max(np.abs(ref.sig))
    assert np.max(np.abs(fe.u - ref.u)) < 1e-6 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 1e-6 * s
    assert np.max(np.abs(fe._state('epl') - ref.epl)) < 1e-6 * np.max(np.abs(ref.eps))
    assert np.max(np.abs(fe.sgl - ref.sgl)) < 1e-6 * s


def test_odd_coarse_grid_chebyshev_vs_oracle():
    """50 x 30 elements halve once to 25 x 15 (odd): the coarsest level has 832 DOFs, no dense inverse -> fixed-degree
    Jacobi-Chebyshev coarse solver inside the V-cycle; plastic tension against the oracle's sparse direct solve"""
    import pylabfea_amd as FE
    from oracle.solve_ref import RefSolver

    def build():
        mat = FE.Material()
        mat.elasticity(E=200.e3, nu=0.3)
        mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
        soft = FE.Material(num=2)
        soft.elasticity(E=1.e3, nu=0.27)
        fe = FE.Model(dim=2, planestress=False)
        fe.geom(sect=2, LX=5., LY=3.)
        fe.assign([mat, soft])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.002 * fe.leny, 'disp')
        el = np.ones((50, 30))
        el[20:30, 10:20] = 2
        fe.mesh(elmts=el, NX=50, NY=30)
        return fe
    fe = build()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=5)
    kind, levels = fe._engine.precond_info()
    assert kind == 1 and levels == 2
    assert max(s[0] for s in fe.solver_stats) < 60      # multigrid-type iteration counts, not Jacobi's hundreds
    ref = RefSolver(build()).solve(min_step=5)
    assert fe.nsteps == ref.nsteps and list(fe.niter) == list(ref.niter)
    s = np.max(np.abs(ref.sig))
    assert np.max(fe._state('epl')) > 0.
    assert np.max(np.abs(fe.u - ref.u)) < 1e-6 * np.max(np.abs(ref.u))
    assert np.max(np.abs(fe._state('sig') - ref.sig)) < 1e-6 * s
    assert np.max(np.abs(fe.sgl - ref.sgl)) < 1e-6 * s


# tangent of an SVC element after the least-squares correction step of Material.response (material.py:324-338), as met in
# BASELINE config 5 at full size (upper triangle by rows): the yy entry is NEGATIVE -- such tangents are what the reference
# hands to its LU solver
BAD_TANGENT_21 = [3.06119e+05, 2.30987e+05, 2.44365e+05, -7.10713e+02, 8.16221e+02, -3.89722e+01, -5.30574e+05, -2.01886e+05,
                  8.08981e+02, -9.29004e+02, 4.43106e+01, 2.03386e+05, -4.23220e+01, 4.86228e+01, -2.33304e+00, 5.81516e+04,
                  1.14637e+01, -5.47451e-01, 5.81484e+04, 6.34734e-01, 5.81615e+04]


@pytest.mark.parametrize('solver', ['sqmr', 'surrogate', 'minres', 'gmres'])
@pytest.mark.parametrize('nx,ny,mg', [(64, 64, True), (48, 24, True), (13, 6, False)])
def test_indefinite_tangent_solved_like_the_reference(nx, ny, mg, solver, monkeypatch):
    """A stiffness matrix with negative eigenvalues: PCG meets a direction of negative curvature and the solve is completed
    by right-preconditioned GMRES(400) (default), by SQMR (round 5: short recurrences, a symmetric indefinite V-cycle is
    admissible), by preconditioned MINRES with the V-cycle rebuilt on the SPD surrogate
    operator (every indefinite element matrix shifted by its most negative eigenvalue; needs no GMRES here), or by MINRES
    with the V-cycle of the indefinite operator itself (GMRES takes over when that is not positive definite); the solution
    must be the one a direct solver (the reference's numpy.linalg.solve) finds."""
    monkeypatch.setenv('PLFX_INDEFINITE_SOLVER', solver)
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import pylabfea_amd as FE
    from pylabfea_amd import _lib
    mat = FE.Material()
    mat.elasticity(E=151220., nu=0.3)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * ny / nx)
    fe.assign([mat])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.002 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    eng = fe._ensure_engine()
    assert (eng.precond_info()[0] == 1) == mg
    CV = fe._element_CV(mat)
    D = np.tile(CV, (fe.Nel, 1, 1))
    bad = np.zeros((6, 6))
    bad[np.triu_indices(6)] = BAD_TANGENT_21
    bad = bad + bad.T - np.diag(np.diag(bad))
    # isolated elements, as in config 5 (2 of 4.2 million): the diagonal of K stays positive, K itself does not stay definite
    for cx, cy in ((nx // 3, ny // 2), (2 * nx // 3, ny // 4), (nx // 2, (3 * ny) // 4)):
        D[cx * ny + cy] = bad
    eng.state_set(_lib.ST_ELSTIFF, D.reshape(fe.Nel, 36))
    eng.assemble()
    z, d = np.zeros(2), np.array([0., 0.002 * fe.leny])
    presc, first, w, fext = fe._bc_data(z, z, z, d, None)
    eng.apply_bc(presc, first, w, fext)
    n0 = eng.solve_fallbacks()
    it, rr, ok = eng.solve(1e-10, 20000, False)
    assert ok and rr <= 1e-10
    assert eng.solve_fallbacks() == n0 + 1            # PCG gave up, MINRES finished the solve
    info = eng.indefinite_info()
    assert info['solves'] == 1
    if solver == 'surrogate' and mg:
        # the three planted element matrices were found and replaced, MINRES with the SPD V-cycle finished: no GMRES
        assert info == dict(solves=1, by_minres_surrogate=1, by_gmres=0, surrogates_built=1, elements_shifted=3)
        # a second system on the same operator (other boundary values) reuses the surrogate hierarchy
        presc2, first2, w2, fext2 = fe._bc_data(z, z, z, 0.5 * d, None)
        eng.apply_bc(presc2, first2, w2, fext2)
        it2, rr2, ok2 = eng.solve(1e-10, 20000, False)
        assert ok2 and rr2 <= 1e-10
        du2 = eng.state_get(_lib.ST_DU)
        assert eng.indefinite_info()['surrogates_built'] == 1 and eng.indefinite_info()['by_gmres'] == 0
        eng.apply_bc(presc, first, w, fext)
        it, rr, ok = eng.solve(1e-10, 20000, False)
        assert ok and rr <= 1e-10
        assert np.max(np.abs(du2 - 0.5 * eng.state_get(_lib.ST_DU))) < 1e-7 * np.max(np.abs(du2))   # linear system: half the load
        # a new assembly (tangents unchanged in content, but rewritten) goes back to the true operator first
        eng.state_set(_lib.ST_ELSTIFF, D.reshape(fe.Nel, 36))
        eng.assemble()
        eng.apply_bc(presc, first, w, fext)
        it, rr, ok = eng.solve(1e-10, 20000, False)
        assert ok and eng.indefinite_info()['surrogates_built'] == 2
    elif solver == 'gmres':
        assert info['by_gmres'] == 1 and info['surrogates_built'] == 0
    elif solver == 'sqmr':
        # CG-like recurrences that do not need definiteness, V-cycle of the operator as it is; no GMRES on this problem
        assert eng.sqmr_info() == 1 and info['by_gmres'] == 0 and info['surrogates_built'] == 0
    du = eng.state_get(_lib.ST_DU)
    K = eng.get_csr().tocsr()
    free = np.setdiff1d(np.arange(fe.Ndof), presc)
    wfull = np.zeros(fe.Ndof)
    wfull[presc] = w
    rhs = (-(K @ wfull) if fext is None else fext - K @ wfull)[free]
    Kff = K[free][:, free].tocsc()
    assert Kff.diagonal().min() > 0.
    lam = spla.eigsh(Kff, k=1, which='SA', tol=1e-4, return_eigenvectors=False)
    assert lam[0] < -1e-3 * Kff.diagonal().max()       # really indefinite
    ref = spla.spsolve(Kff, rhs)
    assert np.max(np.abs(du[free] - ref)) < 1e-6 * np.max(np.abs(ref))
    assert np.array_equal(du[presc], first)
