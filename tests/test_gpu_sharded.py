"""Multi-rank runs with 2 to 4 ranks on ONE GPU (replicated-solve mode and the strip-local engine): every rank is a process with its own libplfx context on cuda:0 and owns
one x-strip of elements; the collectives (stiffness generators after a sweep, sweep flags, calc_scf statistics,
calc_global sums) go through the host-staged transport (plfx_comm_init_callback) over a gloo process group, because RCCL
refuses two ranks on one device.  Everything else -- strip ownership inside the kernels, zeroing of the foreign
generators, the replicated solve, the native load step -- is the code that runs with RCCL on N GPUs.
Reference: traces of the reference's solve (tests/golden/solve.npz) and the single-rank run of the same model."""
import os
import socket
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def collect(q, procs, world, limit=900.):
    """results of the workers; fails as soon as one of them has died without reporting (instead of waiting out the limit)"""
    import queue
    import time
    res, t0 = {}, time.time()
    while len(res) < world:
        try:
            r, d = q.get(timeout=2.)
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 >= limit:
                for p in procs:          # peers of a dead rank wait in a collective for ever: end the processes we started
                    if p.is_alive():
                        p.terminate()
                raise AssertionError('worker exit codes %s' % dead if dead else 'workers did not report within %g s' % limit)
            continue
        if isinstance(d, str):
            for p in procs:
                if p.is_alive():
                    p.terminate()
            raise AssertionError(d)
        res[r] = d
    return res


def build(case, golden_dir):
    import pylabfea_amd as FE
    mat = FE.Material()
    mat.elasticity(E=200.e3, nu=0.3)
    mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2, planestress=False)
    if case == 'hill6_12':          # golden trace of the reference
        fe.geom([4.], LY=4.)
        fe.assign([mat])
        n, eps, ms = 12, 0.003, 8
    elif case == 'laminate_np':     # non-proportional laminate: per-column widths, multigrid of the uniform grid
        mb = FE.Material(num=2)
        mb.elasticity(E=120.e3, nu=0.33)
        mb.plasticity(sy=90., hill=[0.8, 1.1, 1.3, 1., 0.9, 1.2], khard=300., sdim=6)
        fe.geom([2, 1, 2, 1, 2], LY=4.)
        fe.assign([mat, mb, mat, mb, mat])
        n, eps, ms = 52, 0.003, 6
    elif case == 'inclusion':       # heterogeneous: strips see different branches
        soft = FE.Material(num=2)
        soft.elasticity(E=1.e3, nu=0.27)
        fe.geom(sect=2, LX=4., LY=4.)
        fe.assign([mat, soft])
        n, eps, ms = 24, 0.002, 6
    else:                           # J2 + SVC laminate: wave-per-element SVC kernels on strips
        z = np.load(os.path.join(golden_dir, 'svc_shear.npz'))
        mb = FE.Material(name='ML', num=2)
        mb.elasticity(CV=z['par_CV'])
        mb.plasticity(sy=float(z['par_sy']), sdim=6)
        mb.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']),
                   float(z['par_scale_seq']))
        fe.geom([1, 1, 1, 1], LY=2.)
        fe.assign([mat, mb, mat, mb])
        n, eps, ms = 16, 0.0015, 4
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(eps * fe.leny, 'disp')
    if case == 'inclusion':
        el = np.ones((n, n))
        el[8:16, 8:16] = 2
        fe.mesh(elmts=el, NX=n, NY=n)
    elif case == 'hill6_12':
        fe.mesh(NX=n, NY=n)
    elif case == 'laminate_np':
        fe.mesh(NX=n, NY=8)
    else:
        fe.mesh(NX=n, NY=8)
    return fe, ms


def _worker(rank, world, port, case, golden_dir, q, mode='replicated', level=None):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pylabfea_amd as FE
        fe, ms = build(case, golden_dir)
        fe.distribute(rank, world, None, host_allreduce=FE.host_transport(dist, rank, world), mode=mode, coarse_level=level)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=ms)
        eng = fe._engine
        assert eng.comm_info() == (rank, world, True)
        e0, e1 = fe._e0, fe._e1
        strip = fe._strip
        q.put((rank, dict(nsteps=fe.nsteps, niter=list(fe.niter), u=fe.u, f=fe.f, sgl=fe.sgl, e0=e0, e1=e1,
                          sig=fe._state('sig')[e0:e1], epl=fe._state('epl')[e0:e1], native=fe._native_step and fe._dev_coll,
                          strip=strip, strip_info=eng.strip_info(), its=[s[0] for s in fe.solver_stats], fallbacks=eng.solve_fallbacks(),
                          glob={k: v for k, v in fe.glob.items() if np.ndim(v) == 0 and v is not None})))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, 'ERROR: ' + traceback.format_exc()))
        raise exc
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case,world', [('hill6_12', 2), ('inclusion', 3), ('laminate_svc', 2), ('laminate_np', 2)])
def test_sharded_ranks_on_one_gpu(golden_dir, case, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = collect(q, procs, world, 600.)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-rank run of the same model in this process
    fe, ms = build(case, golden_dir)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=ms)
    sig1, epl1 = fe._state('sig'), fe._state('epl')
    assert np.max(epl1) > 0.
    if case == 'laminate_np':
        assert fe._engine.precond_info()[0] == 1 and np.ptp(fe._grid['dx_col']) > 0.01 * np.mean(fe._grid['dx_col'])
    for r in range(world):
        d = res[r]
        assert d['native']                                    # the native load step ran with device-side collectives
        assert d['nsteps'] == fe.nsteps and d['niter'] == list(fe.niter)
        assert np.max(np.abs(d['u'] - fe.u)) <= 1e-9 * np.max(np.abs(fe.u))
        assert np.max(np.abs(d['f'] - fe.f)) <= 1e-8 * np.max(np.abs(fe.f))
        assert np.max(np.abs(d['sgl'] - fe.sgl)) <= 1e-9 * np.max(np.abs(fe.sgl))
        e0, e1 = d['e0'], d['e1']
        assert e1 > e0
        assert np.max(np.abs(d['sig'] - sig1[e0:e1])) <= 1e-8 * np.max(np.abs(sig1))
        assert np.max(np.abs(d['epl'] - epl1[e0:e1])) <= 1e-8 * max(np.max(np.abs(epl1)), 1e-30)
    assert sorted(res[r]['e0'] for r in res) == sorted(fe.strip_range(r, world)[0] for r in range(world))
    if case == 'hill6_12':   # and the reference's own trace
        g = np.load(os.path.join(golden_dir, 'solve.npz'))
        assert res[0]['nsteps'] == int(g['hill6_12_nsteps']) and res[0]['niter'] == list(g['hill6_12_niter'])
        assert np.max(np.abs(res[0]['u'] - g['hill6_12_u'])) <= 1e-6 * np.max(np.abs(g['hill6_12_u']))


# ---------------------------------------------------------------------------------------------------------------------
# strip-local engine (plfx_set_strip): every rank holds its x-strip + halo as a standalone local problem
def build_strip(case, golden_dir):
    import pylabfea_amd as FE
    mat = FE.Material()
    mat.elasticity(E=200.e3, nu=0.3)
    mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = FE.Model(dim=2, planestress=False)
    if case == 'tension':                # homogeneous Hill tension, 128 x 32 elements (lenx : leny = 4 : 1, square elements)
        fe.geom([16.], LY=4.)
        fe.assign([mat])
        NX, NY, eps, ms = 128, 32, 0.003, 8
        el = None
    elif case == 'inclusion':            # soft inclusion across a strip boundary: strips take different branches
        soft = FE.Material(num=2)
        soft.elasticity(E=1.e3, nu=0.27)
        fe.geom(sect=2, LX=12., LY=4.)
        fe.assign([mat, soft])
        NX, NY, eps, ms = 192, 64, 0.002, 6
        el = np.ones((NX, NY))
        el[80:112, 20:44] = 2
    else:                                # J2 + SVC laminate (wave-per-element SVC kernels on strips)
        z = np.load(os.path.join(golden_dir, 'svc_shear.npz'))
        mb = FE.Material(name='ML', num=2)
        mb.elasticity(CV=z['par_CV'])
        mb.plasticity(sy=float(z['par_sy']), sdim=6)
        mb.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']),
                   float(z['par_scale_seq']))
        fe.geom([2, 1, 2, 1, 2], LY=2.)
        fe.assign([mat, mb, mat, mb, mat])
        NX, NY, eps, ms = 64, 16, 0.0015, 4
        el = None
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(eps * fe.leny, 'disp')
    if el is not None:
        fe.mesh(elmts=el, NX=NX, NY=NY)
    else:
        fe.mesh(NX=NX, NY=NY)
    return fe, ms


def _strip_worker(rank, world, port, case, golden_dir, q):
    global build
    build = build_strip            # the worker above takes the model from build()
    if case.endswith('+python'):   # the statement-by-statement Python transcription of the load step over the C-ABI calls
        os.environ['PLFX_NATIVE_STEP'] = '0'
    # hand-over levels as deep as the small test meshes allow (the default follows the plan's cost model)
    level = {'tension': {2: 3, 4: 2, 8: 1}.get(world), 'inclusion': 3, 'laminate_svc': 2}[case.split('+')[0]]
    if case.endswith('+default'):
        level = None
    if case.endswith('+sweephalo'):  # the redundant variant: halo elements swept locally, no generator exchange
        os.environ['PLFX_STRIP_SWEEP_HALO'] = '1'
    if case.endswith('+mgcap'):      # multigrid-PCG capped at 2 iterations: every solve ends in the Jacobi-PCG fall-back
        os.environ['PLFX_MG_MAXIT'] = '2'
    _worker(rank, world, port, case.split('+')[0], golden_dir, q, mode='strip', level=level)


@pytest.mark.parametrize('case,world', [('tension', 2), ('tension', 4), ('tension', 8), ('inclusion', 3), ('laminate_svc', 2),
                                        ('tension+python', 2), ('tension+default', 2), ('inclusion+sweephalo', 3),
                                        ('laminate_svc+default', 4), ('tension+mgcap', 2)])
def test_strip_local_engine_on_one_gpu(golden_dir, case, world, monkeypatch):
    """Strips + halo on 2..8 ranks (processes on cuda:0, host-staged transport over gloo: halo refresh of r / x, coarse
    right-hand side, partial sums, flags, statistics) against the single-rank run of the same model: identical load-step,
    K-iteration AND PCG-iteration counts (the V-cycle is arithmetically the single-GPU one), fields to 1e-9."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, world, port, case, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = collect(q, procs, world)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if case.endswith('+mgcap'):
        monkeypatch.setenv('PLFX_MG_MAXIT', '2')
    mgcap = case.endswith('+mgcap')
    python_driver = case.endswith('+python')
    default_level = case.endswith('+default')
    sweep_halo = case.endswith('+sweephalo')
    case = case.split('+')[0]
    fe, ms = build_strip(case, golden_dir)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=ms)
    sig1, epl1 = fe._state('sig'), fe._state('epl')
    assert np.max(epl1) > 0.
    its1 = [s[0] for s in fe.solver_stats]
    nyn = fe.NnodeY
    for r in range(world):
        d = res[r]
        st = d['strip']
        assert st is not None and d['native'] == (not python_driver)
        active, halo, Ld, clev, nh, nc, npart, ngen = d['strip_info']
        assert active and halo == st['W'] == 4 << Ld and clev >= 2
        if default_level and case == 'tension':
            assert Ld == 2          # the cheapest slowest strip of the plan's cost model on 2 x 64 columns
        if default_level and case == 'laminate_svc':
            # boundaries follow the cost of the columns: every strip gets its share of the SVC columns, widths differ
            widths = [res[q]['strip']['c1'] - res[q]['strip']['c0'] for q in range(world)]
            assert len(set(widths)) > 1 and min(widths) >= st['W'], widths
        assert nh > 0 and nc > 0 and npart > 0
        assert (ngen == 0) if sweep_halo else (ngen > 0)   # halo generators: received from their owners / recomputed locally
        assert d['nsteps'] == fe.nsteps and d['niter'] == list(fe.niter)
        if not mgcap:
            assert d['its'] == its1                                  # same PCG iterations in every solve
        else:
            assert d['fallbacks'] > 0 and fe._engine.solve_fallbacks() > 0
        lo, hi = 2 * st['c0'] * nyn, 2 * (st['c1'] + 1) * nyn        # nodes of the owned columns
        su = np.max(np.abs(fe.u))
        slack = 300. if mgcap else 1.     # two different Jacobi-PCG loops, each converged to rtol = 1e-10 of the residual
        assert np.max(np.abs(d['u'][lo:hi] - fe.u[lo:hi])) <= 1e-9 * slack * su
        flo, fhi = 2 * st['own_nodes'][0], 2 * st['own_nodes'][1]
        assert np.max(np.abs(d['f'][flo:fhi] - fe.f[flo:fhi])) <= 1e-8 * slack * np.max(np.abs(fe.f))
        assert np.max(np.abs(d['sgl'] - fe.sgl)) <= 1e-9 * slack * np.max(np.abs(fe.sgl))
        smax = np.max(np.abs(fe.sgl))
        for k, v in d['glob'].items():
            # entries that vanish in exact arithmetic (stress on the force-free boundary) are compared on the scale of the
            # stresses of the run, not on an absolute floor below the solver tolerance
            floor = 1e-3 * smax if k.startswith('s') else 1e-3
            assert np.max(np.abs(v - fe.glob[k])) <= 1e-8 * slack * max(floor, np.max(np.abs(fe.glob[k]))), k
        e0, e1 = d['e0'], d['e1']
        assert (e0, e1) == (st['c0'] * fe._NY, st['c1'] * fe._NY)
        assert np.max(np.abs(d['sig'] - sig1[e0:e1])) <= 1e-8 * slack * np.max(np.abs(sig1))
        assert np.max(np.abs(d['epl'] - epl1[e0:e1])) <= 1e-8 * slack * max(np.max(np.abs(epl1)), 1e-30)
    assert sorted(res[r]['e0'] for r in res)[0] == 0 and max(res[r]['e1'] for r in res) == fe.Nel


def test_bench_one_rank_with_rccl_collectives_forced():
    """The code path the driver launches for --gpus N, at N = 1 on real RCCL: the whole bench.py with a
    1-rank NCCL process group, the model distributed as ONE strip and every collective of the strip engine forced on
    (PLFX_STRIP_FORCE_COLL=1: ncclAllReduce of partial sums / coarse right-hand sides / coarse generators, the ncclGroup of
    the halo refresh, the flag and calc_global all-reduces) -- same sweeps, solves and PCG iterations as the plain single-GPU
    run, a JSON line with the per-rank budget, and the collective time on the stream reported."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ['--steps', '3', '--warmup', '1', '--mesh', '256', '--no-cpu', '--svc-mesh', '16', '--no-inclusion', '--no-2048', '--no-reuse-off',
              '--config5-leg-mesh', '128']
    env = dict(os.environ, PLFX_FORCE_DIST='1', PLFX_STRIP_FORCE_COLL='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), PLFX_COLL_TIMEOUT='120')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1'] + common, env=env, capture_output=True,
                         text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and 'strip-local engine x1' in d['config']['parallelism']
    sc = d['strip_collectives']
    assert sc['halo_refreshes'] > 0 and sc['coarse_gathers'] > 0 and sc['partial_sum_allreduces'] > 0
    assert d['per_rank'][0]['collectives_per_step'] > 0 and d['collective_ms_per_step'] > 0.
    assert d['config5_leg']['sweeps'] > 0 and 'error' not in d['config5_leg']
    one = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + common, capture_output=True, text=True, timeout=900, cwd=root)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith('{')][-1])
    assert (d['sweeps'], d['solves'], d['pcg_iterations']) == (d1['sweeps'], d1['solves'], d1['pcg_iterations'])
    # the two-solution initial guess ran in the strip as in the plain engine: sums over the owned columns, all-reduced
    assert d['initial_guess_from_two_solutions'] == d1['initial_guess_from_two_solutions'] and d1['initial_guess_from_two_solutions']['accepted_as_solution'] > 0
    assert abs(d['config5_leg']['sgl_yy'] - d1['config5_leg']['sgl_yy']) < 1e-8 * abs(d1['config5_leg']['sgl_yy'])


@pytest.mark.parametrize('mode', ['strong', 'weak', 'strong4'])
def test_bench_two_ranks(tmp_path, mode):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), here with both ranks on
    ONE GPU over the host-staged transport (PLFX_BENCH_TRANSPORT=host; RCCL refuses two ranks on a device).  Default =
    STRONG scaling (north star: the SAME mesh cut into x-strips, contiguous in the reference's x-major numbering,
    model.py:893, 935): `config.workload` identical to the single-GPU line, `scaling: "strong"`; `--weak`: 2 strips of
    128 x 128 elements side by side.  Either way: JSON contract, honest labels, per-rank rooflines, and the same sweeps /
    solves / PCG iterations as the single-GPU run of the same mesh."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = free_port()
    env = dict(os.environ, PLFX_BENCH_TRANSPORT='host')
    # 'strong4': four strips of 128 owned + 2 x 32 halo columns, hand-over level 3 -- the geometry of the driver's 8-GPU run of
    # the 1024 x 1024 mesh (profiles/r03u_*: that run itself, 4 and 8 ranks on one GPU)
    nr, mesh = (4, 512) if mode == 'strong4' else (2, 128)
    mode = 'strong' if mode == 'strong4' else mode
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nr), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', str(nr), '--steps', '3', '--warmup', '1',
           '--mesh', str(mesh), '--config5-leg-mesh', '256' if nr == 2 else '0'] + (['--weak'] if mode == 'weak' else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    assert out.stdout.strip().splitlines()[-1] == line          # the JSON line is the last thing printed
    d = json.loads(line)
    strips = 2 if mode == 'weak' else 1
    assert d['n_gpus'] == nr and d['scaling'] == mode and d['steps'] == 3 and d['dtype'] == 'f64'
    assert d['config']['elements'] == strips * mesh * mesh and ('strip-local engine x%d' % nr) in d['config']['parallelism']
    assert ('%dx%d Q4' % (strips * mesh, mesh)) in d['config']['workload'] and (mode + ' scaling') in d['config']['parallelism']
    assert d['strip_collectives']['halo_refreshes'] > 0 and d['strip_collectives']['coarse_gathers'] > 0
    assert d['roofline'] is not None and d['cpu_baseline'] is None
    assert [r['rank'] for r in d['per_rank']] == list(range(nr)) and all(r['roofline'] is not None for r in d['per_rank'])
    cols = [r['owned_columns'] for r in d['per_rank']]
    assert cols[0][0] == 0 and cols[-1][1] == strips * mesh and all(a[1] == b[0] for a, b in zip(cols[:-1], cols[1:]))   # the strips tile the mesh
    if nr == 4:
        assert all(r['halo_columns'] == 32 for r in d['per_rank']) and [c[1] - c[0] for c in cols] == [128] * 4
    # where every rank's load step goes, and the Amdahl arithmetic that follows from it
    for r in d['per_rank']:
        b = r['time_budget_ms_per_step']
        # (vcycles may be 0: with the interpolated start the steady load steps of this window need no PCG iteration)
        assert b['load_step'] > 0. and b['vcycles'] >= 0. and 0. < b['divisible_by_strips'] < b['load_step']
    assert d['amdahl']['estimated_one_gpu_ms_per_step'] > 0.
    if mode == 'strong' and nr == 2:   # the sweep-dominated leg (BASELINE config 5, here on 256 x 256 elements) in the same line
        leg = d['config5_leg']
        assert leg['sweeps'] > 0 and leg['value'] > 0. and '256x256 laminate' in leg['workload'] and leg['sgl_yy'] > 100.
    # the same workload on one rank: identical counts (and, for strong scaling, the identical workload string)
    sys.path.insert(0, root)
    one = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '1', '--mesh', str(mesh),
                          '--no-cpu', '--no-svc', '--no-inclusion', '--no-2048'], capture_output=True, text=True, timeout=900, cwd=root)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith('{')][-1])
    if mode == 'strong':
        assert d1['config']['workload'] == d['config']['workload'] and d1['config']['elements'] == d['config']['elements']
        assert (d['sweeps'], d['solves'], d['pcg_iterations']) == (d1['sweeps'], d1['solves'], d1['pcg_iterations'])
    else:
        import pylabfea_amd as FE
        import bench
        fe = bench.tension_model(FE, bench.hill_material(FE), 128, 0.005, strips=2)
        ninc, pre = bench.schedule(3, 1)
        marks = {}

        def hook(il):
            if il == pre + 1:
                marks['s0'], marks['q0'] = fe.n_sweeps, len(fe.solver_stats)
            if il == pre + 4:
                marks['s1'], marks['q1'] = fe.n_sweeps, len(fe.solver_stats)
        fe._step_hook = hook
        fe._max_load_steps = pre + 4
        fe.solve(min_step=ninc)
        its = [s[0] for s in fe.solver_stats[marks['q0']:marks['q1']]]
        assert d['sweeps'] == marks['s1'] - marks['s0'] and d['solves'] == len(its) and d['pcg_iterations'] == sum(its)

def test_bench_inclusion_variant_is_reproducible():
    """bench.py's heterogeneous leg (soft inclusion: long PCG solves, half of the matrix elements on the 50-sub-step corrector): two
    runs of the same window in one process give the same sweeps, solves, PCG iterations and the same number of solves completed
    by a fall-back solver (that count differed between the driver's window and the builder's -- other load steps;
    for ONE window it is deterministic: integer atomics only, fixed-order sums)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import pylabfea_amd as FE
    a = bench.inclusion_variant(FE, 256, 4, 1)
    b = bench.inclusion_variant(FE, 256, 4, 1)
    for k in ('sweeps', 'solves', 'pcg_iterations', 'solves_completed_by_fallback_solver', 'elements_on_50_substep_corrector_last_sweep'):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a['pcg_iterations'] > 100 and a['sweeps'] > 4


def test_bench_config5_option(tmp_path):
    """bench.py --config 5 (BASELINE config 5: laminate of J2 + the Goss-Barlat-trained SVC) on a reduced mesh: one rank, and
    two ranks on one GPU over the host transport with cost-balanced strips -- same sweeps / solves / iterations, one JSON line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [os.path.join(root, 'bench.py'), '--config', '5', '--mesh', '256', '--steps', '1', '--warmup', '0', '--no-cpu']
    one = subprocess.run([sys.executable] + base, capture_output=True, text=True, timeout=900, cwd=root)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith('{')][-1])
    assert 'two-phase laminate' in d1['config']['workload'] and d1['config']['elements'] == 256 * 256
    assert d1['n_gpus'] == 1 and d1['sweeps'] > 0 and d1['value'] > 0. and 'roofline_2048' not in d1
    port = free_port()
    env = dict(os.environ, PLFX_BENCH_TRANSPORT='host')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port)] + base + ['--gpus', '2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['scaling'] == 'strong' and d['config']['workload'] == d1['config']['workload']
    assert (d['sweeps'], d['solves']) == (d1['sweeps'], d1['solves'])
    cols = [r['owned_columns'] for r in d['per_rank']]
    assert cols[0][0] == 0 and cols[0][1] == cols[1][0] and cols[1][1] == 256


# ---------------------------------------------------------------------------------------------------------------------
# indefinite tangent stiffness on strips: the GMRES / MINRES fall-backs with owned-only sums and halo exchanges
BAD_TANGENT_21 = [3.06119e+05, 2.30987e+05, 2.44365e+05, -7.10713e+02, 8.16221e+02, -3.89722e+01, -5.30574e+05, -2.01886e+05,
                  8.08981e+02, -9.29004e+02, 4.43106e+01, 2.03386e+05, -4.23220e+01, 4.86228e+01, -2.33304e+00, 5.81516e+04,
                  1.14637e+01, -5.47451e-01, 5.81484e+04, 6.34734e-01, 5.81615e+04]


def indefinite_solve(rank=None, world=None, dist=None):
    """elastic 128 x 32 mesh with the tangent of tests/test_gpu_random.py planted into three elements -- one of them in the
    column next to the strip boundary, so that the neighbour's operator needs it through the generator exchange"""
    import pylabfea_amd as FE
    from pylabfea_amd import _lib
    mat = FE.Material()
    mat.elasticity(E=151220., nu=0.3)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([16.], LY=4.)
    fe.assign([mat])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.002 * fe.leny, 'disp')
    fe.mesh(NX=128, NY=32)
    if dist is not None:
        fe.distribute(rank, world, None, host_allreduce=FE.host_transport(dist, rank, world), mode='strip', coarse_level=2)
    eng = fe._ensure_engine()
    NY = fe._NY
    bad = np.zeros((6, 6))
    bad[np.triu_indices(6)] = BAD_TANGENT_21
    bad = bad + bad.T - np.diag(np.diag(bad))
    D = np.tile(fe._element_CV(mat), (fe.Nel, 1, 1))
    for cx, cy in ((40, 16), (63, 8), (90, 24)):
        D[cx * NY + cy] = bad
    eng.state_set(_lib.ST_ELSTIFF, D[fe._e0:fe._e1].reshape(-1, 36))
    eng.assemble()
    z, d = np.zeros(2), np.array([0., 0.002 * fe.leny])
    fe._bc_apply(eng, z, z, z, d, None)
    n0 = eng.solve_fallbacks()
    it, rr, ok = eng.solve(1e-10, 20000, False)
    assert ok and rr <= 1e-10 and eng.solve_fallbacks() == n0 + 1
    return fe, it, fe._nodal(eng.state_get(_lib.ST_DU))


def _indef_worker(rank, world, port, solver, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['PLFX_INDEFINITE_SOLVER'] = solver
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fe, it, du = indefinite_solve(rank, world, dist)
        q.put((rank, dict(it=it, du=du, strip=fe._strip)))
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, 'ERROR: ' + traceback.format_exc()))
        raise exc
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('solver', ['gmres', 'minres'])
def test_strip_indefinite_tangent(solver, monkeypatch):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_indef_worker, args=(r, world, port, solver, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = collect(q, procs, world, 300.)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    monkeypatch.setenv('PLFX_INDEFINITE_SOLVER', solver)
    fe, it1, du1 = indefinite_solve()
    nyn = fe.NnodeY
    for r in range(world):
        st = res[r]['strip']
        assert abs(res[r]['it'] - it1) <= 3, (res[r]['it'], it1)     # same Krylov method; sums in another order
        lo, hi = 2 * st['c0'] * nyn, 2 * (st['c1'] + 1) * nyn
        assert np.max(np.abs(res[r]['du'][lo:hi] - du1[lo:hi])) <= 1e-7 * np.max(np.abs(du1))
