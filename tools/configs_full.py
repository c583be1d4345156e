#!/usr/bin/env python3
"""Whole Model.solve() wall-clock of the BASELINE.json configs on one GPU (full load schedules).

Launched through ``python -m torch.distributed.run --nproc-per-node N tools/configs_full.py 5full`` the model is distributed
over N ranks (strip-local engine): RCCL with one GPU per rank, or -- PLFX_TOOL_TRANSPORT=host -- the host-staged transport
with all ranks on the GPUs that exist (functional full-size check on a single-GPU box; the wall-clock then says nothing)."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pylabfea_amd as FE  # noqa: E402


def tension(mat, n, eps):
    fe = FE.Model(dim=2)
    fe.geom([4.], LY=4.)
    fe.assign([mat])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(eps * fe.leny, 'disp')
    fe.mesh(NX=n, NY=n)
    return fe


RANK, WORLD = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
DIST = None
if WORLD > 1:
    import torch
    import torch.distributed as DIST
    HOST = os.environ.get('PLFX_TOOL_TRANSPORT') == 'host'
    DEV = int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(DEV)
    DIST.init_process_group('gloo' if HOST else 'nccl')


def distribute(fe):
    if DIST is None:
        return
    fe.device = DEV
    if HOST:
        fe.distribute(RANK, WORLD, None, host_allreduce=FE.host_transport(DIST, RANK, WORLD))
    else:
        from pylabfea_amd import _lib
        uid = [_lib.Context(DEV).comm_unique_id() if RANK == 0 else None]
        DIST.broadcast_object_list(uid, src=0)
        fe.distribute(RANK, WORLD, uid[0])


def run(name, fe, ms):
    distribute(fe)
    eng = fe._ensure_engine()
    if DIST is not None:
        DIST.barrier()
    eng.sync()
    t = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=ms)
    eng.sync()
    dt = time.perf_counter() - t
    its = [s[0] for s in fe.solver_stats]
    if DIST is not None:
        st = fe._strip
        svc = np.array([getattr(m, 'ML_yf', False) for m in fe.mat])[fe._mat_id[fe._e0:fe._e1]]
        heavy = int(np.sum(fe._state('max_steps')[fe._e0:fe._e1] == 49))
        rows = [None] * WORLD
        DIST.all_gather_object(rows, (RANK, st and (st['c0'], st['c1'], st['W'], st['Ld']), int(svc.sum()), heavy, dt))
        if RANK == 0:
            for r in rows:
                print('    rank %d: owned columns / halo / level %s, SVC elements owned %d, on the 50-sub-step corrector in the last sweep %d, %.1f s'
                      % (r[0], r[1], r[2], r[3], r[4]))
        if RANK != 0:
            return
    if os.environ.get('CFG_VERBOSE') == '1':
        print('    K-iterations per load step: %s' % list(fe.niter))
        print('    PCG iterations per solve: %s' % its)
    rel = np.array([s[1] for s in fe.solver_stats])
    print('    solves above rtol %g: %d (worst relative residual %.3g); load steps with unconverged K-iterations: %s; sgl_yy per load step: %s'
          % (fe.cg_rtol, int(np.sum(rel > fe.cg_rtol)), rel.max() if len(rel) else 0., [int(v) for v in np.asarray(fe.co_nconv).ravel()],
             np.round([s[1] for s in fe.sgl], 3).tolist()))
    print('%-34s %8.3f s  load steps %3d  K-iterations %4d  sweeps %4d  solves %4d  PCG its %5d (max %d, Jacobi fall-backs %d)  updates/s %.3g  sgl_yy %.6f'
          % (name, dt, fe.nsteps, sum(max(n, 0) + 1 for n in fe.niter), fe.n_sweeps, len(its), sum(its), max(its), eng.solve_fallbacks(),
             fe.Nel * fe.n_sweeps / dt if fe.n_sweeps else 0., fe.sgl[-1][1]))
    sys.stdout.flush()


which = sys.argv[1:] or ['1', '2', '3', '4']
if '1' in which:
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    run('config 1: 32x32 elastic', tension(m, 32, 0.001), None)
if '2' in which:
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=150., khard=500., sdim=6)
    run('config 2: 256x256 J2, min_step=20', tension(m, 256, 0.004), 20)
if '3' in which:
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    run('config 3: 1024x1024 Hill, min_step=50', tension(m, 1024, 0.005), 50)
if '4' in which:
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_hill.npz'))
    m = FE.Material(name='ML-Hill')
    m.elasticity(CV=z['par_CV'])
    m.plasticity(sy=float(z['par_sy']), sdim=6)
    m.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
    run('config 4: 512x512 SVC, min_step=10', tension(m, 512, 0.001), 10)


def inclusion(mat, n, eps):
    """configs 2-3 with the central soft inclusion of examples/inclusion.py scaled to the mesh (branch
    divergence, heterogeneous states)"""
    soft = FE.Material(num=2)
    soft.elasticity(E=1.e3, nu=0.27)
    fe = FE.Model(dim=2)
    fe.geom(sect=2, LX=4., LY=4.)
    fe.assign([mat, soft])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(eps * fe.leny, 'disp')
    el = np.ones((n, n))
    el[n // 3:2 * (n // 3), n // 3:2 * (n // 3)] = 2
    fe.mesh(elmts=el, NX=n, NY=n)
    return fe


if '2i' in which:
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=150., khard=500., sdim=6)
    run('config 2 + inclusion: 256x256 J2', inclusion(m, 256, 0.004), 20)
if '3i' in which:
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    fe = inclusion(m, 1024, 0.005)
    run('config 3 + inclusion: 1024x1024 Hill', fe, 50)
    ms = fe._state('max_steps')
    print('    elements that ran the 50-sub-step corrector at least once: %d of %d' % (int(np.sum(ms == 49)), fe.Nel))
if '5' in which or '5full' in which:
    # config 5 geometry on ONE GPU: laminate [2,1,2,1,2], J2 + SVC phases, 2048 x 2048, first load steps
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_gossbarlat.npz'))   # the SVC trained on Barlat Yld2004-18p (Goss)
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
    n5x, n5y = int(os.environ.get('CFG5_NX', '2048')), int(os.environ.get('CFG5_NY', '2048'))   # experiments: smaller meshes
    fe.mesh(NX=n5x, NY=n5y)
    if os.environ.get('CFG5_EQUAL') == '1':
        fe.strip_weights = np.ones(n5x)                     # strips of equal width instead of equal cost
    if '5full' in which:
        run('config 5: %dx%d laminate J2 + Goss-Barlat SVC, all 20 load steps' % (n5x, n5y), fe, 20)
        if RANK == 0 and DIST is None:
            print('    SVC elements on the 50-sub-step corrector at least once: %d' % int(np.sum(fe._state('max_steps') == 49)))
    else:
        fe._max_load_steps = int(os.environ.get('CFG5_STEPS', '8'))
        run('config 5: 2048x2048 laminate J2 + Goss-Barlat SVC, first 8 of 20 steps', fe, 20)
if DIST is not None:
    DIST.barrier()
    DIST.destroy_process_group()
