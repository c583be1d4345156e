#!/usr/bin/env python3
"""Whole Model.solve() wall-clock of the BASELINE.json configs on one GPU (full load schedules)."""
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


def run(name, fe, ms):
    eng = fe._ensure_engine()
    eng.sync()
    t = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=ms)
    eng.sync()
    dt = time.perf_counter() - t
    its = [s[0] for s in fe.solver_stats]
    print('%-34s %8.3f s  load steps %3d  K-iterations %4d  sweeps %4d  solves %4d  PCG its %5d  updates/s %.3g  sgl_yy %.6f'
          % (name, dt, fe.nsteps, sum(max(n, 0) + 1 for n in fe.niter), fe.n_sweeps, len(its), sum(its),
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
    fe.mesh(NX=2048, NY=2048)
    if '5full' in which:
        run('config 5: 2048x2048 laminate J2 + Goss-Barlat SVC, all 20 load steps', fe, 20)
        print('    SVC elements on the 50-sub-step corrector at least once: %d' % int(np.sum(fe._state('max_steps') == 49)))
    else:
        fe._max_load_steps = 8
        run('config 5: 2048x2048 laminate J2 + Goss-Barlat SVC, first 8 of 20 steps', fe, 20)
