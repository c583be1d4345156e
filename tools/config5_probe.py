#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: two-phase laminate [2,1,2,1,2] (LY=8 so that elements are square),
phase A = J2 (sy=150, khard=500), phase B = SVC surrogate (fixture svc_hill; the Barlat-trained SVC of
examples/train_goss_barlat.py needs training data that is out of scope).  config5_probe.py <n> <steps>"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pylabfea_amd as FE  # noqa: E402
from pylabfea_amd import _lib  # noqa: E402

n = int(sys.argv[1])
steps = int(sys.argv[2])
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_hill.npz'))
ma = FE.Material(name='J2')
ma.elasticity(E=200.e3, nu=0.3)
ma.plasticity(sy=150., khard=500., sdim=6)
mb = FE.Material(name='ML')
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
t = time.perf_counter()
fe.mesh(NX=n, NY=n)
eng = fe._ensure_engine()
eng.sync()
print('mesh+engine %.2f s, precond %s' % (time.perf_counter() - t, eng.precond_info()))
fe._max_load_steps = steps
eng.timing_enable(True)
t = time.perf_counter()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fe.solve(min_step=20)
eng.sync()
dt = time.perf_counter() - t
its = [s[0] for s in fe.solver_stats]
ms, cnt = eng.timing_get(_lib.T_SWEEP)
print('config5 %dx%d: %d load steps in %.2f s (%.1f ms/step), niter %s, %d solves, PCG its %d (max %d), sweeps %d avg %.2f ms'
      % (n, n, fe.nsteps, dt, 1e3 * dt / fe.nsteps, fe.niter, len(its), sum(its), max(its), cnt, ms / max(cnt, 1)))
print('  sgl[-1]', fe.sgl[-1][:2], 'max relres %.1e' % max(s[1] for s in fe.solver_stats))
