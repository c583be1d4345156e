#!/usr/bin/env python3
"""BASELINE config 4 (SVC yield function, examples/train_hill.py parameters shipped as fixture):
time a few load steps at mesh size n.  svc_probe.py <n> <steps>"""
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
m = FE.Material(name='ML-Hill')
m.elasticity(CV=z['par_CV'])
m.plasticity(sy=float(z['par_sy']), sdim=6)
m.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
fe = FE.Model(dim=2)
fe.geom([4.], LY=4.)
fe.assign([m])
fe.bcleft(0.)
fe.bcbot(0.)
fe.bcright(0., 'force')
fe.bctop(0.001 * fe.leny, 'disp')
fe.mesh(NX=n, NY=n)
fe._max_load_steps = steps
eng = fe._ensure_engine()
eng.timing_enable(True)
t = time.perf_counter()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fe.solve(min_step=10)
eng.sync()
dt = time.perf_counter() - t
ms, cnt = eng.timing_get(_lib.T_SWEEP)
print('SVC %dx%d: %d load steps in %.2f s, niter %s, sweeps %d, sweep avg %.2f ms -> %.3g element-updates/s in the sweep'
      % (n, n, fe.nsteps, dt, fe.niter, cnt, ms / max(cnt, 1), fe.Nel * cnt / (ms * 1e-3)))
print('  max_steps histogram', np.bincount(fe._state('max_steps').astype(int))[[0, -1]], 'sgl', fe.sgl[-1][:2])
