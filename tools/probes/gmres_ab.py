#!/usr/bin/env python3
"""A/B of the two Gram-Schmidt forms of GMRES (PLFX_GMRES_ORTH=cgs2 | default) on config 5's laminate: per-solve iteration counts
and relative residuals of the fall-back solves.   python tools/probes/gmres_ab.py NX NY [max_load_steps]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_configs import laminate_cfg5  # noqa: E402

nx, ny = int(sys.argv[1]), int(sys.argv[2])
fe = laminate_cfg5(os.path.join(ROOT, 'tests', 'golden'), nx, ny)
if len(sys.argv) > 3:
    fe._max_load_steps = int(sys.argv[3])
t = time.time()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fe.solve(min_step=20)
fe._engine.sync()
dt = time.time() - t
its = [q[0] for q in fe.solver_stats]
rel = [q[1] for q in fe.solver_stats]
print('%s  %dx%d  %.2f s  solves %d  iterations %d (max %d)  fall-backs %d  worst residual %.2e  sgl_yy %.6f  niter %s'
      % (os.environ.get('PLFX_GMRES_ORTH', 'dcgs2'), nx, ny, dt, len(its), sum(its), max(its), fe._engine.solve_fallbacks(), max(rel),
         fe.sgl[-1][1], list(fe.niter)))
print('iterations per solve:', its)
