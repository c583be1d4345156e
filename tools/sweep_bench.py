#!/usr/bin/env python3
"""Micro-benchmark of the return-mapping sweep kernel on a realistic state: run the bench workload into
the plastic regime, then re-launch the sweep on the converged increment.  Usage: sweep_bench.py [n] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pylabfea_amd as FE  # noqa: E402
from pylabfea_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mat = bench.hill_material(FE)
fe = bench.tension_model(FE, mat, n, 0.005)
fe._max_load_steps = steps
fe.solve(min_step=50)
eng = fe._engine
ns = eng.state_get(_lib.ST_MAXSTEPS)
print('state: load steps %d, niter %s, max_steps histogram %s' % (fe.nsteps, fe.niter, np.bincount(ns.astype(int))[:3]))
eng.timing_enable(True)
for rep in range(3):
    eng.timing_reset()
    t = time.perf_counter()
    for i in range(20):
        eng.sweep(1)
    eng.sync()
    dt = (time.perf_counter() - t) / 20
    ms, cnt = eng.timing_get(_lib.T_SWEEP)
    us = 1e3 * ms / cnt
    print('sweep: %.1f us/launch (events), %.1f us wall;  412 B/el -> %.0f GB/s (%.1f%% of 8 TB/s), %.2f G updates/s'
          % (us, dt * 1e6, 412. * fe.Nel / us / 1e3, 412. * fe.Nel / us / 1e3 / 80., fe.Nel / us / 1e3))
