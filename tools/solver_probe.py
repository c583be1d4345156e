#!/usr/bin/env python3
"""Compare the PCG preconditioners (0 Jacobi, 1 multigrid) on the bench workload:
solver_probe.py <steps> <n> [<n> ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pylabfea_amd as FE  # noqa: E402
from pylabfea_amd import _lib  # noqa: E402

steps = int(sys.argv[1])
for n in [int(a) for a in sys.argv[2:]]:
    res = {}
    for pc in ((0, 1) if os.environ.get('PROBE_JACOBI') else (1,)):
        mat = bench.hill_material(FE)
        fe = bench.tension_model(FE, mat, n, 0.005)
        fe.precond = pc
        fe._max_load_steps = steps
        fe.cg_maxit = 20000 if pc == 0 else 300
        eng = fe._ensure_engine()
        if pc == 1 and os.environ.get('MG_NU'):
            eng.set_precond(1, float(os.environ.get('MG_OMEGA', '0.7')), int(os.environ['MG_NU']))
        eng.sync()
        eng.timing_enable(True)
        t = time.perf_counter()
        fe.solve(min_step=50)
        eng.sync()
        dt = time.perf_counter() - t
        its = [s[0] for s in fe.solver_stats]
        rr = [s[1] for s in fe.solver_stats]
        print('n=%d precond=%d(%s) solve(%d steps) %.3fs  solves %d  iters sum %d max %d  max relres %.1e  niter %s'
              % (n, pc, eng.precond_info(), steps, dt, len(its), sum(its), max(its), max(rr), fe.niter))
        for w, name in enumerate(['sweep', 'spmv', 'cgupd', 'assemble', 'vcycle', 'smooth0']):
            ms, cnt = eng.timing_get(w)
            if cnt:
                print('     %-8s %9.3f ms over %6d = %8.2f us' % (name, ms, cnt, 1e3 * ms / cnt))
        res[pc] = (fe.u.copy(), fe._state('sig').copy(), fe.sgl.copy())
        sys.stdout.flush()
    if 0 in res:
        du = np.max(np.abs(res[0][0] - res[1][0])) / np.max(np.abs(res[0][0]))
        ds = np.max(np.abs(res[0][1] - res[1][1])) / np.max(np.abs(res[0][1]))
        print('   Jacobi vs MG: rel diff u %.2e sig %.2e' % (du, ds))
