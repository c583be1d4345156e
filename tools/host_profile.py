#!/usr/bin/env python3
"""Host-side time split of Model.solve() on the bench workload: seconds inside each C-ABI call (which includes waiting for
the GPU at the synchronising ones) against pure Python time between the calls.  host_profile.py [mesh] [steps]"""
import os
import sys
import time
import collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pylabfea_amd as FE
from pylabfea_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
acc = collections.defaultdict(float)
cnt = collections.Counter()


def wrap(name):
    f = getattr(_lib.Context, name)

    def g(self, *a, **k):
        t = time.perf_counter()
        try:
            return f(self, *a, **k)
        finally:
            acc[name] += time.perf_counter() - t
            cnt[name] += 1
    setattr(_lib.Context, name, g)


for name in ('load_step', 'set_bc_plan', 'set_bc_sources', 'set_finish_set', 'assemble', 'apply_bc', 'apply_bc_plan', 'solve', 'sweep', 'scf_all', 'scf_stats', 'scf_sumsq', 'update_state',
             'finish_step', 'gather', 'global_sums', 'state_get'):
    wrap(name)
fe = bench.tension_model(FE, bench.hill_material(FE), n, 0.005, device=0)
eng = fe._ensure_engine()
fe._max_load_steps = 6
fe.solve(min_step=50)          # elastic pre-roll
acc.clear(); cnt.clear()
fe._max_load_steps = steps
eng.sync()
t0 = time.perf_counter()
fe.solve(min_step=50 - 6)
eng.sync()
dt = time.perf_counter() - t0
ns = fe.nsteps
print('%d load steps, %.3f ms per step wall' % (ns, 1e3 * dt / ns))
tot = 0.
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-14s %6d calls  %8.1f us per step  (%6.1f us per call)' % (k, cnt[k], 1e6 * v / ns, 1e6 * v / cnt[k]))
    tot += v
print('  %-14s               %8.1f us per step' % ('python between', 1e6 * (dt - tot) / ns))
