#!/usr/bin/env python3
"""Tight-loop V-cycle timing (plfx_precond_bench) on the hierarchy of the bench workload after three load steps; knobs come
from the environment (one process per setting).  python tools/probes/vcycle_ab.py [mesh] [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
import pylabfea_amd as FE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fe = bench.tension_model(FE, bench.hill_material(FE), n, 0.005)
fe._max_load_steps = 3
fe.solve(min_step=50)
eng = fe._ensure_engine()
best = None
for rep in range(3):
    a, b = eng.precond_bench(reps)
    best = (a, b) if best is None or a < best[0] else best
knobs = {k: v for k, v in os.environ.items() if k.startswith('PLFX_')}
print('mesh %d levels %d  V-cycle %.1f us  (levels >= 1 with transfers: %.1f us)  knobs %s' % (n, eng.precond_info()[1], best[0], best[1], knobs))
