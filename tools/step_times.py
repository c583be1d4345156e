#!/usr/bin/env python3
"""Wall-clock of every load step of the bench workload (GPU synchronised at the step boundaries), without and with the
HIP-event instrumentation that bench.py switches on for its timed region."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pylabfea_amd as FE


def run(instr_from):
    fe = bench.tension_model(FE, bench.hill_material(FE), 1024, 0.005, device=0)
    eng = fe._ensure_engine()
    ts = []

    def hook(il):
        if il == instr_from:
            eng.timing_reset()
            eng.timing_enable(True)
        eng.sync()
        ts.append((il, time.perf_counter(), fe.n_sweeps, len(fe.solver_stats)))
    fe._step_hook = hook
    fe._max_load_steps = 22
    fe.solve(min_step=50)
    return fe, ts


for instr in (None, 7):
    fe, ts = run(instr)
    print('event instrumentation from step', instr)
    for a, b in zip(ts[:-1], ts[1:]):
        print('  step %2d: %.3f ms  sweeps %d solves %d its %s' % (
            b[0], 1e3 * (b[1] - a[1]), b[2] - a[2], b[3] - a[3], [s[0] for s in fe.solver_stats[a[3]:b[3]]]))
