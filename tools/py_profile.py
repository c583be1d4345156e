#!/usr/bin/env python3
"""cProfile of the Python side of Model.solve() on the bench workload (where the per-step host time goes)."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pylabfea_amd as FE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fe = bench.tension_model(FE, bench.hill_material(FE), n, 0.005, device=0)
eng = fe._ensure_engine()
fe._max_load_steps = 6
fe.solve(min_step=50)
fe._max_load_steps = steps
pr = cProfile.Profile()
pr.enable()
fe._solve_steps(min_step=50 - 6)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
