#!/usr/bin/env python3
"""Host-side profile of Model.solve() on the bench workload (where does the Python / ctypes time go?)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pylabfea_amd as FE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
fe = bench.tension_model(FE, bench.hill_material(FE), n, 0.005, device=0)
eng = fe._ensure_engine()
fe._max_load_steps = 4
fe.solve(min_step=50)          # warm-up (elastic steps)
fe._max_load_steps = steps
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
fe.solve(min_step=50 - 4)
pr.disable()
eng.sync()
dt = time.perf_counter() - t0
print('steps %d  wall %.2f ms/step  sweeps %d' % (fe.nsteps, 1e3 * dt / fe.nsteps, fe.n_sweeps))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
