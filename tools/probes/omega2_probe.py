#!/usr/bin/env python3
"""PCG iteration counts of the heterogeneous variant and of the computed window for smoother weights (PLFX_MG_OMEGA2=w1,w2 set by the caller)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import pylabfea_amd as FE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 1)
a = bench.inclusion_variant(FE, n, K, W)
b = bench.window_run(FE, n, K, W, reuse=False)
print(os.environ.get('PLFX_MG_OMEGA2', 'one weight 0.65'), '| inclusion: its %d (%.1f per computed solve) %.1f ms/step fallbacks %d | computed window: its %d %.2f ms/step'
      % (a['pcg_iterations'], a['pcg_iterations_per_computed_solve'], a['ms_per_step'], a['solves_completed_by_fallback_solver'], b['pcg_iterations'], b['ms_per_step']))
