"""BASELINE config 5 (laminate J2 + Goss-Barlat SVC) on a reduced mesh: how uniform along y are the fields of its solves, and
which solves cost the iterations.  python tools/probes/cfg5_uniformity.py [n] [steps]"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
fe, nsv = bench.laminate_model(FE, n)
marks = []
def hook(il):
    marks.append(len(fe.solver_stats))
    from pylabfea_amd import _lib
    u = np.asarray(fe._engine.state_get(_lib.ST_DU)).reshape(n + 1, n + 1, 2)   # the increment of this load step
    ux, uy = u[:, :, 0], u[:, :, 1]
    sx = np.max(np.abs(ux - ux[:, [n // 2]])) / max(np.max(np.abs(ux)), 1e-300)
    ylin = uy - uy[:, [n]] * (np.arange(n + 1)[None, :] / n)
    sy = np.max(np.abs(ylin)) / max(np.max(np.abs(uy)), 1e-300)
    a = marks[-2] if len(marks) > 1 else 0
    its = [q[0] for q in fe.solver_stats[a:marks[-1]]]
    print('step %2d: u_x spread along y %.1e, u_y departure from y-linear %.1e | solves %d iterations %s' % (il, sx, sy, len(its), its), flush=True)
fe._step_hook = hook
fe._max_load_steps = steps
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fe.solve(min_step=20)
print('fallbacks', fe._engine.indefinite_info() if hasattr(fe._engine, 'indefinite_info') else None)
