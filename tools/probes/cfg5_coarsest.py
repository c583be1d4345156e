"""config 5 at 2048^2, first N load steps: sgl_yy history, linear-solve residuals and fall-back counts (knobs from the environment).
python tools/probes/cfg5_coarsest.py [steps]"""
import sys, os, warnings, time, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_configs as T
g = os.path.join(ROOT, 'tests', 'golden')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 17
fe = T.laminate_cfg5(g, 2048, 2048)
fe._max_load_steps = steps
t0 = time.time()
with warnings.catch_warnings():
    warnings.simplefilter('ignore'); fe.solve(min_step=20)
rel = np.array([s[1] for s in fe.solver_stats]); its = np.array([s[0] for s in fe.solver_stats])
print('knobs', {k: v for k, v in os.environ.items() if k.startswith('PLFX_')})
print('%.1f s; sgl_yy %s' % (time.time() - t0, np.round(np.array(fe.sgl)[:, 1], 3).tolist()))
print('solves %d, worst rel. residual %.3e, solves above 1.0000001 rtol: %s' % (len(rel), rel.max(), np.nonzero(rel > 1.0000001 * fe.cg_rtol)[0].tolist()))
print('fall-backs %d, %s, iterations %d (max %d)' % (fe._engine.solve_fallbacks(), fe._engine.indefinite_info(), its.sum(), its.max()))
print('niter', list(fe.niter), 'co_nconv', list(fe.co_nconv))
