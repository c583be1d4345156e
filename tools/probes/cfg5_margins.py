"""Margins of the size-independent property tests of BASELINE config 5 (tests/test_gpu_configs.py): measured differences
against their tolerances.  python tools/probes/cfg5_margins.py"""
import sys, os, warnings, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_configs as T
g = os.path.join(ROOT, 'tests', 'golden')
def run(nx, ny, steps=None):
    fe = T.laminate_cfg5(g, nx, ny)
    if steps: fe._max_load_steps = steps
    with warnings.catch_warnings():
        warnings.simplefilter('ignore'); fe.solve(min_step=20)
    return fe
a, b = run(256, 32), run(512, 128)
sa, sb = np.array(a.sgl), np.array(b.sgl)
print('cfg5 mesh independence: max rel diff %.3e (tol 2e-4), final sgl_yy %.4f / %.4f (ref 144.133 +- 0.02)' % (np.max(np.abs(sa - sb)) / np.max(np.abs(sa)), sa[-1][1], sb[-1][1]))
big, small = run(2048, 2048, 12), run(512, 64, 12)
sB, sS = np.array(big.sgl), np.array(small.sgl)
print('cfg5 full size 12 steps: max rel diff sgl %.3e egl %.3e epgl %.3e (tol 2e-4)' % (np.max(np.abs(sB - sS)) / np.max(np.abs(sS)),
      np.max(np.abs(np.array(big.egl) - np.array(small.egl))) / np.max(np.abs(small.egl)), np.max(np.abs(np.array(big.epgl) - np.array(small.epgl))) / np.max(np.abs(small.egl))))
