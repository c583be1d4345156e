import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from test_gpu_configs import laminate_cfg5
G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden')
mid = np.load(os.path.join(G, 'mid_configs_128.npz'))
p = 'cfg5_256x128'
fe = laminate_cfg5(G, 256, 128)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fe.solve(min_step=20)
s = np.max(np.abs(mid[p + '_sig'])); e = np.max(np.abs(mid[p + '_eps']))
for name, a, b, sc in (('sig', fe._state('sig'), mid[p + '_sig'], s), ('epl', fe._state('epl'), mid[p + '_epl'], e), ('eps', fe._state('eps'), mid[p + '_eps'], e)):
    d = np.abs(a - b) / sc
    de = d.max(axis=1)
    print(name, 'max %.2e rms %.2e  elements above 2e-5: %d of %d, above 1e-5: %d; worst elements' % (d.max(), np.sqrt((d ** 2).mean()), (de > 2e-5).sum(), len(de), (de > 1e-5).sum()), np.argsort(de)[-5:], 'mat of worst', fe._mat_id[np.argsort(de)[-5:]])
    cols = np.argsort(de)[-20:] // 128
    print('   columns of the 20 worst elements:', sorted(set(cols.tolist())))
print('sgl per step diff', np.max(np.abs(np.asarray(fe.sgl) - mid[p + '_sgl']), axis=1) / s)
print('co_nconv', list(fe.co_nconv), list(mid[p + '_co_nconv']))
