"""Config 4 (SVC, perfect plasticity): how does the last, non-converged load step depend on mesh size / PCG tolerance?"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
from test_gpu_model import svc_material, tension_model
G = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')
g = np.load(os.path.join(G, 'solve_configs.npz'))
ref = g['cfg4_svc_4_sgl']
for n, rtol in ((4, 1e-10), (16, 1e-10), (16, 1e-13), (64, 1e-10), (64, 1e-13), (256, 1e-10), (256, 1e-13)):
    fe = tension_model(svc_material(G, 'hill'), n, 0.001)
    fe.cg_rtol = rtol
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
    sig = fe._state('sig')
    dev = np.max(np.abs(fe.sgl - ref), axis=1) / np.max(np.abs(ref))
    print(n, rtol, 'nsteps', fe.nsteps, 'niter', fe.niter[-1], 'co_nconv', fe.co_nconv[-1], 'dev per step', np.array2string(dev, precision=2),
          'nonuniformity', np.max(np.abs(sig - sig[0])) / np.max(np.abs(sig)), 'its', sum(s[0] for s in fe.solver_stats), flush=True)
