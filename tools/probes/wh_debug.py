import sys, os, warnings, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_workhard_svc as T
z = np.load('/root/repo/tests/golden/svc_workhard.npz')
fe = T.wh_model(z, 4)
with warnings.catch_warnings():
    warnings.simplefilter('ignore'); fe.solve(min_step=8)
print(fe._engine.wh_info(), fe.nsteps, fe.niter)
