#!/usr/bin/env python3
"""PCG iterations of the plastic inclusion case (config 2 + inclusion) for smoother settings (omega, nu).
`python tools/mg_probe2.py [n] [steps]`"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pylabfea_amd as FE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
variants = [(0., 0), (0.65, 3), (0.65, 4), (0.55, 2), (0.75, 2), (0.8, 2), (0.75, 3)]
for om, nu in variants:
    m = FE.Material()
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=150., khard=500., sdim=6)
    soft = FE.Material(num=2)
    soft.elasticity(E=1.e3, nu=0.27)
    fe = FE.Model(dim=2)
    fe.geom(sect=2, LX=4., LY=4.)
    fe.assign([m, soft])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.004 * fe.leny, 'disp')
    el = np.ones((n, n))
    el[n // 3:2 * (n // 3), n // 3:2 * (n // 3)] = 2
    fe.mesh(elmts=el, NX=n, NY=n)
    eng = fe._ensure_engine()
    if om > 0.:
        eng.set_precond(1, om, nu)
    fe._max_load_steps = steps
    eng.sync()
    t = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=20)
    eng.sync()
    dt = time.perf_counter() - t
    its = [s[0] for s in fe.solver_stats]
    print('n=%d omega=%.2f nu=%d: %.3f s, %d solves, %d PCG its (max %d), steps %d, sgl_yy %.6f'
          % (n, om, nu, dt, len(its), sum(its), max(its), fe.nsteps, fe.sgl[-1][1]))
    sys.stdout.flush()
