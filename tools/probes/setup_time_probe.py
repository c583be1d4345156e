#!/usr/bin/env python3
"""Set-up latency: Model.mesh() -> first load step, per stage.  The first model of a process also pays the
HIP context; the second one shows the steady state."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import pylabfea_amd as FE
from pylabfea_amd import _lib
for n in (1024, 1024, 2048):
    mat = FE.Material(); mat.elasticity(E=200.e3, nu=0.3)
    mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    t0 = time.perf_counter()
    fe = FE.Model(dim=2, planestress=False); fe.geom([4.], LY=4.); fe.assign([mat])
    fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.005 * fe.leny, 'disp')
    fe.mesh(NX=n, NY=n)
    t1 = time.perf_counter()
    eng = fe._ensure_engine(); eng.sync()
    t2 = time.perf_counter()
    fe._max_load_steps = 1
    fe.solve(min_step=50); eng.sync()
    t3 = time.perf_counter()
    print('%d^2: Model.mesh() %.1f ms | engine (context, materials, plfx_set_mesh_structured, plfx_set_grid) %.1f ms | solve() with 1 load step %.1f ms '
          '| mesh() + first solve() call = %.1f ms' % (n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0)), flush=True)
    fe._drop_engine()
