#!/usr/bin/env python3
"""Where does Model.mesh() -> first solve() set-up time go?  (host NumPy index work vs plfx_set_mesh / plfx_set_grid)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import pylabfea_amd as FE
from pylabfea_amd import _lib
for n in (1024, 2048):
    mat = FE.Material(); mat.elasticity(E=200.e3, nu=0.3)
    mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    t0 = time.perf_counter()
    fe = FE.Model(dim=2, planestress=False); fe.geom([4.], LY=4.); fe.assign([mat])
    fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.005 * fe.leny, 'disp')
    fe.mesh(NX=n, NY=n)
    t1 = time.perf_counter()
    eng = _lib.Context(0)
    eng.set_materials([mat._record(fe._element_CV(mat))])
    t2 = time.perf_counter()
    eng.set_mesh(fe._conn, fe._mat_id, fe._lxy, fe.Nnode, fe.thick, fe.planestress)
    eng.sync(); t3 = time.perf_counter()
    eng.set_grid(n, n)
    eng.sync(); t4 = time.perf_counter()
    eng.close()
    fe._max_load_steps = 1
    t5 = time.perf_counter()
    fe.solve(min_step=50)
    t6 = time.perf_counter()
    print('%d^2: Model.mesh() %.0f ms | context + materials %.0f ms | plfx_set_mesh %.0f ms | plfx_set_grid %.0f ms | first solve() call with 1 load step (incl. engine set-up again) %.0f ms'
          % (n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t6 - t5)), flush=True)
