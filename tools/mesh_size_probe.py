#!/usr/bin/env python3
"""Multigrid on awkward mesh sizes: PCG iterations and time per solve for NX x NY Hill tension, a few load steps.
mesh_size_probe.py NX[,NY] ..."""
import os
import sys
import time
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pylabfea_amd as FE

for arg in sys.argv[1:]:
    nx, ny = (int(v) for v in (arg.split(',') + [arg])[:2])
    mat = bench.hill_material(FE)
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * ny / nx)
    fe.assign([mat])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.005 * fe.leny, 'disp')
    fe.mesh(NX=nx, NY=ny)
    fe._max_load_steps = 10
    eng = fe._ensure_engine()
    eng.sync()
    t = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=50)
    eng.sync()
    dt = time.perf_counter() - t
    its = [s[0] for s in fe.solver_stats]
    print('%5d x %-5d precond %s levels %d: %d load steps %.3f s, %d solves, PCG its total %d (max %d), niter %s, sgl_yy %.6f'
          % (nx, ny, eng.precond_info()[0], eng.precond_info()[1], fe.nsteps, dt, len(its), sum(its), max(its), fe.niter[-3:],
             fe.sgl[-1][1]))
