"""Wall-clock of the reference's everyday sizes on the GPU engine: calc_properties (2 x 2 elements, four load cases) and an
8 x 8 / 32 x 32 Hill tension solve with 50 increments -- per load step and per K-iteration.  python tools/probes/small_mesh_latency.py"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE
warnings.simplefilter('ignore')
m = FE.Material(name='hill')
m.elasticity(E=200.e3, nu=0.3)
m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
for rep in range(2):
    t = time.perf_counter()
    m.calc_properties(eps=0.01, sigeps=True, min_step=12)
    print('calc_properties (4 load cases, 2x2 elements, min_step=12): %.3f s' % (time.perf_counter() - t))
for n in (8, 32, 128):
    for rep in range(2):
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([4.], LY=4.); fe.assign([m]); fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.005 * fe.leny, 'disp')
        fe.mesh(NX=n, NY=n)
        t = time.perf_counter()
        fe.solve(min_step=50)
        dt = time.perf_counter() - t
    print('%3d x %3d Hill tension, 50 increments: %.3f s  (%d load steps, %d sweeps, %d solves) = %.2f ms per load step, %.2f ms per sweep+solve'
          % (n, n, dt, fe.nsteps, fe.n_sweeps, len(fe.solver_stats), 1e3 * dt / fe.nsteps, 1e3 * dt / max(fe.n_sweeps, 1)))
