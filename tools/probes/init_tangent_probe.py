#!/usr/bin/env python3
"""Why does k_init_tangent take 21 ms per launch under rocprofv3?  Host-side timing of plfx_state_reset
(memsets + k_init_tangent + sync) on a 1024^2 mesh, first and repeated calls, without a profiler attached."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import pylabfea_amd as FE
mat = FE.Material(); mat.elasticity(E=200.e3, nu=0.3)
mat.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
for trial in range(2):
    fe = FE.Model(dim=2, planestress=False); fe.geom([4.], LY=4.); fe.assign([mat])
    fe.bcleft(0.); fe.bcbot(0.); fe.bcright(0., 'force'); fe.bctop(0.005 * fe.leny, 'disp')
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    fe.mesh(NX=N, NY=N)
    t0 = time.perf_counter()
    eng = fe._ensure_engine()
    eng.sync()
    t1 = time.perf_counter()
    ts = []
    for rep in range(4):
        a = time.perf_counter(); eng.state_reset(); eng.sync(); ts.append(1e3 * (time.perf_counter() - a))
    print('model %d: engine set-up (set_mesh incl. the first state_reset) %.1f ms; state_reset again: %s ms'
          % (trial, 1e3 * (t1 - t0), ['%.2f' % t for t in ts]), flush=True)
    fe._drop_engine()
