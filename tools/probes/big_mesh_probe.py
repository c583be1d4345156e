"""Mesh-independence probe: a y-laminate under uniaxial tension has fields that are uniform along y and piecewise constant per
section, so sgl must not depend on the mesh as long as the section boundaries fall on element edges."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pylabfea_amd as FE  # noqa: E402


def model(NX, NY, kind):
    a = FE.Material(num=1)
    a.elasticity(E=200.e3, nu=0.3)
    a.plasticity(sy=150., khard=500., sdim=6)
    b = FE.Material(num=2)
    if kind == 'svc':
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_gossbarlat.npz'))
        b.elasticity(CV=z['par_CV'])
        b.plasticity(sy=float(z['par_sy']), sdim=6)
        b.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
    else:
        b.elasticity(E=151220., nu=0.3)
        b.plasticity(sy=46.76, khard=0. if kind == 'ideal' else 200., sdim=6)
    fe = FE.Model(dim=2)
    fe.geom([2, 1, 2, 1, 2], LY=8.)
    fe.assign([a, b, a, b, a])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.003 * fe.leny, 'disp')
    fe.mesh(NX=NX, NY=NY)
    return fe


for arg in sys.argv[1:]:
    kind, nx, ny = arg.split(',')
    fe = model(int(nx), int(ny), kind)
    t = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=20)
    dt = time.perf_counter() - t
    its = [s[0] for s in fe.solver_stats]
    print('%-8s %5s x %-5s %7.1f s  K-its %4d  PCG its %6d (max %5d)  fall-backs %3d  heavy %8d  sgl_yy %s'
          % (kind, nx, ny, dt, sum(max(n, 0) + 1 for n in fe.niter), sum(its), max(its), fe._engine.solve_fallbacks(),
             int(np.sum(fe._state('max_steps') == 49)), np.round([s[1] for s in fe.sgl][-4:], 4).tolist()), flush=True)
    fe._drop_engine()
