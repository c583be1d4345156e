#!/usr/bin/env python3
"""Robustness sweep (GPU): the multigrid / matrix-free default path against the Jacobi-PCG / assembled path of the same
library on mesh sizes that the parity fixtures do not visit (odd, prime, strongly non-square, one element thick): same
plastic tension problem, a few load steps each; prints the largest difference of the global stress / strain histories and
the iteration bookkeeping.  The two paths share the sweep kernels but nothing of the solver.
    python tools/probes/size_sweep.py"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pylabfea_amd as FE  # noqa: E402


def run(nx, ny, precond, operator, incl):
    m = FE.Material(name='hill')
    m.elasticity(E=200.e3, nu=0.3)
    m.plasticity(sy=100., hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
    mats = [m]
    fe = FE.Model(dim=2, planestress=False)
    fe.precond, fe.operator = precond, operator
    if incl:
        soft = FE.Material(name='soft')
        soft.elasticity(E=1.e3, nu=0.3)
        mats.append(soft)
        fe.geom([2., 2.], LY=4. * ny / nx)
    else:
        fe.geom([4.], LY=4. * ny / nx)
    fe.assign(mats)
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.004 * 4. * ny / nx, 'disp')
    if incl:
        el = np.ones((nx, ny), dtype=int)
        el[nx // 3:max(nx // 3 + 1, 2 * nx // 3), ny // 3:max(ny // 3 + 1, 2 * ny // 3)] = 2
        fe.mesh(elmts=el, NX=nx, NY=ny)
    else:
        fe.mesh(NX=nx, NY=ny)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        fe.solve(min_step=8)
    return fe, [str(x.message)[:80] for x in w]


cases = [(1, 1), (2, 1), (1, 7), (3, 3), (5, 9), (17, 17), (31, 2), (2, 64), (37, 41), (96, 96), (100, 100), (127, 129), (250, 10),
         (257, 255), (384, 384), (500, 500), (513, 511)]
worst = 0.
for incl in (False, True):
    for nx, ny in cases:
        if incl and min(nx, ny) < 3:
            continue
        t = time.time()
        a, wa = run(nx, ny, None, None, incl)
        b, wb = run(nx, ny, 0, 0, incl)
        d = max(np.max(np.abs(a.sgl - b.sgl)) / max(1., np.max(np.abs(b.sgl))), np.max(np.abs(a.egl - b.egl)) / np.max(np.abs(b.egl)))
        same = (a.nsteps == b.nsteps and list(a.niter) == list(b.niter))
        worst = max(worst, d)
        its = lambda f: sum(i for i, _ in f.solver_stats)
        print('%s %4d x %-4d steps %2d  K-its %-28s  same bookkeeping %s  max rel diff %.2e  PCG its default %5d / jacobi %6d  %.1fs %s'
              % ('incl' if incl else 'homo', nx, ny, a.nsteps, list(a.niter), same, d, its(a), its(b), time.time() - t,
                 (wa + wb)[:2] if (wa or wb) else ''), flush=True)
print('worst relative difference %.2e' % worst)
