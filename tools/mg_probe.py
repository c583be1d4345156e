#!/usr/bin/env python3
"""PCG iteration counts of the multigrid preconditioner on ELASTIC two-material meshes: stiffness contrast and
alignment of the inclusion with the coarse grids.  `python tools/mg_probe.py [n] [omega] [nu]`."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pylabfea_amd as FE  # noqa: E402


def model(n, e2, lo, hi):
    a = FE.Material(num=1)
    a.elasticity(E=200.e3, nu=0.3)
    b = FE.Material(num=2)
    b.elasticity(E=e2, nu=0.27)
    fe = FE.Model(dim=2)
    fe.geom(sect=2, LX=4., LY=4.)
    fe.assign([a, b])
    fe.bcleft(0.)
    fe.bcbot(0.)
    fe.bcright(0., 'force')
    fe.bctop(0.001 * fe.leny, 'disp')
    el = np.ones((n, n))
    el[lo:hi, lo:hi] = 2
    fe.mesh(elmts=el, NX=n, NY=n)
    return fe


n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
omega = float(sys.argv[2]) if len(sys.argv) > 2 else 0.
nu = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for name, lo, hi in (('aligned n/4..3n/4', n // 4, 3 * n // 4), ('unaligned n/3..2n/3', n // 3, 2 * (n // 3)),
                     ('odd n/4+1..3n/4-1', n // 4 + 1, 3 * n // 4 - 1)):
    for e2 in (200.e3, 100.e3, 20.e3, 1.e3, 4.e7):
        fe = model(n, e2, lo, hi)
        eng = fe._ensure_engine()
        if omega > 0. or nu > 0:
            eng.set_precond(1, omega, nu)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve()
        print('n=%d %-22s E2/E1 = %-8g PCG its per solve %s' % (n, name, e2 / 200.e3, [s[0] for s in fe.solver_stats]))
        sys.stdout.flush()
