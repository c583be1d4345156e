#!/usr/bin/env python3
"""Two builds of libplfx.so on the same solves, compared BIT FOR BIT (u, sig, epl, sgl, PCG iterations per solve) -- for
changes that must not move a number (a kernel rewritten with the same sums in the same order).

    python tools/probes/lib_ab.py pylabfea_amd/libplfx_prev.so pylabfea_amd/libplfx.so

Cases: even mesh with a soft inclusion (fine + coarse generators, the whole V-cycle), odd mesh 201 x 199 (levels with a wider
last column / row: area-scaled diagonal, k_mg_coarsen_M), non-proportional laminate (per-column widths)."""
import hashlib
import os
import subprocess
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import pylabfea_amd as FE

    def digest(fe):
        h = hashlib.sha256()
        for a in (fe.u, fe._state('sig'), fe._state('epl'), np.asarray(fe.sgl), np.asarray(fe.egl)):
            h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        return h.hexdigest()[:16], [q[0] for q in fe.solver_stats]

    def hill(num=1, sy=100.):
        m = FE.Material(num=num)
        m.elasticity(E=200.e3, nu=0.3)
        m.plasticity(sy=sy, hill=[0.7, 1., 1.4, 1., 1.2, 0.8], khard=100., sdim=6)
        return m

    def finish(fe, nx, ny, steps, min_step):
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.004 * fe.leny, 'disp')
        fe.mesh(NX=nx, NY=ny)
        fe._max_load_steps = steps
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            fe.solve(min_step=min_step)
        return digest(fe)

    out = []
    fe = FE.Model(dim=2, planestress=False)   # three sections, the middle one soft: heterogeneous tangents
    fe.geom([2, 1, 2], LY=5.)
    fe.assign([hill(1), hill(2, 40.), hill(1)])
    out.append(('sections 320x256', finish(fe, 320, 256, 8, 12)))
    fe = FE.Model(dim=2, planestress=False)
    fe.geom([4.], LY=4. * 199 / 201)
    fe.assign([hill()])
    out.append(('odd 201x199', finish(fe, 201, 199, 5, 6)))
    ma, mb = hill(1, 150.), hill(2, 90.)
    fe = FE.Model(dim=2, planestress=True)
    fe.geom([3, 1, 2, 1, 2], LY=9. * 256 / 320)
    fe.assign([ma, mb, ma, mb, ma])
    out.append(('laminate 320x256', finish(fe, 320, 256, 6, 20)))
    for name, (d, its) in out:
        print('%s|%s|%s' % (name, d, ','.join(map(str, its))))


if __name__ == '__main__':
    if len(sys.argv) == 2 and sys.argv[1] == '--child':
        child()
        sys.exit(0)
    res = []
    for lib in sys.argv[1:3]:
        env = dict(os.environ, PLFX_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
        if o.returncode:
            print(o.stdout, o.stderr)
            sys.exit(1)
        res.append([l for l in o.stdout.splitlines() if l.count('|') == 2])
    bad = 0
    for a, b in zip(*res):
        same = a == b
        bad += not same
        print(('identical  ' if same else 'DIFFERENT  ') + a + ('' if same else '\n           ' + b))
    sys.exit(1 if bad or not res[0] or len(res[0]) != len(res[1]) else 0)
