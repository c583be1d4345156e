#!/usr/bin/env python3
"""Mid-size fixtures from the PINNED ORACLE (test infrastructure; build container, no GPU): BASELINE configs 4 and 5 on meshes
between the 32-element-wide oracle comparisons the GPU suite runs live and the full sizes -- config 4 on 64 x 64 (all 11 load
steps) and config 5 on 128 x 64 (all 20 load steps) -- solved by oracle/solve_ref.py (sparse direct solve + the C restatement
of Material.response, itself pinned against the reference's vectors by tests/test_oracle_golden.py / test_oracle_solve.py).
The reference cannot produce these: 0.3 - 0.45 s per sub-stepped SVC response call, 65 000 / 350 000 of them.  Running the oracle
inside the GPU suite would take 1.5 / 13 minutes of host time; its result is a fixture instead (tests/golden/mid_configs.npz,
fields in float64, a few hundred KB).

    python oracle/gen_mid_configs.py [4] [5]
"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle.solve_ref import RefSolver  # noqa: E402
from test_gpu_model import svc_material, tension_model  # noqa: E402
from test_gpu_configs import laminate_cfg5  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
OUT = os.path.join(GOLD, 'mid_configs.npz')


def record(rec, p, ref, dt):
    rec[p + '_nsteps'] = np.array(ref.nsteps)
    rec[p + '_niter'] = np.asarray(ref.niter)
    rec[p + '_co_nconv'] = np.asarray(ref.co_nconv)
    for k in ('u', 'sig', 'epl', 'eps', 'sgl', 'egl', 'epgl'):
        rec[p + '_' + k] = np.asarray(getattr(ref, k), dtype=float)
    rec[p + '_seconds'] = np.array(dt)


def larger(which):
    """One octave up (round 6, end): config 4 on 128 x 128 and config 5 on 256 x 128 -> tests/golden/mid_configs_128.npz
    (about 20 and 40 minutes of the oracle on 8 cores); fields in float32 steps would lose the comparison: float64, compressed."""
    out = os.path.join(GOLD, 'mid_configs_128.npz')
    rec = dict(np.load(out)) if os.path.exists(out) else {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if 4 in which:
            t = time.time()
            ref = RefSolver(tension_model(svc_material(GOLD, 'hill'), 128, 0.001)).solve(min_step=10)
            record(rec, 'cfg4_128', ref, time.time() - t)
            print('config 4 on 128 x 128: %.0f s, %d load steps, niter %s, sgl_yy %.6f' % (time.time() - t, ref.nsteps, list(ref.niter), ref.sgl[-1][1]), flush=True)
            np.savez_compressed(out, **rec)
        if 5 in which:
            t = time.time()
            ref = RefSolver(laminate_cfg5(GOLD, 256, 128)).solve(min_step=20)
            record(rec, 'cfg5_256x128', ref, time.time() - t)
            print('config 5 on 256 x 128: %.0f s, %d load steps, niter %s, sgl_yy %.6f' % (time.time() - t, ref.nsteps, list(ref.niter), ref.sgl[-1][1]), flush=True)
            np.savez_compressed(out, **rec)


def main(which):
    rec = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if 4 in which:
            t = time.time()
            ref = RefSolver(tension_model(svc_material(GOLD, 'hill'), 64, 0.001)).solve(min_step=10)
            record(rec, 'cfg4_64', ref, time.time() - t)
            print('config 4 on 64 x 64: %.0f s, %d load steps, niter %s, sgl_yy %.6f' % (time.time() - t, ref.nsteps, list(ref.niter), ref.sgl[-1][1]))
            np.savez_compressed(OUT, **rec)
        if 5 in which:
            t = time.time()
            ref = RefSolver(laminate_cfg5(GOLD, 128, 64)).solve(min_step=20)
            record(rec, 'cfg5_128x64', ref, time.time() - t)
            print('config 5 on 128 x 64: %.0f s, %d load steps, niter %s, sgl_yy %.6f' % (time.time() - t, ref.nsteps, list(ref.niter), ref.sgl[-1][1]))
            np.savez_compressed(OUT, **rec)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--larger':
        larger([int(a) for a in sys.argv[2:]] or [4, 5])
    else:
        main([int(a) for a in sys.argv[1:]] or [4, 5])
