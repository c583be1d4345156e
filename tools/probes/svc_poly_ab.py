"""A/B of the SVC ray search: sampled-ray form (PLFX_SVC_POLY=1, default) against the FP32-screened evaluations of rounds 2-4
(PLFX_SVC_POLY=0) on a bounded config-4 sample.  Each variant runs in a process of its own (the knob is read once).
usage: python tools/probes/svc_poly_ab.py [n=128 [va vb]]    -> runs variants va and vb (default 0 and 2), compares fields, prints kernel times
       python tools/probes/svc_poly_ab.py run <n> <out.npz>  -> one variant (env decides)"""
import os
import subprocess
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(n, out):
    import warnings
    import pylabfea_amd as FE
    from pylabfea_amd import _lib
    import bench
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_hill.npz'))
    m = FE.Material(name='ML-Hill-p1')
    m.elasticity(CV=z['par_CV'])
    m.plasticity(sy=float(z['par_sy']), sdim=6)
    m.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
    fe = bench.tension_model(FE, m, n, 0.001)
    eng = fe._ensure_engine()
    eng.timing_reset()
    eng.timing_select((_lib.T_SWEEP, _lib.T_SWEEP_HEAVY))
    eng.timing_enable(True)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
    eng.sync()
    dt = time.perf_counter() - t0
    ms_l, n_l = eng.timing_get(_lib.T_SWEEP)
    ms_h, n_h = eng.timing_get(_lib.T_SWEEP_HEAVY)
    sig = np.array([el.sig for el in fe.element[:: max(1, fe.Nel // 4096)]])
    epl = np.array([el.epl for el in fe.element[:: max(1, fe.Nel // 4096)]])
    np.savez(out, sgl=np.array(fe.sgl), egl=np.array(fe.egl), epgl=np.array(fe.epgl), niter=np.array(fe.niter), u=fe.u, f=fe.f,
             sig=sig, epl=epl, times=np.array([dt, ms_l, n_l, ms_h, n_h]), nsteps=fe.nsteps, sweeps=fe.n_sweeps)
    print('POLY=%s n=%d: solve %.3f s, streaming %.1f ms / %d launches, corrector %.1f ms / %d launches, sweeps %d, niter %s'
          % (os.environ.get('PLFX_SVC_POLY', '1'), n, dt, ms_l, n_l, ms_h, n_h, fe.n_sweeps, list(fe.niter)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'run':
        return run(int(sys.argv[2]), sys.argv[3])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    va, vb = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ('0', '2')
    outs = {}
    for v in (va, vb):
        out = '/tmp/svc_poly_%s.npz' % v
        env = dict(os.environ, PLFX_SVC_POLY=v)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'run', str(n), out], env=env)
        outs[v] = np.load(out)
    a, b = outs[va], outs[vb]
    print('nsteps', int(a['nsteps']), int(b['nsteps']), 'niter equal', np.array_equal(a['niter'], b['niter']), 'sweeps', int(a['sweeps']), int(b['sweeps']))
    for k in ('sgl', 'egl', 'epgl', 'u', 'f', 'sig', 'epl'):
        if a[k].shape != b[k].shape:
            print(k, 'shape differs', a[k].shape, b[k].shape)
            continue
        sc = np.max(np.abs(a[k])) + 1e-300
        print('%-5s max |diff| / max |value| = %.3e' % (k, np.max(np.abs(a[k] - b[k])) / sc))
    ta, tb = a['times'], b['times']
    print('corrector ms: %.1f -> %.1f (x%.2f)   streaming ms: %.1f -> %.1f (x%.2f)   solve s: %.3f -> %.3f'
          % (ta[3], tb[3], ta[3] / tb[3], ta[1], tb[1], ta[1] / tb[1], ta[0], tb[0]))


if __name__ == '__main__':
    main()
