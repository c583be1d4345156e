"""ML_full_yf on N points: one wave per point (k_full_yf_wave) against one thread per point (k_point_eval, PLFX_FULL_YF_WAVE=0);
states = the corrector's: stresses on rays through the yield locus, 0.9 ... 1.3 of the yield stress.  python tools/probes/full_yf_probe.py [N]"""
import os, sys, time, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import pylabfea_amd as FE
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_hill.npz'))
m = FE.Material(name='ML-Hill'); m.elasticity(CV=z['par_CV']); m.plasticity(sy=float(z['par_sy']), sdim=6)
m.set_svc(z['par_sv'], z['par_dual'], float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(3)
u = rng.normal(size=(N, 6)); u[:, :3] -= u[:, :3].mean(axis=1)[:, None]
seq = m.calc_seq(u)
sig = u / seq[:, None] * (m.sy * rng.uniform(0.9, 1.3, size=N))[:, None]
f = m.ML_full_yf(sig[:256], verb=False)          # warm-up (context, tables)
t = time.perf_counter(); f = m.ML_full_yf(sig, verb=False); dt = time.perf_counter() - t
print('%s: %d points in %.3f s = %.2f us per ML_full_yf call; checksum %.9e' % (os.environ.get('PLFX_FULL_YF_WAVE', '1'), N, dt, 1e6 * dt / N, float(np.sum(f))))
np.save('/tmp/fyf_%s.npy' % os.environ.get('PLFX_FULL_YF_WAVE', '1'), f)
