"""How much of the SVC corrector is NOT support-vector work?  The same decision function with more support vectors (every third
vector listed twice with half its dual coefficient: 1585 -> 1982 vectors, still in LDS) runs the same control flow; the time
difference prices 397 vectors, the rest of a launch is the part that does not scale with the vector count.
usage: python tools/probes/svc_scalar_share.py [n=128]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pylabfea_amd as FE
from pylabfea_amd import _lib
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_hill.npz'))
res = {}
extras = [int(a) for a in sys.argv[2:]] or [0, 397]
for extra in extras:
    tag = str(1585 + extra)
    sv, dual = z['par_sv'], z['par_dual']
    if extra:
        rep = np.ones(len(dual), dtype=int)
        rep[:extra] = 2
        sv = np.repeat(sv, rep, axis=0)
        dual = np.repeat(dual / rep, rep)
    m = FE.Material(name='ML-Hill-' + tag)
    m.elasticity(CV=z['par_CV'])
    m.plasticity(sy=float(z['par_sy']), sdim=6)
    m.set_svc(sv, dual, float(z['par_intercept']), float(z['par_gamma']), float(z['par_scale_seq']))
    fe = bench.tension_model(FE, m, n, 0.001)
    eng = fe._ensure_engine()
    eng.timing_reset(); eng.timing_select((_lib.T_SWEEP, _lib.T_SWEEP_HEAVY)); eng.timing_enable(True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fe.solve(min_step=10)
    eng.sync()
    ms_l, _ = eng.timing_get(_lib.T_SWEEP); ms_h, _ = eng.timing_get(_lib.T_SWEEP_HEAVY)
    res[tag] = (len(dual), ms_l, ms_h, list(fe.niter), fe.sgl[-1][1])
    print('%s vectors (padded to %d): streaming %.2f ms, corrector %.2f ms, niter %s, sgl_yy %.9f' % (len(dual), (len(dual) + 63) // 64 * 64, ms_l, ms_h, list(fe.niter), fe.sgl[-1][1]))
    fe._drop_engine()
(na, la, ha, _, _), (nb, lb, hb, _, _) = res[str(1585 + extras[0])], res[str(1585 + extras[-1])]
pa, pb = (na + 63) // 64 * 64, (nb + 63) // 64 * 64
for name, a, b in (('streaming', la, lb), ('corrector', ha, hb)):
    per = (b - a) / (pb - pa)
    print('%s: %.4f ms per padded vector -> %.1f ms of the %.1f ms with %d vectors are support-vector loops (%.0f %%), %.1f ms are not'
          % (name, per, per * pa, a, pa, 100. * per * pa / a, a - per * pa))
