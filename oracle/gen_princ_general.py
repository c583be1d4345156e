"""Fixture generator: equivalent stress / yield function of PRINCIPAL-stress materials (sdim = 3, 3-parameter Hill) on general
3-d stress states with out-of-plane shear, computed by the UNMODIFIED reference imported from /root/reference/src.  There
calc_seq reduces every Voigt stress with basic.sig_princ (np.linalg.eig + the axis-tracking re-ordering, basic.py:153-175),
whose order depends on LAPACK for such states and enters the 3-parameter Hill form (material.py:667-670).
Writes tests/golden/princ_general.npz.  Test infrastructure; needs /root/reference (build container only).

    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src python oracle/gen_princ_general.py
"""
import os
import sys
import warnings

import numpy as np

os.environ.setdefault('MPLBACKEND', 'Agg')
import pylabfea as FE  # noqa: E402  (the reference)
from pylabfea.basic import sig_princ  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def main():
    rng = np.random.default_rng(77)
    n = 400
    sig = rng.normal(size=(n, 6)) * 60.
    sig[:40, 3:5] = 0.                      # plane states among them
    sig[40:80, 3:] *= 1e-3                  # nearly diagonal: dominated axes
    sig[80:100, 0:3] = sig[80:100, 0:1]     # equal normal stresses + shear
    rec = {'sig': sig}
    rec['princ'], rec['evec'] = sig_princ(sig)
    hill = [0.7, 1.0, 1.4]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = FE.Material()
        m.elasticity(E=200.e3, nu=0.3)
        m.plasticity(sy=100., hill=hill, khard=100., drucker=0.05, sdim=3)
        rec['par'] = np.array([200.e3, 0.3, 100., 100., 0.05])
        rec['hill'] = np.array(hill)
        rec['seq'] = m.calc_seq(sig)
        epl = rng.normal(size=(n, 6)) * 1e-3
        rec['epl'] = epl
        rec['yf'] = np.array([m.calc_yf(sig[i], epl=epl[i]) for i in range(n)])
        rec['seq_single'] = np.array([m.calc_seq(sig[i]) for i in range(0, n, 7)])
        mt = FE.Material()
        mt.elasticity(E=200.e3, nu=0.3)
        mt.plasticity(sy=100., tresca=True, sdim=3)
        rec['tresca_seq'] = mt.calc_seq(sig)
        # 2-feature SVC of sdim = 3 (fixture svc_hill3d.npz: test_ml_plasticity's training): the features (J2 stress, polar angle
        # on the deviatoric plane) come from the reference's create_scaled_input -> sig_princ, the decision function from the
        # fixture's support vectors by its formula (pinned against scikit-learn by tests/test_oracle_golden.py)
        zs = np.load(os.path.join(ROOT, 'tests', 'golden', 'svc_hill3d.npz'))
        ml = FE.Material()
        ml.elasticity(E=float(zs['par_E']), nu=float(zs['par_nu']))
        ml.plasticity(sy=float(zs['par_sy']), hill=list(zs['par_hill']), sdim=3)
        ml.scale_seq = float(zs['par_scale_seq'])
        ml.Ndof = 2
        x = ml.create_scaled_input(sig)
        rec['ml3_x'] = x
        d2 = np.sum((x[:, None, :] - zs['par_sv'][None, :, :]) ** 2, axis=2)
        rec['ml3_yf'] = np.exp(-float(zs['par_gamma']) * d2) @ zs['par_dual'] + float(zs['par_intercept'])
        # ---- round 5: calc_fgrad with a (6,) stress, ML_full_yf and response on the same kind of states
        # calc_fgrad(sig (6,)) of the analytic sdim = 3 material: seq in sig_princ's order over the deviator of the VOIGT
        # normals (material.py:834-838), and the same with the `seq` argument
        rec['fgrad6'] = np.array([m.calc_fgrad(sig[i]) for i in range(n)])
        rec['fgrad6_seq_in'] = rng.uniform(40., 160., size=n)
        rec['fgrad6_seq'] = np.array([m.calc_fgrad(sig[i], seq=rec['fgrad6_seq_in'][i]) for i in range(n)])
        m6 = FE.Material()
        m6.elasticity(E=200.e3, nu=0.3)
        m6.plasticity(sy=100., hill=[0.7, 1.0, 1.4, 1.2, 0.8, 1.1], khard=100., drucker=0.05, sdim=6)
        rec['hill6'] = np.array([0.7, 1.0, 1.4, 1.2, 0.8, 1.1])
        rec['fgrad6_seq_sdim6'] = np.array([m6.calc_fgrad(sig[i], seq=rec['fgrad6_seq_in'][i]) for i in range(n)])
        # ML_full_yf of the 2-feature SVC (a stand-in object with the fixture's decision function takes the place of the
        # scikit-learn estimator; everything else is the reference's code)

        class Svm:
            def decision_function(self, x):
                d2 = np.sum((x[:, None, :] - zs['par_sv'][None, :, :]) ** 2, axis=2)
                return np.exp(-float(zs['par_gamma']) * d2) @ zs['par_dual'] + float(zs['par_intercept'])
        ml.svm_yf = Svm()
        ml.ML_yf = True
        ml.ML_grad = False
        nf = 160
        sf = sig[:nf] / np.array([m.calc_seq(sig[i]) for i in range(nf)])[:, None] \
            * (float(zs['par_sy']) * rng.uniform(0.3, 1.6, size=nf))[:, None]
        rec['ml3_full_sig'] = sf
        rec['ml3_full_yf'] = np.array([ml.ML_full_yf(sf[i], verb=False) for i in range(nf)])
        rec['ml3_full_yf_check'] = np.array([ml.calc_yf(sf[i]) for i in range(nf)])
        # response of the analytic material on general 3-d states (plane strain stiffness): every calc_yf / calc_seq inside
        # re-orders the principal stresses with LAPACK's output of that moment
        from gen_golden import gen_response_inputs, run_response, element_CV
        CVr = element_CV(m, False)
        s, e, d = gen_response_inputs(m, CVr, np.random.default_rng(5), 160, False)
        fy, so, dp, ct, ns = run_response(m, s, e, d, CVr)
        print('response on general states: nsteps histogram', np.bincount(ns))
        rec['r_CV'], rec['r_sig'], rec['r_epl'], rec['r_deps'] = CVr, s, e, d
        rec['r_fy'], rec['r_sig_out'], rec['r_depl'], rec['r_ct'], rec['r_nsteps'] = fy, so, dp, ct, ns
        # epl_dot / C_tan as point functions on the same inputs: ONE principal-stress reduction each (material.py:1044-1047,
        # 1079-1081), the normal taken w.r.t. the principal stresses -- sdim = 3 and, for comparison, the sdim = 6 material
        rec['ed_pdot'] = np.array([m.epl_dot(s[i], e[i], CVr, d[i]) for i in range(len(s))])
        rec['ed_ctan'] = np.array([m.C_tan(s[i], CVr, epl=e[i]).reshape(36) for i in range(len(s))])
        rec['ed_pdot6'] = np.array([m6.epl_dot(s[i], e[i], CVr, d[i]) for i in range(len(s))])
        rec['ed_ctan6'] = np.array([m6.C_tan(s[i], CVr, epl=e[i]).reshape(36) for i in range(len(s))])
        # ---- round 6: Material.response(..., maxit) with maxit != 50 (material.py:207, 288-291) on the same inputs: the sdim = 3
        # material (every sub-step re-orders the principal stresses) and the sdim = 6 one
        for mi in (20, 7):
            for tag, mat in (('', m), ('6', m6)):
                fy = np.zeros(len(s)); so = np.zeros((len(s), 6)); dp = np.zeros((len(s), 6)); ct = np.zeros((len(s), 36))
                ns = np.zeros(len(s), dtype=np.int32)
                for i in range(len(s)):
                    mat.msg['nsteps'] = -1
                    f, so[i], dp[i], c = mat.response(s[i], e[i], d[i], CVr, maxit=mi)
                    fy[i], ct[i], ns[i] = f, np.asarray(c).reshape(36), mat.msg['nsteps']
                k = 'rm%d%s_' % (mi, tag)
                rec[k + 'fy'], rec[k + 'sig_out'], rec[k + 'depl'], rec[k + 'ct'], rec[k + 'nsteps'] = fy, so, dp, ct, ns
                print('response maxit=%d sdim %s: nsteps histogram' % (mi, tag or '3'), np.bincount(ns))
    out = os.path.join(ROOT, 'tests', 'golden', 'princ_general.npz')
    old = dict(np.load(out)) if os.path.exists(out) else {}
    for k, v in old.items():   # the keys of earlier rounds must not move (same seeds, same library)
        assert k in rec and np.array_equal(np.asarray(rec[k]), v, equal_nan=True), k
    np.savez_compressed(out, **rec)
    print('wrote', out, {k: v.shape for k, v in rec.items()})


if __name__ == '__main__':
    main()
