"""Fixture generator: equivalent stress / yield function of PRINCIPAL-stress materials (sdim = 3, 3-parameter Hill) on general
3-d stress states with out-of-plane shear, computed by the UNMODIFIED reference imported from /root/reference/src.  There
calc_seq reduces every Voigt stress with basic.sig_princ (np.linalg.eig + the axis-tracking re-ordering, basic.py:153-175),
whose order depends on LAPACK for such states and enters the 3-parameter Hill form (material.py:667-670).
Writes tests/golden/princ_general.npz.  Test infrastructure; needs /root/reference (build container only).

    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src python oracle/gen_princ_general.py
"""
import os
import warnings

import numpy as np

os.environ.setdefault('MPLBACKEND', 'Agg')
import pylabfea as FE  # noqa: E402  (the reference)
from pylabfea.basic import sig_princ  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    out = os.path.join(ROOT, 'tests', 'golden', 'princ_general.npz')
    np.savez_compressed(out, **rec)
    print('wrote', out, {k: v.shape for k, v in rec.items()})


if __name__ == '__main__':
    main()
