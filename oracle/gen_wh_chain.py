#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, build container only): a Model.solve trace of the unmodified reference in which
the SEQUENTIAL carry of Material.khard through the element loop matters (material.py:808-814, model.py:1340-1359) -- simple
shear of a 4 x 4 mesh with the work-hardening SVC of oracle/gen_golden.py:train_hardening: the gradient evaluations of different
elements leave different, positive hardening moduli behind (in the uniaxial case of svc_workhard.npz every call leaves 0).

    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src python oracle/gen_wh_chain.py

Writes tests/golden/svc_workhard_chain.npz: the trained parameters (par_*) and the trace (whs_*), plus -- captured from the
reference's own stack -- the modulus the Material object held after every response() call of the run (whs_khard_calls); and a
second trace (whl_*): a 6 x 4 laminate [SVC | J2 | the same SVC object] under the same shear."""
import contextlib
import io
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
FE = G.FE


def main():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with contextlib.redirect_stdout(io.StringIO()):
            ml, _ = G.train_hardening()
        rec = {('par_' + k): v for k, v in G.svc_params(ml).items()}
        rec['par_scale_wh'] = np.array(float(ml.scale_wh))
        rec['par_ind_wh'] = np.array(int(ml.ind_wh))
        rec['par_epc'] = np.array(float(ml.epc))
        ml.khard = 0.
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([4.], LY=4.)
        fe.assign([ml])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.006 * fe.leny, 'disp', 'x')
        fe.mesh(NX=4, NY=4)
        calls = []
        orig = ml.response

        def spy(*a, **k):
            out = orig(*a, **k)
            calls.append(float(ml.khard))
            return out
        ml.response = spy
        t = time.time()
        with G.SolveTracer() as tr:
            fe.solve(min_step=8)
        G.solve_record(fe, 'whs', rec, time.time() - t)
        tr.store(rec, 'whs')
        rec['whs_khard_final'] = np.array(float(ml.khard))
        rec['whs_khard_calls'] = np.array(calls)
        print('whs %.1fs' % rec['whs_tsolve'], fe.nsteps, fe.niter, fe.sgl[-1], 'khard', ml.khard,
              'calls', len(calls), 'with khard > 0:', int(np.sum(np.array(calls) > 0)))
        # second trace: a laminate [work-hardening SVC | J2 | the SAME SVC object] -- assign() stores the object twice, so the modulus
        # runs from the last element of the first section straight into the first element of the third (the chain skips the elements
        # of the other object)
        ml.response = orig
        ml.khard = 0.
        j2 = FE.Material(name='J2', num=2)
        j2.elasticity(E=200.e3, nu=0.3)
        j2.plasticity(sy=60., khard=1000., sdim=6)
        fe = FE.Model(dim=2, planestress=False)
        fe.geom([2., 2., 2.], LY=4.)
        fe.assign([ml, j2, ml])
        fe.bcleft(0.)
        fe.bcbot(0.)
        fe.bcright(0., 'force')
        fe.bctop(0.006 * fe.leny, 'disp', 'x')
        fe.mesh(NX=6, NY=4)
        calls2 = []

        def spy2(*a, **k):
            out = orig(*a, **k)
            calls2.append(float(ml.khard))
            return out
        ml.response = spy2
        t = time.time()
        with G.SolveTracer() as tr:
            fe.solve(min_step=6)
        G.solve_record(fe, 'whl', rec, time.time() - t)
        tr.store(rec, 'whl')
        rec['whl_khard_final'] = np.array(float(ml.khard))
        rec['whl_khard_calls'] = np.array(calls2)
        print('whl %.1fs' % rec['whl_tsolve'], fe.nsteps, fe.niter, fe.sgl[-1], 'khard', ml.khard, 'calls', len(calls2))
    np.savez_compressed(os.path.join(G.OUT, 'svc_workhard_chain.npz'), **rec)


if __name__ == '__main__':
    main()
