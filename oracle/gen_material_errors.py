#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not product code): records the exception type and message of the REFERENCE's point functions
(/root/reference/src/pylabfea/material.py: `response` :207-346, `calc_yf` :348-412, `ML_full_yf` :414-516, `calc_seq` :576-676,
`calc_seqB` :678-702, `calc_fgrad` :704-858) for malformed arguments and unsupported materials -- the argument checks that run
BEFORE any arithmetic -- as the fixture `tests/golden/material_errors.json` for `tests/test_material_errors.py`.
Run in the build container only:
    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src python oracle/gen_material_errors.py
"""
import json
import os
import warnings

import numpy as np

EL = dict(E=200e3, nu=0.3)
MATS = {
    'elastic': (EL, None),
    'j2': (EL, dict(sy=150., khard=500., sdim=6)),
    'tresca': (EL, dict(sy=150., tresca=True, sdim=3)),
    'barlat': (EL, dict(sy=150., barlat=np.linspace(0.8, 1.2, 18).tolist(), barlat_exp=8, sdim=6)),
}
Z6 = [0.] * 6
S6 = [10., 80., 30., 0., 0., 5.]
CALLS = [
    ('j2', 'response', dict(sig=[Z6, Z6], epl=Z6, deps=Z6, CV=None)),
    ('j2', 'response', dict(sig=[0.] * 5, epl=Z6, deps=Z6, CV=None)),
    ('elastic', 'response', dict(sig=S6, epl=Z6, deps=[1e-4] * 6, CV=None)),
    ('j2', 'calc_seq', dict(sig=[0.] * 5)),
    ('j2', 'calc_seq', dict(sig=[[0.] * 4] * 2)),
    ('j2', 'calc_fgrad', dict(sig=S6, epl=[0.] * 3)),
    ('j2', 'calc_fgrad', dict(sig=[[0.] * 4] * 2)),
    ('tresca', 'calc_fgrad', dict(sig=[10., 80., 30.])),
    ('barlat', 'calc_fgrad', dict(sig=S6)),
    ('j2', 'ML_full_yf', dict(sig=[S6, S6])),
    ('j2', 'calc_seqB', dict(sv=S6)),
    ('j2', 'export_MLparam', dict(sname='x')),
]


def run(Material, mat, method, kw):
    el, pl = MATS[mat]
    m = Material(name='m')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.elasticity(**el)
        if pl is not None:
            m.plasticity(**pl)
        kw = {k: (np.array(v, dtype=float) if isinstance(v, list) else v) for k, v in kw.items()}
        if 'CV' in kw and kw['CV'] is None:
            kw['CV'] = m.CV
        try:
            getattr(m, method)(**kw)
        except Exception as e:  # noqa: BLE001 -- the type and text ARE the behaviour recorded
            return {'exception': type(e).__name__, 'message': str(e)}
    return {'exception': None}


if __name__ == '__main__':
    import pylabfea as REF
    rec = [{'material': a, 'method': b, 'args': c, 'outcome': run(REF.Material, a, b, c)} for a, b, c in CALLS]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'material_errors.json')
    with open(path, 'w') as fp:
        json.dump(rec, fp, indent=0)
    for r in rec:
        print(r['material'], r['method'], '->', r['outcome'])
