#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not product code): records what the REFERENCE's `Material.elasticity()` / `Material.plasticity()`
(/root/reference/src/pylabfea/material.py:2401-2594) do for a list of argument combinations -- resulting attributes,
warnings, printed text, or the exception type and message -- as the fixture `tests/golden/material_api.json`, which
`tests/test_material_api.py` holds the package's `Material` to (same names, argument meaning and error behaviour).
Run in the build container only:
    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src python oracle/gen_material_api.py
"""
import contextlib
import copy
import io
import json
import os
import warnings

import numpy as np

ATTRS = ('E', 'nu', 'C11', 'C12', 'C44', 'CV', 'sy', 'sy0', 'khard', 'drucker', 'hill', 'hill_3p', 'hill_6p', 'sdim', 'tresca',
         'barlat')


def cases():
    el0 = dict(E=200e3, nu=0.3)
    cv = (np.diag([3., 3., 3., 1., 1., 1.]) + 0.5 * (np.ones((6, 6)) - np.eye(6)) * (np.arange(6)[:, None] < 3) * (np.arange(6)[None, :] < 3)).tolist()
    els = [el0, dict(C11=170e3, C12=120e3, C44=75e3), dict(CV=cv), dict(E=200e3), dict(E=200e3, nu=0.3, C11=1.),
           dict(C11=1., nu=0.3), dict(C11=1.), dict(), dict(C11=170e3, C12=120e3, C44=75e3, CV=np.eye(6).tolist())]
    h6, h3 = [0.7, 1., 1.4, 1., 1.2, 0.8], [0.7, 1., 1.4]
    pls = [None, dict(sy=100.), dict(sy=-1.), dict(sy=100., khard=-5.), dict(sy=100., sdim=3), dict(sy=100., sdim=4),
           dict(sy=100., hill=h6), dict(sy=100., hill=h3), dict(sy=100., hill=[1., 1., 1.]), dict(sy=100., hill=[1.] * 6),
           dict(sy=100., hill=h3, sdim=3), dict(sy=100., hill=h6, sdim=3), dict(sy=100., hill=[1., 1., 1.], sdim=3),
           dict(sy=100., rv=[1.2, 1., 0.8, 1., 1., 1.]), dict(sy=100., rv=[1.2, 1., 0.8], sdim=3), dict(sy=100., rv=[1.2, 1., 0.8]),
           dict(sy=100., rv=[1.2, 1., 0.8, 1.1, 0.9, 1.05]), dict(sy=100., rv=[1.2, 1., 0.8, 1., 1., 1.], hill=h6),
           dict(sy=100., hill=[0.7, 1., 1.4, 1.]), dict(sy=100., hill=h3, hill_6p=True), dict(sy=100., hill=h6, hill_3p=True),
           dict(sy=100., hill=h3, hill_3p=True), dict(sy=100., hill=h6, hill_6p=True), dict(sy=100., hill=h6, hill_6p=True, sdim=3),
           dict(sy=100., hill=h3, hill_3p=True, sdim=3), dict(sy=100., hill=h3, hill_3p=False, hill_6p=False),
           dict(sy=100., tresca=True, sdim=3), dict(sy=100., drucker=0.1, khard=50.),
           dict(sy=100., barlat=np.linspace(0.8, 1.2, 18).tolist(), barlat_exp=8), dict(sy=150., khard=500., sdim=6)]
    out = [(el0, p) for p in pls]
    out += [(e, p) for e in els[1:] for p in (None, dict(sy=100.))]
    return out


def run(Material, el, pl):
    """outcome of `m.elasticity(**el); m.plasticity(**pl)` on a fresh Material of the given class"""
    el, pl = copy.deepcopy(el), copy.deepcopy(pl)   # the reference extends the caller's `hill` list in place
    m = Material(name='m')
    buf = io.StringIO()
    with warnings.catch_warnings(record=True) as w, contextlib.redirect_stdout(buf):
        warnings.simplefilter('always')
        try:
            m.elasticity(**el)
            if pl is not None:
                m.plasticity(**pl)
        except Exception as e:  # noqa: BLE001 -- the type and text ARE the behaviour recorded
            return {'exception': type(e).__name__, 'message': str(e)}
    out = {}
    for k in ATTRS:
        v = getattr(m, k, None)
        out[k] = None if v is None else np.array(v, dtype=float).tolist()
    out['warnings'] = sorted(str(x.message) for x in w)
    out['printed'] = buf.getvalue()
    return out


if __name__ == '__main__':
    import pylabfea as REF
    rec = [{'elasticity': e, 'plasticity': p, 'outcome': run(REF.Material, e, p)} for e, p in cases()]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'material_api.json')
    with open(path, 'w') as fp:
        json.dump(rec, fp, indent=0)
    print(len(rec), 'cases ->', os.path.normpath(path))
