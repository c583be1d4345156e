#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not product code): records what the REFERENCE's pre-processing methods of `Model`
(/root/reference/src/pylabfea/model.py:514-757 `geom / assign / bcleft / bcright / bcbot / bctop / bcnode`, and the argument
checks of `mesh`, :758-830) do for a list of call sequences -- resulting attributes, warnings, or the exception type and
message -- as the fixture `tests/golden/model_api.json`, which `tests/test_model_api.py` holds the package's `Model` to.
Run in the build container only:
    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src python oracle/gen_model_api.py
"""
import contextlib
import io
import json
import os
import warnings

import numpy as np

ATTRS = ('dim', 'planestress', 'Nsec', 'LS', 'lenx', 'leny', 'thick', 'nonlin', 'bcl', 'bcr', 'bcb', 'bct', 'bcn', 'ubcleft',
         'ubcright', 'ubcbot', 'ubctop', 'ubcn', 'Nnode', 'NnodeX', 'NnodeY', 'Nel', 'Ndof', 'shapefact')

EL = ('material', dict(E=200e3, nu=0.3))                       # elastic material
PL = ('material', dict(E=200e3, nu=0.3), dict(sy=150., khard=500., sdim=6))  # plastic material


def cases():
    g = [('geom', dict(sect=[2., 1., 2.], LY=4.))]
    g1 = [('geom', dict(sect=1, LX=4., LY=4.))]
    return [
        g, g1, [('geom', dict(sect=3, LX=6., LY=2., LZ=0.5))], [('geom', dict(sect=0, LX=1.))], [('geom', dict(sect=2))],
        [('geom', dict(sect=1.5, LX=1.))], [('geom', dict(sect=(1., 2.)))],
        g + [('assign', [EL, PL, EL])], g + [('assign', [EL, EL, EL])], g + [('assign', [EL, PL])], g1 + [('assign', [PL])],
        g1 + [('bcleft', dict(val=0.))], g1 + [('bcleft', dict(val=0.1, bctype='disp', bcdir='y'))],
        g1 + [('bcleft', dict(val=0., bctype='force'))], g1 + [('bcleft', dict(val=1., bctype='force'))],
        g1 + [('bcleft', dict(val=0., bctype='pressure'))], g1 + [('bcleft', dict(val=0., bctype='DISP', bcdir='X'))],
        g1 + [('bcleft', dict(val=0., bcdir='z'))], g1 + [('bcleft', dict(val=0., bcdir=1))],
        g1 + [('bcbot', dict(val=0.))], g1 + [('bcbot', dict(val=0.2, bctype='disp', bcdir='x'))],
        g1 + [('bcbot', dict(val=2., bctype='force'))], g1 + [('bcbot', dict(val=0., bctype='none'))],
        g1 + [('bcright', dict(val=0., bctype='force'))], g1 + [('bcright', dict(val=0.01, bctype='disp'))],
        g1 + [('bcright', dict(val=5., bctype='force', bcdir='y'))], g1 + [('bcright', dict(val=0., bctype='fixed'))],
        g1 + [('bctop', dict(val=0.004, bctype='disp'))], g1 + [('bctop', dict(val=10., bctype='force'))],
        g1 + [('bctop', dict(val=0.004, bctype='disp', bcdir='x'))], g1 + [('bctop', dict(val=0., bctype='what'))],
        g1 + [('bctop', dict(val=0., bctype='disp', bcdir='q'))],
        g1 + [('assign', [EL]), ('bcleft', dict(val=0.)), ('bcbot', dict(val=0.)), ('bcright', dict(val=0., bctype='force')),
              ('bctop', dict(val=0.004, bctype='disp')), ('mesh', dict(NX=4, NY=3))],
        g + [('assign', [EL, PL, EL]), ('mesh', dict(NX=10, NY=2))], g + [('assign', [EL, PL, EL]), ('mesh', dict(NX=2, NY=2))],
        g1 + [('assign', [EL]), ('mesh', dict(NX=3, NY=3, SF=2))],
        g1 + [('assign', [EL]), ('mesh', dict(NX=2, NY=2)), ('bcnode', dict(node=4, val=0.01, bctype='disp', bcdir='y'))],
        g1 + [('assign', [EL]), ('mesh', dict(NX=2, NY=2)), ('bcnode', dict(node=[4, 5], val=3., bctype='force', bcdir='x'))],
        g1 + [('assign', [EL]), ('mesh', dict(NX=2, NY=2)), ('bcnode', dict(node=4, val=0., bctype='bad', bcdir='x'))],
        g1 + [('solve', dict())],
    ]


def run(pkg, seq, dim=2, planestress=False):
    """outcome of the call sequence on a fresh Model of the given package (`pkg.Model`, `pkg.Material`)"""
    def make(spec):
        m = pkg.Material(name='m')
        m.elasticity(**spec[1])
        if len(spec) > 2:
            m.plasticity(**spec[2])
        return m
    buf = io.StringIO()
    with warnings.catch_warnings(record=True) as w, contextlib.redirect_stdout(buf):
        warnings.simplefilter('always')
        try:
            fe = pkg.Model(dim=dim, planestress=planestress)
            for name, arg in seq:
                if name == 'assign':
                    fe.assign([make(s) for s in arg])
                else:
                    getattr(fe, name)(**arg)
        except Exception as e:  # noqa: BLE001 -- the type and text ARE the behaviour recorded
            return {'exception': type(e).__name__, 'message': str(e)}
    out = {}
    for k in ATTRS:
        v = getattr(fe, k, None)
        out[k] = None if v is None else np.array(v, dtype=float).tolist()
    if getattr(fe, 'noset', None) is not None:
        out['noset'] = np.ravel(np.array(fe.noset, dtype=float)).tolist()
    out['warnings'] = sorted(str(x.message) for x in w)
    return out


if __name__ == '__main__':
    import pylabfea as REF
    rec = [{'calls': seq, 'outcome': run(REF, seq)} for seq in cases()]
    rec.append({'calls': 'Model(dim=3)', 'outcome': run(REF, [], dim=3)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'model_api.json')
    with open(path, 'w') as fp:
        json.dump(rec, fp, indent=0)
    print(len(rec), 'cases ->', os.path.normpath(path), ' exceptions:', sum('exception' in r['outcome'] for r in rec))
