"""The drop-in boundary on the model side: `Model.geom / assign / bcleft / bcright / bcbot / bctop / bcnode` and the argument
checks of `mesh` / `solve` against what the REFERENCE's methods (model.py:514-830, 1024-1025) did for the same call sequences
-- attributes, warnings, exception type and message.  Fixture: tests/golden/model_api.json, recorded from the reference by
oracle/gen_model_api.py.  No GPU: nothing here reaches the engine."""
import json
import os

import pytest

from oracle.gen_model_api import cases, run  # test infrastructure: the recorder's own call list and harness, applied to the package

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'model_api.json')) as fp:
    CASES = json.load(fp)

CALLS = cases()   # the Python objects (a tuple stays a tuple); the fixture holds the same list as JSON
assert len(CALLS) == len(CASES) - 1 and all(json.loads(json.dumps(a)) == c['calls'] for a, c in zip(CALLS, CASES))

# Differences that are deliberate, by index into the call list: what is compared instead
TYPE_ONLY = {
    31: "bctop's message for an unknown direction names 'bcleft' in the reference (model.py:705, copied from bcleft)",
    35: "the reference's message lacks a blank ('elementswith', model.py:361-362)",
}
EXTENSION = {
    18: "bcdir given as 0 / 1 is accepted here; the reference calls .lower() on it and dies with AttributeError",
}


@pytest.mark.parametrize('k', range(len(CASES)))
def test_model_preprocessing_matches_reference(k):
    import pylabfea_amd as pkg
    c = CASES[k]
    got = run(pkg, [], dim=3) if isinstance(c['calls'], str) else run(pkg, CALLS[k])
    want = c['outcome']
    if k in EXTENSION:
        assert want['exception'] == 'AttributeError' and 'exception' not in got, EXTENSION[k]
        assert got['bcl'] == [0., 0.] and got['ubcleft'] == [1., 1.]
    elif k in TYPE_ONLY:
        assert got['exception'] == want['exception'] and got['message'] != want['message'], TYPE_ONLY[k]
    else:
        assert got == want, c['calls']
