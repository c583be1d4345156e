"""Argument checks of the point functions (`response, calc_seq, calc_seqB, calc_fgrad, ML_full_yf, export_MLparam`) against the
exception type and message the REFERENCE raises for the same malformed call (material.py:207-858).  Fixture:
tests/golden/material_errors.json, recorded by oracle/gen_material_errors.py.  No GPU: every check runs before the engine is
touched (which is itself part of the behaviour tested: a malformed call must not die with a missing-device error)."""
import json
import os

import pytest

from oracle.gen_material_errors import CALLS, run  # test infrastructure: the recorder's own harness, applied to the package

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'material_errors.json')) as fp:
    CASES = json.load(fp)

# where the reference dies somewhere inside its arithmetic, the package raises on entry: same class of error, own message
OWN_MESSAGE = {
    2: 'elastic material: the reference fails on `None * float` inside get_sflow (TypeError); the package says what is wrong',
    9: 'ML_full_yf accepts (N, 6) batches here (extension); the non-ML material is what is refused',
    10: "calc_seqB without Barlat parameters: the reference's AttributeError comes from the missing attribute Bar_m1",
}


@pytest.mark.parametrize('k', range(len(CASES)))
def test_point_function_argument_checks_match_reference(k, monkeypatch):
    from pylabfea_amd import Material
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    mat, method, kw = CALLS[k]
    assert CASES[k]['material'] == mat and CASES[k]['method'] == method
    got, want = run(Material, mat, method, kw), CASES[k]['outcome']
    assert got['exception'] is not None
    if k in OWN_MESSAGE:
        assert got['exception'] in (want['exception'], 'AttributeError', 'ValueError'), OWN_MESSAGE[k]
        assert 'device' not in got['message'].lower() and 'libplfx' not in got['message'].lower()
    else:
        assert got == want
