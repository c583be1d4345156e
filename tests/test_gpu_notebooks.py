"""The reference's tutorial scenarios (tests/notebook_cases.py) through the drop-in Model / Material API on the GPU against
the snapshots the unmodified reference produced for them (tests/golden/notebook_cases.npz, oracle/gen_notebook_cases.py):
force-controlled loads, shear loads with bcdir, chained solve() calls, u = None resets, re-meshing, bcnode."""
import os
import warnings

import numpy as np
import pytest

import notebook_cases as NC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', NC.CASES, ids=[c.__name__ for c in NC.CASES])
def test_notebook_case_vs_reference(golden_dir, case):
    import pylabfea_amd as FE
    g = np.load(os.path.join(golden_dir, 'notebook_cases.npz'))
    name = case.__name__
    seen = []

    def snap(fe, tag):
        seen.append(tag)
        got = NC.snapshot(fe)
        ref = {k: g['%s__%s__%s' % (name, tag, k)] for k in got}
        where = '%s[%s]' % (name, tag)
        # identical control flow: load steps and K-iterations per step
        assert int(got['nsteps']) == int(ref['nsteps']), where
        assert np.array_equal(got['niter'], ref['niter']), (where, got['niter'], ref['niter'])
        s = max(np.max(np.abs(ref['sig'])), 1e-30)
        e = max(np.max(np.abs(ref['eps'])), 1e-30)
        for k, scale in (('u', np.max(np.abs(ref['u']))), ('f', max(np.max(np.abs(ref['f'])), 1e-30)),
                         ('sig', s), ('sgl', s), ('eps', e), ('egl', e), ('epl', e), ('epgl', e)):
            assert got[k].shape == ref[k].shape, (where, k, got[k].shape, ref[k].shape)
            err = np.max(np.abs(got[k] - ref[k])) if ref[k].size else 0.
            assert err < 2e-6 * scale, (where, k, err, scale)
        for row, scale in zip(range(3), (s, e, e)):
            assert np.max(np.abs(got['glob'][row] - ref['glob'][row])) < 2e-6 * scale, (where, 'glob', row)
        assert np.max(np.abs(got['globbc'][:2] - ref['globbc'][:2])) < 2e-6 * e, where
        assert np.max(np.abs(got['globbc'][2:] - ref['globbc'][2:])) < 2e-6 * s, where

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        case(FE, snap)
    assert seen == list(g[name + '__tags'])
