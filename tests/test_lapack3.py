"""The LAPACK replay behind the principal-stress materials (pylabfea_amd/csrc/plfx_lapack3.hpp), through the library's HOST
entry points plfx_sig_princ_host / plfx_eig3_host (no GPU): the order of np.linalg.eig's eigenpairs for symmetric 3 x 3 matrices
and the reference's axis-tracking re-ordering (basic.py:153-175), against (i) the reference's own sig_princ output on 400 general
stress states (tests/golden/princ_general.npz, written by the unmodified reference) and (ii) numpy itself -- the library the
reference calls -- on structured families of matrices: every zero pattern dgebal distinguishes, nearly diagonal matrices (where
dlahqr's deflation tests are decided by the last bits), equal diagonals, pure shear."""
import os

import numpy as np
import pytest

from pylabfea_amd import _lib


def mat(s):
    return np.array([[s[0], s[5], s[4]], [s[5], s[1], s[3]], [s[4], s[3], s[2]]])


def ref_sig_princ(s):
    """basic.py:153-172 restated on one Voigt stress (np.linalg.eig is the reference's own call)"""
    sp, ev = np.linalg.eig(mat(s))
    iev = np.argmax(np.abs(ev), axis=1)
    j = [i for c in range(3) for i in range(3) if iev[i] == c]
    return np.array([sp[j[0]], sp[j[1]], sp[j[2]]]), sp, ev


def test_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, 'princ_general.npz'))
    sp = _lib.sig_princ_host(z['sig'])
    assert np.max(np.abs(sp - z['princ'])) < 1e-12 * np.max(np.abs(z['princ']))
    g = np.load(os.path.join(golden_dir, 'basic.npz'))
    assert np.max(np.abs(_lib.sig_princ_host(g['sig']) - g['princ'])) < 1e-12 * np.max(np.abs(g['princ']))
    # the reference's documented examples
    assert np.allclose(_lib.sig_princ_host([[1., 5., 3., 0., 0., 0.]])[0], [1., 5., 3.])
    assert np.allclose(_lib.sig_princ_host([[0., 0., 0., 0., 0., 10.]])[0], [10., -10., 0.])


FAMILIES = {
    'full': lambda r, n: r.normal(size=(n, 6)) * 100,
    's23 only (row 1 isolated)': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 1, 0, 0]),
    's13 only (row 2 isolated)': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 0, 1, 0]),
    's12 only (plane state)': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 0, 0, 1]),
    's23 and s13': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 1, 1, 0]),
    's23 and s12': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 1, 0, 1]),
    's13 and s12': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 0, 1, 1]),
    'diagonal': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 0, 0, 0]),
    'pure shear': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([0, 0, 0, 1, 1, 1]),
    'nearly diagonal': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 1e-6, 1e-6, 1e-6]),
    'weakly coupled row': lambda r, n: r.normal(size=(n, 6)) * 100 * np.array([1, 1, 1, 1, 1e-9, 1e-9]),
}


@pytest.mark.parametrize('family', sorted(FAMILIES))
def test_against_numpy(family):
    rng = np.random.default_rng(sum(map(ord, family)))
    S = FAMILIES[family](rng, 4000)
    if family == 'full':
        S[:500, 1] = S[:500, 0]                                       # equal normal stresses
        S[500:1000, 0:3] = S[500:1000, 0:1] + 1e-3 * rng.normal(size=(500, 3))   # nearly hydrostatic + shear
    w, V = _lib.eig3_host(S)
    sp = _lib.sig_princ_host(S)
    n_order = n_princ = 0
    for i, s in enumerate(S):
        rsp, rw, rev = ref_sig_princ(s)
        assert not np.iscomplexobj(rw)
        sc = max(1e-300, np.max(np.abs(rw)))
        n_order += bool(np.max(np.abs(w[i] - rw)) > 1e-12 * sc)         # same eigenvalue in every position
        n_princ += bool(np.max(np.abs(sp[i] - rsp)) > 1e-12 * sc)
        if np.max(np.abs(w[i] - rw)) <= 1e-12 * sc:
            assert np.max(np.abs(np.abs(V[i]) - np.abs(rev))) < 1e-9    # same eigenvectors up to sign
    # every family agrees in every position on the machine the fixtures were written on (OpenBLAS 0.3.29 picking its Haswell
    # kernels); where the deflation tests are decided by the last bits ('nearly diagonal', 'weakly coupled row') another CPU
    # dispatch of the same library may move a handful -- and would move the reference with it
    slack = 4 if family in ('nearly diagonal', 'weakly coupled row') else 0
    assert n_order <= slack and n_princ <= slack, (family, n_order, n_princ)


def test_exact_ties_are_the_only_open_cases():
    """small-integer matrices have eigenvectors whose components are EQUAL in exact arithmetic ((1,1,1)/sqrt 3 ...): the
    row-argmax of basic.py:157 is then decided by the last bit of LAPACK's eigenvectors.  The replay follows the arithmetic of
    the library's kernels (see the header) and still leaves a few of those: bounded here, everything else must agree."""
    rng = np.random.default_rng(3)
    S = np.round(rng.normal(size=(20000, 6)) * 3)
    w, _ = _lib.eig3_host(S)
    sp = _lib.sig_princ_host(S)
    bad_w = bad_sp = 0
    for i, s in enumerate(S):
        rsp, rw, rev = ref_sig_princ(s)
        if np.iscomplexobj(rw):
            continue
        sc = max(1e-300, np.max(np.abs(rw)))
        bad_w += bool(np.max(np.abs(w[i] - rw)) > 1e-9 * sc)
        bad_sp += bool(np.max(np.abs(sp[i] - rsp)) > 1e-9 * sc)
    assert bad_w <= 20 and bad_sp <= 40, (bad_w, bad_sp)               # (measured here: 0 and 1 of 20 000)
