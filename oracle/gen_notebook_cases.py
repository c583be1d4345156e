"""Fixture generator: runs the scenarios of tests/notebook_cases.py with the UNMODIFIED reference package imported from
/root/reference/src and stores every snapshot in tests/golden/notebook_cases.npz (keys ``<case>__<tag>__<field>``).
Test infrastructure; needs /root/reference (build container only).

    MPLBACKEND=Agg PYTHONPATH=oracle/_refshim:/root/reference/src python oracle/gen_notebook_cases.py
"""
import io
import os
import sys
import time
import warnings
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('MPLBACKEND', 'Agg')
import pylabfea as FE  # noqa: E402  (the reference)
import notebook_cases as NC  # noqa: E402


def main():
    rec = {}
    for case in NC.CASES:
        t0 = time.time()
        tags = []

        def snap(fe, tag, case=case):
            tags.append(tag)
            for k, v in NC.snapshot(fe).items():
                rec['%s__%s__%s' % (case.__name__, tag, k)] = v
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            with redirect_stdout(io.StringIO()):
                case(FE, snap)
        rec[case.__name__ + '__tags'] = np.array(tags)
        last = '%s__%s__' % (case.__name__, tags[-1])
        print('%-44s %5.1f s  snapshots %-28s nsteps %s  sgl[-1] %s' % (case.__name__, time.time() - t0, ','.join(tags),
              rec[last + 'nsteps'], np.round(rec[last + 'sgl'][-1][:2], 4)), flush=True)
    out = os.path.join(ROOT, 'tests', 'golden', 'notebook_cases.npz')
    np.savez_compressed(out, **rec)
    print('wrote', out, '%.1f KB' % (os.path.getsize(out) / 1e3))


if __name__ == '__main__':
    main()
