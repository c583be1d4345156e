"""Test-infrastructure shim (NOT product code).

The reference package calls importlib.metadata.version('pylabfea') at import
time (/root/reference/src/pylabfea/__init__.py:15,19) and is not pip-installed
in the build container.  Putting this directory on PYTHONPATH lets
oracle/gen_golden.py import the reference straight from /root/reference/src.
Only used in the build container; never on the GPU box.
"""
import importlib.metadata as _md

_orig_version = _md.version


def _version(name):
    if name == 'pylabfea':
        return '4.4.2'
    return _orig_version(name)


_md.version = _version
