"""AddressSanitizer / UndefinedBehaviorSanitizer runs: the CPU oracle's C file and the HOST
side of libplfx (built without device code) are compiled with -fsanitize=address,undefined and driven by the CPU tests that
already exist for them, in a child process that preloads the sanitizer runtime.  A report of either sanitizer aborts the
child (non-zero exit)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, preload, args):
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1',
               LD_PRELOAD=preload, **env_extra)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider'] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    return r


def _tool(*names):
    """first of the candidates that exists (absolute path or on PATH), else skip: a box without the toolchain skips, it does not fail"""
    for n in names:
        p = n if os.path.isabs(n) and os.path.exists(n) else shutil.which(n)
        if p:
            return p
    pytest.skip('toolchain not found: ' + ' / '.join(names))


def _make(path, target):
    try:
        subprocess.check_call(['make', '-s', '-C', path, target])
    except (subprocess.CalledProcessError, OSError) as exc:
        pytest.skip('sanitizer build not possible here: %s' % exc)


def test_oracle_under_asan_ubsan():
    gcc = _tool('gcc')
    _tool('make')
    _make(os.path.join(ROOT, 'oracle'), 'asan')
    lib = os.path.join(ROOT, 'oracle', 'libplfx_oracle_asan.so')
    asan = subprocess.check_output([gcc, '-print-file-name=libasan.so'], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('gcc has no libasan.so here')
    r = _run({'PLFO_LIB': lib}, asan, ['tests/test_oracle_golden.py'])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert ' passed' in r.stdout and 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr


def test_libplfx_host_code_under_asan_ubsan():
    """needs ROCm (hipcc + its clang): skipped on a box without it"""
    _tool('/opt/rocm/bin/hipcc', 'hipcc')
    clang = _tool('/opt/rocm/lib/llvm/bin/clang', 'amdclang', 'clang')
    _tool('make')
    _make(os.path.join(ROOT, 'pylabfea_amd', 'csrc'), 'asan')
    lib = os.path.join(ROOT, 'pylabfea_amd', 'libplfx_host_asan.so')
    rt = subprocess.check_output([clang, '--print-file-name=libclang_rt.asan-x86_64.so'], text=True).strip()
    if not os.path.exists(rt):
        pytest.skip('no clang AddressSanitizer runtime in this image')
    # the host-only entry points and everything the binding does without a device: symbol table, structured-grid generator
    # against the reference's fixtures, closed-form block-ELL pattern against the generic derivation, argument checks
    r = _run({'PLFX_LIB': lib}, rt, ['tests/test_abi.py', 'tests/test_mesh.py', '-m', 'not gpu'])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert ' passed' in r.stdout and 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr
