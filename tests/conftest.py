"""pytest configuration: markers, import paths, shared fixtures."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'l-giremi_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    _native_stack_on_abort()


def _native_stack_on_abort():
    """a crash inside the native libraries (an abort of the HIP runtime, a glibc heap check, a GPU fault surfacing as
    SIGABRT) otherwise leaves only Python's "Fatal Python error: Aborted": tools/src/abort_bt.c prints the native stack
    first.  Best effort — no compiler, no helper."""
    import ctypes
    import subprocess
    import tempfile
    src = os.path.join(ROOT, 'tools', 'src', 'abort_bt.c')
    out = os.path.join(tempfile.gettempdir(), 'lgmi_abort_bt_%d.so' % os.getuid())
    try:
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.run(['gcc', '-shared', '-fPIC', '-o', out, src], check=True, capture_output=True, timeout=60)
        ctypes.CDLL(out).lgmi_abort_bt_install()
    except Exception:                                  # noqa: BLE001
        pass


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def sites_to_mismatches(sites):
    """golden 'sites' list -> the reference's mismatches[strand] dict (insertion order kept)."""
    mm = {}
    for pos, type_, depth, nt in sites:
        mm[pos] = {'ref': nt[0][0] if nt else '', 'type': type_,
                   'depth': {a: d for a, d in depth},
                   'nt': {a: list(names) for a, names in nt},
                   'neighbor': {}, 'up': 'A', 'down': 'C'}
    return mm


def all_pair_cases():
    cases = []
    for fn in ('pairs_edge.json', 'pairs_random.json', 'pairs_banded_cfg1.json'):
        cases.extend(load_golden(fn)['cases'])
    return cases


@pytest.fixture(scope='session')
def pair_cases():
    return all_pair_cases()
