"""INTEGRATION.md section 4 is executable documentation: the fenced ctypes stub is extracted from the file and
  - on the CPU: executed up to its own ABI / struct-size assertions against the built library, and its struct
    declarations are held field by field against lgmi/_lib.py's (the binding the product uses);
  - on a GPU: run through the raw C ABI on the reference-generated edge cases (tests/golden/pairs_edge.json) and
    compared with the reference's rows.
A stub that drifts from include/lgmi.h (round 3: ABI 3 structs against an ABI 4 library) fails here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from lgmi import _lib


def stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## 4. The ctypes stub'):]
    sec = sec[:sec.index('\n## 5.')]
    blocks = re.findall(r'```python\n(.*?)```', sec, flags=re.S)
    assert len(blocks) == 1, 'section 4 must hold exactly one python block'
    return blocks[0]


def load_stub():
    ns = {'LGMI_SO': _lib.LIB_PATH}
    exec(compile(stub_source(), 'INTEGRATION.md#4', 'exec'), ns)      # runs its ABI-version and struct-size assertions
    return ns


def test_stub_declares_the_structs_of_the_shipped_library():
    ns = load_stub()
    for name in ('Batch', 'Params', 'Result'):
        mine, theirs = ns[name], getattr(_lib, name)
        assert C.sizeof(mine) == C.sizeof(theirs), name
        assert [(f[0], getattr(mine, f[0]).offset, getattr(mine, f[0]).size) for f in mine._fields_] == \
               [(f[0], getattr(theirs, f[0]).offset, getattr(theirs, f[0]).size) for f in theirs._fields_], name
    lib = _lib.load()
    assert lib.lgmi_abi_version() == 6 and 'lgmi_abi_version() == 6' in stub_source()
    assert [lib.lgmi_struct_size(k) for k in range(9)] == [96, 32, 144, 128, 48, 152, 8, 540, 0]


def test_struct_size_handshake_refuses_a_stale_declaration():
    """what the handshake is for: the ABI 3 Params of round 3's document (24 bytes) against this library"""
    class OldParams(C.Structure):
        _fields_ = [('min_common', C.c_uint32), ('n_shuffles', C.c_uint32), ('seed', C.c_uint64),
                    ('het_only', C.c_uint8), ('emit_counts', C.c_uint8), ('exact_2x2', C.c_uint8), ('reserved0', C.c_uint8),
                    ('shard_rank', C.c_uint16), ('shard_world', C.c_uint16)]
    assert _lib.load().lgmi_struct_size(1) != C.sizeof(OldParams)


def class_matrix(sites):
    """the packing rules at the end of INTEGRATION.md section 4, written out independently of lgmi/pack.py:
    -> (pos, type, cls[P, R]) with cls 0 not covered, 1 minor, 2 major, 3 other"""
    sites = sorted(sites, key=lambda s: s[0])
    reads = {}
    for _pos, _typ, _depth, nt in sites:
        for _allele, names in nt:
            for r in names:
                reads.setdefault(r, len(reads))
    cls = np.zeros((len(sites), max(len(reads), 1)), np.uint8)
    for s, (_pos, _typ, depth, nt) in enumerate(sites):
        order = sorted(range(len(depth)), key=lambda k: -depth[k][1])             # stable: ties keep dict order
        rank = {depth[k][0]: n for n, k in enumerate(order)}
        for allele, names in nt:                                                   # a later allele overwrites: last one wins
            c = {0: 2, 1: 1}.get(rank.get(allele), 3)          # absent from depth: 'other' too (defaultdict(int), :33-38)
            for r in names:
                cls[s, reads[r]] = c
    typ = np.array([{'mismatch': 0, 'snp': 1, 'het_snp': 2}[s[1]] for s in sites], np.uint8)
    return np.array([s[0] for s in sites], np.int64), typ, cls


@pytest.mark.gpu
def test_stub_runs_the_reference_edge_cases_through_the_raw_c_abi():
    ns = load_stub()
    cases = [c for c in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'pairs_edge.json')))['cases']
             if len(c['sites']) >= 2 and all(len(s[2]) >= 2 for s in c['sites'])]
    assert len(cases) >= 8
    names = {0: 'mismatch', 1: 'snp', 2: 'het_snp'}
    for mc in sorted({c['min_common'] for c in cases}):
        group = [c for c in cases if c['min_common'] == mc]
        blocks = [class_matrix(c['sites']) for c in group]
        blk, pi, pj, mi = ns['run_blocks'](blocks, min_common=mc, het_only=0)     # several blocks in one call
        for b, c in enumerate(group):
            m = blk == b
            typ = dict(zip(blocks[b][0].tolist(), blocks[b][1].tolist()))
            got = [[int(a), names[typ[int(a)]], int(d), names[typ[int(d)]]] for a, d in zip(pi[m], pj[m])]
            assert got == [r[:4] for r in c['rows']], c['name']
            assert np.allclose(mi[m], [r[4] for r in c['rows']], rtol=0, atol=1e-6), c['name']
