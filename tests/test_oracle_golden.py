"""The Python oracle (oracle/mi_oracle.py) against the reference's own outputs
(tests/golden/*.json, produced by tests/golden/gen_golden.py from /root/reference)."""
import math

import pytest

from conftest import all_pair_cases, load_golden, sites_to_mismatches
from oracle import mi_oracle

MI_TOL = 1e-6  # BASELINE.json north_star tolerance; observed ~1e-15


@pytest.mark.parametrize('case', all_pair_cases(), ids=lambda c: c['name'])
def test_pair_rows_match_reference(case):
    mm = sites_to_mismatches(case['sites'])
    rows, tables = mi_oracle.pair_rows(mm, case['min_common'], with_counts=True)
    assert [r[:4] for r in rows] == [r[:4] for r in case['rows']]
    assert [sum(t, []) for t in tables] == case['tables']          # counts: bit-exact
    for got, exp in zip(rows, case['rows']):
        assert abs(got[4] - exp[4]) <= MI_TOL
        assert abs(got[4] - exp[4]) <= 1e-12                        # and in practice far tighter
        if exp[4] == 0.0:
            assert got[4] == 0.0                                    # zero-entropy shortcut is exact


@pytest.mark.parametrize('case', all_pair_cases(), ids=lambda c: c['name'])
def test_mean_rows_match_reference(case):
    for key, rows in (('mean_all', case['rows']),
                      ('mean_het', [r for r in case['rows'] if r[1] == 'het_snp' or r[3] == 'het_snp'])):
        got = mi_oracle.mean_rows(rows) if rows else []
        assert [g[0] for g in got] == [e[0] for e in case[key]]
        for g, e in zip(got, case[key]):
            assert abs(g[1] - e[1]) <= 1e-12


def test_ecdf_matches_reference():
    for c in load_golden('ecdf.json')['cases']:
        f = mi_oracle.ecdf_strict(c['sample'])
        for q, v in zip(c['query'], c['value']):
            assert math.isclose(f(q), v, abs_tol=1e-12)


def test_too_few_alleles_raises_indexerror():
    mm = {1: {'type': 'mismatch', 'depth': {'A': 9}, 'nt': {'A': ['a', 'b']}},
          2: {'type': 'mismatch', 'depth': {'A': 9, 'C': 2}, 'nt': {'A': ['a'], 'C': ['b']}}}
    with pytest.raises(IndexError):
        mi_oracle.pair_rows(mm, 1)
