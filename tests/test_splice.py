"""N4 — site x splice-site MI (lgmi.splice) against the reference utility run on the same tables
(tests/golden/splice.json, produced by running calculate_site_splice_mi.py with runpy)."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden

CASES = load_golden('splice.json')['cases']


def tables(case):
    site = pd.DataFrame(case['site'], columns=['read_name', 'chromosome', 'pos', 'seq'])
    splice = pd.DataFrame(case['splice'], columns=['read_name', 'chromosome', 'pos', 'type', 'corrected_pos', 'annotation'])
    exp = pd.DataFrame(case['out']['data'], columns=case['out']['columns'])
    return site, splice, exp


@pytest.mark.parametrize('case', CASES, ids=lambda c: c['name'])
def test_pair_discovery_matches_reference(case):
    from lgmi.splice import site_splice_pairs
    site, splice, exp = tables(case)
    got = site_splice_pairs(site, splice)
    assert got['chromosome'].tolist() == exp['chromosome'].tolist()
    assert got['site_pos'].astype(int).tolist() == exp['site_pos'].tolist()
    assert got['seq'].tolist() == exp['seq'].tolist()
    assert got['splice_pos'].astype(int).tolist() == exp['splice_pos'].tolist()
    assert got['count'].tolist() == exp['count'].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=lambda c: c['name'])
def test_site_splice_mi_matches_reference(case):
    import lgmi
    from lgmi.splice import site_splice_mi
    site, splice, exp = tables(case)
    got = site_splice_mi(site, splice, engine=lgmi.default_engine())
    assert len(got) == len(exp)
    assert np.max(np.abs(got['mi'].values - exp['mi'].values)) <= 1e-6
    assert ((got['mi'].values == 0.0) == (exp['mi'].values == 0.0)).all()
