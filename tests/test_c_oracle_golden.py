"""The C oracle (oracle/lgmi_oracle.c) and the host packer (lgmi/pack.py) against the
reference's own outputs (tests/golden/*.json).  CPU only."""
import numpy as np
import pytest

from conftest import all_pair_cases, sites_to_mismatches
from lgmi.pack import pack_blocks
from oracle import c_oracle

MI_TOL = 1e-6  # north_star tolerance; counts are bit-exact


def rows_from_oracle(pb, out):
    pos, names = pb.site_pos.tolist(), pb.type_names
    return [[pos[i], names[i], pos[j], names[j], mi]
            for i, j, mi in zip(out['row_i'].tolist(), out['row_j'].tolist(), out['row_mi'].tolist())]


@pytest.mark.parametrize('case', all_pair_cases(), ids=lambda c: c['name'])
def test_all_pairs_match_reference(case):
    mm = sites_to_mismatches(case['sites'])
    pb = pack_blocks([mm])
    out = c_oracle.run(pb, min_common=case['min_common'], het_only=False)
    rows = rows_from_oracle(pb, out)
    assert [r[:4] for r in rows] == [r[:4] for r in case['rows']]
    assert out['row_counts'].reshape(-1, 9).tolist() == case['tables']
    if rows:
        assert np.max(np.abs(np.array([r[4] for r in rows]) - np.array([r[4] for r in case['rows']]))) <= 1e-12
    for got, exp in zip(rows, case['rows']):
        if exp[4] == 0.0:
            assert got[4] == 0.0
    # per-site mean over all rows (mutual_information.py:48-60)
    exp_mean = dict((p, m) for p, m in case['mean_all'])
    got_mean = {p: m for p, m, c in zip(pb.site_pos.tolist(), out['site_mean_mi'].tolist(),
                                        out['site_n_pairs'].tolist()) if c}
    assert set(got_mean) == set(exp_mean)
    for p in exp_mean:
        assert abs(got_mean[p] - exp_mean[p]) <= 1e-12


@pytest.mark.parametrize('case', all_pair_cases(), ids=lambda c: c['name'])
def test_het_only_matches_reference_filter(case):
    """mismatch.py:392-400: keep rows with a het_snp side, mean over the kept rows"""
    mm = sites_to_mismatches(case['sites'])
    pb = pack_blocks([mm])
    out = c_oracle.run(pb, min_common=case['min_common'], het_only=True)
    exp = [r for r in case['rows'] if r[1] == 'het_snp' or r[3] == 'het_snp']
    rows = rows_from_oracle(pb, out)
    assert [r[:4] for r in rows] == [r[:4] for r in exp]
    for got, e in zip(rows, exp):
        assert abs(got[4] - e[4]) <= 1e-12
    exp_mean = dict((p, m) for p, m in case['mean_het'])
    got_mean = {p: m for p, m, c in zip(pb.site_pos.tolist(), out['site_mean_mi'].tolist(),
                                        out['site_n_pairs'].tolist()) if c}
    assert set(got_mean) == set(exp_mean)
    for p in exp_mean:
        assert abs(got_mean[p] - exp_mean[p]) <= 1e-12


def test_two_blocks_do_not_mix():
    cases = [c for c in all_pair_cases() if c['name'] in ('two_allele_linkage', 'three_allele')]
    mms = [sites_to_mismatches(c['sites']) for c in cases]
    pb = pack_blocks(mms)
    out = c_oracle.run(pb, min_common=5, het_only=False)
    n0 = len(cases[0]['rows'])
    assert len(out['row_i']) == n0 + len(cases[1]['rows'])
    b1 = int(pb.block_site_begin[1])
    assert (out['row_j'][:n0] < b1).all() and (out['row_i'][n0:] >= b1).all()


def test_zero_common_with_min_common_zero_raises():
    mm = sites_to_mismatches([c for c in all_pair_cases() if c['name'] == 'below_min_common'][0]['sites'])
    mm[99999] = {'type': 'snp', 'depth': {'A': 5, 'G': 3}, 'nt': {'A': ['x1'], 'G': ['x2']}}
    with pytest.raises(ValueError):
        c_oracle.run(pack_blocks([mm]), min_common=0, het_only=False)
