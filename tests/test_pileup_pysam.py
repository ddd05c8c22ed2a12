"""lgmi.io's pile-up against pysam's own answer for the same BAM (tests/golden/pileup_pysam.json, made by
tools/make_pysam_pileup_golden.py where pysam is installed).  The build container has neither pysam nor htslib, so the
fixture may be absent: the test is then skipped, and the pile-up semantics stay "pysam's defaults as documented"
(include/lgmi_io.h) — VERDICT r3: the only way they ever get pinned."""
import json
import os
import sys

import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, 'tests', 'golden', 'pileup_pysam.json')


@pytest.mark.skipif(not os.path.exists(GOLD), reason='tests/golden/pileup_pysam.json not generated (needs pysam: tools/make_pysam_pileup_golden.py)')
def test_pileup_matches_pysam(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
    from lgmi.io import BamReader
    from pileup_bam import write_bam
    gold = json.load(open(GOLD))
    write_bam(tmp_path / 'p.bam')
    rd = BamReader(str(tmp_path / 'p.bam'))
    for reg in gold['regions']:
        got = [[c.pos, c.get_query_names(), c.get_query_sequences()] for c in rd.pileup(reg['contig'], reg['start'], reg['stop'])]
        want = reg['columns']
        assert [c[0] for c in got] == [c[0] for c in want], 'columns differ in %s:%d-%d' % (reg['contig'], reg['start'], reg['stop'])
        for g, w in zip(got, want):
            # (pysam upper / lower-cases bases by strand in get_query_sequences(); the reference compares case-insensitively
            #  only through its own cs walk, mismatch.py:170-188 uses the names of the column and the base as given)
            assert sorted(zip(g[1], [b.upper() for b in g[2]])) == sorted(zip(w[1], [b.upper() for b in w[2]])), g[0]


def test_the_fixture_bam_is_deterministic(tmp_path):
    """the generator and this test must be looking at the same bytes"""
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
    from pileup_bam import write_bam
    a = write_bam(tmp_path / 'a.bam')
    b = write_bam(tmp_path / 'b.bam')
    assert a == b
    ha, hb = (hashlib.sha256(open(tmp_path / n, 'rb').read()).hexdigest() for n in ('a.bam', 'b.bam'))
    assert ha == hb
    # and lgmi's own pile-up of it is not empty in any fixture region
    from lgmi.io import BamReader
    from pileup_bam import REGIONS
    rd = BamReader(str(tmp_path / 'a.bam'))
    for contig, s, e in REGIONS:
        assert sum(1 for _ in rd.pileup(contig, s, e)) > 50
