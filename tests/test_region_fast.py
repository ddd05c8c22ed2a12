"""The native site extraction (lgio_bam_region_sites through lgmi.region.region_sites_native) against its
specification, the Python routine get_region_mismatches_with_filters: same surviving sites (same dict order, same
allele order, same read lists, same window counts), same removed table (same order, same reasons) — on simulated
genes at several error rates and filter settings, on hand-made records for the quirks, and refused (None, so the
caller runs the Python routine) for the inputs the native walk does not cover."""
import os

import numpy as np
import pytest

from fakes import simulate_region
from test_cli import write_inputs


def _plain(sites):
    return {k: dict(v, depth=dict(v['depth']), nt=dict(v['nt']), neighbor=dict(v['neighbor'])) for k, v in sites.items()}


def compare(sam, genome, chrom, start, end, **kw):
    from lgmi.region import get_region_mismatches_with_filters, region_sites_native, _compact_gone
    want_sites, want_gone = get_region_mismatches_with_filters(chromosome=chrom, start_pos=start, end_pos=end, sam=sam,
                                                               genome=genome, **kw)
    got = region_sites_native(chromosome=chrom, start_pos=start, end_pos=end, sam=sam, genome=genome, **kw)
    assert got is not None
    got_sites, (got_gone, got_reasons) = got
    want_compact, want_reasons = _compact_gone(want_gone)
    n_window = 0
    for strand in '+-':
        a, b = _plain(want_sites[strand]), _plain(got_sites[strand])
        assert list(a) == list(b)                                   # dict order = row order of the site table
        for pos in a:
            assert list(a[pos]['nt']) == list(b[pos]['nt']) and list(a[pos]['depth']) == list(b[pos]['depth'])
            assert a[pos] == b[pos], (strand, pos)
        wp, wc = want_compact[strand]
        gp, gc = got_gone[strand]
        assert wp.tolist() == gp.tolist()
        w = [want_reasons[c] for c in wc]
        assert w == [got_reasons[c] for c in gc]
        n_window += w.count('too many window mismatches')
    return want_sites, n_window


def _open(tmp_path, regions):
    from lgmi.io import open_alignment, open_fasta
    bam, fa, _vcf = write_inputs(tmp_path, regions)
    return open_alignment(bam), open_fasta(fa)


SETTINGS = [
    dict(),
    dict(min_total_depth=0, min_allele_depth=1, min_allele_ratio=0.0),
    dict(keep_non_spliced_read=True, min_dist_from_splice=0),
    dict(mismatch_window_size=30, max_window_mismatch=2, max_window_mismatch_type=1),
    dict(mismatch_window_size=7, max_window_mismatch=1, max_window_mismatch_type=0, min_total_depth=0),
    dict(min_dist_from_splice=25, min_allele_ratio=0.3, min_total_depth=12.5, min_allele_depth=2.0),
]


@pytest.mark.parametrize('err', [0.004, 0.03, 0.12])
def test_native_extraction_equals_the_python_routine(tmp_path, err):
    regions = {}
    for k, contig in enumerate(['chrA', 'chrB']):
        reads, genome, snps, _ = simulate_region(seed=900 + k + int(err * 1000), n_reads=90 + 40 * k, err=err)
        regions[contig] = (reads, genome, snps)
    sam, genome = _open(tmp_path, regions)
    windowed = 0
    for contig, (_r, g, snps) in regions.items():
        for kw in SETTINGS:
            for (a, b) in [(0, len(g)), (150, 500), (400, 401)]:
                _s, nw = compare(sam, genome, contig, a, b, snp_positions=snps, simple_repeat_intervals=[[250, 280]], **kw)
                windowed += nw
    if err >= 0.03:
        assert windowed > 0             # the window filter (and the sites it brings back empty) was exercised


def _write_records(tmp_path, records, length=400):
    """records: (start, name, reverse, cigar, seq, cs, flag, quality)"""
    from lgmi.io import BamWriter, open_alignment, open_fasta
    rng = np.random.default_rng(5)
    genome = ''.join(rng.choice(list('ACGT'), length))
    bam, fa = str(tmp_path / 'q.bam'), str(tmp_path / 'q.fa')
    w = BamWriter(bam, [('c', length)])
    for start, name, rev, cigar, seq, cs, flag, qual in sorted(records, key=lambda r: r[0]):
        w.write('c', start, name, rev, cigar, seq, cs, flag=flag, quality=qual)
    w.close()
    with open(fa, 'w') as f:
        f.write('>c\n%s\n' % genome)
    return open_alignment(bam), open_fasta(fa)


def _spliced(start, name, rev, sub_at, alt='g', flag=0, qual=40, ref='a'):
    """60M 100N 60M with one substitution `sub_at` bases into the first exon"""
    seq = ['A'] * 120
    seq[sub_at] = alt.upper()
    cs = ':%d*%s%s:%d~gt100ag:60' % (sub_at, ref, alt, 59 - sub_at)
    return (start, name, rev, [(0, 60), (3, 100), (0, 60)], ''.join(seq), cs, flag, qual)


def test_native_extraction_quirks(tmp_path):
    recs = []
    for k in range(12):                                             # a site at 30 with reference and alternative reads
        recs.append(_spliced(0, 'r%02d' % k, k % 2 == 1, 30, alt='g' if k < 8 else 'c'))
    for k in range(6):                                              # reference-allele reads: no substitution at 30
        recs.append((0, 'm%02d' % k, k % 2 == 1, [(0, 60), (3, 100), (0, 60)], 'A' * 120, ':60~gt100ag:60', 0, 40))
    # the same NAME on both strands: the first record decides the strand of both (read_strand_dict)
    recs.append(_spliced(0, 'dup', False, 30))
    recs.append(_spliced(5, 'dup', True, 25))
    # secondary / duplicate records are walked for substitutions but are not in the pile-up
    recs.append(_spliced(2, 'sec', False, 28, flag=256))
    recs.append(_spliced(2, 'pcr', True, 28, flag=1024))
    # a low-quality read: its bases drop out of the pile-up (min_base_quality 13), its substitutions do not
    recs.append(_spliced(1, 'lowq', False, 29, qual=5))
    # a deletion and an insertion: 20M 3D 37M 100N 60M / 20M 2I 40M 100N 60M
    recs.append((0, 'del', False, [(0, 20), (2, 3), (0, 37), (3, 100), (0, 60)], 'A' * 117, ':20-acg:7*ag:29~gt100ag:60', 0, 40))
    recs.append((0, 'ins', True, [(0, 20), (1, 2), (0, 40), (3, 100), (0, 60)], 'A' * 122, ':20+tt:10*ag:29~gt100ag:60', 0, 40))
    # an unspliced read (skipped unless keep_non_spliced_read) and one whose substitution sits next to the junction
    recs.append((10, 'flat', False, [(0, 80)], 'A' * 80, ':20*ag:59', 0, 40))
    recs.append(_spliced(0, 'edge', False, 58))
    sam, genome = _write_records(tmp_path, recs)
    for kw in SETTINGS:
        sites, _ = compare(sam, genome, 'c', 0, 400, **kw)
    sites, _ = compare(sam, genome, 'c', 0, 400, min_total_depth=2, min_allele_depth=1)
    assert 30 in sites['+'] and 30 in sites['-']
    assert 'dup' in sites['+'][30]['nt']['G'] and 'dup' not in sites['-'][30]['nt'].get('G', [])


def test_packing_from_read_ids_equals_packing_from_names(tmp_path):
    """the run's pipeline packs a footprint from integer read ids (lgio_sites.read_uid: the first record of the read's
    NAME) instead of name strings: the packed block — read numbering by first appearance, last-allele-wins for a read
    listed twice, bit planes — must be the one the names give, also when a name occurs on two records"""
    from lgmi.pack import pack_blocks
    from lgmi.region import region_sites_native
    recs = []
    for k in range(14):
        recs.append(_spliced(0, 'r%02d' % k, False, 30, alt='g' if k < 9 else 'c'))
    for k in range(6):
        recs.append((0, 'm%02d' % k, False, [(0, 60), (3, 100), (0, 60)], 'A' * 120, ':60~gt100ag:60', 0, 40))
    for k in range(8):                                              # a second site at 45, overlapping read sets
        recs.append(_spliced(2, 's%02d' % k, False, 43, alt='t'))
    # one NAME on two records (a supplementary alignment): one read to the reference — listed under both alleles of site 30
    recs.append(_spliced(0, 'twice', False, 30, alt='g'))
    recs.append((0, 'twice', False, [(0, 60), (3, 100), (0, 60)], 'A' * 120, ':60~gt100ag:60', 2048, 40))
    sam, genome = _write_records(tmp_path, recs)
    kw = dict(chromosome='c', start_pos=0, end_pos=400, sam=sam, genome=genome, min_total_depth=2, min_allele_depth=1, min_allele_ratio=0.0)
    by_name, _ = region_sites_native(**kw)
    by_id, _ = region_sites_native(read_ids=True, **kw)
    assert list(by_name['+']) == list(by_id['+']) and len(by_name['+']) >= 2
    assert any('twice' in names for site in by_name['+'].values() for names in site['nt'].values())
    a, b = pack_blocks([by_name['+'], by_name['-']]), pack_blocks([by_id['+'], by_id['-']])
    for f in ('block_site_begin', 'block_n_reads', 'site_pos', 'site_type', 'site_word_off', 'site_n_words', 'site_plane_off', 'planes', 'site_tri'):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)


@pytest.mark.parametrize('cs', [None, ':30*an:29~gt100ag:60', ':30*na:29~gt100ag:60', '=AAAA*ag=AAAA', ':30*ag:29~gt100:60', ':60~100:60',
                                ':3x', '*agt:4'])
def test_native_extraction_leaves_the_rest_to_python(tmp_path, cs):
    from lgmi.io import BamWriter, open_alignment, open_fasta
    from lgmi.region import region_sites_native
    bam, fa = str(tmp_path / 'q.bam'), str(tmp_path / 'q.fa')
    w = BamWriter(bam, [('c', 400)])
    w.write('c', 0, 'ok', False, [(0, 60), (3, 100), (0, 60)], 'A' * 120, ':30*ag:29~gt100ag:60')
    if cs is None:
        w.write('c', 1, 'bare', False, [(0, 60)], 'A' * 60, ':60', cs_tag=False)
    else:
        w.write('c', 1, 'odd', False, [(0, 60), (3, 100), (0, 60)], 'A' * 120, cs)
    w.close()
    with open(fa, 'w') as f:
        f.write('>c\n%s\n' % ('ACGT' * 100))
    sam, genome = open_alignment(bam), open_fasta(fa)
    assert region_sites_native(chromosome='c', start_pos=0, end_pos=400, sam=sam, genome=genome) is None
    # ... and so do the callers' cases the native entry point has no argument for
    w = BamWriter(bam, [('c', 400)])
    w.write('c', 0, 'ok', False, [(0, 60), (3, 100), (0, 60)], 'A' * 120, ':30*ag:29~gt100ag:60')
    w.close()
    sam = open_alignment(bam)
    assert region_sites_native(chromosome='c', start_pos=0, end_pos=400, sam=sam, genome=genome) is not None
    assert region_sites_native(chromosome='c', start_pos=0, end_pos=400, sam=sam, genome=genome, read_strand_dict={}) is None
    assert region_sites_native(chromosome='c', start_pos=0, end_pos=400, sam=sam, genome=genome, min_dist_from_splice=2.5) is None
    assert region_sites_native(chromosome='nope', start_pos=0, end_pos=400, sam=sam, genome=genome) is None


def test_whole_run_extraction_uses_the_native_walk(tmp_path, monkeypatch):
    """regions_mismatch_analysis(concat=True) — the CLI's path — goes through the native walk; LGMI_PY_SITES=1 forces the
    Python routine; both give the same two site tables"""
    import lgmi.region as region
    regions = {}
    for k, contig in enumerate(['chrA', 'chrB']):
        reads, genome, snps, _ = simulate_region(seed=77 + k, n_reads=80)
        regions[contig] = (reads, genome, snps)
    sam, genome = _open(tmp_path, regions)
    jobs = [{'chromosome': c, 'start': 0, 'end': len(g), 'snp_positions': s, 'simple_repeat_intervals': [],
             'read_strand_dict': None} for c, (_r, g, s) in regions.items()]
    calls = {'native': 0, 'python': 0}
    native, python = region.region_sites_native, region.get_region_mismatches_with_filters
    monkeypatch.setattr(region, 'region_sites_native', lambda **kw: (calls.__setitem__('native', calls['native'] + 1), native(**kw))[1])
    monkeypatch.setattr(region, 'get_region_mismatches_with_filters',
                        lambda **kw: (calls.__setitem__('python', calls['python'] + 1), python(**kw))[1])
    fast = region._extract_chunk((None, jobs, {}, True), sam, genome)
    assert calls == {'native': 2, 'python': 0}
    monkeypatch.setenv('LGMI_PY_SITES', '1')
    slow = region._extract_chunk((None, jobs, {}, True), sam, genome)
    assert calls == {'native': 2, 'python': 2}
    for (c1, s1, (g1, r1)), (c2, s2, (g2, r2)) in zip(fast, slow):
        assert c1 == c2
        for strand in '+-':
            assert _plain(s1[strand]) == _plain(s2[strand]) and list(s1[strand]) == list(s2[strand])
            assert g1[strand][0].tolist() == g2[strand][0].tolist()
            assert [r1[c] for c in g1[strand][1]] == [r2[c] for c in g2[strand][1]]


def test_native_entry_point_rejects_bad_arguments(tmp_path):
    from lgmi.io import BamWriter, SiteParams, open_alignment
    bam = str(tmp_path / 'q.bam')
    w = BamWriter(bam, [('c', 400)])
    w.write('c', 0, 'ok', False, [(0, 60), (3, 100), (0, 60)], 'A' * 120, ':30*ag:29~gt100ag:60')
    w.close()
    sam = open_alignment(bam)
    params = SiteParams(min_base_quality=13, max_depth=8000, min_dist_from_splice=4, half_window=50, min_allele_depth=3,
                        min_allele_ratio=0.1, min_total_depth=6, max_window_mismatch=10, max_window_mismatch_type=3)
    with pytest.raises(ValueError):
        sam.region_sites('c', 10, 5, params)            # end before start
    with pytest.raises(ValueError):
        sam.region_sites('c', -1, 5, params)
    got = sam.region_sites('c', 0, 0, params)            # an empty interval: no reads, no sites
    assert got is not None and len(got['pos']) == 0 and all(len(p) == 0 for p, _c in got['removed'])
    got = sam.region_sites('c', 300, 400, params)        # nothing aligned there
    assert got is not None and len(got['pos']) == 0


def test_native_extraction_on_random_records(tmp_path):
    """random CIGARs, random cs strings (well-formed but unrelated to the CIGAR, so substitutions land anywhere — also
    outside the positions the read covers), random flags, qualities and duplicate names: the native walk either declines
    (None) or returns exactly what the Python routine returns"""
    from lgmi.io import BamWriter, open_alignment, open_fasta
    from lgmi.region import region_sites_native
    rng = np.random.default_rng(77)
    length = 3000
    fa = str(tmp_path / 'g.fa')
    with open(fa, 'w') as f:
        f.write('>c\n%s\n' % ''.join(rng.choice(list('ACGT'), length)))
    genome = open_fasta(fa)
    declined = compared = 0
    for trial in range(12):
        bam = str(tmp_path / ('r%d.bam' % trial))
        w = BamWriter(bam, [('c', length)])
        recs = []
        for k in range(int(rng.integers(5, 120))):
            start = int(rng.integers(0, 1500))
            cigar, qlen = [], 0
            for _ in range(int(rng.integers(1, 7))):
                op = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
                n = int(rng.integers(1, 400 if op == 3 else 60))
                if cigar and cigar[-1][0] == op:
                    continue
                cigar.append((op, n))
                qlen += n if op in (0, 1, 4) else 0
            if not any(op == 0 for op, _n in cigar):
                cigar.append((0, 20)); qlen += 20
            cs = []
            for _ in range(int(rng.integers(1, 12))):
                kind = int(rng.integers(0, 6))
                if kind <= 1:
                    cs.append(':%d' % int(rng.integers(1, 80)))
                elif kind == 2:
                    a, b = rng.choice(list('acgt'), 2, replace=False)
                    cs.append('*%s%s' % (a, b))
                elif kind == 3:
                    cs.append(('+' if rng.random() < 0.5 else '-') + ''.join(rng.choice(list('acgt'), int(rng.integers(1, 5)))))
                elif kind == 4:
                    cs.append('~gt%dag' % int(rng.integers(20, 400)))
                elif trial % 4 == 3:
                    cs.append(str(rng.choice(['*an', '=ACGT', ':', '*a', '~gt12'])))      # declined by the native walk
            flag = int(rng.choice([0, 0, 0, 256, 1024, 512, 2048]))
            qual = [int(q) for q in rng.integers(0, 41, qlen)] if rng.random() < 0.3 else 40
            name = 'r%d' % int(rng.integers(0, 60))                                       # duplicate names on purpose
            recs.append((start, name, bool(rng.random() < 0.5), cigar, ''.join(rng.choice(list('ACGT'), qlen)), ''.join(cs), flag, qual))
        for start, name, rev, cigar, seq, cs, flag, qual in sorted(recs, key=lambda r: r[0]):
            w.write('c', start, name, rev, cigar, seq, cs, flag=flag, quality=qual)
        w.close()
        sam = open_alignment(bam)
        for (a, b) in [(0, length), (200, 900)]:
            for kw in (SETTINGS[1], SETTINGS[2], SETTINGS[4]):
                got = region_sites_native(chromosome='c', start_pos=a, end_pos=b, sam=sam, genome=genome, **kw)
                if got is None:
                    declined += 1
                    continue
                compare(sam, genome, 'c', a, b, **kw)
                compared += 1
    assert compared > 20 and declined > 0


def test_repeat_intervals_outside_a_footprint_never_remove_a_site(tmp_path):
    """what lgmi.cli relies on when it hands every footprint an EMPTY repeat list: the reference gives a footprint the
    repeat intervals with a > end or b < start (script/giremi.py:55-59), and none of those can contain a site of the
    footprint — the routine returns the same with them as without"""
    from lgmi.region import get_region_mismatches_with_filters
    reads, genome_seq, snps, _ = simulate_region(seed=31, n_reads=90)
    sam, genome = _open(tmp_path, {'chrA': (reads, genome_seq, snps)})
    start = min(r.reference_start for r in reads)
    end = max(r.reference_start + len(r._blocks) + 400 for r in reads)       # at least the footprint's end
    s0, e0 = sam.intervals('chrA')
    start, end = int(s0.min()), int(e0.max())
    rng = np.random.default_rng(4)
    table = [[int(a), int(a + w)] for a, w in zip(rng.integers(-500, end + 2000, 400), rng.integers(1, 300, 400))]
    outside = [[a, b] for a, b in table if a > end or b < start]
    assert 10 < len(outside) < len(table)
    kw = dict(chromosome='chrA', start_pos=start, end_pos=end, sam=sam, genome=genome, snp_positions=snps, min_total_depth=2)
    with_, gone_w = get_region_mismatches_with_filters(simple_repeat_intervals=outside, **kw)
    without, gone_wo = get_region_mismatches_with_filters(simple_repeat_intervals=[], **kw)
    for strand in '+-':
        assert _plain(with_[strand]) == _plain(without[strand]) and len(without[strand]) > 0
        assert list(gone_w[strand]) == list(gone_wo[strand])
        assert not any(v['removed'] == 'in simple repeat regions' for v in gone_w[strand].values())
    # ... while an interval INSIDE the footprint does remove sites (the filter itself works)
    inside, _g = get_region_mismatches_with_filters(simple_repeat_intervals=[[start, end]], **kw)
    assert all(len(inside[strand]) == 0 for strand in '+-')


def _part_or_die(job):
    if job == 'die':
        os._exit(3)                                    # a worker killed from outside (the kernel's out-of-memory killer)
    return job


def test_pipeline_stops_when_a_worker_dies():
    """region._ordered_parts: results in job order; a worker's death ends the run with an error — a plain Pool.imap waits
    for the lost job for ever, while the pool forks a replacement from the parent (which holds the GPU context by then)"""
    import multiprocessing as mp
    import time
    from lgmi.region import _ordered_parts
    with mp.get_context('fork').Pool(3) as pool:
        assert list(_ordered_parts(pool, _part_or_die, list(range(40)), poll_s=0.05)) == list(range(40))
    with mp.get_context('fork').Pool(3) as pool:
        t0 = time.time()
        got = []
        with pytest.raises(RuntimeError, match='extraction worker died'):
            for x in _ordered_parts(pool, _part_or_die, [0, 1, 'die', 3, 4], poll_s=0.05):
                got.append(x)
        assert got == [0, 1][:len(got)] and time.time() - t0 < 20          # (stopped at once: what had come back is not handed on)
