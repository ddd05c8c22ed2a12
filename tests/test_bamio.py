"""liblgmi_io.so (csrc/bamio.cpp): the streaming, indexed BAM reader behind the CLI — region queries against a
brute-force scan of every record, the BAI written by the library against the format's own rules, pile-up filters
(pysam defaults), and that a region query touches only a small part of the file."""
import os
import struct

import numpy as np
import pytest

from lgmi.io import BamReader, BamWriter, FastaReader


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def make_bam(path, seed=3, n_reads=4000, length=3_000_000, index=True):
    rng = np.random.default_rng(seed)
    w = BamWriter(str(path), [('chr1', length), ('chrEmpty', 1000), ('chr2', length // 2)], index=index)
    recs = []
    for contig, ln, n in (('chr1', length, n_reads), ('chr2', length // 2, n_reads // 3)):
        starts = np.sort(rng.integers(0, ln - 30_000, n))
        for k, st in enumerate(starts.tolist()):
            e1, intron, e2 = int(rng.integers(40, 400)), int(rng.integers(50, 20_000)), int(rng.integers(40, 400))
            cig = [(0, e1), (3, intron), (0, e2)] if k % 3 else [(4, 5), (0, e1), (1, 2), (0, e2), (2, 3), (0, 7)]
            qlen = sum(n_ for op, n_ in cig if op in (0, 1, 4))
            seq = ''.join(rng.choice(list('ACGT'), qlen))
            flag = [0, 0, 0, 256, 1024, 4][k % 6] if k % 50 == 0 else 0
            name = '%s_r%05d' % (contig, k)
            w.write(contig, st, name, bool(k & 1), cig, seq, ':%d' % qlen, quality=rng.integers(5, 41, qlen).astype(np.uint8),
                    flag=flag)
            end = st + sum(n_ for op, n_ in cig if op in (0, 2, 3))
            recs.append((contig, st, end, name, flag, cig, seq))
    w.close()
    return recs


def test_region_queries_match_a_full_scan(tmp_path):
    recs = make_bam(tmp_path / 'a.bam')
    rd = BamReader(str(tmp_path / 'a.bam'))
    assert rd.has_index_file and rd.references == ['chr1', 'chrEmpty', 'chr2']
    rng = np.random.default_rng(0)
    for contig in ('chr1', 'chr2', 'chrEmpty'):
        mine = [r for r in recs if r[0] == contig and not r[4] & 4]
        got = list(rd.fetch(contig))
        assert [g.query_name for g in got] == [r[3] for r in mine]
        for _ in range(25):
            a = int(rng.integers(0, 3_000_000))
            b = a + int(rng.integers(1, 200_000))
            exp = [r[3] for r in mine if r[2] > a and r[1] < b]
            got = list(rd.fetch(contig, a, b))
            assert [g.query_name for g in got] == exp
            for g in got[:5]:
                r = next(x for x in mine if x[3] == g.query_name)
                assert (g.reference_start, g.reference_end, g.cigartuples, g.query_sequence) == (r[1], r[2], r[5], r[6])
                assert g.get_tag('cs') == ':%d' % len(r[6]) and g.has_tag('cs') and not g.has_tag('NM')
    assert list(rd.fetch('chrNone')) == []
    starts, ends = rd.intervals('chr2')
    assert starts.tolist() == [r[1] for r in recs if r[0] == 'chr2' and not r[4] & 4]
    assert ends.tolist() == [r[2] for r in recs if r[0] == 'chr2' and not r[4] & 4]
    with pytest.raises(KeyError):
        rd.intervals('chrNone')


def test_a_region_query_reads_a_small_part_of_the_file(tmp_path):
    make_bam(tmp_path / 'b.bam', n_reads=20_000)
    size = os.path.getsize(tmp_path / 'b.bam')
    rd = BamReader(str(tmp_path / 'b.bam'))
    before = rd.bytes_read                               # header + index file only: no scan
    assert before < 0.05 * size
    got = list(rd.fetch('chr1', 1_500_000, 1_520_000))
    assert len(got) > 10
    assert rd.bytes_read - before < 0.06 * size          # a few 64-KiB blocks, not the file


def test_index_file_equals_in_memory_index_and_follows_the_format(tmp_path):
    recs = make_bam(tmp_path / 'c.bam', n_reads=3000, index=False)
    path = str(tmp_path / 'c.bam')
    rd0 = BamReader(path)                                # no .bai: one streaming pass builds the index in memory
    assert not rd0.has_index_file
    BamReader.build_index(path)
    rd1 = BamReader(path)
    assert rd1.has_index_file
    for a, b in ((0, 10_000), (700_000, 900_000), (2_900_000, 3_000_000)):
        assert [g.query_name for g in rd0.fetch('chr1', a, b)] == [g.query_name for g in rd1.fetch('chr1', a, b)]
    # the .bai on disk: magic, one entry per reference, every bin is the reg2bin of a read that lives in it, chunk
    # offsets increase, linear index entries do not decrease
    raw = open(path + '.bai', 'rb').read()
    assert raw[:4] == b'BAI\x01' and struct.unpack_from('<i', raw, 4)[0] == 3
    at = 8
    bins_seen = []
    for ref in range(3):
        n_bin = struct.unpack_from('<i', raw, at)[0]; at += 4
        bins = set()
        for _ in range(n_bin):
            bn, n_chunk = struct.unpack_from('<Ii', raw, at); at += 8
            chunks = struct.unpack_from('<%dQ' % (2 * n_chunk), raw, at); at += 16 * n_chunk
            assert all(chunks[2 * k] < chunks[2 * k + 1] for k in range(n_chunk))
            assert all(chunks[2 * k + 1] <= chunks[2 * k + 2] for k in range(n_chunk - 1))
            bins.add(bn)
        n_intv = struct.unpack_from('<i', raw, at)[0]; at += 4
        lin = struct.unpack_from('<%dQ' % n_intv, raw, at); at += 8 * n_intv
        assert all(lin[k] <= lin[k + 1] for k in range(n_intv - 1))
        bins_seen.append(bins)
    assert at == len(raw)
    assert bins_seen[0] == {reg2bin(r[1], r[2]) for r in recs if r[0] == 'chr1'}
    assert bins_seen[1] == set()
    assert bins_seen[2] == {reg2bin(r[1], r[2]) for r in recs if r[0] == 'chr2'}


def test_pileup_defaults(tmp_path):
    recs = make_bam(tmp_path / 'd.bam', n_reads=1500, length=400_000)
    rd = BamReader(str(tmp_path / 'd.bam'))
    a, b = 100_000, 101_000
    cols = {c.pos: (c.get_query_names(), c.get_query_sequences()) for c in rd.pileup('chr1', a, b)}
    # brute force with the documented rules
    exp = {}
    path = str(tmp_path / 'd.bam')
    by_name = {g.query_name: g for g in BamReader(path).fetch('chr1', a, b)}
    for contig, st, end, name, flag, cig, seq in recs:
        if contig != 'chr1' or not (end > a and st < b) or flag & (4 | 256 | 512 | 1024):
            continue
        qual = by_name[name].query_qualities
        ref, q = st, 0
        for op, n in cig:
            if op == 0:
                for k in range(n):
                    if qual[q + k] >= 13:
                        exp.setdefault(ref + k, ([], []))
                        exp[ref + k][0].append(name); exp[ref + k][1].append(seq[q + k])
                ref += n; q += n
            elif op in (2, 3):
                for k in range(n):
                    exp.setdefault(ref + k, ([], []))
                    exp[ref + k][0].append(name); exp[ref + k][1].append('')
                ref += n
            else:
                q += n
    assert sorted(cols) == sorted(exp)                          # columns are not truncated to [a, b)
    assert min(cols) < a and max(cols) >= b
    for p in exp:
        assert cols[p] == exp[p]
    capped = {c.pos: len(c.get_query_names()) for c in rd.pileup('chr1', a, b, max_depth=2)}
    assert max(capped.values()) == 2 and set(capped) == set(cols)
    allq = {c.pos: len(c.get_query_names()) for c in rd.pileup('chr1', a, b, min_base_quality=0)}
    assert sum(allq.values()) > sum(len(v[0]) for v in cols.values())


def test_errors_are_reported(tmp_path):
    with pytest.raises(OSError):
        BamReader(str(tmp_path / 'missing.bam'))
    bad = tmp_path / 'bad.bam'
    bad.write_bytes(b'this is not a bam file at all')
    with pytest.raises(ValueError):
        BamReader(str(bad))


def test_fasta_random_access(tmp_path):
    rng = np.random.default_rng(1)
    seqs = {'c1': ''.join(rng.choice(list('ACGTacgt'), 1234)), 'c2 extra words': ''.join(rng.choice(list('ACGT'), 61)), 'c3': 'A'}
    fa = tmp_path / 'g.fa'
    with open(fa, 'w') as f:
        for name, s in seqs.items():
            f.write('>%s\n' % name)
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + '\n')
    g = FastaReader(str(fa))
    assert g.references == ['c1', 'c2', 'c3']
    for name, s in (('c1', seqs['c1']), ('c2', seqs['c2 extra words']), ('c3', 'A')):
        assert g.fetch(name) == s
        for a, b in ((0, 1), (59, 61), (60, 120), (100, 1300), (-5, 10), (1233, 1234)):
            assert g.fetch(name, a, b) == s[max(a, 0):b]
    # with a samtools-style .fai on disk
    with open(str(fa) + '.fai', 'w') as f:
        for name, (ln, off, lb, lw) in g._fai.items():
            f.write('%s\t%d\t%d\t%d\t%d\n' % (name, ln, off, lb, lw))
    g2 = FastaReader(str(fa))
    assert g2.fetch('c1', 100, 200) == seqs['c1'][100:200]


@pytest.mark.parametrize('index', [True, False])
def test_interval_scan_with_threads_equals_the_sequential_fetch(tmp_path, index):
    """lgio_bam_ref_intervals (blocks inflated by several threads, records walked in place, straddling records
    assembled) against the start / end columns of a plain fetch of the whole contig — with a .bai and with the index
    built in memory; corrupt input comes back as an error from whichever thread meets it"""
    recs = make_bam(tmp_path / 'p.bam', n_reads=12_000, index=index)
    rd = BamReader(str(tmp_path / 'p.bam'))
    for contig in ('chr1', 'chr2', 'chrEmpty'):
        want = rd._table(contig, None, None, 0)
        for threads in (1, 3, 8, 100):
            s, e = rd.intervals(contig, threads=threads)
            assert np.array_equal(s, want.start) and np.array_equal(e, want.end)
        assert len(want.start) == sum(1 for r in recs if r[0] == contig and not r[4] & 4)
    s, e = rd.intervals('chr1')                          # default thread count
    assert len(s) == len(rd._table('chr1', None, None, 0).start)
    # a flipped byte in the middle of the file: CRC / inflate failure, reported, not a crash
    raw = bytearray(open(tmp_path / 'p.bam', 'rb').read())
    raw[len(raw) // 2] ^= 0x5A
    bad = tmp_path / 'bad.bam'
    bad.write_bytes(bytes(raw))
    if index:
        import shutil
        shutil.copy(str(tmp_path / 'p.bam') + '.bai', str(bad) + '.bai')
        with pytest.raises(ValueError):
            BamReader(str(bad)).intervals('chr1', threads=4)


def test_a_record_size_completed_across_blocks_is_checked_before_it_is_believed(tmp_path):
    """crafted input for the straddling-record path of lgio_bam_ref_intervals (advice r3): the 4-byte size word of a record
    is split over two BGZF blocks and says 0 — the completion test `stream.size() == 4 + size` then holds with nothing
    behind the word.  The reader must answer with a format error, not walk a record that is not there (tools/asan_io.sh
    runs this under AddressSanitizer)."""
    import gzip
    from lgmi.io import _bgzf_block
    w = BamWriter(str(tmp_path / 'ok.bam'), [('chr1', 100_000)], index=False)
    for k in range(3):
        w.write('chr1', 100 + 50 * k, 'r%d' % k, False, [(0, 40)], 'A' * 40, ':40')
    w.close()
    raw = gzip.decompress(open(tmp_path / 'ok.bam', 'rb').read())          # BGZF is a multi-member gzip file
    l_text = struct.unpack_from('<i', raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from('<i', raw, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from('<i', raw, p)[0]
        p += 4 + l_name + 4
    first = p + 4 + struct.unpack_from('<i', raw, p)[0]                      # end of the first record
    for size_word in (0, 7, 1 << 30):
        body = raw[:first] + struct.pack('<I', size_word) + b'\0' * 64
        cut = first + 2                                                       # the size word straddles the two blocks
        bad = tmp_path / ('straddle_%d.bam' % size_word)
        bad.write_bytes(_bgzf_block(body[:cut]) + _bgzf_block(body[cut:]) +
                        bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000'))
        with pytest.raises(ValueError):
            BamReader(str(bad)).intervals('chr1', threads=2)


# ---------------------------------------------------------------- read filters of the pile-up, flag by flag
# (flag, in the pile-up?) — the rule and where it is documented:
PILEUP_FLAG_CASES = [
    ('plain', 0, True),
    ('reverse_0x10', 16, True),
    # samtools mpileup, --excl-flags: "default UNMAP,SECONDARY,QCFAIL,DUP"; pysam AlignmentFile.pileup, `flag_filter`:
    # "ignore reads where any of the bits in the flag are set. The default is BAM_FUNMAP | BAM_FSECONDARY | BAM_FQCFAIL | BAM_FDUP"
    ('unmapped_0x4', 4, False),
    ('secondary_0x100', 256, False),
    ('qcfail_0x200', 512, False),
    ('duplicate_0x400', 1024, False),
    # ... and nothing else: a supplementary alignment (0x800) is not in that mask
    ('supplementary_0x800', 2048, True),
    # pysam pileup, `ignore_orphans`: "ignore orphans (paired reads that are not in a proper pair)", default True
    # (samtools mpileup -A / --count-orphans is off by default: "Do not skip anomalous read pairs in variant calling")
    ('paired_not_proper_0x1', 1, False),
    ('paired_proper_0x3', 3, True),
    ('paired_proper_mate_reverse_0x23', 0x23, True),
]


@pytest.mark.parametrize('name,flag,kept', PILEUP_FLAG_CASES, ids=[c[0] for c in PILEUP_FLAG_CASES])
def test_pileup_read_filter_by_flag(tmp_path, name, flag, kept):
    """what the reference sees through sam.pileup(contig, start, stop) with pysam's defaults (src/giremi/mismatch.py:160-163):
    one read with the flag under test between two plain ones.  NOT pinned against pysam (it is not installed here): these
    are the documented rules, quoted above"""
    w = BamWriter(str(tmp_path / 'f.bam'), [('c', 1000)])
    w.write('c', 100, 'before', False, [(0, 50)], 'A' * 50, ':50')
    w.write('c', 110, 'probe', False, [(0, 50)], 'C' * 50, ':50', flag=flag)
    w.write('c', 120, 'after', True, [(0, 50)], 'G' * 50, ':50')
    w.close()
    rd = BamReader(str(tmp_path / 'f.bam'))
    cols = {c.pos: (c.get_query_names(), c.get_query_sequences()) for c in rd.pileup('c', 0, 1000)}
    assert sorted(cols) == list(range(100, 170))
    want = (['before', 'probe', 'after'], ['A', 'C', 'G']) if kept else (['before', 'after'], ['A', 'G'])
    assert cols[130] == want
    if not kept:
        assert all('probe' not in names for names, _ in cols.values())
    # the native site extraction applies the same filter to its pile-up step (bamio.cpp: region_sites_impl step 2):
    # 'probe' as a reference-allele read of a candidate site at 130
    from lgmi.io import SiteParams
    w = BamWriter(str(tmp_path / 'g.bam'), [('c', 1000)])
    for k in range(4):
        w.write('c', 100, 'alt%d' % k, False, [(0, 50)], 'C' * 30 + 'T' + 'C' * 19, ':30*ct:19')
    for k in range(2):
        w.write('c', 105, 'ref%d' % k, False, [(0, 50)], 'C' * 50, ':50')
    w.write('c', 110, 'probe', False, [(0, 50)], 'C' * 50, ':50', flag=flag)
    w.close()
    rd = BamReader(str(tmp_path / 'g.bam'))
    raw = rd.region_sites('c', 0, 1000, SiteParams(keep_non_spliced_read=1, min_base_quality=13, max_depth=8000, min_dist_from_splice=0,
                                                   half_window=50, min_allele_depth=1, min_allele_ratio=0.0, min_total_depth=0,
                                                   max_window_mismatch=10, max_window_mismatch_type=3))
    assert raw is not None and raw['pos'].tolist() == [130] and raw['ref'].decode() == 'C'
    alleles = raw['allele_nt'].decode()
    noff, pool = raw['name_off'], raw['names']
    names = lambda ids: [pool[noff[i]:noff[i + 1]].decode() for i in ids]
    by_nt = {alleles[a]: names(raw['reads'][raw['reads_off'][a]:raw['reads_off'][a + 1]].tolist()) for a in range(len(alleles))}
    assert by_nt['T'] == ['alt0', 'alt1', 'alt2', 'alt3']
    # (a read on the other strand is in the column but is not a read of this strand's site: mismatch.py:176-180)
    assert by_nt['C'] == ['ref0', 'ref1'] + (['probe'] if kept and not flag & 16 else [])


def test_pileup_base_quality_threshold_and_deletions(tmp_path):
    """pysam pileup, `min_base_quality`: "Minimum base quality. Bases below the minimum quality will not be output", default 13
    — a base of quality 12 is not in its column, one of quality 13 is; the read's other bases are unaffected.
    A deletion spanning the column: pysam's PileupColumn.get_query_sequences() yields an empty string for the read (no
    base, `add_indels` off) and get_query_names() still lists it; the reference then compares '' with the reference base
    (src/giremi/mismatch.py:169-175) and never counts the read for the reference allele.  Same for a reference skip (N).
    Known difference, documented in include/lgmi_io.h: htslib tests min_base_quality at the query position next to a
    deletion and drops the entry when that base is below it; here the entry stays — it changes nothing the reference
    computes, since '' never equals a base."""
    q12 = np.full(50, 40, np.uint8); q12[20] = 12
    q13 = np.full(50, 40, np.uint8); q13[20] = 13
    qdel = np.full(45, 40, np.uint8); qdel[19] = 5                              # the base before the deletion is poor
    w = BamWriter(str(tmp_path / 'q.bam'), [('c', 1000)])
    w.write('c', 100, 'q12', False, [(0, 50)], 'A' * 50, ':50', quality=q12)
    w.write('c', 100, 'q13', False, [(0, 50)], 'A' * 50, ':50', quality=q13)
    w.write('c', 100, 'del', False, [(0, 20), (2, 5), (0, 25)], 'A' * 45, ':20-ccccc:25', quality=qdel)
    w.write('c', 100, 'skip', True, [(0, 10), (3, 30), (0, 10)], 'A' * 20, ':10~gt30ag:10')
    w.write('c', 100, 'noqual', False, [(0, 50)], 'A' * 50, ':50', quality=np.full(50, 255, np.uint8))   # '*' in SAM: 0xFF bytes
    w.close()
    rd = BamReader(str(tmp_path / 'q.bam'))
    cols = {c.pos: dict(zip(c.get_query_names(), c.get_query_sequences())) for c in rd.pileup('c', 0, 1000)}
    assert cols[120] == {'q13': 'A', 'del': '', 'skip': '', 'noqual': 'A'}                       # q12 dropped here ...
    assert cols[119] == {'q12': 'A', 'q13': 'A', 'skip': '', 'noqual': 'A'}                      # ... only here; del's poor base too
    assert cols[121]['q12'] == 'A' and cols[124]['del'] == '' and cols[125]['del'] == 'A'
    assert cols[109]['skip'] == 'A' and cols[110]['skip'] == '' and cols[139]['skip'] == '' and cols[140]['skip'] == 'A'
    loose = {c.pos: c.get_query_names() for c in rd.pileup('c', 0, 1000, min_base_quality=0)}
    assert loose[120] == ['q12', 'q13', 'del', 'skip', 'noqual'] and loose[119] == ['q12', 'q13', 'del', 'skip', 'noqual']


def test_pileup_max_depth_default(tmp_path):
    """pysam pileup, `max_depth`: "Maximum read depth permitted. The default limit is '8000'" — 8,005 reads over one column:
    the first 8,000 in file order are in it.  (htslib caps the reads entering the pile-up, this reader caps each column:
    the same on reads that all start before the column, as here; include/lgmi_io.h)"""
    w = BamWriter(str(tmp_path / 'd.bam'), [('c', 1000)])
    for k in range(8005):
        w.write('c', 100 + (k >= 8000), 'r%04d' % k, False, [(0, 30)], 'A' * 30, ':30')
    w.close()
    rd = BamReader(str(tmp_path / 'd.bam'))
    cols = {c.pos: c.get_query_names() for c in rd.pileup('c', 0, 1000)}
    assert len(cols[100]) == 8000 and len(cols[110]) == 8000 and cols[110] == ['r%04d' % k for k in range(8000)]
    assert len(cols[130]) == 5                                                   # only the five late starters reach 130
    assert len({c.pos: c for c in rd.pileup('c', 0, 1000, max_depth=9000)}[110].get_query_names()) == 8005


def test_site_extraction_beyond_max_depth_takes_the_first_reads_of_a_column(tmp_path):
    """lgio_bam_region_sites counts a read at a column only while the column holds fewer than max_depth reads (pysam's
    `max_depth`; the fast, segment-wise form of its pile-up step is exact only below that and hands over to the base-by-base
    form otherwise): with max_depth 5 the column of the site at 130 takes alt0..alt3 and ref0 — ref1 and ref2 come too late"""
    from lgmi.io import SiteParams
    w = BamWriter(str(tmp_path / 'm.bam'), [('c', 1000)])
    for k in range(4):
        w.write('c', 100, 'alt%d' % k, False, [(0, 50)], 'C' * 30 + 'T' + 'C' * 19, ':30*ct:19')
    for k in range(3):
        w.write('c', 105 + k, 'ref%d' % k, False, [(0, 50)], 'C' * 50, ':50')
    w.close()
    rd = BamReader(str(tmp_path / 'm.bam'))
    got = {}
    for depth in (5, 8000):
        raw = rd.region_sites('c', 0, 1000, SiteParams(keep_non_spliced_read=1, min_base_quality=13, max_depth=depth, min_dist_from_splice=0,
                                                       half_window=50, min_allele_depth=1, min_allele_ratio=0.0, min_total_depth=0,
                                                       max_window_mismatch=10, max_window_mismatch_type=3))
        assert raw['pos'].tolist() == [130]
        alleles = raw['allele_nt'].decode()
        noff, pool = raw['name_off'], raw['names']
        got[depth] = {alleles[a]: [pool[noff[i]:noff[i + 1]].decode() for i in raw['reads'][raw['reads_off'][a]:raw['reads_off'][a + 1]].tolist()]
                      for a in range(len(alleles))}
    assert got[8000] == {'T': ['alt0', 'alt1', 'alt2', 'alt3'], 'C': ['ref0', 'ref1', 'ref2']}
    assert got[5] == {'T': ['alt0', 'alt1', 'alt2', 'alt3'], 'C': ['ref0']}


def test_native_removed_table_writer_equals_pandas(tmp_path):
    """lgio_write_removed_table against what the CLI wrote before (pandas / pyarrow, lgmi.cli._write_removed): the same bytes —
    header or not, appended parts, one thread or several, an empty part; names that would need quoting are refused"""
    import pandas as pd
    from lgmi.cli import _write_removed
    from lgmi.io import write_removed_table
    rng = np.random.default_rng(8)
    chroms, reasons = ['chr1', 'chrX', 'scaffold_12.1'], ['too many window mismatches', 'too few usable reads after filters', 'in homopoly regions']
    n = 300_000
    cc, st = rng.integers(0, 3, n).astype(np.int32), rng.integers(0, 2, n).astype(np.int8)
    ps, rc = rng.integers(0, 250_000_000, n).astype(np.int64), rng.integers(0, 3, n).astype(np.int8)
    ps[:3] = [0, 9, 10]
    df = pd.DataFrame({'chromosome': pd.Categorical.from_codes(cc, categories=chroms), 'strand': pd.Categorical.from_codes(st, categories=list('+-')),
                       'pos': ps, 'removed': pd.Categorical.from_codes(rc, categories=reasons)})
    a, b = str(tmp_path / 'a.txt'), str(tmp_path / 'b.txt')
    _write_removed(df, a)
    for threads in (1, 5):
        write_removed_table(b, chroms, reasons, cc, st, ps, rc, threads=threads)
        assert open(a, 'rb').read() == open(b, 'rb').read()
    # in parts, the way the run writes it
    cut = 123_457
    write_removed_table(b, chroms, reasons, cc[:cut], st[:cut], ps[:cut], rc[:cut], header=True, append=False)
    write_removed_table(b, chroms, reasons, cc[:0], st[:0], ps[:0], rc[:0], header=False, append=True)
    write_removed_table(b, chroms[::-1], reasons, 2 - cc[cut:], st[cut:], ps[cut:], rc[cut:], header=False, append=True)   # (a part's own name order)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    with pytest.raises(ValueError):
        write_removed_table(b, ['chr\t1'], reasons, cc[:5] * 0, st[:5], ps[:5], rc[:5])
    with pytest.raises(ValueError):
        write_removed_table(b, chroms, reasons, cc[:5] + 7, st[:5], ps[:5], rc[:5])


def test_removed_writer_queues_parts_in_order(tmp_path):
    """cli._RemovedWriter.write_arrays hands the parts to one writer thread: the file is the whole table in the order the
    parts were handed over — also when one part falls back to pandas (a name pandas would quote) and when a DataFrame part
    comes in between; a part the native writer refuses for good surfaces from close(), and abort() leaves nothing"""
    import filecmp
    import pandas as pd
    from lgmi.cli import _RemovedWriter, _write_removed
    rng = np.random.default_rng(11)
    chroms, reasons = ['chr1', 'chr2'], ['in homopoly regions', 'too few usable reads after filters']

    def part(n, names=chroms):
        return (list(names), list(reasons), rng.integers(0, 2, n).astype(np.int32), rng.integers(0, 2, n).astype(np.int8),
                rng.integers(1, 10 ** 8, n).astype(np.int64), rng.integers(0, 2, n).astype(np.int8))

    def frame(parts):
        cols = {'chromosome': [], 'strand': [], 'pos': [], 'removed': []}
        for names, rs, cc, st, ps, rc in parts:
            cols['chromosome'] += [names[k] for k in cc]
            cols['strand'] += ['+-'[k] for k in st]
            cols['pos'] += ps.tolist()
            cols['removed'] += [rs[k] for k in rc]
        return pd.DataFrame(cols)

    parts = [part(4000), part(1), part(0), part(30, ['chr\t1', 'chr2']), part(2500)]
    w = _RemovedWriter(str(tmp_path / 'r.txt'))
    for k, p in enumerate(parts):
        w.write_arrays(p)
        if k == 1:
            w(frame([part0 := part(7)]))                                  # a DataFrame part between two queued ones
            parts_with_frame = parts[:2] + [part0]
    assert not os.path.exists(w.final)
    w.close()
    want = str(tmp_path / 'want.txt')
    _write_removed(frame(parts_with_frame + parts[2:]), want)
    assert filecmp.cmp(want, w.final, shallow=False) and not os.path.exists(w.path)
    bad = _RemovedWriter(str(tmp_path / 'bad.txt'))
    bad.write_arrays(part(10))
    wrong = list(part(10))
    wrong[2] = wrong[2] + 9                                               # chromosome codes outside the name list
    bad.write_arrays(tuple(wrong))
    with pytest.raises(Exception):
        bad.close()
    bad.abort()
    assert not os.path.exists(bad.path) and not os.path.exists(bad.final)


def test_fasta_index_without_fai_matches_the_line_by_line_pass(tmp_path):
    """FastaReader._scan_fast (mmap.find + bytes.count) builds the index _scan builds line by line: LF and CRLF files, an
    unterminated last line, short lines, lines before the first header, an empty sequence, an empty first line (fallback),
    '>' inside a line — and fetch() through it returns the bases"""
    import random
    rnd = random.Random(5)
    seqs = [(b'chr1', 1000), (b'chr2', 0), (b'chrX', 61), (b'c4', 60), (b'c5', 1)]

    def make(path, eol=b'\n', last_eol=True, width=60, blank_first=False, junk=False):
        truth = {}
        with open(path, 'wb') as f:
            if junk:
                f.write(b'junk line' + eol)
            for k, (name, n) in enumerate(seqs):
                f.write(b'>' + name + b' some description' + eol)
                if blank_first:
                    f.write(eol)
                s = bytes(rnd.choice(b'ACGTN') for _ in range(n))
                truth[name.decode()] = s.decode()
                f.write(eol.join(s[i:i + width] for i in range(0, n, width)))
                if n and (k + 1 < len(seqs) or last_eol):
                    f.write(eol)
        return truth

    for kw in (dict(), dict(eol=b'\r\n'), dict(last_eol=False), dict(width=7), dict(junk=True), dict(blank_first=True)):
        path = str(tmp_path / 'x.fa')
        truth = make(path, **kw)
        assert FastaReader._scan_fast(path) == FastaReader._scan(path), kw
        if not kw.get('blank_first'):
            fa = FastaReader(path)
            assert fa.references == [n.decode() for n, _ in seqs]
            assert fa.fetch('chr1', 100, 333) == truth['chr1'][100:333] and fa.fetch('chrX') == truth['chrX'] and fa.fetch('chr2') == ''
            again = FastaReader(path, fai=fa.index)                    # a worker's reader on the parent's index
            assert again.fetch('c4', 3, 60) == truth['c4'][3:60]
    for raw in (b'>only', b'', b'>a\nAC>GT\n>b\n\n', b'no header at all\n'):
        path = str(tmp_path / 'y.fa')
        with open(path, 'wb') as f:
            f.write(raw)
        assert FastaReader._scan_fast(path) == FastaReader._scan(path), raw


def test_native_float_text_equals_numpy():
    """lgio_format_doubles (what lgio_write_table writes for a float64) against numpy's astype(str) — what pandas' to_csv
    writes: random bit patterns, magnitudes from 1e-12 to 1e20, uniform values, whole numbers, p-values k / 1001, and the
    edges of the positional / scientific switch"""
    from lgmi.io import format_doubles
    rng = np.random.default_rng(1)
    specials = np.array([0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999999999999999e-5, 1e-5, 1e16, 9999999999999998.0, 1e15,
                         123456789012345.0, 1e22, 1e100, 1e-100, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                         np.inf, -np.inf, np.nan, 0.000999000999000999, 1 / 3, 2 / 3, 1e-7, 1.5e-7, 123456.789, 100.0, 1e23, 8.41e21])
    bits = rng.integers(0, 2 ** 63, 60000, dtype=np.int64).view(np.float64)
    x = np.concatenate([specials, bits[np.isfinite(bits)], 10.0 ** rng.uniform(-12, 20, 60000) * rng.choice([-1, 1], 60000),
                        rng.random(60000), rng.integers(-10 ** 6, 10 ** 6, 20000).astype(np.float64), np.round(rng.random(20000), 3),
                        rng.integers(1, 1002, 20000) / 1001.0])
    want = x.astype(str)
    want[np.isnan(x)] = ''
    got = format_doubles(x)
    bad = [(a, b, c) for a, b, c in zip(x.tolist(), got, want.tolist()) if b != c]
    assert not bad, bad[:5]
