"""Duck-typed stand-ins for pysam objects (pysam/htslib are not installed here) and a small
simulator of spliced long reads with minimap2-style cs tags.  Used both by the golden-vector
generator (which drives the REFERENCE with them) and by the tests (which drive lgmi.region)."""
import numpy as np

BASES = 'ACGT'


class FakeRead:
    def __init__(self, name, start, is_reverse, cs, blocks):
        self.query_name = name
        self.reference_start = start
        self.is_reverse = is_reverse
        self._cs = cs
        self._blocks = blocks          # [(ref_pos, read_base)] for every aligned base
        self.reference_end = blocks[-1][0] + 1 if blocks else start

    def get_tag(self, tag):
        assert tag == 'cs'
        return self._cs


class FakeColumn:
    def __init__(self, pos, names, bases):
        self.pos = pos
        self._names, self._bases = names, bases

    def get_query_names(self):
        return list(self._names)

    def get_query_sequences(self):
        return list(self._bases)


class FakeSam:
    def __init__(self, reads):
        self.reads = reads

    def fetch(self, contig=None, start=None, stop=None):
        for r in self.reads:
            if start is None or (r.reference_end > start and r.reference_start < stop):
                yield r

    def pileup(self, contig=None, start=None, stop=None):
        cols = {}
        for r in self.reads:
            for pos, base in r._blocks:
                if start <= pos < stop:
                    cols.setdefault(pos, ([], []))
                    cols[pos][0].append(r.query_name)
                    cols[pos][1].append(base)
        for pos in sorted(cols):
            yield FakeColumn(pos, cols[pos][0], cols[pos][1])


class FakeSamSkips(FakeSam):
    """pile-up as pysam produces it on a real BAM with default arguments: a read that has a reference skip (intron)
    at a column is listed there with an EMPTY base string, and columns are not truncated to [start, stop) — the two
    pysam behaviours the path's look-ups depend on (mismatch.py:160-190).  Used to run the REFERENCE on exactly what
    the CLI's BAM reader hands over for the same reads."""

    def pileup(self, contig=None, start=None, stop=None):
        cols = {}
        for r in self.reads:
            if not (r.reference_end > start and r.reference_start < stop):
                continue
            aligned = dict(r._blocks)
            for pos in range(r.reference_start, r.reference_end):
                cols.setdefault(pos, ([], []))
                cols[pos][0].append(r.query_name)
                cols[pos][1].append(aligned.get(pos, ''))
        for pos in sorted(cols):
            yield FakeColumn(pos, cols[pos][0], cols[pos][1])


class FakeGenome:
    def __init__(self, seq):
        self.seq = seq

    def fetch(self, contig, start, end):
        return self.seq[max(start, 0):end]


def simulate_region(seed, n_reads=60, length=900, n_snps=6, n_edits=8, err=0.004, lower_case_ref=False, scale=1):
    """one gene with three exons; reads start anywhere, all spliced; two haplotypes differing at n_snps
    positions; n_edits A>G editing sites hit with a site-specific probability; uniform sequencing errors.
    scale > 1 stretches the gene (length and exon coordinates times scale) for cfg1-sized cases.
    Returns (reads, genome string, snp positions, [start, end])."""
    rng = np.random.default_rng(seed)
    length = length * scale
    seq = ''.join(rng.choice(list(BASES), length))
    # a homopolymer stretch and a fixed exon structure
    seq = seq[:200] + 'AAAAAAA' + seq[207:]
    exons = [(20 * scale, 300 * scale), (380 * scale, 620 * scale), (700 * scale, 880 * scale)]
    exonic = [p for a, b in exons for p in range(a + 12, b - 12)]
    snp_pos = sorted(int(p) for p in rng.choice(exonic, n_snps, replace=False))
    snp_alt = {p: rng.choice([b for b in BASES if b != seq[p]]) for p in snp_pos}
    a_sites = [p for p in exonic if seq[p] == 'A' and p not in snp_alt]
    edit_pos = sorted(int(p) for p in rng.choice(a_sites, min(n_edits, len(a_sites)), replace=False))
    edit_rate = {p: float(rng.uniform(0.1, 0.6)) for p in edit_pos}
    reads = []
    for k in range(n_reads):
        hap = int(rng.integers(0, 2))
        reverse = bool(rng.random() < 0.5)
        first = int(rng.integers(exons[0][0], exons[0][1] - 40))
        last = int(rng.integers(exons[2][0] + 40, exons[2][1]))
        cs, blocks, run = [], [], 0
        for ei, (a, b) in enumerate(exons):
            lo = first if ei == 0 else a
            hi = last if ei == 2 else b
            if ei > 0:
                if run:
                    cs.append(':%d' % run)
                    run = 0
                pa, pb = exons[ei - 1][1], a
                cs.append('~%s%d%s' % (seq[pa:pa + 2].lower(), pb - pa, seq[pb - 2:pb].lower()))
            for p in range(lo, hi):
                base = seq[p]
                if p in snp_alt and hap == 1:
                    base = snp_alt[p]
                elif p in edit_rate and rng.random() < edit_rate[p]:
                    base = 'G'
                if rng.random() < err:
                    base = rng.choice([x for x in BASES if x != base])
                blocks.append((p, base))
                if base == seq[p]:
                    run += 1
                else:
                    if run:
                        cs.append(':%d' % run)
                        run = 0
                    cs.append('*%s%s' % (seq[p].lower(), base.lower()))
        if run:
            cs.append(':%d' % run)
        reads.append(FakeRead('read%03d' % k, first, reverse, ''.join(cs), blocks))
    reads.sort(key=lambda r: r.reference_start)
    genome = seq.lower() if lower_case_ref else seq
    return reads, genome, snp_pos, [0, length]
