#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

    python3 -B tests/golden/gen_golden.py            # rewrites tests/golden/*.json

The reference (gxiaolab/L-GIREMI, /root/reference, read-only) is imported by file
path — `import giremi` fails because the dist is not pip-installed
(src/giremi/__init__.py:3) — and driven with synthetic `mismatches` dicts built
here.  Only DATA leaves this script: inputs and the reference's outputs.  The
3x3 tables are captured from the label vectors the reference hands to
sklearn.metrics.mutual_info_score (mutual_information.py:41), by wrapping that
one name in the imported module.

/root/reference does not exist on the GPU box; tests read the JSON only.
"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src/giremi'
NTS = 'ACGT'


def load_ref(name):
    spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_mi = load_ref('mutual_information')
ref_stat = load_ref('stat')

_captured = []
_orig_mis = ref_mi.mutual_info_score


def _recording_mis(l1, l2):
    t = [0] * 9
    for a, b in zip(l1, l2):
        t[3 * a + b] += 1
    _captured.append(t)
    return _orig_mis(l1, l2)


ref_mi.mutual_info_score = _recording_mis


def run_ref(mismatches, min_common):
    _captured.clear()
    rows = ref_mi.mismatch_pair_mutual_info(mismatches, min_common_reads=min_common)
    tables = [list(t) for t in _captured]
    assert len(tables) == len(rows)
    het = [r for r in rows if r[1] == 'het_snp' or r[3] == 'het_snp']
    mean_all = ref_mi.mean_mismatch_pair_mutual_info(rows) if rows else []
    mean_het = ref_mi.mean_mismatch_pair_mutual_info(het) if het else []
    return rows, tables, mean_all, mean_het


def site(type_, alleles):
    """alleles: list of (nt, [read names]) in insertion order; depth = list lengths."""
    return {'ref': alleles[0][0], 'type': type_,
            'depth': {nt: len(names) for nt, names in alleles},
            'nt': {nt: list(names) for nt, names in alleles},
            'neighbor': {}, 'up': 'A', 'down': 'C'}


def ser_mismatches(mm):
    return [[int(pos), s['type'], [[nt, int(d)] for nt, d in s['depth'].items()],
             [[nt, list(names)] for nt, names in s['nt'].items()]]
            for pos, s in mm.items()]


def case(name, mm, min_common):
    rows, tables, mean_all, mean_het = run_ref(mm, min_common)
    return {'name': name, 'min_common': min_common, 'sites': ser_mismatches(mm),
            'rows': [[int(r[0]), r[1], int(r[2]), r[3], float(r[4])] for r in rows],
            'tables': tables,
            'mean_all': [[int(a), float(b)] for a, b in mean_all],
            'mean_het': [[int(a), float(b)] for a, b in mean_het]}


def R(prefix, idx):
    return ['%s%d' % (prefix, i) for i in idx]


def edge_cases():
    out = []
    r = lambda a, b: R('r', range(a, b))
    # 2-allele x 2-allele, partial and perfect linkage
    mm = {100: site('het_snp', [('A', r(0, 10)), ('G', r(10, 20))]),
          200: site('mismatch', [('A', r(0, 7) + r(10, 14)), ('G', r(7, 10) + r(14, 20))]),
          300: site('mismatch', [('C', r(0, 10)), ('T', r(10, 20))])}
    out.append(case('two_allele_linkage', mm, 5))
    # a 3-allele site (3x2 and 3x3 tables)
    mm = {10: site('het_snp', [('A', r(0, 12)), ('G', r(12, 21)), ('T', r(21, 27))]),
          20: site('mismatch', [('C', r(0, 5) + r(12, 18) + r(21, 23)), ('T', r(5, 12) + r(18, 21) + r(23, 27))]),
          30: site('snp', [('A', r(0, 9) + r(21, 24)), ('C', r(9, 15)), ('G', r(15, 21) + r(24, 27))])}
    out.append(case('three_allele', mm, 5))
    # one site monomorphic among the COMMON reads -> exactly 0.0, row still emitted
    mm = {10: site('het_snp', [('A', r(0, 8)), ('G', r(8, 16))]),
          30: site('mismatch', [('C', r(0, 8)), ('T', r(20, 30))])}
    out.append(case('monomorphic_in_common', mm, 5))
    # common reads below the threshold -> pair skipped
    mm = {10: site('het_snp', [('A', r(0, 8)), ('G', r(8, 16))]),
          30: site('mismatch', [('C', r(0, 2)), ('T', r(14, 16) + r(20, 30))]),
          40: site('mismatch', [('C', r(0, 3)), ('T', r(13, 16) + r(20, 30))])}
    out.append(case('below_min_common', mm, 6))
    out.append(case('below_min_common_mc1', mm, 1))
    # a read listed under two alleles of one site: the last allele in nt order wins
    mm = {10: site('het_snp', [('A', r(0, 8)), ('G', r(6, 16))]),
          30: site('mismatch', [('C', r(0, 5) + r(10, 13)), ('T', r(5, 10) + r(13, 16))])}
    out.append(case('duplicate_read_last_allele_wins', mm, 5))
    # depth ties with 4 alleles: stable sort on depth decides which two fall to class 0
    mm = {10: site('mismatch', [('A', r(0, 6)), ('C', r(6, 12)), ('G', r(12, 18)), ('T', r(18, 24))]),
          20: site('het_snp', [('A', r(0, 3) + r(6, 9) + r(12, 15) + r(18, 21)),
                               ('G', r(3, 6) + r(9, 12) + r(15, 18) + r(21, 24))]),
          25: site('het_snp', [('T', r(0, 4) + r(12, 20)), ('C', r(4, 12) + r(20, 24))])}
    out.append(case('depth_ties_four_alleles', mm, 5))
    # depth dict that disagrees with the nt lists (ranking follows depth, not len(nt))
    s = site('mismatch', [('A', r(0, 9)), ('G', r(9, 14)), ('T', r(14, 20))])
    s['depth'] = {'A': 2, 'G': 50, 'T': 7}
    mm = {5: s, 9: site('het_snp', [('C', r(0, 4) + r(9, 12) + r(14, 17)), ('T', r(4, 9) + r(12, 14) + r(17, 20))])}
    out.append(case('depth_dict_overrides', mm, 5))
    # allele present in nt but absent from depth -> class 0
    s = site('mismatch', [('A', r(0, 9)), ('G', r(9, 14)), ('T', r(14, 20))])
    s['depth'] = {'A': 9, 'G': 5}
    mm = {5: s, 9: site('het_snp', [('C', r(0, 4) + r(9, 12) + r(14, 17)), ('T', r(4, 9) + r(12, 14) + r(17, 20))])}
    out.append(case('allele_missing_from_depth', mm, 5))
    # single site / empty -> no rows
    out.append(case('single_site', {7: site('het_snp', [('A', r(0, 8)), ('G', r(8, 16))])}, 5))
    out.append(case('empty', {}, 5))
    # exactly min_common common reads; n words boundary 64/65 reads
    mm = {1: site('het_snp', [('A', r(0, 32)), ('G', r(32, 64))]),
          2: site('mismatch', [('A', r(0, 20) + r(32, 40)), ('G', r(20, 32) + r(40, 65))]),
          3: site('mismatch', [('C', r(59, 62)), ('T', r(62, 65))])}
    out.append(case('word_boundary_64_65', mm, 5))
    # perfectly linked het pair, big counts
    mm = {1: site('het_snp', [('A', r(0, 500)), ('G', r(500, 1000))]),
          2: site('het_snp', [('C', r(0, 500)), ('T', r(500, 1000))]),
          3: site('mismatch', [('C', r(0, 1000, )[::2]), ('T', r(0, 1000)[1::2])])}
    out.append(case('perfect_linkage_1000', mm, 5))
    return out


def random_block(rng, n_sites, n_reads, max_alleles, cover, het_frac=0.3):
    names = R('q', range(n_reads))
    order = rng.permutation(n_reads)
    names = [names[i] for i in order]  # first-seen order differs from sorted order
    hap = rng.integers(0, 2, n_reads)
    mm = {}
    pos = 1000
    for _s in range(n_sites):
        pos += int(rng.integers(1, 60))
        k = int(rng.integers(2, max_alleles + 1))
        covered = np.nonzero(rng.random(n_reads) < cover)[0]
        if len(covered) < 2:
            covered = np.arange(min(n_reads, 4))
        is_het = rng.random() < het_frac
        if is_het:
            al = (hap[covered] ^ (rng.random(len(covered)) < 0.05)).astype(int)
            if k > 2:
                third = rng.random(len(covered)) < 0.1
                al = np.where(third, rng.integers(2, k, len(covered)), al)
        else:
            p = rng.dirichlet(np.ones(k) * 1.5)
            al = rng.choice(k, size=len(covered), p=p)
        nts = list(rng.permutation(list(NTS))[:k])
        alleles = []
        for a in range(k):
            members = [names[i] for i in covered[al == a]]
            if members:
                alleles.append((nts[a], members))
        if len(alleles) < 2:
            half = len(covered) // 2
            alleles = [(nts[0], [names[i] for i in covered[:half]]),
                       (nts[1], [names[i] for i in covered[half:]])]
        t = 'het_snp' if is_het else ('snp' if rng.random() < 0.1 else 'mismatch')
        mm[pos] = site(t, alleles)
    return mm


def random_cases():
    out = []
    rng = np.random.Generator(np.random.PCG64(20250808))
    for i in range(40):
        n_sites = int(rng.integers(2, 41))
        n_reads = int(rng.integers(6, 301))
        max_alleles = int(rng.integers(2, 5))
        cover = float(rng.uniform(0.15, 1.0))
        mc = int(rng.choice([1, 5, 6, 20]))
        mm = random_block(rng, n_sites, n_reads, max_alleles, cover)
        out.append(case('random_%02d_P%d_R%d_A%d_mc%d' % (i, n_sites, n_reads, max_alleles, mc), mm, mc))
    return out


def banded_block(seed, n_sites, n_reads, mean_span=20):
    """cfg1-like: reads sorted by start, each spans a window of sites."""
    rng = np.random.Generator(np.random.PCG64(seed))
    start = np.sort(rng.integers(0, n_sites, n_reads))
    span = 1 + rng.geometric(1.0 / mean_span, n_reads)
    hap = rng.integers(0, 2, n_reads)
    lists = {}
    for s in range(n_sites):
        is_het = (s % 5 == 0)
        e = rng.uniform(0.05, 0.5)
        tri = rng.random() < 0.02
        cov = np.nonzero((start <= s) & (s < start + span) & (rng.random(n_reads) >= 0.10))[0]
        if is_het:
            al = hap[cov] ^ (rng.random(len(cov)) < 0.02)
        else:
            al = (rng.random(len(cov)) < e).astype(int)
        if tri:
            al = np.where(rng.random(len(cov)) < 0.05, 2, al)
        groups = [(nt, ['b%d' % i for i in cov[al == a]]) for a, nt in enumerate('AGT')]
        tot = sum(len(g[1]) for g in groups)
        groups = [g for g in groups if len(g[1]) >= 3 and len(g[1]) / max(tot, 1) >= 0.05]
        if len(groups) < 2:
            continue
        t = 'het_snp' if is_het else ('snp' if rng.random() < 0.01 else 'mismatch')
        lists[10_000 + 37 * s] = site(t, groups)
    return lists


def time_reference(out_path):
    """wall time of the reference's own MI step, this container."""
    res = {'host': '8 vCPU Intel Xeon @ 2.10GHz container', 'python': sys.version.split()[0]}
    import sklearn
    res['sklearn'] = sklearn.__version__
    blocks = []
    mm = banded_block(20250809, 500, 2000)
    t0 = time.perf_counter()
    rows = _orig_call(mm, 6)
    dt = time.perf_counter() - t0
    p = len(mm)
    blocks.append({'block': 'cfg1 banded 500x2000 (%d sites kept)' % p, 'pairs_examined': p * (p - 1) // 2,
                   'pairs_emitted': len(rows), 'wall_s': dt})
    rng = np.random.Generator(np.random.PCG64(7))
    mm = random_block(rng, 60, 2000, 2, 0.8)
    t0 = time.perf_counter()
    rows = _orig_call(mm, 6)
    dt = time.perf_counter() - t0
    blocks.append({'block': 'dense 60x2000 cover 0.8', 'pairs_examined': 60 * 59 // 2,
                   'pairs_emitted': len(rows), 'wall_s': dt})
    res['blocks'] = blocks
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)


LARGE_CONFIGS = [('cfg2 10k x 50k', 50_000), ('cfg3 22 x (9,091 x 45,455)', 45_455), ('north-star 50k x 200k', 200_000),
                 ('cfg5 depth x 4 (181,820 reads per chromosome)', 181_820)]


def _time_block(args):
    """worker of time_reference_large (also run through multiprocessing.Pool): one dense 64-site block = 2,016 pairs"""
    seed, n_reads = args
    rng = np.random.Generator(np.random.PCG64(seed))
    mm = random_block(rng, 64, n_reads, 2, 0.9)
    t0 = time.perf_counter()
    rows = _orig_call(mm, 6)
    return time.perf_counter() - t0, len(rows)


def time_reference_large(out_path):
    """a fixed-seed ~2,000-pair subsample (one dense block of 64 sites = 2,016 pairs at the
    config's read depth, 10 % dropout) of every larger config, single process and multiprocessing.Pool(8) over 8
    such blocks — the reference's own parallelism (script/giremi.py:375-380).  Appends to the timing JSON."""
    import multiprocessing as mp
    with open(out_path) as f:
        res = json.load(f)
    res['subsamples'] = []
    for k, (name, n_reads) in enumerate(LARGE_CONFIGS):
        dt, n_rows = _time_block((1000 + k, n_reads))
        with mp.Pool(8) as pool:
            t0 = time.perf_counter()
            outs = pool.map(_time_block, [(2000 + 10 * k + j, n_reads) for j in range(8)])
            wall8 = time.perf_counter() - t0
        res['subsamples'].append({'config': name, 'n_reads': n_reads, 'pairs_per_block': 2016,
                                  'single_process': {'wall_s': dt, 'pairs_per_s': 2016 / dt, 'pairs_emitted': n_rows},
                                  'pool8_over_8_blocks': {'wall_s': wall8, 'pairs_per_s': 8 * 2016 / wall8,
                                                          'per_block_wall_s': [o[0] for o in outs]}})
        with open(out_path, 'w') as f:
            json.dump(res, f, indent=1)


def _orig_call(mm, mc):
    ref_mi.mutual_info_score = _orig_mis
    try:
        return ref_mi.mismatch_pair_mutual_info(mm, min_common_reads=mc)
    finally:
        ref_mi.mutual_info_score = _recording_mis


def ecdf_cases():
    rng = np.random.Generator(np.random.PCG64(99))
    out = []
    for n in (1, 4, 50):
        sample = [float(x) for x in np.round(rng.random(n), 3)]
        f = ref_stat.ecdf(sample)
        q = sample + [0.0, 1.0, 0.5, -1.0, 2.0]
        out.append({'sample': sample, 'query': q, 'value': [float(f(v)) for v in q]})
    return out


def load_ref_mismatch():
    """giremi.mismatch imports giremi.cs / utils / mutual_information but not pysam; the package itself
    cannot be imported (src/giremi/__init__.py:3 needs pip metadata), so register an empty stub package
    whose __path__ points at the reference sources"""
    import importlib
    import types
    if 'giremi' not in sys.modules:
        pkg = types.ModuleType('giremi')
        pkg.__path__ = [REF]
        sys.modules['giremi'] = pkg
    return importlib.import_module('giremi.mismatch')


REGION_CASES = [
    # name, simulate_region kwargs, region_mismatch_analysis kwargs
    ('default', dict(seed=11), dict()),
    ('more_reads', dict(seed=12, n_reads=150), dict(min_common_reads=6)),
    ('keep_non_spliced_no_splice_filter', dict(seed=13), dict(keep_non_spliced_read=True, min_dist_from_splice=0)),
    ('noisy_window_filter', dict(seed=14, n_reads=80, err=0.05),
     dict(mismatch_window_size=60, max_window_mismatch=4, max_window_mismatch_type=2, min_allele_depth=2)),
    ('lowercase_genome_repeats', dict(seed=15, lower_case_ref=True),
     dict(simple_repeat_intervals=[[100, 140], [400, 430]], homopoly_length=4)),
    ('strand_override', dict(seed=16, n_reads=90), dict(strand_override=True, min_allele_ratio=0.05)),
    ('strict_filters', dict(seed=17, n_reads=120), dict(min_allele_depth=5, min_total_depth=20, min_het_snp_ratio=0.4,
                                                         max_het_snp_ratio=0.6, min_common_reads=10)),
]


def frame_json(df):
    """full-precision records (DataFrame.to_json would round floats to 10 digits); NaN stays NaN"""
    def py(v):
        if isinstance(v, (np.integer,)):
            return int(v)
        if isinstance(v, (np.floating,)):
            return float(v)
        return v
    return {'columns': list(df.columns), 'data': [[py(v) for v in row] for row in df.values.tolist()]}


def region_cases():
    import hashlib
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from fakes import FakeGenome, FakeSam, simulate_region
    ref_mm = load_ref_mismatch()
    out = []
    for name, sim_kw, kw in REGION_CASES:
        reads, genome, snps, (a, b) = simulate_region(**sim_kw)
        kw = dict(kw)
        strand_dict = None
        if kw.pop('strand_override', False):
            strand_dict = {r.query_name: '+' for r in reads[::3]}
        dfs = ref_mm.region_mismatch_analysis('chrS', a, b, FakeSam(reads), FakeGenome(genome), snp_positions=snps,
                                              read_strand_dict=strand_dict, **kw)
        digest = hashlib.sha256(('|'.join(r.query_name + r._cs for r in reads) + genome).encode()).hexdigest()
        out.append({'name': name, 'sim': sim_kw, 'kwargs': {k: v for k, v in kw.items()},
                    'strand_override': strand_dict is not None, 'input_sha256': digest,
                    'mismatch': frame_json(dfs[0]), 'pair_mi': frame_json(dfs[1]), 'removed': frame_json(dfs[2])})
    return out


# the two contigs of the CLI end-to-end test (tests/test_cli.py: regions_fixture) — name, simulate_region kwargs
CLI_REGIONS = [('chrA', dict(seed=200, n_reads=70)), ('chrB', dict(seed=201, n_reads=100))]
# the CLI's defaults (src/giremi/script/giremi.py:140-321 as passed on at :62-81)
CLI_KWARGS = dict(keep_non_spliced_read=False, min_dist_from_splice=4, min_allele_depth=3, min_allele_ratio=0.05,
                  min_total_depth=2, homopoly_length=5, min_het_snp_ratio=0.35, max_het_snp_ratio=0.65,
                  mismatch_window_size=100, max_window_mismatch=10, max_window_mismatch_type=3, min_common_reads=6,
                  mode='cs', read_strand_dict=None)


# BASELINE.json configs[0]: "Synthetic 1-chrom BAM, 500 sites x 2k reads, --mi_calculation_only": one stretched gene,
# (10.8 kb), 2,000 spliced reads (half per strand), 50 SNP positions and 420 editing sites: ~350 sites per strand block
CFG1_REGION = ('chrC', dict(seed=300, n_reads=2000, n_snps=50, n_edits=420, err=0.0001, scale=12))


def cli_cfg1_case():
    """the reference on the cfg1-sized footprint (same construction as cli_cases); the site table is left out"""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from fakes import FakeGenome, FakeSamSkips, simulate_region
    ref_mm = load_ref_mismatch()
    contig, sim_kw = CFG1_REGION
    reads, genome, snps, _ = simulate_region(**sim_kw)
    lo, hi = min(r.reference_start for r in reads), max(r.reference_end for r in reads)
    snp_in = sorted(p for p in snps if lo <= p < hi)
    t0 = time.perf_counter()
    dfs = ref_mm.region_mismatch_analysis(contig, lo, hi, FakeSamSkips(reads), FakeGenome(genome),
                                          simple_repeat_intervals=[], snp_positions=snp_in, **CLI_KWARGS)
    dt = time.perf_counter() - t0
    return {'contig': contig, 'sim': sim_kw, 'footprint': [lo, hi], 'pair_mi': frame_json(dfs[1]),
            'removed': frame_json(dfs[2]), 'n_sites': int(len(dfs[0])), 'reference_wall_s': dt}


def cli_cases():
    """what the reference's footprint_bulk_calculation (script/giremi.py:20-93) computes for the CLI test's two
    footprints: region_mismatch_analysis with the CLI's defaults, on a pysam-like view of the same reads (pile-up
    with empty strings inside introns, untruncated columns; footprint = [first read start, last read end))"""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from fakes import FakeGenome, FakeSamSkips, simulate_region
    ref_mm = load_ref_mismatch()
    out = []
    for contig, sim_kw in CLI_REGIONS:
        reads, genome, snps, _ = simulate_region(**sim_kw)
        lo, hi = min(r.reference_start for r in reads), max(r.reference_end for r in reads)
        snp_in = sorted(p for p in snps if lo <= p < hi)
        dfs = ref_mm.region_mismatch_analysis(contig, lo, hi, FakeSamSkips(reads), FakeGenome(genome),
                                              simple_repeat_intervals=[], snp_positions=snp_in, **CLI_KWARGS)
        out.append({'contig': contig, 'sim': sim_kw, 'footprint': [lo, hi], 'pair_mi': frame_json(dfs[1]),
                    'removed': frame_json(dfs[2]), 'mismatch': frame_json(dfs[0])})
    return out


def cli_mip_cases(with_cfg1=True):
    """the mismatch table with the reference's own p-value column `mip` as script/giremi.py:415-429 builds it just before
    the GLM: region_mismatch_analysis per footprint, frames concatenated in footprint order (:79-88, :381-394), then
    miecdf = stat.ecdf(mean_mi of the het_snp rows that have one) applied to every row's mean_mi (NaN stays NaN).
    The ten lines of main() that do this cannot be imported (script/giremi.py needs pysam): they are restated here
    around the reference's OWN stat.ecdf and mismatch.region_mismatch_analysis.  Round 3."""
    import pandas as pd
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from fakes import FakeGenome, FakeSamSkips, simulate_region
    ref_mm = load_ref_mismatch()

    def run(regions):
        frames = []
        for contig, sim_kw in regions:
            reads, genome, snps, _ = simulate_region(**sim_kw)
            lo, hi = min(r.reference_start for r in reads), max(r.reference_end for r in reads)
            snp_in = sorted(p for p in snps if lo <= p < hi)
            dfs = ref_mm.region_mismatch_analysis(contig, lo, hi, FakeSamSkips(reads), FakeGenome(genome),
                                                  simple_repeat_intervals=[], snp_positions=snp_in, **CLI_KWARGS)
            frames.append(dfs[0])
        df = pd.concat(frames, axis=0)
        df.loc[:, 'mip'] = np.nan
        if df['mean_mi'].notna().sum() > 0:
            miecdf = ref_stat.ecdf(df.loc[df['mean_mi'].notna() & (df['type'] == 'het_snp'), 'mean_mi'])
            df.loc[:, 'mip'] = df.apply(lambda a: miecdf(a['mean_mi']) if not np.isnan(a['mean_mi']) else np.nan, axis=1)
        return frame_json(df)
    out = {'cli': {'regions': [c for c, _k in CLI_REGIONS], 'mismatch_mip': run(CLI_REGIONS)}}
    if with_cfg1:
        out['cfg1'] = {'regions': [CFG1_REGION[0]], 'mismatch_mip': run([CFG1_REGION])}
    return out


def splice_tables(seed, n_reads, n_sites, n_splice, dup=False):
    """synthetic read-site / read-splice tables for calculate_site_splice_mi.py"""
    import pandas as pd
    rng = np.random.default_rng(seed)
    reads = ['rd%04d' % i for i in range(n_reads)]
    rows = []
    for c in ('chr1', 'chr2'):
        for s in range(n_sites):
            pos = 100 + 37 * s
            for r in reads:
                if rng.random() < 0.5:
                    rows.append([r, c, pos, str(rng.choice(['A', 'G', 'T'], p=[0.5, 0.4, 0.1]))])
                    if dup and rng.random() < 0.05:
                        rows.append([r, c, pos, str(rng.choice(['A', 'G']))])
    site = pd.DataFrame(rows, columns=['read_name', 'chromosome', 'pos', 'seq'])
    rows = []
    for c in ('chr1', 'chr2'):
        for r in reads:
            for k in range(n_splice):
                if rng.random() < 0.4:
                    rows.append([r, c, 1000 + k * 50 + int(rng.integers(0, 3)), 'l', 1000 + k * 50, 'annot'])
    splice = pd.DataFrame(rows, columns=['read_name', 'chromosome', 'pos', 'type', 'corrected_pos', 'annotation'])
    return site, splice


def splice_cases():
    """the reference utility is a script (argparse + main): run it with runpy on TSV files"""
    import runpy
    import tempfile
    import warnings
    import pandas as pd
    warnings.simplefilter('ignore')
    cases = []
    for name, kw in (('small', dict(seed=1, n_reads=30, n_sites=3, n_splice=2)),
                     ('dups', dict(seed=2, n_reads=50, n_sites=4, n_splice=3, dup=True)),
                     ('two_chunks', dict(seed=3, n_reads=700, n_sites=2, n_splice=20))):
        site, splice = splice_tables(**kw)
        d = tempfile.mkdtemp()
        site.to_csv(d + '/site.tsv', sep='\t', index=False)
        splice.to_csv(d + '/splice.tsv', sep='\t', index=False)
        argv, sys.argv = sys.argv, ['x', '-m', d + '/site.tsv', '-s', d + '/splice.tsv', '-o', d + '/out']
        try:
            runpy.run_path(os.path.join(REF, 'script', 'calculate_site_splice_mi.py'), run_name='__main__')
        finally:
            sys.argv = argv
        out = pd.read_table(d + '/out.site_splice_pair', sep='\t')
        cases.append({'name': name, 'site': site.values.tolist(), 'splice': splice.values.tolist(),
                      'out': {'columns': list(out.columns), 'data': out.values.tolist()}})
    return cases


def main():
    import sklearn
    import scipy
    meta = {'reference': 'gxiaolab/L-GIREMI v0.2.4 imported by file path',
            'sklearn': sklearn.__version__, 'numpy': np.__version__, 'scipy': scipy.__version__}
    if '--time-large' in sys.argv:      # added in round 2
        time_reference_large(os.path.join(HERE, 'reference_timing.json'))
        return
    if '--only-cli-cfg1' in sys.argv:   # added in round 2 (several minutes: the reference examines ~1e5 pairs at depth 2,000)
        with open(os.path.join(HERE, 'cli_cfg1.json'), 'w') as f:
            json.dump({'meta': meta, 'case': cli_cfg1_case()}, f)
        return
    if '--only-cli-mip' in sys.argv:    # added in round 3 (the cfg1 part takes the reference several minutes)
        with open(os.path.join(HERE, 'cli_mip.json'), 'w') as f:
            json.dump({'meta': meta, 'cases': cli_mip_cases('--no-cfg1' not in sys.argv)}, f)
        return
    if '--only-cli' in sys.argv:        # added in round 2: leaves the round-1 fixtures byte-for-byte as they are
        with open(os.path.join(HERE, 'cli.json'), 'w') as f:
            json.dump({'meta': meta, 'cases': cli_cases()}, f)
        return
    with open(os.path.join(HERE, 'pairs_edge.json'), 'w') as f:
        json.dump({'meta': meta, 'cases': edge_cases()}, f)
    with open(os.path.join(HERE, 'pairs_random.json'), 'w') as f:
        json.dump({'meta': meta, 'cases': random_cases()}, f)
    mm = banded_block(20250809, 500, 2000)
    with open(os.path.join(HERE, 'pairs_banded_cfg1.json'), 'w') as f:
        json.dump({'meta': meta, 'cases': [case('cfg1_banded_500x2000', mm, 6)]}, f)
    with open(os.path.join(HERE, 'ecdf.json'), 'w') as f:
        json.dump({'meta': meta, 'cases': ecdf_cases()}, f)
    with open(os.path.join(HERE, 'splice.json'), 'w') as f:
        json.dump({'meta': meta, 'cases': splice_cases()}, f)
    with open(os.path.join(HERE, 'region.json'), 'w') as f:
        json.dump({'meta': meta, 'cases': region_cases()}, f)
    with open(os.path.join(HERE, 'cli.json'), 'w') as f:
        json.dump({'meta': meta, 'cases': cli_cases()}, f)
    if '--time' in sys.argv:
        time_reference(os.path.join(HERE, 'reference_timing.json'))


if __name__ == '__main__':
    main()
