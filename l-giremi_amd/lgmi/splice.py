"""Site x splice-site mutual information on the GPU.

Drop-in for the computation of ``l-giremi``'s ``calculate_site_splice_mi`` utility
(src/giremi/script/calculate_site_splice_mi.py:34-130): for every (site, allele, splice
position) that at least one read supports, the MI between "read carries the allele at the
site" and "read uses the splice position", over the reads that cover the site.  The
reference evaluates sklearn's ``mutual_info_score`` once per pandas row (:89-125); here
every site becomes one block of the packed batch — its reads are the universe, every allele
and every candidate splice position a column — and all pairs are counted in one launch.
"""
from __future__ import annotations

from collections import Counter, defaultdict
from typing import Optional

import numpy as np
import pandas as pd

from . import _lib
from .engine import Engine, default_engine
from .pack import PackedBatch


def site_splice_pairs(rsite: pd.DataFrame, rsplice: pd.DataFrame, chunksize: int = 10000) -> pd.DataFrame:
    """(chromosome, site_pos, seq, splice_pos, count) in the reference's order: the splice table is
    merged chunk by chunk (:50-72), a chunk contributes its pairs in groupby (sorted) order and a pair
    keeps the place of its first appearance; all key columns come out as strings (:79-85)."""
    seen: Counter = Counter()
    for lo in range(0, len(rsplice), chunksize):
        ck = rsplice.iloc[lo:lo + chunksize]
        m = pd.merge(rsite, ck[['read_name', 'chromosome', 'corrected_pos']], how='inner',
                     on=['read_name', 'chromosome'])
        m.columns = ['read_name', 'chromosome', 'site_pos', 'seq', 'splice_pos']
        counts = m.groupby(['chromosome', 'site_pos', 'seq', 'splice_pos'])['read_name'].count().reset_index()
        for row in counts.itertuples(index=False):
            seen['\t'.join(str(v) for v in row[:4])] += row[4]
    keys = list(seen.keys())
    cols = [k.split('\t') for k in keys]
    out = pd.DataFrame(cols, columns=['chromosome', 'site_pos', 'seq', 'splice_pos']) if cols else \
        pd.DataFrame(columns=['chromosome', 'site_pos', 'seq', 'splice_pos'])
    out['count'] = [seen[k] for k in keys]
    return out


def site_splice_mi(rsite: pd.DataFrame, rsplice: pd.DataFrame, chunksize: int = 10000,
                   engine: Optional[Engine] = None) -> pd.DataFrame:
    """-> DataFrame [chromosome, site_pos, seq, splice_pos, count, mi] (what the reference writes to
    PREFIX.site_splice_pair).  ``rsite``: read_name, chromosome, pos, seq; ``rsplice``: read_name,
    chromosome, pos, type, corrected_pos, annotation."""
    pairs = site_splice_pairs(rsite, rsplice, chunksize)
    if len(pairs) == 0:
        pairs['mi'] = []
        return pairs
    # reads of every site, per allele, in table order (:88-93); duplicates are kept — the reference's
    # universe is the concatenation of the allele lists (:117-120), so a read listed twice counts twice
    site_reads = defaultdict(lambda: defaultdict(list))
    for name, chrom, pos, seq in zip(rsite['read_name'], rsite['chromosome'], rsite['pos'], rsite['seq']):
        site_reads['%s:%s' % (chrom, pos)][seq].append(name)
    splice_reads = defaultdict(set)
    for name, chrom, cpos in zip(rsplice['read_name'], rsplice['chromosome'], rsplice['corrected_pos']):
        splice_reads['%s:%s' % (chrom, cpos)].add(name)

    # one block per site that occurs in `pairs`: columns = its alleles (typed het_snp so that every
    # allele x splice pair is "het-involved") followed by the splice positions paired with it
    by_site = defaultdict(lambda: ([], []))            # site label -> (alleles, splice labels), first-seen order
    for chrom, spos, seq, sp in zip(pairs['chromosome'], pairs['site_pos'], pairs['seq'], pairs['splice_pos']):
        alleles, splices = by_site['%s:%s' % (chrom, spos)]
        if seq not in alleles:
            alleles.append(seq)
        lab = '%s:%s' % (chrom, sp)
        if lab not in splices:
            splices.append(lab)
    bsb, n_reads, typ, woff, nwords, poff, chunks = [0], [], [], [], [], [], []
    col_index = {}
    total = 0
    one = np.uint64(1)
    for label, (alleles, splices) in by_site.items():
        lists = site_reads[label]
        universe = [n for seq in lists for n in lists[seq]]            # with duplicates
        R = len(universe)
        W = (R + 63) // 64
        idx = np.arange(R)
        word, bit = idx >> 6, np.left_shift(one, (idx & 63).astype(np.uint64))
        base = bsb[-1]
        members = [set(lists[str_seq]) if str_seq in lists else set(_lookup(lists, str_seq)) for str_seq in alleles]
        members += [splice_reads.get(lab, set()) for lab in splices]
        for k, mem in enumerate(members):
            has = np.fromiter((n in mem for n in universe), bool, R)
            lo = np.zeros(W, np.uint64)
            hi = np.zeros(W, np.uint64)
            np.bitwise_or.at(hi, word[has], bit[has])                   # class 2: in the set
            np.bitwise_or.at(lo, word[~has], bit[~has])                 # class 1: not in the set
            typ.append(_lib.TYPE_HET_SNP if k < len(alleles) else _lib.TYPE_MISMATCH)
            woff.append(0)
            nwords.append(W)
            poff.append(total)
            chunks.extend((lo, hi))
            total += 2 * W
        for k, a in enumerate(alleles):
            col_index[(label, 'a', str(a))] = base + k
        for k, lab in enumerate(splices):
            col_index[(label, 's', lab)] = base + len(alleles) + k
        bsb.append(base + len(members))
        n_reads.append(R)
    ns = bsb[-1]
    pos = np.concatenate([np.arange(b - a) for a, b in zip(bsb[:-1], bsb[1:])]).astype(np.int64)
    batch = PackedBatch(np.asarray(bsb, np.uint64), np.asarray(n_reads, np.uint32), pos, np.asarray(typ, np.uint8),
                        np.asarray(woff, np.uint32), np.asarray(nwords, np.uint32), np.asarray(poff, np.uint64),
                        np.ascontiguousarray(np.concatenate(chunks), np.uint64), ['x'] * ns, np.zeros(ns, bool))
    eng = engine or default_engine()
    res = eng.run(batch, min_common=1, het_only=True)
    mi_of = {(int(i), int(j)): float(m) for i, j, m in zip(res.row_i, res.row_j, res.row_mi)}
    out_mi = []
    for chrom, spos, seq, sp in zip(pairs['chromosome'], pairs['site_pos'], pairs['seq'], pairs['splice_pos']):
        label = '%s:%s' % (chrom, spos)
        i, j = col_index[(label, 'a', str(seq))], col_index[(label, 's', '%s:%s' % (chrom, sp))]
        out_mi.append(mi_of[(i, j)])
    pairs['mi'] = out_mi
    return pairs


def _lookup(lists, key_as_str):
    """`pairs` carries allele keys as strings (they went through '\\t'.join / split); find the original key"""
    for k in lists:
        if str(k) == key_as_str:
            return lists[k]
    return []
