"""Host-side synthetic chromosomes in the BANDED (long-read-like) regime of SURVEY §8d:
reads sorted by start, each spanning a short window of sites, one (footprint, strand)
block per gene.  numpy only; the dense regime is generated on the device
(lgmi_synth_dense)."""
from __future__ import annotations

import numpy as np

from . import _lib
from .pack import PackedBatch

NAMES = ['mismatch', 'snp', 'het_snp']


def banded_chromosome(n_sites: int, n_reads: int, seed: int = 20250809, mean_span: int = 20,
                      gene_sites: int = 2000, dropout: float = 0.10) -> PackedBatch:
    """P sites at pos 10000 + 37 s; read r covers the window [start_r, start_r + span_r) of its gene
    (span ~ 1 + Geometric(1/mean_span)); each covered site dropped with prob. `dropout`; every 5th
    site het_snp (allele = haplotype xor Bern(0.02)), the others Bern(e_s), e_s ~ U(0.05, 0.5), 1 % typed
    snp; 2 % of the sites carry a third allele at 5 %; alleles with depth < 3 or ratio < 0.05 are removed
    and sites left with < 2 alleles dropped (mirrors src/giremi/mismatch.py:242-282).  One block per
    gene of `gene_sites` sites (strands alternate, reads never cross a gene)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P, R = int(n_sites), int(n_reads)
    start = np.sort(rng.integers(0, P, R))
    span = 1 + rng.geometric(1.0 / mean_span, R)
    gene = start // gene_sites
    end = np.minimum(np.minimum(start + span, (gene + 1) * gene_sites), P)
    length = end - start
    hap = rng.integers(0, 2, R).astype(np.int8)
    # incidences (read, site)
    tot = int(length.sum())
    r_idx = np.repeat(np.arange(R), length)
    first = np.cumsum(length) - length
    s_idx = np.arange(tot) - np.repeat(first, length) + np.repeat(start, length)
    keep = rng.random(tot) >= dropout
    r_idx, s_idx = r_idx[keep], s_idx[keep]
    tot = len(r_idx)
    is_het = (np.arange(P) % 5 == 0)
    e_s = rng.uniform(0.05, 0.5, P)
    is_tri = rng.random(P) < 0.02
    is_snp = (~is_het) & (rng.random(P) < 0.01)
    u = rng.random(tot)
    allele = np.where(is_het[s_idx], hap[r_idx] ^ (u < 0.02), (u < e_s[s_idx])).astype(np.int8)
    third = is_tri[s_idx] & (rng.random(tot) < 0.05)
    allele[third] = 2
    # allele filters, then class by depth rank (stable, insertion order 0,1,2)
    depth = np.bincount(s_idx * 3 + allele, minlength=3 * P).reshape(P, 3)
    total = depth.sum(axis=1)
    ok = (depth >= 3) & (depth / np.maximum(total, 1)[:, None] >= 0.05)
    depth_f = np.where(ok, depth, 0)
    site_ok = ok.sum(axis=1) >= 2
    inc_ok = ok[s_idx, allele] & site_ok[s_idx]
    r_idx, s_idx, allele = r_idx[inc_ok], s_idx[inc_ok], allele[inc_ok]
    order = np.argsort(-depth_f, axis=1, kind='stable')           # rank 0 = major
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(3)[None, :].repeat(P, 0), axis=1)
    cls = np.array([2, 1, 0], np.int8)[rank[s_idx, allele]]
    # blocks = genes; reads numbered inside their gene in start order
    n_genes = (P + gene_sites - 1) // gene_sites
    gene_first_read = np.searchsorted(gene, np.arange(n_genes))
    gene_n_reads = np.diff(np.append(gene_first_read, R))
    local_r = r_idx - gene_first_read[gene[r_idx]]
    kept_sites = np.nonzero(site_ok)[0]
    new_id = np.full(P, -1, np.int64)
    new_id[kept_sites] = np.arange(len(kept_sites))
    sid = new_id[s_idx]
    word = local_r >> 6
    ns = len(kept_sites)
    w0 = np.full(ns, np.iinfo(np.int64).max)
    w1 = np.zeros(ns, np.int64)
    np.minimum.at(w0, sid, word)
    np.maximum.at(w1, sid, word + 1)
    nw = w1 - w0
    poff = 2 * (np.cumsum(nw) - nw)
    planes = np.zeros(int(2 * nw.sum()), np.uint64)
    bit = np.left_shift(np.uint64(1), (local_r & 63).astype(np.uint64))
    lo_at = poff[sid] + (word - w0[sid])
    hi_at = lo_at + nw[sid]
    m_lo, m_hi = cls != 2, cls != 1
    np.bitwise_or.at(planes, lo_at[m_lo], bit[m_lo])
    np.bitwise_or.at(planes, hi_at[m_hi], bit[m_hi])
    site_gene = kept_sites // gene_sites
    bsb = np.searchsorted(site_gene, np.arange(n_genes + 1)).astype(np.uint64)
    typ = np.where(is_het[kept_sites], _lib.TYPE_HET_SNP,
                   np.where(is_snp[kept_sites], _lib.TYPE_SNP, _lib.TYPE_MISMATCH)).astype(np.uint8)
    return PackedBatch(bsb, gene_n_reads.astype(np.uint32), (10_000 + 37 * kept_sites).astype(np.int64), typ,
                       w0.astype(np.uint32), nw.astype(np.uint32), poff.astype(np.uint64), planes,
                       [NAMES[t] for t in typ], np.zeros(ns, bool))


def _footprint_chunk(rng, block_P, block_R, site_id0, dropout):
    """one chunk of footprint blocks -> PackedBatch (the pipeline of banded_chromosome with blocks of their own sizes)"""
    nb = len(block_P)
    sb = np.concatenate([[0], np.cumsum(block_P)]).astype(np.int64)          # first site of each block
    rb = np.concatenate([[0], np.cumsum(block_R)]).astype(np.int64)          # first read of each block
    P, R = int(sb[-1]), int(rb[-1])
    rblock = np.repeat(np.arange(nb), block_R)
    # reads sorted by start inside their block (long reads sorted by position: a site's covering reads are a narrow
    # index range, which is what the band storage relies on)
    start_in = np.floor(rng.random(R) * block_P[rblock]).astype(np.int64)
    order = np.lexsort((start_in, rblock))
    start_in = start_in[order]
    mean_span = np.maximum(2.0, block_P[rblock] / 2.0)
    span = 1 + rng.geometric(1.0 / mean_span)
    start = sb[rblock] + start_in
    end = np.minimum(start + span, sb[rblock + 1])
    length = end - start
    hap = rng.integers(0, 2, R).astype(np.int8)
    tot = int(length.sum())
    r_idx = np.repeat(np.arange(R), length)
    first = np.cumsum(length) - length
    s_idx = np.arange(tot) - np.repeat(first, length) + np.repeat(start, length)
    keep = rng.random(tot) >= dropout
    r_idx, s_idx = r_idx[keep], s_idx[keep]
    tot = len(r_idx)
    site_in_block = np.arange(P) - np.repeat(sb[:-1], block_P)
    is_het = (site_in_block % 5 == 0)                                        # every 5th site of a footprint
    e_s = rng.uniform(0.05, 0.5, P)
    is_tri = rng.random(P) < 0.02
    is_snp = (~is_het) & (rng.random(P) < 0.01)
    u = rng.random(tot)
    allele = np.where(is_het[s_idx], hap[r_idx] ^ (u < 0.02), (u < e_s[s_idx])).astype(np.int8)
    third = is_tri[s_idx] & (rng.random(tot) < 0.05)
    allele[third] = 2
    depth = np.bincount(s_idx * 3 + allele, minlength=3 * P).reshape(P, 3)
    total = depth.sum(axis=1)
    ok = (depth >= 3) & (depth / np.maximum(total, 1)[:, None] >= 0.05)
    depth_f = np.where(ok, depth, 0)
    site_ok = ok.sum(axis=1) >= 2
    inc_ok = ok[s_idx, allele] & site_ok[s_idx]
    r_idx, s_idx, allele = r_idx[inc_ok], s_idx[inc_ok], allele[inc_ok]
    rank_order = np.argsort(-depth_f, axis=1, kind='stable')
    rank = np.empty_like(rank_order)
    np.put_along_axis(rank, rank_order, np.arange(3)[None, :].repeat(P, 0), axis=1)
    cls = np.array([2, 1, 0], np.int8)[rank[s_idx, allele]]
    local_r = r_idx - rb[rblock[r_idx]]
    kept_sites = np.nonzero(site_ok)[0]
    new_id = np.full(P, -1, np.int64)
    new_id[kept_sites] = np.arange(len(kept_sites))
    sid = new_id[s_idx]
    word = local_r >> 6
    ns = len(kept_sites)
    # per-site band and the OR of the bits of every plane word, by sorting (ufunc.at is an order of magnitude slower)
    so = np.argsort(sid, kind='stable')
    sid_s, word_s = sid[so], word[so]
    cuts = np.flatnonzero(np.diff(sid_s)) + 1
    starts = np.concatenate([[0], cuts])
    present = sid_s[starts]
    w0 = np.zeros(ns, np.int64)
    w1 = np.zeros(ns, np.int64)
    w0[present] = np.minimum.reduceat(word_s, starts)
    w1[present] = np.maximum.reduceat(word_s, starts) + 1
    nw = w1 - w0
    poff = 2 * (np.cumsum(nw) - nw)
    planes = np.zeros(int(2 * nw.sum()), np.uint64)
    bit = np.left_shift(np.uint64(1), (local_r & 63).astype(np.uint64))
    lo_at = poff[sid] + (word - w0[sid])
    hi_at = lo_at + nw[sid]
    for at, m in ((lo_at, cls != 2), (hi_at, cls != 1)):
        k, v = at[m], bit[m]
        if len(k):
            o = np.argsort(k, kind='stable')
            k, v = k[o], v[o]
            st = np.concatenate([[0], np.flatnonzero(np.diff(k)) + 1])
            planes[k[st]] |= np.bitwise_or.reduceat(v, st)
    site_block = np.searchsorted(sb, kept_sites, side='right') - 1
    bsb = np.searchsorted(site_block, np.arange(nb + 1)).astype(np.uint64)
    typ = np.where(is_het[kept_sites], _lib.TYPE_HET_SNP,
                   np.where(is_snp[kept_sites], _lib.TYPE_SNP, _lib.TYPE_MISMATCH)).astype(np.uint8)
    return PackedBatch(bsb, np.asarray(block_R, np.uint32), (10_000 + 37 * (site_id0 + kept_sites)).astype(np.int64), typ,
                       w0.astype(np.uint32), nw.astype(np.uint32), poff.astype(np.uint64), planes,
                       [NAMES[t] for t in typ], np.zeros(ns, bool))


def _footprint_job(job):
    seed_seq, bp, br, site_id0, dropout = job
    return _footprint_chunk(np.random.Generator(np.random.PCG64(seed_seq)), bp, br, site_id0, dropout)


def footprint_blocks(n_blocks: int = 20_000, seed: int = 20250810, sites=(5, 120), reads=(50, 3000),
                     dropout: float = 0.10, chunk: int = 250, reads_dist: str = 'loguniform', workers: int = 0,
                     cache_dir: str = None) -> PackedBatch:
    """The shape real L-GIREMI input has (src/giremi/footprint.py:6-28, script/giremi.py:32,60-78): tens of thousands of
    small (footprint, strand) blocks — P ~ U(sites) sites x R reads each (R log-uniform over `reads` by default: most
    footprints are shallow, a few are deep; 'uniform' for U(reads)), reads sorted by start and spanning
    1 + Geometric(mean P / 2) consecutive sites of their footprint, 10 % dropout, every 5th site of a footprint a
    haplotype-linked het SNP, the others independent mismatches (1 % typed snp), 2 % of the sites with a third allele at
    5 %, the reference's allele filters applied (banded_chromosome's site model).  Deterministic: block sizes come from
    PCG64(seed), chunk c of `chunk` blocks from the c-th child of SeedSequence(seed) — so the chunks can be built by
    `workers` processes (0: one per core, at most 16) and the result does not depend on how many.  `cache_dir`: keep the
    packed batch there as an .npz keyed by every argument (a bench session builds it once)."""
    import os
    from .pack import concat_batches
    key = 'lgmi_footprints_%d_%d_%d-%d_%d-%d_%g_%d_%s.npz' % (n_blocks, seed, sites[0], sites[1], reads[0], reads[1],
                                                               dropout, chunk, reads_dist)
    path = os.path.join(cache_dir, key) if cache_dir else None
    fields = ('block_site_begin', 'block_n_reads', 'site_pos', 'site_type', 'site_word_off', 'site_n_words',
              'site_plane_off', 'planes')
    if path and os.path.exists(path):
        with np.load(path) as z:
            arr = [z[f] for f in fields]
        return PackedBatch(*arr, [NAMES[t] for t in arr[3]], np.zeros(len(arr[3]), bool))
    rng = np.random.Generator(np.random.PCG64(seed))
    block_P = rng.integers(sites[0], sites[1] + 1, n_blocks)
    if reads_dist == 'uniform':
        block_R = rng.integers(reads[0], reads[1] + 1, n_blocks)
    else:
        block_R = np.exp(rng.uniform(np.log(reads[0]), np.log(reads[1] + 1), n_blocks)).astype(np.int64)
    starts = list(range(0, n_blocks, chunk))
    seeds = np.random.SeedSequence(seed).spawn(len(starts))
    site0 = np.concatenate([[0], np.cumsum(block_P)])
    jobs = [(seeds[c], block_P[b0:b0 + chunk], block_R[b0:b0 + chunk], int(site0[b0]), dropout) for c, b0 in enumerate(starts)]
    n_workers = workers or min(16, os.cpu_count() or 1)
    if n_workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context('fork').Pool(min(n_workers, len(jobs))) as pool:      # (before any HIP context exists: the bench builds the batch first)
            parts = pool.map(_footprint_job, jobs)
    else:
        parts = [_footprint_job(j) for j in jobs]
    pb = concat_batches(parts)
    if path:
        tmp = path + '.%d.tmp.npz' % os.getpid()
        np.savez(tmp, **{f: getattr(pb, f) for f in fields})
        os.replace(tmp, path)
    return pb
