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
