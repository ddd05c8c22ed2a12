"""Synthetic packed batches for parity tests (numpy, deterministic)."""
import numpy as np

from lgmi.pack import PackedBatch

NAMES = ['mismatch', 'snp', 'het_snp']


def pack_class_matrix(blocks):
    """blocks: list of (pos[int64 P], type[uint8 P], cls[int8 P x R]) with cls -1 = not covered,
    0/1/2 = class.  Each site's band is the minimal word range holding its covered reads."""
    bsb, nreads, pos, typ, woff, nwords, poff, chunks = [0], [], [], [], [], [], [], []
    total = 0
    for p, t, cls in blocks:
        P, R = cls.shape
        W = (R + 63) // 64
        pad = np.full((P, W * 64), -1, np.int8)
        pad[:, :R] = cls
        lo_bits = (pad == 1) | (pad == 0)
        hi_bits = (pad == 2) | (pad == 0)
        weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))
        lo = (lo_bits.reshape(P, W, 64).astype(np.uint64) * weights).sum(axis=2, dtype=np.uint64)
        hi = (hi_bits.reshape(P, W, 64).astype(np.uint64) * weights).sum(axis=2, dtype=np.uint64)
        for s in range(P):
            nz = np.nonzero(lo[s] | hi[s])[0]
            w0, w1 = (int(nz[0]), int(nz[-1]) + 1) if nz.size else (0, 0)
            woff.append(w0)
            nwords.append(w1 - w0)
            poff.append(total)
            chunks.append(lo[s, w0:w1])
            chunks.append(hi[s, w0:w1])
            total += 2 * (w1 - w0)
        pos.extend(int(x) for x in p)
        typ.extend(int(x) for x in t)
        bsb.append(len(pos))
        nreads.append(R)
    planes = np.concatenate(chunks) if chunks else np.zeros(0, np.uint64)
    typ = np.asarray(typ, np.uint8)
    return PackedBatch(np.asarray(bsb, np.uint64), np.asarray(nreads, np.uint32), np.asarray(pos, np.int64), typ,
                       np.asarray(woff, np.uint32), np.asarray(nwords, np.uint32), np.asarray(poff, np.uint64),
                       np.ascontiguousarray(planes, np.uint64), [NAMES[t] for t in typ], np.zeros(len(typ), bool))


def random_block(rng, P, R, banded=False, tri_frac=0.1, het_frac=0.25, cover=0.8, mean_span=12):
    pos = 1000 + np.cumsum(rng.integers(1, 50, P))
    typ = np.where(rng.random(P) < het_frac, 2, np.where(rng.random(P) < 0.1, 1, 0)).astype(np.uint8)
    hap = rng.integers(0, 2, R)
    cls = np.full((P, R), -1, np.int8)
    if banded:
        start = np.sort(rng.integers(0, P, R))
        span = 1 + rng.geometric(1.0 / mean_span, R)
    for s in range(P):
        if banded:
            cov = (start <= s) & (s < start + span) & (rng.random(R) < 0.9)
        else:
            cov = rng.random(R) < cover
        if typ[s] == 2:
            a = hap ^ (rng.random(R) < 0.05)
        else:
            a = (rng.random(R) < rng.uniform(0.05, 0.5)).astype(int)
        c = np.where(a == 1, 1, 2).astype(np.int8)
        if rng.random() < 0.5:
            c = 3 - c                                  # either allele can be the major one
        if rng.random() < tri_frac:
            c = np.where(rng.random(R) < 0.12, 0, c)
        cls[s] = np.where(cov, c, -1)
    return pos, typ, cls


def random_batch(seed, n_blocks=3, P=(2, 90), R=(6, 700), banded=None, tri_frac=0.1):
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks = []
    for b in range(n_blocks):
        p = int(rng.integers(P[0], P[1] + 1))
        r = int(rng.integers(R[0], R[1] + 1))
        bd = bool(rng.integers(0, 2)) if banded is None else banded
        blocks.append(random_block(rng, p, r, banded=bd, tri_frac=tri_frac))
    return pack_class_matrix(blocks)
