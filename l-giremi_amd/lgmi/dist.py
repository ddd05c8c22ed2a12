"""Host-side helpers of the multi-GPU path (one process per GPU).

Site pairs never cross a (footprint, strand) block (src/giremi/mismatch.py:387-391;
blocks come from footprints, src/giremi/script/giremi.py:32,60), so blocks are dealt
to ranks and nothing is exchanged while computing; the reference's own parallelism is
the same shape (mp.Pool.map over footprint chunks, script/giremi.py:375-380).  The
only communication is the final gather.  `dist` is a torch.distributed-like module
(gloo on CPU in the tests, nccl = RCCL on the GPUs)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np


def block_costs(block_site_begin: Sequence[int], block_n_reads: Sequence[int], het_counts: Sequence[int]) -> np.ndarray:
    """examined pairs x words: H*(P-H) + H*(H-1)/2 pairs of ceil(R/64) words each"""
    bsb = np.asarray(block_site_begin, np.int64)
    P = bsb[1:] - bsb[:-1]
    H = np.asarray(het_counts, np.int64)
    W = (np.asarray(block_n_reads, np.int64) + 63) // 64
    return (H * (P - H) + H * (H - 1) // 2) * np.maximum(W, 1)


def shard_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """longest-processing-time-first: heaviest block to the least loaded rank.
    Deterministic (ties by block index), so every rank computes the same plan."""
    order = sorted(range(len(costs)), key=lambda b: (-float(costs[b]), b))
    load = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for b in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(b)
        load[r] += float(costs[b])
    for s in shards:
        s.sort()            # keep the reference's block (footprint) order inside a rank
    return shards


def exchange_unique_id(dist, make_id: Optional[Callable[[], bytes]], src: int = 0) -> bytes:
    """rank `src` calls make_id(); everybody returns the same 128 bytes"""
    box = [make_id() if make_id is not None else None]
    dist.broadcast_object_list(box, src=src)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError('unique id exchange failed')
    return bytes(uid)


def gather_tables_host(dist, table: dict, root: int = 0):
    """host-side gather of per-rank row tables (dict of equal-length numpy arrays) in
    rank order — what writing one .mi.txt on rank 0 needs when rows already sit in
    host memory.  Returns the concatenation on root, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    box = [None] * world if rank == root else None
    dist.gather_object(table, box, dst=root)
    if rank != root:
        return None
    keys = list(table.keys())
    return {k: np.concatenate([np.asarray(t[k]) for t in box]) for k in keys}
