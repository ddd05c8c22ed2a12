"""Tile-level sharding of one batch, checked on the CPU through the host-only planner of the C ABI
(lgmi_plan_shard — the same code lgmi_run_device plans with): a shard is a contiguous, cost-balanced range of the
result rows in reference order, and the count tiles it keeps cover every slot its rows read.  The rows themselves
come from the oracle; the GPU side of the same property is tests/test_gpu_shard.py."""
import numpy as np
import pytest

from oracle import c_oracle
from util_synth import pack_class_matrix, random_batch, random_block

import lgmi
from lgmi._lib import EMIT_SEG, EMIT_SEG_Q, NONE, LgmiError


def item_of_rows(pb, plan, row_i, row_j):
    """work item (site, segment) of every row -> index into the plan's item list"""
    xrow, xnext = plan['site_xrow'].astype(np.int64), plan['site_xnext'].astype(np.int64)
    i, j = row_i.astype(np.int64), row_j.astype(np.int64)
    i_is_x = xrow[i] != NONE
    q = np.where(i_is_x, j - i - 1, xnext[j] - 1 - xnext[i])              # (x rank of j = xnext[j] - 1; rows hold pseudo rows in between)
    assert (q >= 0).all()
    seg = np.where(i_is_x, q // EMIT_SEG, q // EMIT_SEG_Q)       # (other sites' rows: segments of EMIT_SEG_Q partners, round 5)
    key = {(int(s), int(g)): k for k, (s, g) in enumerate(zip(plan['item_site'], plan['item_seg']))}
    return np.array([key[(int(a), int(b))] for a, b in zip(i, seg)], np.int64)


def covered(plan, block, xr, yc):
    m = plan['tile_block'] == block
    x0, y0, e = plan['tile_x0'][m].astype(np.int64), plan['tile_y0'][m].astype(np.int64), plan['tile_edge'][m].astype(np.int64)
    return bool(((x0 <= xr) & (xr < x0 + e) & (y0 <= yc) & (yc < y0 + e)).any())


def check_shards(pb, het_only, world, min_common=1):
    ora = c_oracle.run(pb, min_common=min_common, het_only=het_only)
    whole = lgmi.plan_shard(pb, het_only, (0, 1))
    assert whole['item_begin'] == 0 and whole['item_end'] == whole['n_items_total']
    assert whole['n_examined'] == whole['n_examined_total'] == ora['n_examined']
    assert whole['n_tiles'] == whole['n_tiles_total']
    items = item_of_rows(pb, whole, ora['row_i'], ora['row_j']) if len(ora['row_i']) else np.zeros(0, np.int64)
    assert (np.diff(items) >= 0).all()                      # rows are in item order: a range of items is a range of rows
    bsb = pb.block_site_begin.astype(np.int64)
    block_of = np.searchsorted(bsb, np.arange(len(pb.site_pos)), side='right') - 1
    prev_end, examined, pieces, tiles = 0, 0, [], 0
    for r in range(world):
        pl = lgmi.plan_shard(pb, het_only, (r, world))
        assert pl['item_begin'] == prev_end and pl['item_end'] >= pl['item_begin']
        prev_end = pl['item_end']
        examined += pl['n_examined']
        tiles += pl['n_tiles']
        assert pl['n_tiles'] <= pl['n_tiles_total'] == whole['n_tiles_total']
        np.testing.assert_array_equal(pl['item_site'], whole['item_site'])
        mine = np.nonzero((items >= pl['item_begin']) & (items < pl['item_end']))[0]
        pieces.append(mine)
        xrow, ycol, prow, pcol = pl['site_xrow'], pl['site_ycol'], pl['site_prow'], pl['site_pcol']
        for k in mine:
            i, j = int(ora['row_i'][k]), int(ora['row_j'][k])
            x, y = (i, j) if xrow[i] != NONE else (j, i)
            b = int(block_of[i])
            assert covered(pl, b, int(xrow[x]), int(ycol[y])), (r, i, j)
            if prow[x] != NONE:
                assert covered(pl, b, int(prow[x]), int(ycol[y])), (r, i, j, 'pseudo row')
            if pcol[y] != NONE:
                assert covered(pl, b, int(xrow[x]), int(pcol[y])), (r, i, j, 'pseudo col')
            if prow[x] != NONE and pcol[y] != NONE:
                assert covered(pl, b, int(prow[x]), int(pcol[y])), (r, i, j, 'pseudo both')
    assert prev_end == whole['n_items_total']
    assert examined == whole['n_examined_total']
    merged = np.concatenate(pieces) if pieces else np.zeros(0, np.int64)
    np.testing.assert_array_equal(merged, np.arange(len(ora['row_i'])))     # rank order == single-rank order
    return whole, tiles


@pytest.mark.parametrize('seed', [1, 2, 3])
@pytest.mark.parametrize('world', [2, 3, 8])
@pytest.mark.parametrize('het_only', [True, False])
def test_shards_partition_rows_and_cover_their_slots(seed, world, het_only):
    pb = random_batch(100 + seed, n_blocks=4, P=(2, 70), R=(6, 300), tri_frac=0.3)
    check_shards(pb, het_only, world)


def test_one_block_with_matrix_core_tiles_and_many_shards():
    """a block large enough for 128 x 128 tiles (nx, ny >= 96, >= 32 words): boundary tiles are kept by both
    neighbours, everything else by exactly one shard"""
    rng = np.random.Generator(np.random.PCG64(77))
    pb = pack_class_matrix([random_block(rng, 420, 2100, tri_frac=0.2, het_frac=0.5)])
    whole, tiles = check_shards(pb, True, 4, min_common=5)
    assert (lgmi.plan_shard(pb, True, (0, 4))['tile_edge'] == 128).all()
    assert tiles >= whole['n_tiles_total']        # (a 4 x 5-tile block: every shard touches most tiles)


def test_tile_redundancy_at_scale():
    """planner only (no rows computed): a dense 20,000-site block cut 8 ways.  A shard's slots form an L-shaped band
    of the slot matrix; only the tiles the band boundaries cross are computed twice."""
    from lgmi.pack import PackedBatch
    P, R = 20_000, 6_400
    W = (R + 63) // 64
    typ = np.where(np.arange(P) % 5 == 0, 2, 0).astype(np.uint8)
    pb = PackedBatch(np.array([0, P], np.uint64), np.array([R], np.uint32), (1000 + 37 * np.arange(P)).astype(np.int64),
                     typ, np.zeros(P, np.uint32), np.full(P, W, np.uint32), (2 * W * np.arange(P)).astype(np.uint64),
                     np.zeros(2 * W * P, np.uint64), ['mismatch'] * P, np.zeros(P, bool))
    for world in (2, 4, 8):
        parts = [lgmi.plan_shard(pb, True, (r, world)) for r in range(world)]
        total = parts[0]['n_tiles_total']
        assert total <= sum(p['n_tiles'] for p in parts) <= 1.25 * total
        assert max(p['n_tiles'] for p in parts) <= 1.45 * total / world
        # (rows are NOT equal across the shards since round 4: a row of a non-x site reads its slot by a column walk that
        #  costs more the further it reaches — plan.cpp: COST_WALK, fitted on the GPU — so the first shards get fewer rows;
        #  the shards still partition the items, and no shard is far from its share)
        ex = [p['n_examined'] for p in parts]
        assert sum(ex) == parts[0]['n_examined_total']
        assert max(ex) <= 1.35 * sum(ex) / world and min(ex) >= 0.65 * sum(ex) / world
        assert ex == sorted(ex)                               # more rows to the later shards


def test_rows_longer_than_one_segment():
    """a site row with more than EMIT_SEG partners is cut into several work items: a shard boundary may fall
    inside a row.  9,000 sites of which 60 are covered, so the oracle stays cheap."""
    rng = np.random.Generator(np.random.PCG64(5))
    P, R = 9000, 50
    cls = np.full((P, R), -1, np.int8)
    live = np.sort(rng.choice(P, 60, replace=False))
    live[0], live[1] = 3, 11                                   # early sites: their rows span two segments
    for s in live:
        cls[s] = rng.integers(1, 3, R)
        cls[s, rng.random(R) < 0.2] = -1
    typ = np.zeros(P, np.uint8)
    typ[live[::2]] = 2
    typ[rng.choice(P, 300, replace=False)] = 2
    pb = pack_class_matrix([(1000 + 3 * np.arange(P), typ, cls)])
    for het_only in (True, False):
        whole, _ = check_shards(pb, het_only, 3, min_common=5)
        assert (whole['item_seg'] > 0).any()


def test_balance_and_degenerate_inputs():
    pb = random_batch(9, n_blocks=6, P=(20, 90), R=(100, 700))
    whole = lgmi.plan_shard(pb, True, (0, 1))
    parts = [lgmi.plan_shard(pb, True, (r, 5)) for r in range(5)]
    share = whole['n_examined_total'] / 5
    assert max(p['n_examined'] for p in parts) <= 2.5 * share + EMIT_SEG
    # more shards than work items: the extra shards are empty, nothing is lost
    tiny = random_batch(4, n_blocks=1, P=(3, 3), R=(20, 20))
    got = [lgmi.plan_shard(tiny, False, (r, 16)) for r in range(16)]
    assert sum(p['n_examined'] for p in got) == got[0]['n_examined_total'] == 3
    assert sum(p['item_end'] - p['item_begin'] for p in got) == got[0]['n_items_total']
    with pytest.raises(lgmi._lib.LgmiError):
        lgmi.plan_shard(tiny, False, (3, 3))


def _wide_batch(n_blocks=2200, sites=32, reads=64):
    """a batch large enough (>= 2^16 sites) for the validation to run on several threads: n_blocks blocks of `sites` sites
    over one word of reads, every plane word zero"""
    from lgmi.pack import PackedBatch
    ns = n_blocks * sites
    return PackedBatch(block_site_begin=np.arange(0, ns + 1, sites, dtype=np.uint64),
                       block_n_reads=np.full(n_blocks, reads, np.uint32),
                       site_pos=np.tile(np.arange(sites, dtype=np.int64) * 7 + 100, n_blocks),
                       site_type=np.zeros(ns, np.uint8), site_word_off=np.zeros(ns, np.uint32),
                       site_n_words=np.ones(ns, np.uint32), site_plane_off=np.arange(ns, dtype=np.uint64) * 2,
                       planes=np.zeros(2 * ns, np.uint64))


def test_batch_validation_on_several_threads_reports_what_the_serial_loop_would():
    """round 4: a million sites were 1 ms of a one-shot call, so batches of 2^16 sites and more are validated block range by
    block range on a thread team; the EARLIEST defect is reported, in the words of the one-thread loop (small batches)"""
    pb = _wide_batch()
    assert pb.n_sites >= 1 << 16
    assert lgmi.plan_shard(pb, True, (0, 1))['n_items_total'] == 0          # valid (and no het site: nothing to do)

    def message(mutate, batch=None):
        b = batch if batch is not None else _wide_batch()
        mutate(b)
        with pytest.raises(LgmiError) as e:
            lgmi.plan_shard(b, True, (0, 1))
        return str(e.value)

    def two(b, late=True):                                # two defects: the earlier one is the one reported
        b.site_type[123] = 7
        if late:
            b.site_word_off[60000] = 5
    assert 'site 123: type 7' in message(two)
    assert 'site 60000: band [5,+1) exceeds 1 words of block 1875' in message(lambda b: b.site_word_off.__setitem__(60000, 5))
    assert 'site 70399: planes exceed n_plane_words' in message(lambda b: b.site_plane_off.__setitem__(70399, 2 * 70400))
    assert 'site 33: positions must increase strictly inside a block' in message(lambda b: b.site_pos.__setitem__(33, 0))
    assert 'site 32' not in message(lambda b: b.site_pos.__setitem__(33, 0))     # (a block's first site has no predecessor)

    def swap(b):                                          # a block table that goes backwards: found whatever the ranges were
        b.block_site_begin[1001] = b.block_site_begin[999]
    assert 'block_site_begin not monotone at block 1000' in message(swap)
    # the same defects in a batch small enough for the one-thread loop: the same words
    small = lambda: _wide_batch(n_blocks=8)               # noqa: E731
    assert 'site 123: type 7' in message(lambda b: two(b, late=False), small())
    assert 'site 33: positions must increase strictly inside a block' in message(lambda b: b.site_pos.__setitem__(33, 0), small())
