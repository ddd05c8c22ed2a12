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


def _enc(obj):
    """what the ranks exchange — None, bool, int, float, str, bytes, lists / tuples, str-keyed dicts and numpy arrays —
    as plain JSON (bytes and arrays base64-tagged).  No pickle: a frame read from a socket is data, never code."""
    import base64
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (bytes, bytearray)):
        return {'__b': base64.b64encode(bytes(obj)).decode('ascii')}
    if isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError('object arrays do not travel')
        a = np.ascontiguousarray(obj)
        return {'__nd': a.dtype.str, 'shape': list(a.shape), 'data': base64.b64encode(a.tobytes()).decode('ascii')}
    if isinstance(obj, (list, tuple)):
        return {'__t': [_enc(x) for x in obj]} if isinstance(obj, tuple) else [_enc(x) for x in obj]
    if isinstance(obj, dict):
        if any(not isinstance(k, str) or k.startswith('__') for k in obj):
            raise TypeError('only str-keyed dicts travel (keys must not start with "__")')
        return {k: _enc(v) for k, v in obj.items()}
    raise TypeError('cannot send a %s between ranks' % type(obj).__name__)


def _dec(obj):
    import base64
    if isinstance(obj, list):
        return [_dec(x) for x in obj]
    if isinstance(obj, dict):
        if '__b' in obj:
            return base64.b64decode(obj['__b'])
        if '__t' in obj:
            return tuple(_dec(x) for x in obj['__t'])
        if '__nd' in obj:
            dt = np.dtype(obj['__nd'])
            if dt.hasobject:
                raise ValueError('object arrays do not travel')
            return np.frombuffer(base64.b64decode(obj['data']), dtype=dt).reshape(obj['shape']).copy()
        return {k: _dec(v) for k, v in obj.items()}
    return obj


def dumps(obj) -> bytes:
    import json
    return json.dumps(_enc(obj), separators=(',', ':'), allow_nan=True).encode('utf-8')


def loads(data: bytes):
    import json
    return _dec(json.loads(data.decode('utf-8')))


class SocketGroup:
    """The few host-side exchanges a multi-GPU run needs — the 128-byte RCCL id, a barrier, a max over ranks, small
    object gathers — over plain TCP sockets, so that neither the library nor the bench needs torch.distributed.
    Rank 0 listens; every collective is "everyone sends to rank 0, rank 0 answers everyone": fine for a node's 8
    ranks.  Where rank 0 listens:
      port > 0        on (addr, port) — the launcher guarantees the port is free (mpirun-style launch, tests);
      port == 0       on an ephemeral port that it publishes in `rdzv_file` (one node, shared /tmp): what
                      group_from_env() picks under torch.distributed.run, whose agent already owns MASTER_PORT.
    Every connection must present the group's token (32 hex digits) before rank 0 keeps it: with a rendezvous file the
    token is random and travels in the file (created 0600, and a reader refuses a file that is not its own user's);
    with a fixed port it is `token` / LGMI_RDZV_TOKEN when the launcher provides a secret, else a digest of
    (addr, port, world, run id) — which keeps strangers' stray connections out but is no secret: on a network you do
    not trust, set LGMI_RDZV_TOKEN.  Frames are JSON (dumps / loads above), never pickle.  A hello with a rank outside
    1 .. world-1 or one already taken is dropped, and a connection that breaks during the hello does not end rank 0."""

    def __init__(self, rank: int, world: int, addr: str = '127.0.0.1', port: int = 29500, timeout: float = 300.0,
                 rdzv_file: Optional[str] = None, token: Optional[str] = None):
        import hashlib
        import os
        import socket
        import time
        self.rank, self.world = int(rank), int(world)
        self.peers = {}
        self.sock = None
        self.rdzv_file = None
        if world == 1:
            return
        if port == 0 and not rdzv_file:
            raise ValueError('an ephemeral port needs a rendezvous file')
        if not (0 <= self.rank < self.world):
            raise ValueError('rank %d outside 0 .. %d' % (self.rank, self.world - 1))
        if port != 0:
            secret = token or os.environ.get('LGMI_RDZV_TOKEN') or 'lgmi|%s|%d|%d|%s' % (
                addr, port, world, os.environ.get('TORCHELASTIC_RUN_ID', ''))
            token = hashlib.sha256(secret.encode('utf-8')).hexdigest()[:32]
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            if port == 0:
                token = os.urandom(16).hex()
                tmp = '%s.%d.tmp' % (rdzv_file, os.getpid())
                fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
                with os.fdopen(fd, 'w') as f:
                    f.write('%d %s\n' % (srv.getsockname()[1], token))
                os.replace(tmp, rdzv_file)                       # atomic: readers see nothing or the whole line
                self.rdzv_file = rdzv_file
            try:
                while len(self.peers) < world - 1:
                    conn, _ = srv.accept()                       # (a timeout here ends the group: a rank never came)
                    try:
                        conn.settimeout(min(timeout, 10.0))
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        hello = self._recv_exact(conn, 36)
                        r = int.from_bytes(hello[:4], 'little')
                        if hello[4:].decode('ascii', 'replace') != token or not (1 <= r < world) or r in self.peers:
                            conn.close()                         # a stranger, a stale rendezvous, a rank twice
                            continue
                        conn.sendall(b'ok')
                        conn.settimeout(timeout)
                        self.peers[r] = conn
                    except (OSError, ConnectionError):
                        conn.close()                             # that connection only
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    to_port, tok = port, token
                    if port == 0:
                        st = os.stat(rdzv_file)
                        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
                            raise ValueError('rendezvous file is not private to this user')
                        with open(rdzv_file) as f:
                            a, tok = f.read().split()
                        to_port = int(a)
                    s = socket.create_connection((addr, to_port), timeout=5.0)
                    s.settimeout(timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.sendall(self.rank.to_bytes(4, 'little') + tok.encode('ascii'))
                    if self._recv_exact(s, 2) == b'ok':
                        break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    pass
                if time.time() > deadline:
                    raise TimeoutError('rank %d could not reach the rendezvous of rank 0' % self.rank)
                time.sleep(0.1)
            self.sock = s

    @staticmethod
    def _recv_exact(conn, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = conn.recv(n - len(buf))
            if not chunk:
                raise ConnectionError('peer closed the rendezvous socket')
            buf += chunk
        return bytes(buf)

    def _send_msg(self, conn, payload: bytes):
        conn.sendall(len(payload).to_bytes(8, 'little') + payload)

    MAX_FRAME = 1 << 32

    def _recv_msg(self, conn) -> bytes:
        n = int.from_bytes(self._recv_exact(conn, 8), 'little')
        if n > self.MAX_FRAME:
            raise ConnectionError('frame of %d bytes refused' % n)
        return self._recv_exact(conn, n)

    def gather(self, obj, root: int = 0):
        """-> list of every rank's object on rank 0 (root must be 0), None elsewhere"""
        assert root == 0
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [None] * (self.world - 1)
            for r, conn in self.peers.items():
                out[r] = loads(self._recv_msg(conn))
            return out
        self._send_msg(self.sock, dumps(obj))
        return None

    def broadcast(self, obj, src: int = 0):
        assert src == 0
        if self.world == 1:
            return obj
        if self.rank == 0:
            data = dumps(obj)
            for conn in self.peers.values():
                self._send_msg(conn, data)
            return obj
        return loads(self._recv_msg(self.sock))

    def allgather(self, obj):
        return self.broadcast(self.gather(obj))

    def barrier(self):
        self.allgather(None)

    def allreduce_max(self, value: float) -> float:
        return max(self.allgather(float(value)))

    def broadcast_object_list(self, box, src: int = 0):      # the one torch.distributed call exchange_unique_id() uses
        box[0] = self.broadcast(box[0], src)

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.rank

    def gather_object(self, obj, box, dst: int = 0):         # what gather_tables_host() uses
        got = self.gather(obj, dst)
        if got is not None:
            box[:] = got

    def close(self):
        import os
        for conn in self.peers.values():
            conn.close()
        if self.sock is not None:
            self.sock.close()
        if self.rdzv_file:
            try:
                os.remove(self.rdzv_file)
            except OSError:
                pass
        self.peers, self.sock, self.rdzv_file = {}, None, None


def group_from_env(timeout: float = 300.0) -> SocketGroup:
    """SocketGroup from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).
    LGMI_RDZV_PORT=<port>: rank 0 listens there (any launcher, several nodes).  Under torch.distributed.run
    (TORCHELASTIC_USE_AGENT_STORE=True: the agent itself listens on MASTER_PORT) rank 0 takes an ephemeral port
    and publishes it in a file under the temp directory keyed by MASTER_PORT, the run id and the agent's pid
    (the ranks of one node share all three).  Otherwise MASTER_PORT is free and rank 0 listens on it."""
    import os
    import tempfile
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    if os.environ.get('LGMI_RDZV_PORT'):
        return SocketGroup(rank, world, addr, int(os.environ['LGMI_RDZV_PORT']), timeout)
    if os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True':
        key = 'lgmi_rdzv_%s_%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'),
                                     os.getppid())
        return SocketGroup(rank, world, '127.0.0.1' if addr in ('localhost', '') else addr, 0, timeout,
                           rdzv_file=os.path.join(tempfile.gettempdir(), key))
    return SocketGroup(rank, world, addr, int(os.environ.get('MASTER_PORT', '29500')), timeout)
