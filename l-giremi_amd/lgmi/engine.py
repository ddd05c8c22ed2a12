"""Engine: one MI355X, one HIP stream — thin object wrapper over the C ABI."""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check
from .pack import PackedBatch


class MIResult:
    """rows in reference order; indices are global site indices of the batch.

    ``row_p``: the permutation p of every row, or None without p-values.  The library is asked not to make the array
    when it is a function of ``row_exceed`` (Monte-Carlo estimates: ``(1 + row_exceed) / (n_shuffles + 1)``,
    lgmi_params.no_row_p) — 8 of 28 bytes per row that need not cross PCIe; it is derived here on first use.

    ``row_i`` / ``row_j`` / ``row_exceed``: with the compact row form (include/lgmi.h, ABI 6: what ``Engine.run`` asks
    for) the library ships per-site row offsets instead of ``row_i``, no partner at all for a site whose every candidate
    pair was emitted, and 16-bit counts; the plain arrays are made on first use (``lgmi_result_expand_rows``), the way
    ``row_p`` is.  ``row_begin`` / ``site_row_full`` / ``row_j_listed`` are the compact arrays themselves (None otherwise)."""

    def __init__(self, row_i, row_j, row_mi, row_p, row_exceed, row_counts, site_mean_mi, site_n_pairs, info,
                 n_shuffles=0, p_derived=False, compact=None):
        self._row_i, self._row_j, self.row_mi = row_i, row_j, row_mi
        self._row_p, self._row_exceed = row_p, row_exceed
        self.row_counts = row_counts             # (n_rows, 3, 3): [class at i][class at j]
        self.site_mean_mi, self.site_n_pairs, self.info = site_mean_mi, site_n_pairs, info
        self.n_shuffles, self._p_derived = int(n_shuffles), bool(p_derived)
        # compact: dict(row_begin, site_row_full, row_j_listed, row_exceed16, expand) or None
        self._compact = compact
        self.row_begin = compact['row_begin'] if compact else None
        self.site_row_full = compact['site_row_full'] if compact else None
        self.row_j_listed = compact['row_j_listed'] if compact else None

    @property
    def compact(self):
        return self._compact is not None

    def _expand(self):
        if self._row_i is None and self._compact is not None:
            self._row_i, self._row_j = self._compact['expand']()

    @property
    def row_i(self):
        self._expand()
        return self._row_i

    @property
    def row_j(self):
        self._expand()
        return self._row_j

    @property
    def row_exceed(self):
        if self._row_exceed is None and self._compact is not None and self._compact.get('row_exceed16') is not None:
            self._row_exceed = self._compact['row_exceed16'].astype(np.uint32)
        return self._row_exceed

    @property
    def row_p(self):
        if self._row_p is None and self._p_derived and self.row_exceed is not None:
            self._row_p = (1.0 + self.row_exceed.astype(np.float64)) / (self.n_shuffles + 1.0)   # the kernels' own expression
        return self._row_p

    @property
    def n_rows(self):
        return len(self.row_mi)


class _HostRows:
    """owner of one lgmi_result: the library's (pinned) host arrays live until the last numpy view of them is gone"""

    def __init__(self, lib, res):
        self.lib, self.res = lib, res

    def __del__(self):
        try:
            self.lib.lgmi_result_free(C.byref(self.res))
        except Exception:
            pass


def _copy_result(res: _lib.Result, info: dict, owner: _HostRows = None) -> MIResult:
    """owner given: the arrays are read-only VIEWS of the library's host buffers (no copy: the rows of a batch of footprints
    are hundreds of megabytes) that keep the owner — and with it the buffers — alive; the small per-site arrays are copies,
    so that keeping one of them does not hold hundreds of megabytes of pinned memory.  Else copies."""
    n, ns = int(res.n_rows), int(res.n_sites)
    ctype = {np.uint32: C.c_uint32, np.float64: C.c_double, np.uint64: C.c_uint64, np.uint8: C.c_uint8, np.uint16: C.c_uint16}

    def a(ptr, count, dt, shape=None, present=True, copy=False):
        if not present:
            return None
        if count == 0 or not ptr:
            out = np.zeros(count if not ptr and count else 0, dt)
        elif owner is not None and not copy:
            buf = (ctype[dt] * count).from_address(C.addressof(ptr.contents))
            buf._owner = owner                       # array -> memoryview -> buf -> owner
            out = np.frombuffer(buf, dtype=dt)
            out.flags.writeable = False              # library-owned (pinned) memory: not the caller's to edit
        else:
            out = np.ctypeslib.as_array(ptr, shape=(count,)).astype(dt, copy=True)
        return out.reshape(shape) if shape else out
    is_compact = bool(res.compact)
    has_e = bool(res.row_exceed) or (n == 0 and info.get('has_p', False) and not res.row_exceed16)
    derived = bool(res.row_p_derived)
    has_p = (bool(res.row_p) or (n == 0 and info.get('has_p', False))) and not derived
    has_c = bool(res.row_counts) or (n == 0 and info.get('has_counts', False))
    compact = None
    if is_compact:
        if owner is None:
            raise ValueError('a compact result needs its owner (lgmi_result_expand_rows reads the library\'s arrays)')

        def expand():
            ri, rj = np.empty(n, np.uint32), np.empty(n, np.uint32)
            check(owner.lib.lgmi_result_expand_rows(C.byref(owner.res), ri.ctypes.data_as(_lib.u32p), rj.ctypes.data_as(_lib.u32p)))
            return ri, rj
        compact = {'row_begin': a(res.row_begin, ns + 1, np.uint64, copy=True), 'site_row_full': a(res.site_row_full, ns, np.uint8, copy=True),
                   'row_j_listed': a(res.row_j_listed, int(res.n_row_j_listed), np.uint32),
                   'row_exceed16': a(res.row_exceed16, n, np.uint16, present=bool(res.row_exceed16)), 'expand': expand}
    return MIResult(a(res.row_i, n, np.uint32, present=not is_compact), a(res.row_j, n, np.uint32, present=not is_compact),
                    a(res.row_mi, n, np.float64),
                    a(res.row_p, n, np.float64, present=has_p), a(res.row_exceed, n, np.uint32, present=has_e),
                    a(res.row_counts, 9 * n, np.uint32, (n, 3, 3), present=has_c),
                    a(res.site_mean_mi, ns, np.float64, copy=True), a(res.site_n_pairs, ns, np.uint32, copy=True), info,
                    n_shuffles=int(res.n_shuffles), p_derived=derived, compact=compact)


def make_params(min_common=5, n_shuffles=0, seed=0, het_only=True, emit_counts=False, exact_2x2=False,
                shard=None, no_row_p=True, stream_site_base=0, compact=False) -> _lib.Params:
    """shard = (rank, world): compute only that contiguous, cost-balanced slice of the result rows.
    no_row_p (default): row_p is not made as an array when it is a function of row_exceed (MIResult.row_p derives it).
    compact: lgmi_run returns the compact row form (MIResult.row_i / row_j / row_exceed expand it on first use)"""
    if min_common < 0:
        min_common = 0
    rank, world = (0, 0) if shard is None else (int(shard[0]), int(shard[1]))
    return _lib.Params(int(min_common), int(n_shuffles), int(seed) & (2**64 - 1),
                       1 if het_only else 0, 1 if emit_counts else 0, 1 if exact_2x2 else 0, 1 if no_row_p else 0, rank, world,
                       int(stream_site_base), 1 if compact else 0, (C.c_uint8 * 3)())


def plan_shard(batch: PackedBatch, het_only=True, shard=(0, 1), n_shuffles=0) -> dict:
    """The planner's view of one shard, computed on the host (no GPU): item range, examined pairs, count tiles
    and the slot-matrix coordinates of every site (include/lgmi.h: lgmi_plan_shard)."""
    lib = _lib.load()
    st, sp = batch.as_struct(), _lib.ShardPlan()
    _lib.check(lib.lgmi_plan_shard(C.byref(st), 1 if het_only else 0, int(n_shuffles), int(shard[0]), int(shard[1]), C.byref(sp)))
    try:
        def a(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n and ptr else np.zeros(0, np.uint32)
        ni, nt, ns = int(sp.n_items_total), int(sp.n_tiles), len(batch.site_pos)
        return {'n_items_total': ni, 'item_begin': int(sp.item_begin), 'item_end': int(sp.item_end),
                'n_examined_total': int(sp.n_examined_total), 'n_examined': int(sp.n_examined),
                'n_tiles_total': int(sp.n_tiles_total), 'n_tiles': nt,
                'item_site': a(sp.item_site, ni), 'item_seg': a(sp.item_seg, ni),
                'tile_block': a(sp.tile_block, nt), 'tile_x0': a(sp.tile_x0, nt), 'tile_y0': a(sp.tile_y0, nt),
                'tile_edge': a(sp.tile_edge, nt),
                'site_xrow': a(sp.site_xrow, ns), 'site_ycol': a(sp.site_ycol, ns), 'site_prow': a(sp.site_prow, ns),
                'site_pcol': a(sp.site_pcol, ns), 'site_xnext': a(sp.site_xnext, ns)}
    finally:
        lib.lgmi_shard_plan_free(C.byref(sp))


def default_synth_spec(n_sites, n_reads, seed=20250808, n_blocks=1) -> _lib.SynthSpec:
    """SURVEY §8d dense regime: 10 % dropout, het SNP every 5th site with 2 % noise,
    2 % tri-allelic sites with a 5 % third allele, 1 % of the other sites typed snp;
    n_blocks chromosomes of n_sites x n_reads each (block c is the one-block batch of seed + c)"""
    return _lib.SynthSpec(int(seed), int(n_sites), int(n_reads), 5, 6554, 1311, 20, 3277, 10, int(n_blocks), 0)


class DeviceBatch:
    def __init__(self, engine, handle):
        self.engine, self.handle = engine, handle
        engine._children.add(self)

    def download(self) -> PackedBatch:
        b = _lib.Batch()
        _lib.check(self.engine.lib.lgmi_dbatch_download(self.handle, C.byref(b)))
        return PackedBatch.from_struct(b)

    def free(self):
        if self.handle and self.engine.handle and os.getpid() == self.engine.pid:
            self.engine.lib.lgmi_dbatch_free(self.handle)   # the C object points into its context
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceResult:
    def __init__(self, engine, handle):
        self.engine, self.handle = engine, handle
        engine._children.add(self)

    def info(self) -> dict:
        ri = _lib.RunInfo()
        _lib.check(self.engine.lib.lgmi_dresult_info(self.handle, C.byref(ri)))
        return ri.as_dict()

    def permute(self):
        """the permutation stage of a result made with run_device(..., rows_only=True)"""
        _lib.check(self.engine.lib.lgmi_dresult_permute(self.engine.handle, self.handle))

    def fetch(self, compact=False) -> MIResult:
        """compact=True: the compact row form crosses PCIe (lgmi_dresult_fetch_compact); MIResult expands it on first use"""
        res = _lib.Result()
        fn = self.engine.lib.lgmi_dresult_fetch_compact if compact else self.engine.lib.lgmi_dresult_fetch
        _lib.check(fn(self.handle, C.byref(res)))
        return _copy_result(res, self.info(), _HostRows(self.engine.lib, res))

    def free(self):
        if self.handle and self.engine.handle and os.getpid() == self.engine.pid:
            self.engine.lib.lgmi_dresult_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GatherInFlight:
    """a gather between lgmi_comm_gather_begin and lgmi_comm_gather_finish (keeps its source result alive)"""

    def __init__(self, engine, handle, source):
        self.engine, self.handle, self.source = engine, handle, source

    def finish(self):
        """-> (DeviceResult on the root / None elsewhere, rank_row_begin list)"""
        if not self.handle:
            raise RuntimeError('gather already finished')
        h, self.handle = self.handle, None
        out = C.c_void_p()
        begins = (C.c_uint64 * (self.engine.world + 1))()
        _lib.check(self.engine.lib.lgmi_comm_gather_finish(h, C.byref(out), begins))
        self.source = None
        return (DeviceResult(self.engine, out) if out else None), [int(b) for b in begins]


class Engine:
    """Owns an ``lgmi_ctx``.  HIP is first touched here, never at import: create the
    engine AFTER any fork (the reference calls the MI step inside multiprocessing
    workers, script/giremi.py:375-380; a HIP context does not survive fork)."""

    def __init__(self, device: Optional[int] = None):
        self.lib = _lib.load()
        if device is None:
            device = int(os.environ.get('LGMI_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        h = C.c_void_p()
        _lib.check(self.lib.lgmi_ctx_create(int(device), C.byref(h)))
        self.handle, self.device, self.pid = h, int(device), os.getpid()
        self.rank, self.world = 0, 1
        self._children = weakref.WeakSet()     # resident batches / results: freed before the context goes

    def _alive(self):
        if not self.handle:
            raise RuntimeError('engine is closed')
        if os.getpid() != self.pid:
            raise RuntimeError('lgmi Engine used in a forked child: create the Engine after fork')

    def close(self):
        if self.handle and os.getpid() == self.pid:
            for child in list(self._children):
                child.free()
            self.lib.lgmi_ctx_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one-shot path (upload + kernels + fetch)
    def synchronize(self):
        self._alive()
        _lib.check(self.lib.lgmi_ctx_synchronize(self.handle))

    def run(self, batch: PackedBatch, min_common=5, n_shuffles=0, seed=0, het_only=True,
            emit_counts=False, exact_2x2=False, shard=None, no_row_p=True, stream_site_base=0, compact=True) -> MIResult:
        self._alive()
        st, prm = batch.as_struct(), make_params(min_common, n_shuffles, seed, het_only, emit_counts, exact_2x2, shard, no_row_p,
                                                 stream_site_base, compact)
        res, info = _lib.Result(), _lib.RunInfo()
        _lib.check(self.lib.lgmi_run(self.handle, C.byref(st), C.byref(prm), C.byref(res), C.byref(info)))
        d = info.as_dict()
        d['has_p'], d['has_counts'] = n_shuffles > 0 or bool(exact_2x2), bool(emit_counts)
        return _copy_result(res, d, _HostRows(self.lib, res))       # views: freed with the last of them

    def run_raw(self, batch: PackedBatch, **kw) -> dict:
        """lgmi_run (upload + kernels + fetch into library-owned host memory) without copying the rows into
        numpy: what the bench times as host-to-host.  -> run info"""
        self._alive()
        st, prm = batch.as_struct(), make_params(**kw)
        res, info = _lib.Result(), _lib.RunInfo()
        _lib.check(self.lib.lgmi_run(self.handle, C.byref(st), C.byref(prm), C.byref(res), C.byref(info)))
        d = info.as_dict()
        d['host_rows'] = int(res.n_rows)
        self.lib.lgmi_result_free(C.byref(res))
        return d

    # ---- resident path (bench, multi-GPU)
    def upload(self, batch: PackedBatch) -> DeviceBatch:
        self._alive()
        st, h = batch.as_struct(), C.c_void_p()
        _lib.check(self.lib.lgmi_batch_upload(self.handle, C.byref(st), C.byref(h)))
        return DeviceBatch(self, h)

    def synth_dense(self, spec: _lib.SynthSpec) -> DeviceBatch:
        self._alive()
        h = C.c_void_p()
        _lib.check(self.lib.lgmi_synth_dense(self.handle, C.byref(spec), C.byref(h)))
        return DeviceBatch(self, h)

    def synth_chromosomes(self, n_blocks, n_sites, n_reads, seed=20250808) -> DeviceBatch:
        """n_blocks dense chromosomes of n_sites x n_reads in one resident batch (BASELINE.json configs[2])"""
        return self.synth_dense(default_synth_spec(n_sites, n_reads, seed=seed, n_blocks=n_blocks))

    def run_device(self, dbatch: DeviceBatch, min_common=5, n_shuffles=0, seed=0, het_only=True,
                   emit_counts=False, exact_2x2=False, shard=None, rows_only=False, no_row_p=True,
                   stream_site_base=0) -> DeviceResult:
        """rows_only=True stops when the rows (i, j, mi, tables, per-site means) are final; DeviceResult.permute()
        runs the permutation stage later — a multi-GPU host starts the row gather in between (comm_gather_begin)"""
        self._alive()
        prm, h = make_params(min_common, n_shuffles, seed, het_only, emit_counts, exact_2x2, shard, no_row_p, stream_site_base), C.c_void_p()
        fn = self.lib.lgmi_run_device_rows if rows_only else self.lib.lgmi_run_device
        _lib.check(fn(self.handle, dbatch.handle, C.byref(prm), C.byref(h)))
        return DeviceResult(self, h)

    def selftest_log(self, x):
        """mi_log (the table-driven logarithm of emit.hip) of every x[k], computed on the device"""
        self._alive()
        x = np.ascontiguousarray(x, np.float64)
        out = np.empty(len(x), np.float64)
        _lib.check(self.lib.lgmi_selftest_log(self.handle, len(x), x.ctypes.data_as(_lib.f64p), out.ctypes.data_as(_lib.f64p)))
        return out

    def selftest_le_exp(self, x2, t):
        """-> (le_exp decisions, det_exp decisions, hardware f32 exp, det_exp) for the pairs (x2[k], t[k])"""
        self._alive()
        x2 = np.ascontiguousarray(x2, np.float64)
        t = np.ascontiguousarray(t, np.float64)
        n = len(t)
        fast, det = np.empty(n, np.uint8), np.empty(n, np.uint8)
        e_hw, e_det = np.empty(n, np.float64), np.empty(n, np.float64)
        _lib.check(self.lib.lgmi_selftest_le_exp(self.handle, n, x2.ctypes.data_as(_lib.f64p), t.ctypes.data_as(_lib.f64p),
                                                 fast.ctypes.data_as(_lib.u8p), det.ctypes.data_as(_lib.u8p),
                                                 e_hw.ctypes.data_as(_lib.f64p), e_det.ctypes.data_as(_lib.f64p)))
        return fast.astype(bool), det.astype(bool), e_hw, e_det

    def site_mean(self, row_i, row_j, row_mi, n_sites):
        self._alive()
        ri = np.ascontiguousarray(row_i, np.uint32)
        rj = np.ascontiguousarray(row_j, np.uint32)
        rm = np.ascontiguousarray(row_mi, np.float64)
        mean = np.empty(n_sites, np.float64)
        cnt = np.empty(n_sites, np.uint32)

        def p(a, t):
            return a.ctypes.data_as(t) if a.size else C.cast(None, t)
        _lib.check(self.lib.lgmi_site_mean(self.handle, len(ri), p(ri, _lib.u32p), p(rj, _lib.u32p),
                                           p(rm, _lib.f64p), n_sites, p(mean, _lib.f64p), p(cnt, _lib.u32p)))
        return mean, cnt


    def ecdf(self, ref, query):
        """#{ref < q} / len(ref) for every q (NaN stays NaN) — stat.ecdf of the reference, on the GPU"""
        self._alive()
        r = np.ascontiguousarray(ref, np.float64).ravel()
        q = np.ascontiguousarray(query, np.float64).ravel()
        if r.size == 0:
            raise ZeroDivisionError('division by zero')      # what stat.ecdf raises (1/n with n == 0)
        out = np.empty(q.size, np.float64)

        def p(a):
            return a.ctypes.data_as(_lib.f64p) if a.size else C.cast(None, _lib.f64p)
        _lib.check(self.lib.lgmi_ecdf(self.handle, r.size, p(r), q.size, p(q), p(out)))
        return out

    # ---- multi-GPU: RCCL is used only for the final gather (csrc/comm.cpp)
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(self.lib.lgmi_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        self._alive()
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError('unique id must be %d bytes' % _lib.UNIQUE_ID_BYTES)
        buf = C.create_string_buffer(unique_id, _lib.UNIQUE_ID_BYTES)
        _lib.check(self.lib.lgmi_comm_init(self.handle, buf, int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    def comm_init_group(self, group, rank=None, world=None):
        """rank 0 creates the RCCL unique id; `group` — lgmi.dist.SocketGroup, or any object with torch.distributed's
        broadcast_object_list / get_rank / get_world_size (nothing here imports torch) — carries the 128 bytes"""
        from .dist import exchange_unique_id
        rank = group.get_rank() if rank is None else int(rank)
        world = group.get_world_size() if world is None else int(world)
        self.comm_init(exchange_unique_id(group, self.comm_unique_id if rank == 0 else None), rank, world)

    def comm_info(self) -> dict:
        """what was bound and what the communicator says about itself (lgmi_comm_info): RCCL version, library path,
        ncclCommCount / ncclCommUserRank, and whether it is the tests' stand-in"""
        self._alive()
        ci = _lib.CommInfo()
        _lib.check(self.lib.lgmi_comm_info(self.handle, C.byref(ci)))
        v = int(ci.rccl_version)
        return {'rccl_version': v, 'rccl_version_str': ('%d.%d.%d' % (v // 10000, v // 100 % 100, v % 100)) if v >= 0 else None,
                'lib_path': ci.lib_path.decode('utf-8', 'replace'), 'nranks': int(ci.nranks), 'rank': int(ci.rank),
                'world_given': int(ci.world_given), 'rank_given': int(ci.rank_given),
                'initialised': bool(ci.initialised), 'stand_in': bool(ci.stand_in)}

    def comm_allgather_u64(self, value: int):
        self._alive()
        out = (C.c_uint64 * self.world)()
        _lib.check(self.lib.lgmi_comm_allgather_u64(self.handle, int(value), out))
        return [int(v) for v in out]

    def comm_allgather_u64v(self, values):
        self._alive()
        n = len(values)
        mine = (C.c_uint64 * n)(*[int(v) for v in values])
        out = (C.c_uint64 * (n * self.world))()
        _lib.check(self.lib.lgmi_comm_allgather_u64v(self.handle, mine, n, out))
        return [[int(out[r * n + k]) for k in range(n)] for r in range(self.world)]

    def comm_gather(self, dresult: 'DeviceResult', root=0, site_base=0, same_batch=False):
        """HBM-to-HBM gather over RCCL: -> (DeviceResult on the root / None elsewhere, rank_row_begin list).
        same_batch=True: the ranks ran shards of one batch (per-site integer sums are added up);
        otherwise site_base shifts this rank's site indices into the global numbering."""
        self._alive()
        opts = _lib.GatherOpts(int(site_base), 1 if same_batch else 0, (C.c_uint8 * 3)())
        h = C.c_void_p()
        begins = (C.c_uint64 * (self.world + 1))()
        _lib.check(self.lib.lgmi_comm_gather(self.handle, dresult.handle, int(root), C.byref(opts), C.byref(h), begins))
        return (DeviceResult(self, h) if h else None), [int(b) for b in begins]

    def comm_gather_begin(self, dresult: 'DeviceResult', root=0, site_base=0, same_batch=False) -> 'GatherInFlight':
        """first half of comm_gather on the communication stream: sizes exchanged, (i, j, mi[, tables]) posted; returns
        while they travel.  Call .finish() after dresult.permute()."""
        self._alive()
        opts = _lib.GatherOpts(int(site_base), 1 if same_batch else 0, (C.c_uint8 * 3)())
        h = C.c_void_p()
        _lib.check(self.lib.lgmi_comm_gather_begin(self.handle, dresult.handle, int(root), C.byref(opts), C.byref(h)))
        return GatherInFlight(self, h, dresult)

    def comm_gather_rows(self, dresult: 'DeviceResult', root=0):
        """every rank's (row_i, row_j, row_mi[, row_p]) concatenated in rank order on `root`"""
        self._alive()
        res = _lib.Result()
        _lib.check(self.lib.lgmi_comm_gather_rows(self.handle, dresult.handle, int(root), C.byref(res)))
        try:
            n = int(res.n_rows)

            def a(ptr, dt):
                if not ptr or n == 0:
                    return np.zeros(0, dt) if (ptr or n == 0) else None
                return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)
            out = {'row_i': a(res.row_i, np.uint32), 'row_j': a(res.row_j, np.uint32), 'row_mi': a(res.row_mi, np.float64),
                   'row_exceed': a(res.row_exceed, np.uint32) if res.row_exceed else None,
                   'row_p': a(res.row_p, np.float64) if res.row_p else None}
            if out['row_p'] is None and res.row_p_derived and out['row_exceed'] is not None:
                out['row_p'] = (1.0 + out['row_exceed'].astype(np.float64)) / (int(res.n_shuffles) + 1.0)
            return out
        finally:
            self.lib.lgmi_result_free(C.byref(res))


_default: Optional[Engine] = None


def default_engine() -> Engine:
    global _default
    if _default is None or _default.pid != os.getpid() or not _default.handle:
        _default = Engine()
    return _default
