#!/usr/bin/env python3
"""bench.py — site-pair MI throughput of the MI355X engine on BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--shuffles S] [--scaling strong|weak]

A "step" is one pass of the hot path (pair counts -> MI -> ordered rows -> per-site mean MI -> permutation p) over
one synthetic batch that is already resident in HBM, ending with the result rows resident in HBM — on ONE GPU at
N = 1; at N > 1 on rank 0's GPU after the final gather.

N > 1 (one process per GPU; the driver launches them with torch.distributed.run, which only has to export RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT — no torch is imported here, the 128-byte RCCL id and the barriers
travel over lgmi.dist.SocketGroup):
  --scaling strong (default; BASELINE.json north_star: "a 50k-site x 200k-read synthetic chromosome at 1/2/4/8
      MI355X"): every rank holds the SAME chromosome (input replicated), computes shard rank/N of its
      site-pair tiles and rows (lgmi_params.shard_*), and the rows of all ranks are gathered HBM-to-HBM over
      RCCL/xGMI onto rank 0 INSIDE the timed region (lgmi_comm_gather, same_batch: per-site integer sums reduced).
  --scaling weak: every rank owns its own chromosome (blocks dealt to ranks), same gather with per-rank site bases.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'l-giremi_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # BASELINE.json north_star target: "50k-site x 200k-read synthetic chromosome"
    'north_star_dense_50kx200k': dict(n_sites=50_000, n_reads=200_000),
    # BASELINE.json configs[1]
    'cfg2_dense_10kx50k': dict(n_sites=10_000, n_reads=50_000),
    'small_dense_2kx20k': dict(n_sites=2_000, n_reads=20_000),
    # the long-read-like regime of SURVEY 8d: 25 (footprint, strand) blocks, each site sees ~70 reads
    'north_star_banded_50kx200k': dict(n_sites=50_000, n_reads=200_000, regime='banded'),
    # BASELINE.json configs[2]: whole-genome scale, 22 chromosomes, 200k sites x 1M reads in one batch
    'cfg3_22x9091x45455': dict(n_sites=9_091, n_reads=45_455, n_blocks=22),
    # BASELINE.json configs[4]: coverage depth x 4, mi_min_common_read = 6, 10,000 shuffles
    'cfg5_dense_depthx4_S10000': dict(n_sites=9_091, n_reads=181_820, n_blocks=22, shuffles=10_000),
    # the shape real L-GIREMI input has (src/giremi/footprint.py:6-28, script/giremi.py:32,60-78): 20,000 small
    # (footprint, strand) blocks, 5-120 sites x 50-3000 reads each (lgmi.synth.footprint_blocks) — not a BASELINE config;
    # reports blocks/s, the host's planning time and the count tiles' utilisation next to pairs/s
    'footprints_20k': dict(n_sites=0, n_reads=0, regime='footprints', n_footprints=20_000),
    # twice the north-star's sites in ONE block: 1.8e9 rows, ~160 GB of HBM in use — the headroom case (not a BASELINE config)
    'headroom_dense_100kx200k': dict(n_sites=100_000, n_reads=200_000),
    # three times the sites: 4.05e9 candidate rows — more than the permutation kernels' 32-bit row numbers and more than one
    # launch sequence's row arrays fit: lgmi_run_device cuts it into sequential shards by itself (lgmi_run_info.n_seq_shards)
    'headroom_dense_150kx200k': dict(n_sites=150_000, n_reads=200_000),
}
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# non-packed VALU: one wave64 instruction per 4 cycles per SIMD (MI355X_MICROARCH.md 'vector-instruction ISSUE
# cost': v_add_f32 4 cyc; the 157.3 TF fp32 peak counts packed 2-wide FMAs).  256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz.
# Measured with tools/ubench_valu.hip: 40.4e12 lane-ops/s for v_and_b32 + v_add_u32 (profiles/r01_ubench_valu.txt).
VALU_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9   # 3.93e13
VALU_WAVE_INSTR_PEAK = 256 * 4 * 2.4e9 / 4  # 6.1e11 wave64 instructions / s (one per 4 cycles per SIMD)
WORD_OP_LANE_OPS = 4                  # one 64-bit AND+POPC = 2 v_and_b32 + 2 v_bcnt_u32_b32
MFMA_I8_PEAK_TOPS = 5000.0            # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate (2.5 PF dense) per clock
MFMA_FP4_PEAK_TOPS = 10000.0          # MI355X_MICROARCH.md: FP4/FP6 MFMA ~10 PF dense


def committed_profile(name):
    """figures rocprofv3 cannot give from inside this process (PMC passes are separate runs, see profiles/README.md):
    the newest committed profiles/rNN_<name>.json"""
    d = os.path.join(ROOT, 'profiles')
    try:
        cands = sorted(f for f in os.listdir(d) if f.endswith('_' + name + '.json'))
        if cands:
            with open(os.path.join(d, cands[-1])) as f:
                out = json.load(f)
            out['_file'] = 'profiles/' + cands[-1]
            return out
    except OSError:
        pass
    return {}


class ClockSampler:
    """sclk / mclk of the device WHILE the timed steps run, read from sysfs (pp_dpm_sclk / pp_dpm_mclk: the level marked
    `*`) by a thread of this process every 50 ms — no child process (a program exec'ed from a process in which the profiler
    has initialised the GPU is refused on the test pool) and no GPU call.  The peaks in this file are priced at 2.4 GHz; what a
    box grants under load is what the count kernel's fraction of the nominal peak follows."""

    def __init__(self, device=0):
        import glob
        import threading
        # the GPUs this process may use are the render nodes its container was given (/dev/dri/renderD*): the device-th of them,
        # not the device-th card of the host (sysfs shows all eight, seven of them other tenants')
        nodes = sorted(glob.glob('/dev/dri/renderD*'), key=lambda n: int(n.rsplit('renderD', 1)[1]))
        self.path = None
        if nodes:
            d = '/sys/class/drm/%s/device' % os.path.basename(nodes[min(device, len(nodes) - 1)])
            if os.path.exists(os.path.join(d, 'pp_dpm_sclk')):
                self.path = d
        if self.path is None:
            cards = sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))
            self.path = os.path.dirname(cards[min(device, len(cards) - 1)]) if cards else None
        self.samples = {'sclk': [], 'mclk': []}
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.path else None

    def _read(self, name):
        try:
            with open(os.path.join(self.path, 'pp_dpm_' + name)) as f:
                for ln in f:
                    if '*' in ln:
                        return int(''.join(ch for ch in ln.split(':')[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def _run(self):
        while not self._stop.is_set():
            for k in self.samples:
                v = self._read(k)
                if v:
                    self.samples[k].append(v)
            self._stop.wait(0.05)

    def start(self):
        if self._thread:
            self._thread.start()
        return self

    def stop(self):
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join(1.0)
        out = {}
        for k, v in self.samples.items():
            if v:
                v = sorted(v)
                out[k + '_mhz'] = {'median': v[len(v) // 2], 'min': v[0], 'max': v[-1], 'samples': len(v)}
        return out or None


def cpu_baseline(eng, wl, min_common, n_shuffles, seed):
    """The CPU oracle (oracle/lgmi_oracle.c, OpenMP) timed on a bounded sample of the
    same workload: same read count, fewer sites, so it finishes in ~10-30 s."""
    import lgmi
    from oracle import c_oracle
    n_reads = wl['n_reads']
    cores = c_oracle.load().lgo_num_threads()
    if wl.get('regime') in ('banded', 'footprints'):
        if wl['regime'] == 'banded':
            from lgmi.synth import banded_chromosome
            pb = banded_chromosome(wl['n_sites'], n_reads, seed=seed)
        else:
            pb = wl['_host_batch']
        t0 = time.perf_counter()
        out = c_oracle.run(pb, min_common=min_common, het_only=True, n_shuffles=n_shuffles, seed=seed, threads=cores)
        dt = time.perf_counter() - t0
        return {'value': out['n_examined'] / dt, 'unit': 'site-pairs/s', 'cores': cores, 'kind': 'port',
                'sample': 'the whole %s workload: %d examined pairs, %d emitted, %.1f s wall'
                          % (wl['regime'], out['n_examined'], len(out['row_i']), dt)}
    # ~2.5e9 pair-words keeps 16 host cores busy for ~10-20 s; the permutation stage of the sample scales with S
    target_pair_words = 2.5e9 * max(1, cores) / 16 / max(1.0, n_shuffles / 1000.0)
    words = (n_reads + 63) // 64
    p = int(max(60, min(wl['n_sites'], (target_pair_words / words / 0.18) ** 0.5)))
    spec = lgmi.default_synth_spec(p, n_reads, seed=seed)
    db = eng.synth_dense(spec)
    pb = db.download()
    db.free()
    t0 = time.perf_counter()
    out = c_oracle.run(pb, min_common=min_common, het_only=True, n_shuffles=n_shuffles, seed=seed, threads=cores)
    dt = time.perf_counter() - t0
    return {'value': out['n_examined'] / dt, 'unit': 'site-pairs/s', 'cores': cores, 'kind': 'port',
            'sample': 'same generator and read count (%d reads), first-principles subsample of %d sites of one '
                      'chromosome: %d examined pairs, %d emitted, %.1f s wall' % (n_reads, p, out['n_examined'],
                                                                                  len(out['row_i']), dt)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (no RANK / WORLD_SIZE in the environment): start the N ranks here, one
    fresh child process per GPU with RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT set — what
    `python -m torch.distributed.run --nproc-per-node N` exports.  Called before this process has loaded liblgmi or made
    any GPU call, and it never makes one: children are started, never exec'ed over a process that has initialised the GPU.
    Rank 0 inherits stdout and prints the one JSON line; the exit code is 0 only if every rank's is."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:                                # a free port for rank 0's rendezvous listener
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # RCCL across processes needs dmabuf IPC on this driver
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [None] * n
    deadline = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if deadline is None and any(c not in (None, 0) for c in codes):
            deadline = time.time() + 30.0                      # a rank failed: the others get half a minute to notice (watchdogs, closed sockets)
        if deadline is not None and time.time() > deadline:
            for r, p in enumerate(procs):                      # exactly the children started above, by pid
                if codes[r] is None:
                    p.send_signal(signal.SIGKILL)
            deadline = time.time() + 1e9
        time.sleep(0.05)
    bad = [c for c in codes if c != 0]
    if bad:
        print('[bench] rank exit codes: %s' % codes, file=sys.stderr)
    return 0 if not bad else (bad[0] if bad[0] > 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='north_star_dense_50kx200k', choices=sorted(WORKLOADS))
    ap.add_argument('--shuffles', type=int, default=None,
                    help='default 1000 (BASELINE.json configs[1]: 1000-shuffle permutation p); cfg5: 10000')
    ap.add_argument('--min-common', type=int, default=6)      # l-giremi CLI default (script/giremi.py:212-216)
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-host-to-host', action='store_true')
    ap.add_argument('--force-gather', action='store_true',
                    help='N = 1: bring the RCCL communicator up with one rank and run the gather and its verification '
                         'anyway (rehearses the N > 1 code path on a one-GPU box)')
    ap.add_argument('--no-gather', action='store_true',
                    help='N > 1: skip the RCCL communicator and the row gather (kernel-only scaling; also lets several '
                         'ranks share one GPU with LGMI_BENCH_DEVICE=0 to rehearse the launch, which RCCL refuses)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and 'RANK' not in os.environ:
            # no launcher: this process becomes one — it starts the N ranks as fresh children and never touches the GPU
            sys.exit(spawn_ranks(args.gpus))
        args.gpus = world

    # stdout carries ONE JSON line.  Libraries loaded below write banners to file descriptor 1 (RCCL prints its version
    # block there when a communicator comes up): keep a private copy of the real stdout for the JSON line and point
    # descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import lgmi
    from lgmi.dist import group_from_env
    wl = WORKLOADS[args.workload]
    if wl.get('regime') == 'footprints':
        # built by a pool of forked workers: before anything touches the GPU
        from lgmi.synth import footprint_blocks
        t_gen = time.perf_counter()
        wl = dict(wl, _host_batch=footprint_blocks(wl['n_footprints'], seed=20250810 + (0 if args.scaling == 'strong' or world == 1 else 1000 * rank),
                                                   cache_dir=os.environ.get('LGMI_BENCH_CACHE', '/tmp')))
        wl['n_sites'] = wl['_host_batch'].n_sites
        wl['n_reads'] = int(wl['_host_batch'].block_n_reads.max())
        print('[bench] %d footprints, %d sites, built in %.1f s' % (wl['n_footprints'], wl['n_sites'], time.perf_counter() - t_gen), file=sys.stderr)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    group = group_from_env()                                   # plain sockets; a no-op object at N = 1
    device = int(os.environ.get('LGMI_BENCH_DEVICE', local_rank))
    import ctypes
    n_dev = ctypes.c_int(0)
    if lgmi._lib.load().lgmi_device_count(ctypes.byref(n_dev)) == 0 and n_dev.value > 0:
        device %= n_dev.value                                  # a launcher that narrows the visible devices per rank

    eng = lgmi.Engine(device)
    n_shuffles = args.shuffles if args.shuffles is not None else wl.get('shuffles', 1000)
    strong = args.scaling == 'strong' or world == 1
    seed = 20250808 + (0 if strong else 1000 * rank)           # strong: the same chromosome on every rank
    n_blocks = wl.get('n_blocks', 1)
    if wl.get('regime') == 'banded':
        from lgmi.synth import banded_chromosome
        db = eng.upload(banded_chromosome(wl['n_sites'], wl['n_reads'], seed=seed))
    elif wl.get('regime') == 'footprints':
        db = eng.upload(wl['_host_batch'])
        n_blocks = wl['_host_batch'].n_blocks
    else:
        db = eng.synth_dense(lgmi.default_synth_spec(wl['n_sites'], wl['n_reads'], seed=seed, n_blocks=n_blocks))
    gather_state = {'on': (world > 1 or args.force_gather) and not args.no_gather, 'note': None}
    hung = []
    shard = (rank, world) if (strong and world > 1) else None
    n_sites_rank = wl['n_sites'] * (1 if wl.get('regime') == 'footprints' else n_blocks)

    def sync():
        eng.synchronize()
        if world > 1:
            group.barrier()
            eng.synchronize()

    def step():
        kw = dict(min_common=args.min_common, n_shuffles=n_shuffles, seed=seed, het_only=True, shard=shard)
        base = 0 if strong else rank * n_sites_rank
        if gather_state['on'] and gather_state.get('overlap'):
            # rows first; their transfer to rank 0 (communication stream) runs under the permutation kernels
            dr = eng.run_device(db, rows_only=True, **kw)
            t0 = time.perf_counter()
            flight = eng.comm_gather_begin(dr, root=0, site_base=base, same_batch=strong)
            t1 = time.perf_counter()
            dr.permute()
            t2 = time.perf_counter()
            g, begins = flight.finish()
            info = dr.info()
            info['ms_gather'] = 1e3 * ((t1 - t0) + (time.perf_counter() - t2))      # host time in the two halves
            info['world_rows'] = [begins[k + 1] - begins[k] for k in range(world)]
            if g is not None:
                info['gathered_rows'] = g.info()['n_rows']
                g.free()
            dr.free()
            return info
        dr = eng.run_device(db, **kw)
        info = dr.info()
        if gather_state['on']:                                 # the final gather: rows HBM -> rank 0's HBM over xGMI
            t0 = time.perf_counter()
            g, begins = eng.comm_gather(dr, root=0, site_base=base, same_batch=strong)
            info['ms_gather'] = 1e3 * (time.perf_counter() - t0)
            info['world_rows'] = [begins[k + 1] - begins[k] for k in range(world)]
            if g is not None:
                info['gathered_rows'] = g.info()['n_rows']
                g.free()
        dr.free()
        return info

    clock_state = {}

    def measure():
        """W untimed + K timed steps between barriers; -> (infos, max-over-ranks seconds, per-rank stage means)"""
        for _ in range(args.warmup):
            step()
        sync()
        sampler = ClockSampler(device).start() if rank == 0 else None
        t0 = time.perf_counter()
        infos_ = []
        for _ in range(args.steps):
            t_ = time.perf_counter()
            infos_.append(step())                      # run_device returns with its stream drained
            infos_[-1]['wall_ms'] = 1e3 * (time.perf_counter() - t_)
        sync()
        elapsed_ = group.allreduce_max(time.perf_counter() - t0)
        if sampler is not None:
            clock_state['clocks'] = sampler.stop()
        per_rank_ = group.gather({k: sum(i[k] for i in infos_) / len(infos_) for k in
                                  ('ms_total', 'ms_count', 'ms_emit', 'ms_perm', 'n_examined', 'n_rows', 'n_tile_pairs')})
        return infos_, elapsed_, per_rank_

    def run_verify():
        """N > 1, strong scaling: one more (untimed) pass whose gathered result rank 0 compares, array by array, with the
        unsharded run of the same chromosome on its own GPU — the multi-GPU path checks itself wherever it runs"""
        import numpy as np
        if gather_state.get('overlap'):
            dr = eng.run_device(db, rows_only=True, min_common=args.min_common, n_shuffles=n_shuffles, seed=seed, het_only=True,
                                shard=shard)
            flight = eng.comm_gather_begin(dr, root=0, same_batch=True)
            dr.permute()
            g, begins = flight.finish()
        else:
            dr = eng.run_device(db, min_common=args.min_common, n_shuffles=n_shuffles, seed=seed, het_only=True, shard=shard)
            g, begins = eng.comm_gather(dr, root=0, same_batch=True)
        dr.free()
        if g is None:
            return None
        got = g.fetch()
        g.free()
        dr0 = eng.run_device(db, min_common=args.min_common, n_shuffles=n_shuffles, seed=seed, het_only=True)
        ref = dr0.fetch()
        dr0.free()
        fields = ['row_i', 'row_j', 'row_mi', 'site_n_pairs', 'site_mean_mi'] + (['row_p', 'row_exceed'] if n_shuffles else [])
        bad = [f for f in fields if not np.array_equal(getattr(got, f), getattr(ref, f), equal_nan=(f in ('site_mean_mi', 'row_p')))]
        return {'gathered_rows': int(got.n_rows), 'unsharded_rows': int(ref.n_rows), 'rank_row_begin': begins,
                'fields_compared': fields, 'fields_differing': bad, 'equal_to_unsharded': not bad}

    def make_out(infos, elapsed, per_rank, verify):
        """the JSON line (rank 0 only); host_to_host and cpu_baseline are added by the caller"""
        info = infos[-1]
        ms_count = sum(i['ms_count'] for i in infos) / len(infos)
        if True:
            examined_job = info['n_examined_total'] if strong else sum(int(p['n_examined']) for p in per_rank)
            rows_job = sum(info['world_rows']) if 'world_rows' in info else sum(int(p['n_rows']) for p in per_rank)
            general_job = info['n_general_rows']                    # this rank's; scaled below for the job-wide rate
            # the count kernel's algorithmic HBM bytes per launch: every column's planes once
            # (16 B per 64-read word) + site metadata + the 4 counts (16 B) of every EXAMINED pair.  (Up to round 4 this
            # credited every slot of every tile, padding included: 7.4x the examined pairs on the footprint batch, where a
            # tile is 14 % full — the padding is traffic, not algorithm; tile_utilisation in the line says how much.)
            mfma = info.get('n_mfma_tiles', 0) > 0
            alg_bytes = info['bytes_in'] + 16 * info['n_examined']
            word_ops = 4 * info['word_pairs']                      # 4 mandatory AND+POPC per pair-word
            secs = ms_count * 1e-3
            out = {
                'metric': 'MI site-pairs/sec (incl. permutation p)' if n_shuffles else 'MI site-pairs/sec',
                'value': examined_job * args.steps / elapsed,
                'unit': 'site-pairs/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': 1e3 * elapsed / args.steps,
                'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
                'dtype': {2: 'u64 bit planes -> fp4 (e2m1) MFMA operands, f32 counts (exact < 2^24), f64 MI',
                          1: 'u64 bit planes -> int8 MFMA operands, i32 counts, f64 MI'}.get(
                              info.get('mfma_dtype', 0), 'u64 bit planes (AND + popcount), u32 counts, f64 MI'),
                'data': 'synthetic',
                'config': {'workload': args.workload, 'n_sites': wl['n_sites'], 'n_reads': wl['n_reads'],
                           'n_blocks': n_blocks, 'regime': wl.get('regime', 'dense'), 'het_every': 5,
                           'min_common': args.min_common, 'n_shuffles': n_shuffles,
                           'examined_pairs_job': examined_job, 'emitted_pairs_job': rows_job,
                           'parallelism': ('1 gpu' if world == 1 else
                                           ('tile shards of one batch x%d, RCCL row gather to rank 0' % world if strong else
                                            'one chromosome per rank x%d, RCCL row gather to rank 0' % world)),
                           'timed_region': 'resident batch -> result rows resident in HBM'
                                           + (' of rank 0 (gather included)' if gather_state['on'] else ''),
                           'gather': ('liblgmi RCCL (lgmi_comm_gather)' if gather_state['on'] else
                                      (gather_state['note'] or ('skipped (--no-gather)' if world > 1 else 'n/a')))},
                'step_wall_ms': [round(i.get('wall_ms', 0.0), 3) for i in infos],      # this rank's; `value` uses the max over ranks of the whole region
                'stage_ms': {k: sum(i.get(k, 0.0) for i in infos) / len(infos)
                             for k in ('ms_total', 'ms_prep', 'ms_plan_host', 'ms_count', 'ms_emit', 'ms_perm', 'ms_perm_fast',
                                       'ms_perm_general', 'ms_perm_exact', 'ms_mean', 'ms_gather')},
                'n_seq_shards': info.get('n_seq_shards', 1),   # > 1: lgmi_run_device split the run to fit its memory budget
            }
            if world > 1 or args.force_gather:
                out['per_rank'] = per_rank
                out['verify'] = verify if verify is not None else 'not run (no gather, or weak scaling)'
            st_ = out['stage_ms']
            ex_ = sum(i.get('ms_perm_exact', 0.0) for i in infos) / len(infos)
            parts = {'count (k_count*)': st_['ms_count'], 'emit (k_emit<1>, scans, k_emit<2>)': st_['ms_emit'],
                     'perm 2 x 2 (k_perm_fast)': st_['ms_perm_fast'], 'perm exact larger tables (k_perm_enum, k_perm_six)': ex_,
                     'perm sampled tables (k_perm_general)': max(st_['ms_perm_general'] - ex_, 0.0), 'prep (k_gather_ops, plan upload)': st_['ms_prep']}
            tot_ = st_['ms_total'] or 1.0
            out['step_breakdown'] = [{'stage': k, 'ms': round(v, 3), 'share': round(v / tot_, 4)}
                                     for k, v in sorted(parts.items(), key=lambda kv: -kv[1])]
            out['dominant_stage'] = out['step_breakdown'][0]['stage']
            out['rates'] = {'emitted_pairs_per_s': rows_job * args.steps / elapsed,
                            'blocks_per_s': n_blocks * (world if not strong else 1) * args.steps / elapsed}
            # how full the count kernels' tiles are: pairs examined / pairs inside the tiles computed (small footprints
            # fill a fraction of a 64 x 64 tile), and what the host spends planning per step
            out['tile_utilisation'] = {'n_examined': info['n_examined'], 'n_tile_pairs': info['n_tile_pairs'],
                                       'frac': info['n_examined'] / info['n_tile_pairs'] if info['n_tile_pairs'] else None,
                                       'ms_plan_host': sum(i['ms_plan_host'] for i in infos) / len(infos)}
            if n_shuffles:
                # the permutation stage, priced on what it really does: a 2 x 2 row costs ONE binomial variate
                # against its exact tail mass (k_perm_fast); a 3 x 2 / 2 x 3 row behind the gate likewise, its mass from the
                # perimeter walk (k_perm_six; small tables by enumeration, k_perm_enum); only what is left draws n_shuffles
                # tables each (k_perm_general).  Instruction counts, lane occupancy and the instruction-class account come
                # from the committed counter passes of this workload.
                sq = committed_profile('pmc_sq_perm').get(args.workload, {})
                cls = committed_profile('pmc_class').get(args.workload, {})
                avg = lambda k: sum(i.get(k, 0.0) for i in infos) / len(infos)
                ms_fast, ms_exact = avg('ms_perm_fast'), avg('ms_perm_exact')
                ms_sampled = max(avg('ms_perm_general') - ms_exact, 0.0)
                six_job = info.get('n_six_rows', 0)
                sampled_rows = max(general_job - six_job, 0)            # (enumerated rows included: a footprint-regime detail)
                draws = sampled_rows * n_shuffles
                pr = {'two_by_two_rows': info['n_rows'] - general_job, 'larger_rows': general_job, 'six_cell_exact_rows': six_job,
                      'sampled_or_enumerated_rows': sampled_rows, 'table_draws_upper': draws,
                      'ms_fast': ms_fast, 'ms_exact': ms_exact, 'ms_sampled': ms_sampled,
                      'two_by_two_rows_per_s': (info['n_rows'] - general_job) / (ms_fast * 1e-3) if ms_fast > 0 else None,
                      'six_cell_rows_per_s': six_job / (ms_exact * 1e-3) if ms_exact > 0 and six_job else None,
                      'table_draws_per_s_lower': draws / (ms_sampled * 1e-3) if ms_sampled > 0 else None,
                      'bound': 'vector-instruction issue at partial lane occupancy for k_perm_fast and k_perm_six (f64 recurrences, '
                               'divergent boundary searches); what bounds each is the instruction-class account below, not a '
                               '4-cycle-per-instruction figure',
                      'unit': 'wave64 VALU instructions/s', 'peak_at_4_cycles': VALU_WAVE_INSTR_PEAK,
                      'counters': sq.get('_file') or committed_profile('pmc_sq_perm').get('_file'),
                      'source': 'the counter-derived fields below come from the committed profiles named in `counters` / '
                                '`class_account.file` (separate rocprofv3 --pmc passes of this workload), NOT from this run; '
                                'times and rates are this run\'s'}
                for k in ('k_perm_fast', 'k_perm_six', 'k_perm_general'):
                    c = sq.get(k)
                    if c and c.get('valu_insts') and c.get('ms'):
                        pr[k] = {'ms_profiled': c['ms'], 'wave_valu_insts': c['valu_insts'],
                                 'valu_issue_frac_at_4_cycles': c['valu_insts'] / (c['ms'] * 1e-3) / VALU_WAVE_INSTR_PEAK,
                                 'active_lane_frac': c.get('active_lanes', 0) / 64.0}
                        if cls.get(k):
                            pr[k]['class_account'] = dict(cls[k], file=committed_profile('pmc_class').get('_file'))
                out['perm_roofline'] = pr
            # the ordered emission (validity count + scan + MI and row write): priced like the permutation kernels, on VALU
            # issue from the committed SQ counter pass — nine f64 logarithms per row are most of its instructions
            sq_e = committed_profile('pmc_sq_perm').get(args.workload, {}).get('k_emit<2>')
            ms_emit = sum(i['ms_emit'] for i in infos) / len(infos)
            if world == 1 and sq_e and sq_e.get('valu_insts') and sq_e.get('ms') and info['n_rows']:
                out['emit_roofline'] = {'kernel': 'k_emit<2>', 'bound': 'vector issue + look-up latency (class_account)', 'ms_emit_stage': ms_emit,
                                        'rows_per_s': info['n_rows'] / (ms_emit * 1e-3) if ms_emit > 0 else None,
                                        'valu_issue_frac_at_4_cycles': sq_e['valu_insts'] / (sq_e['ms'] * 1e-3) / VALU_WAVE_INSTR_PEAK,
                                        'class_account': committed_profile('pmc_class').get(args.workload, {}).get('k_emit<2>'),
                                        'active_lane_frac': sq_e.get('active_lanes', 0) / 64.0,
                                        'valu_insts_per_row': sq_e['valu_insts'] * 64.0 / info['n_rows'],
                                        'peak': VALU_WAVE_INSTR_PEAK, 'unit': 'wave64 VALU instructions/s',
                                        'counters': committed_profile('pmc_sq_perm').get('_file'),
                                        'source': 'valu_* and active_lane_frac: committed profile, not this run'}
            traffic = committed_profile('pmc_k_count').get(args.workload + ('_mfma' if mfma else ''), {}).get('hbm_bytes')
            hbm = {'kernel': 'count', 'bound': 'hbm', 'algorithmic_bytes': alg_bytes, 'achieved': alg_bytes / secs / 1e9,
                   'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg_bytes / secs / 1e9 / HBM_PEAK_GBS,
                   'traffic': traffic, 'traffic_source': 'committed profile (separate --pmc passes), not this run',
                   'note': 'bytes per launch; the count kernel is compute-bound, not HBM-bound.  traffic = L2-miss bytes '
                           '(FETCH_SIZE x2 + WRITE_SIZE, newest profiles/rNN_pmc_k_count.json)'}
            valu = {'kernel': 'k_count', 'bound': 'valu_popcount', 'achieved': word_ops / secs / 1e12,
                    'peak': VALU_LANE_OPS_PEAK / WORD_OP_LANE_OPS / 1e12, 'unit': 'T word-ops/s (64-bit AND+POPC)',
                    'frac': word_ops / secs / (VALU_LANE_OPS_PEAK / WORD_OP_LANE_OPS)}
            if mfma:
                # matrix-core count kernels: the four counts of a pair-word are 4 x 64 multiply-accumulates = 512 ops
                ops = 512.0 * info['word_pairs']
                fp4 = info.get('mfma_dtype', 1) == 2
                peak = MFMA_FP4_PEAK_TOPS if fp4 else MFMA_I8_PEAK_TOPS
                out['roofline'] = {'kernel': 'k_count_mfma_fp4' if fp4 else 'k_count_mfma', 'bound': 'mfma',
                                   'achieved': ops / secs / 1e12, 'peak': peak, 'unit': 'TFLOP/s',
                                   'op_kind': ('fp4 (e2m1)' if fp4 else 'int8') + ' multiply-add ops (tera-ops/s), dense MFMA peak',
                                   'frac': ops / secs / 1e12 / peak, 'traffic': traffic,
                                   'traffic_source': 'committed profile (separate --pmc passes), not this run',
                                   'algorithmic_ops': ops, 'ms_kernel': ms_count,
                                   'note': 'algorithmic ops = 512 x examined pair-words (4 counts x 64 reads x 2) of this '
                                           'rank\'s shard; '
                                           + ('v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands and unit scales, exact in f32 below 2^24 reads; '
                                              if fp4 else 'v_mfma_i32_32x32x32_i8; ')
                                           + 'weighted-bit operands made from the bit planes in registers (one AND per operand dword, the y side\'s place value by the MX block scale; shift-AND for the sign bit)'}
                out['roofline']['share_of_step'] = ms_count / out['stage_ms']['ms_total'] if out['stage_ms']['ms_total'] else None
                out['hbm_roofline'] = hbm
            else:
                out['roofline'] = hbm
                out['valu_roofline'] = valu
            if world == 1 and not args.no_host_to_host:
                # SURVEY §8d's other wall time: packed batch in HOST memory -> result rows in HOST memory (upload + layout
                # prep + kernels + fetch), through lgmi_run.  Never `value`.
                pb = db.download()
                t1 = time.perf_counter()
                eng.run_raw(pb, min_common=args.min_common, n_shuffles=n_shuffles, seed=seed, het_only=True, compact=True)
                dt_first = time.perf_counter() - t1                # pays the pinning of the result buffers (cached after)
                # steady state: the median of four more calls (the second call still pays first-use costs of its own — the
                # runtime's registration of the caller's pages for the DMA — 28.8 ms against 20.4 from the third on, footprint batch)
                later = []
                for _ in range(4 if info['ms_total'] < 50.0 else 2):
                    t1 = time.perf_counter()
                    hi = eng.run_raw(pb, min_common=args.min_common, n_shuffles=n_shuffles, seed=seed, het_only=True, compact=True)
                    later.append(time.perf_counter() - t1)
                dt = sorted(later)[len(later) // 2] if len(later) % 2 else sum(sorted(later)[len(later) // 2 - 1:len(later) // 2 + 1]) / 2.0
                out['host_to_host'] = {'ms': 1e3 * dt, 'ms_min': 1e3 * min(later), 'ms_first_call': 1e3 * dt_first,
                                       'ms_calls_after_the_first': [1e3 * x for x in later],
                                       'site_pairs_per_s': hi['n_examined'] / dt,
                                       'h2d_bytes': int(pb.planes.nbytes + 25 * len(pb.site_pos)),
                                       'd2h_bytes': int(hi['bytes_out'] + 12 * len(pb.site_pos)),
                                       'kernels_ms': hi['ms_total'],
                                       'row_form': 'compact (include/lgmi.h ABI 6): per-site row offsets instead of row_i, no partner for a site '
                                                   'whose every candidate pair was emitted, 16-bit permutation counts',
                                       'note': 'one lgmi_run call from pageable host memory: validation + H2D + layout prep '
                                               '+ kernels + D2H of the compact rows (mi, 16-bit exceed, listed partners, per-site offsets) into pinned buffers the '
                                               'context caches (the first call pins them), the permutation counts following the kernels range by range; '
                                               'ms = median of the calls after the first.  The host side '
                                               '(planner threads, page pinning) shares the box\'s CPUs with seven other tenants under a 16-CPU quota: '
                                               'the same call measured 20 - 60 ms on one box within a minute'}
                # SURVEY 8d's wall time as a rate, next to `value` (which stays the resident-to-resident figure)
                out['value_host_to_host'] = hi['n_examined'] / dt
                del pb
            if not args.no_cpu_baseline and world == 1:        # the CPU leg is timed at N = 1 only
                out['cpu_baseline'] = cpu_baseline(eng, wl, args.min_common, n_shuffles, seed)
            out['clocks'] = clock_state.get('clocks')       # sampled during the timed steps (ClockSampler)
            try:
                med = out['clocks']['sclk_mhz']['median']
                if out['roofline'].get('bound') == 'mfma' and med and out['clocks']['sclk_mhz']['samples'] >= 8 and ms_count >= 10.0:
                    # the nominal peak is priced at 2.4 GHz; under this load the box runs slower (power), and the matrix pipe's
                    # rate follows the clock: the same kernel time against the peak at the clock the box actually granted
                    out['roofline']['frac_at_sampled_sclk'] = out['roofline']['frac'] * 2400.0 / med
                    out['roofline']['sampled_sclk_mhz'] = med
            except (KeyError, TypeError):
                pass
            return out

    def degrade(line, tag, text):
        """the gather could not be measured: the line says so where a machine reads it — `degraded`, no `value` (the
        kernel-only figure moves to `kernel_only_value`) — and the process exits non-zero"""
        line['config']['gather'] = text
        line['degraded'] = tag
        line['kernel_only_value'] = line['value']
        line['value'] = None

    multi = gather_state['on']
    out = None
    exit_code = 0
    if not multi:
        infos, elapsed, per_rank = measure()
        if rank == 0:
            out = make_out(infos, elapsed, per_rank, None)
    else:
        # (1) kernel-only measurement first: it involves no RCCL call, so it always completes and is what rank 0 prints
        #     if the communicator cannot be brought up or the gathered measurement does not finish
        gather_state['on'] = False
        infos_k, elapsed_k, per_rank_k = measure()
        fallback = None
        if rank == 0:
            gather_state['note'] = 'kernel-only: the RCCL gather was not measured'
            fallback = make_out(infos_k, elapsed_k, per_rank_k, None)
        # (2) RCCL communicator inside liblgmi (csrc/comm.cpp).  The 128-byte id travels over the sockets; the collective
        #     ncclCommInitRank runs in a watched thread: a rank that cannot bring RCCL up (or never returns) costs the job
        #     its gather, not its measurement
        import threading
        from lgmi.dist import exchange_unique_id
        uid = exchange_unique_id(group, eng.comm_unique_id if rank == 0 else None)
        box = {}

        def bring_up():
            try:
                eng.comm_init(uid, rank, world)
                box['ok'] = True
            except Exception as e:                              # noqa: BLE001
                box['err'] = repr(e)
        th = threading.Thread(target=bring_up, daemon=True)
        th.start()
        th.join(float(os.environ.get('LGMI_COMM_INIT_TIMEOUT', '240')))
        if th.is_alive():
            hung.append(th)
        states = group.allgather(box.get('err') or ('ok' if box.get('ok') else 'timeout'))
        if any(st != 'ok' for st in states):
            exit_code = 4                                       # the metric (gather included) was NOT measured
            if rank == 0:
                degrade(fallback, 'rccl_init_failed', 'RCCL communicator not available (%s): rows were NOT gathered' % states)
                out = fallback
        else:
            # (3) the measurement proper, gather inside the timed region, under a watchdog: if it does not finish, every
            #     rank leaves and rank 0 prints the kernel-only line saying so
            limit = float(os.environ.get('LGMI_GATHER_TIMEOUT', '600'))

            def give_up():
                if rank == 0:
                    degrade(fallback, 'gather_hang', 'the RCCL gather did not finish within %.0f s: rows were NOT gathered' % limit)
                    os.write(json_fd, (json.dumps(fallback) + '\n').encode())
                os._exit(4 if rank == 0 else 3)                 # never 0: the requested measurement did not happen
            dog = threading.Timer(limit, give_up)
            dog.daemon = True
            dog.start()
            gather_state.update(on=True, note=None)
            rccl = eng.comm_info()                              # what RCCL itself says: version, library, ncclCommCount
            infos, elapsed, per_rank = measure()
            verify = run_verify() if (strong and not os.environ.get('LGMI_BENCH_NO_VERIFY')) else None
            dog.cancel()
            simple = None
            if rank == 0:
                simple = make_out(infos, elapsed, per_rank, verify)
                simple['rccl'] = rccl
                simple['kernel_only'] = {'value': fallback['value'], 'ms_per_step': fallback['ms_per_step'],
                                         'note': 'the same steps without the gather (measured first)'}
                simple['config']['gather'] += ' after the permutation stage'
                out = simple
            # (4) the same with the rows travelling UNDER the permutation stage (run split in two, gather in two halves).
            #     Its own watchdog: if it does not finish, rank 0 prints the line of (3).  It becomes the reported line only
            #     when it finishes, verifies against the unsharded run and is faster.
            if n_shuffles and not os.environ.get('LGMI_BENCH_NO_OVERLAP'):
                def keep_simple():
                    if rank == 0:
                        simple['overlapped_gather'] = 'did not finish within %.0f s' % limit
                        simple['degraded'] = 'overlapped_gather_hang (the reported line is the gather-after-permutation measurement, which completed)'
                        os.write(json_fd, (json.dumps(simple) + '\n').encode())
                    os._exit(5 if rank == 0 else 3)             # a rank is stuck in a collective: say so with the exit code
                dog = threading.Timer(limit, keep_simple)
                dog.daemon = True
                dog.start()
                gather_state['overlap'] = True
                try:
                    infos2, elapsed2, per_rank2 = measure()
                    verify2 = run_verify() if (strong and not os.environ.get('LGMI_BENCH_NO_VERIFY')) else None
                    failed = None
                except Exception as e:                              # noqa: BLE001  (an error on one rank is an error on all: agreed below)
                    failed = repr(e)
                dog.cancel()
                states2 = group.allgather(failed)
                if rank == 0:
                    if any(states2):
                        simple['overlapped_gather'] = 'failed: %s' % [st for st in states2 if st]
                    else:
                        over = make_out(infos2, elapsed2, per_rank2, verify2)
                        over['rccl'] = rccl
                        over['config']['gather'] += ' under the permutation stage (lgmi_run_device_rows, lgmi_comm_gather_begin, lgmi_dresult_permute, lgmi_comm_gather_finish)'
                        ok2 = verify2 is None or verify2.get('equal_to_unsharded')
                        summary = {'value': over['value'], 'ms_per_step': over['ms_per_step'], 'verify': verify2}
                        if ok2 and over['value'] > simple['value']:
                            over['kernel_only'] = simple['kernel_only']
                            over['gather_after_permutation'] = {'value': simple['value'], 'ms_per_step': simple['ms_per_step'],
                                                                'verify': simple.get('verify')}
                            out = over
                        else:
                            simple['overlapped_gather'] = summary
    if hung:
        # a thread is still inside ncclCommInitRank with the context: nothing of it is touched (no free, no close — that
        # would race with the thread), the line goes out and the process leaves with a non-zero code
        if out is not None:
            os.write(json_fd, (json.dumps(out) + '\n').encode())
        os._exit(exit_code or 4)
    db.free()
    if world > 1:
        group.barrier()
    eng.close()
    group.close()
    if out is not None:
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if exit_code:
        sys.exit(exit_code)


if __name__ == '__main__':
    main()
