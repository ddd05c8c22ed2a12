#!/usr/bin/env python3
"""bench.py — site-pair MI throughput of the MI355X engine on BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--shuffles S]

A "step" is one pass of the hot path (pair counts -> MI -> ordered rows -> per-site
mean MI [-> permutation p]) over one synthetic chromosome that is already resident
in HBM.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
owns its own chromosome (weak scaling: blocks never share pairs); the
only collective on the data path is the final RCCL gather of per-rank row counts.
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
}
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# non-packed VALU: one wave64 instruction per 4 cycles per SIMD (MI355X_MICROARCH.md 'vector-instruction ISSUE
# cost': v_add_f32 4 cyc; the 157.3 TF fp32 peak counts packed 2-wide FMAs).  256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz.
# Measured with tools/ubench_valu.hip: 40.4e12 lane-ops/s for v_and_b32 + v_add_u32 (profiles/r01_ubench_valu.txt).
VALU_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9   # 3.93e13
WORD_OP_LANE_OPS = 4                  # one 64-bit AND+POPC = 2 v_and_b32 + 2 v_bcnt_u32_b32
MFMA_I8_PEAK_TOPS = 5000.0            # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate (2.5 PF dense) per clock
MFMA_FP4_PEAK_TOPS = 10000.0          # MI355X_MICROARCH.md: FP4/FP6 MFMA ~10 PF dense


def pmc_traffic(workload, mfma=False):
    """HBM bytes per k_count launch from the committed PMC passes (profiles/r01_pmc_k_count.json): rocprofv3
    cannot be driven from inside this process, so the counters are collected by the separate --pmc runs
    described in that file and looked up here by workload name."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_k_count.json')) as f:
            return json.load(f).get(workload + ('_mfma' if mfma else ''), {}).get('hbm_bytes')
    except OSError:
        return None


def cpu_baseline(eng, wl, min_common, n_shuffles, seed):
    """The CPU oracle (oracle/lgmi_oracle.c, OpenMP) timed on a bounded sample of the
    same workload: same read count, fewer sites, so it finishes in ~10-30 s."""
    import lgmi
    from oracle import c_oracle
    n_reads = wl['n_reads']
    cores = c_oracle.load().lgo_num_threads()
    if wl.get('regime') == 'banded':
        from lgmi.synth import banded_chromosome
        pb = banded_chromosome(wl['n_sites'], n_reads, seed=seed)
        t0 = time.perf_counter()
        out = c_oracle.run(pb, min_common=min_common, het_only=True, n_shuffles=n_shuffles, seed=seed, threads=cores)
        dt = time.perf_counter() - t0
        return {'value': out['n_examined'] / dt, 'unit': 'site-pairs/s', 'cores': cores, 'kind': 'port',
                'sample': 'the whole banded workload: %d examined pairs, %d emitted, %.1f s wall'
                          % (out['n_examined'], len(out['row_i']), dt)}
    # ~2.5e9 pair-words keeps 16 host cores busy for ~10-20 s
    target_pair_words = 2.5e9 * max(1, cores) / 16
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
            'sample': 'same generator and read count (%d reads), first-principles subsample of %d sites: '
                      '%d examined pairs, %d emitted, %.1f s wall' % (n_reads, p, out['n_examined'],
                                                                      len(out['row_i']), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='north_star_dense_50kx200k', choices=sorted(WORKLOADS))
    ap.add_argument('--shuffles', type=int, default=1000)     # BASELINE.json configs[1]: 1000-shuffle permutation p
    ap.add_argument('--min-common', type=int, default=6)      # l-giremi CLI default (script/giremi.py:212-216)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path '
                         'with several ranks on one GPU: set LGMI_BENCH_DEVICE=0)')
    ap.add_argument('--lib-comm', action='store_true',
                    help='final gather through liblgmi\'s own RCCL communicator (lgmi_comm_*) instead of '
                         'torch.distributed (which is RCCL too); only exercised with one rank so far')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
        args.gpus = world

    import torch
    import lgmi
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    tdev = 'cuda' if args.backend == 'nccl' else 'cpu'
    device = int(os.environ.get('LGMI_BENCH_DEVICE', local_rank))

    eng = lgmi.Engine(device)
    wl = WORKLOADS[args.workload]
    seed = 20250808 + 1000 * rank
    if wl.get('regime') == 'banded':
        from lgmi.synth import banded_chromosome
        db = eng.upload(banded_chromosome(wl['n_sites'], wl['n_reads'], seed=seed))
    else:
        spec = lgmi.default_synth_spec(wl['n_sites'], wl['n_reads'], seed=seed)
        db = eng.synth_dense(spec)                             # input resident in HBM before timing
    if world > 1 and args.lib_comm:
        eng.comm_init_torch(dist, rank, world)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        dr = eng.run_device(db, min_common=args.min_common, n_shuffles=args.shuffles, seed=seed, het_only=True)
        info = dr.info()
        if world > 1:                                                      # final gather (RCCL over xGMI)
            if args.lib_comm:
                info['world_rows'] = eng.comm_allgather_u64(info['n_rows'])
            else:
                mine = torch.tensor([info['n_rows']], dtype=torch.int64, device=tdev)
                allr = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allr, mine)
                info['world_rows'] = [int(t.item()) for t in allr]
        dr.free()
        return info

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    infos = [step() for _ in range(args.steps)]
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    info = infos[-1]
    examined = info['n_examined']
    ms_count = sum(i['ms_count'] for i in infos) / len(infos)
    out = None
    if rank == 0:
        # dominant kernel: k_count.  Algorithmic HBM bytes per launch:
        # every column's planes once (16 B per 64-read word) + site metadata + the 4 count planes
        # of every computed slot (16 B).
        mfma = info.get('n_mfma_tiles', 0) > 0
        alg_bytes = info['bytes_in'] + 16 * info['n_tile_pairs']
        word_ops = 4 * info['word_pairs']                      # 4 mandatory AND+POPC per pair-word
        secs = ms_count * 1e-3
        out = {
            'metric': 'MI site-pairs/sec (incl. permutation p)' if args.shuffles else 'MI site-pairs/sec',
            'value': examined * world * args.steps / elapsed,
            'unit': 'site-pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {2: 'u64 bit planes -> fp4 (e2m1) MFMA operands, f32 counts (exact < 2^24), f64 MI',
                      1: 'u64 bit planes -> int8 MFMA operands, i32 counts, f64 MI'}.get(
                          info.get('mfma_dtype', 0), 'u64 bit planes (AND + popcount), u32 counts, f64 MI'),
            'data': 'synthetic',
            'config': {'workload': args.workload, 'n_sites': wl['n_sites'], 'n_reads': wl['n_reads'],
                       'regime': wl.get('regime', 'dense'), 'het_every': 5, 'min_common': args.min_common,
                       'n_shuffles': args.shuffles, 'examined_pairs_per_gpu': examined,
                       'emitted_pairs_per_gpu': info['n_rows'], 'parallelism': 'dp%d' % world},
            'stage_ms': {k: sum(i[k] for i in infos) / len(infos)
                         for k in ('ms_total', 'ms_prep', 'ms_count', 'ms_emit', 'ms_perm', 'ms_mean')},
        }
        # emitted pairs/s and pair-shuffles/s beside the examined-pair rate (whole job)
        rows_all = sum(info.get('world_rows', [info['n_rows']]))
        out['rates'] = {'emitted_pairs_per_s': rows_all * args.steps / elapsed,
                        'pair_shuffles_per_s': rows_all * args.shuffles * args.steps / elapsed}
        hbm = {'kernel': 'count', 'bound': 'hbm', 'algorithmic_bytes': alg_bytes, 'achieved': alg_bytes / secs / 1e9,
               'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg_bytes / secs / 1e9 / HBM_PEAK_GBS,
               'traffic': pmc_traffic(args.workload, mfma),
               'note': 'bytes per launch; the count kernel is compute-bound, not HBM-bound.  traffic = L2-miss bytes '
                       '(FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_k_count.json)'}
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
                               'frac': ops / secs / 1e12 / peak, 'traffic': hbm['traffic'],
                               'algorithmic_ops': ops,
                               'note': 'algorithmic ops = 512 x examined pair-words (4 counts x 64 reads x 2); '
                                       + ('v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands and unit scales, exact in f32 below 2^24 reads; '
                                          if fp4 else 'v_mfma_i32_32x32x32_i8; ')
                                       + 'weighted-bit operands made from the bit planes in registers (AND / shift-AND per operand dword)'}
            out['hbm_roofline'] = hbm
        else:
            out['roofline'] = hbm
            out['valu_roofline'] = valu
        if not args.no_cpu_baseline and world == 1:        # the CPU leg is timed at N = 1 only
            out['cpu_baseline'] = cpu_baseline(eng, wl, args.min_common, args.shuffles, seed)
    db.free()
    if dist is not None:
        dist.barrier()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
