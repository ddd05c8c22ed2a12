#!/usr/bin/env python3
"""Turns rocprofv3 output directories into the small JSON summaries kept under profiles/ (and read back by
bench.py: committed_profile()).

    python tools/pmc_summary.py count  WORKLOAD FETCH_DIR WRITE_DIR OUT.json     # HBM traffic per launch
    python tools/pmc_summary.py sq     WORKLOAD SQ_DIR DRAWS ROWS OUT.json       # VALU issue / active lanes of the perm kernels
    python tools/pmc_summary.py stats  TRACE_DIR OUT.csv                          # per-kernel time table

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a 16-byte-per-lane streaming read
(MI355X_MICROARCH.md, HBM section), so it is doubled; the two counters do not fit one pass and are collected in
separate runs.  Every value is per launch: the runs use --steps 1 --warmup 0."""
import csv
import glob
import json
import os
import sys


def previous(path):
    """the summary already at `path` (several workloads share one file), {} when there is none"""
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True), key=os.path.getmtime)
    if not hits:
        sys.exit('no *%s under %s' % (suffix, d))
    return hits[-1]                 # the newest run when a directory holds several


def counters(d):
    """kernel name -> {counter: summed value}, and kernel name -> (launches, total ns)"""
    vals, time = {}, {}
    seen = set()
    with open(find(d, 'counter_collection.csv')) as f:
        for row in csv.DictReader(f):
            k = row['Kernel_Name']
            vals.setdefault(k, {}).setdefault(row['Counter_Name'], 0.0)
            vals[k][row['Counter_Name']] += float(row['Counter_Value'])
            key = (row['Dispatch_Id'], k)
            if key not in seen:
                seen.add(key)
                n, t = time.get(k, (0, 0))
                time[k] = (n + 1, t + int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    return vals, time


def short(name):
    s = name.split('(')[0].replace('void ', '').replace('lgmi::', '').strip()
    # k_emit<2, true> / k_emit<2, false> (round 5: one kernel per row layout) keep round 4's key
    if s.startswith('k_emit<'):
        s = s.split(',')[0].rstrip('>') + '>'
    return s


def main():
    mode = sys.argv[1]
    if mode == 'count':
        workload, fdir, wdir, out = sys.argv[2:6]
        fv, ft = counters(fdir)
        wv, _ = counters(wdir)
        res = {'_how': 'rocprofv3 --pmc FETCH_SIZE (one pass) and --pmc WRITE_SIZE (another pass) with --kernel-trace on '
                       '`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host`; bytes = KiB x 1024, '
                       'FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md); per launch'}
        entry = {}
        for k in fv:
            if 'FETCH_SIZE' not in fv[k]:
                continue
            n = max(ft[k][0], 1)
            fkb = fv[k]['FETCH_SIZE'] / n
            wkb = wv.get(k, {}).get('WRITE_SIZE', 0.0) / n
            entry[short(k)] = {'launches': n, 'fetch_size_kb': fkb, 'write_size_kb': wkb, 'fetch_correction': 2.0,
                               'hbm_bytes': (2.0 * fkb + wkb) * 1024.0, 'ms': ft[k][1] / n / 1e6}
        res = dict(previous(out), **res)        # other workloads' entries are kept
        res[workload] = entry
        for k, v in entry.items():              # what bench.py looks up: <workload>_mfma -> hbm_bytes of the count kernel
            if k.startswith('k_count_mfma'):
                res[workload + '_mfma'] = dict(v, kernel=k)
            elif k == 'k_count':
                res.setdefault(workload + '_valu', dict(v, kernel=k))
        json.dump(res, open(out, 'w'), indent=1)
    elif mode == 'sq':
        workload, sdir, draws, rows, out = sys.argv[2:7]
        sv, st = counters(sdir)
        res = {'_how': 'rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU '
                       'SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace on `python3 bench.py --steps 1 --warmup 0 '
                       '--no-cpu-baseline --no-host-to-host`; valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz '
                       'x duration); active_lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; units = table draws '
                       '(k_perm_general) or 2 x 2 rows (k_perm_fast) of the run'}
        entry = {}
        for k, c in sv.items():
            n = max(st[k][0], 1)
            ms = st[k][1] / n / 1e6
            e = {'ms': ms, 'counters': {a: b / n for a, b in c.items()}}
            if c.get('SQ_INSTS_VALU') and ms > 0:
                e['valu_insts'] = c['SQ_INSTS_VALU'] / n
                e['valu_issue_frac'] = c['SQ_INSTS_VALU'] / n * 4 / (1024 * 2.4e9 * ms * 1e-3)
            if c.get('SQ_ACTIVE_INST_VALU'):
                e['active_lanes'] = c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU']
            name = short(k)
            if name == 'k_perm_general':
                e['units'] = float(draws)
            if name == 'k_perm_fast':
                e['units'] = float(rows)
            entry[name] = e
        res = dict(previous(out), **res)        # other workloads' entries are kept
        res[workload] = entry
        json.dump(res, open(out, 'w'), indent=1)
    elif mode == 'stats':
        tdir, out = sys.argv[2:4]
        agg = {}
        with open(find(tdir, 'kernel_trace.csv')) as f:
            for row in csv.DictReader(f):
                k = row['Kernel_Name']
                d = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
                a = agg.setdefault(k, [0, 0, 1 << 62, 0])
                a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        total = sum(a[1] for a in agg.values())
        with open(out, 'w') as f:
            w = csv.writer(f)
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
            for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, a[0], a[1], a[1] / a[0], 100.0 * a[1] / total, a[2], a[3]])


if __name__ == '__main__':
    main()
