"""the workload table of DESIGN.md §8 from the committed bench lines:  python tools/design_table.py r05"""
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else 'r05'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = [('north-star dense, S=1000', 'ns'), ('north-star dense, S=0', 'ns_s0'), ('cfg2 dense 10k×50k, S=1000', 'cfg2_dense_10kx50k'),
         ('cfg3 22 × (9,091 × 45,455), S=1000', 'cfg3_22x9091x45455'), ('cfg5 22 × (9,091 × 181,820), S=10,000', 'cfg5_dense_depthx4_S10000'),
         ('north-star banded, 25 blocks', 'north_star_banded_50kx200k'), ('footprints_20k (20,000 blocks, 1.14 M sites)', 'footprints_20k'),
         ('headroom 100k×200k, 60 GB budget, sequential shards', 'headroom_dense_100kx200k_60GB'),
         ('headroom 150k×200k (4.05e9 rows), sequential shards', 'headroom_dense_150kx200k')]
print('| workload | step | count | emit | perm 2×2 | perm exact | perm sampled | pairs/s | host → host (compact rows) |')
print('|---|---|---|---|---|---|---|---|---|')
for title, n in NAMES:
    x = json.load(open(os.path.join(ROOT, 'profiles', '%s_bench_%s.json' % (R, n))))
    st, h = x['stage_ms'], x.get('host_to_host')
    ex = st.get('ms_perm_exact', 0.0)
    step = ('**%.0f ms**' % x['ms_per_step']) if x['ms_per_step'] >= 10 else '%.2f ms' % x['ms_per_step']
    hh = ('%.0f ms (%.3g pairs/s), first call %.0f' % (h['ms'], h['site_pairs_per_s'], h['ms_first_call'])) if h else '—'
    print('| %s | %s | %.1f | %.1f | %.1f | %.1f | %.1f | %.3g | %s |' % (title, step, st['ms_count'], st['ms_emit'], st['ms_perm_fast'], ex,
                                                                          max(st['ms_perm_general'] - ex, 0.0), x['value'], hh))
