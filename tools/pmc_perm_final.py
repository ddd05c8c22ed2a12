#!/usr/bin/env python3
"""profiles/rNN_pmc_perm_general_final.json from the outputs of tools/profile_round.sh (SQ pass, profiles/rNN_pmc_sq_perm.json)
and tools/pmc_perm_l1.sh (gpurun_out/pmc_perm_l1_final.txt): python tools/pmc_perm_final.py r03"""
import ast
import json
import sys

R = sys.argv[1] if len(sys.argv) > 1 else 'r03'
sq = json.load(open('profiles/%s_pmc_sq_perm.json' % R))['north_star_dense_50kx200k']
pg = [v for k, v in sq.items() if 'k_perm_general' in k][0]
l1 = {}
for line in open('gpurun_out/pmc_perm_l1_final.txt'):
    if line[:2] in ('a ', 'b ', 'c ', 'd '):
        l1.update(ast.literal_eval(line[2:]))
draws = int(pg['units'])
cyc = pg['ms'] * 1e-3 * 2.4e9
out = {'_how': 'rocprofv3 --pmc (tools/pmc_perm_l1.sh passes a-d; SQ counters from tools/profile_round.sh, profiles/%s_pmc_sq_perm.json) '
               'on `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host`, north-star dense S=1000: '
               'k_perm_general as it is at the end of the round; assembled by tools/pmc_perm_final.py' % R,
       'ms': pg['ms'], 'sq_counters': pg['counters'], 'l1_counters': l1,
       'derived': {'table_draws': draws,
                   'l1_accesses_per_table_draw': l1['TCP_TOTAL_CACHE_ACCESSES_sum'] / draws,
                   'l1_accesses_per_cu_cycle': l1['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256 / cyc,
                   'l1_miss_frac': l1['TCP_TCC_READ_REQ_sum'] / l1['TCP_TOTAL_CACHE_ACCESSES_sum'],
                   'l1_tagconflict_stall_frac_of_gated_cycles': l1['TCP_READ_TAGCONFLICT_STALL_CYCLES_sum'] / l1['TCP_GATE_EN1_sum'],
                   'l1_pending_stall_frac_of_gated_cycles': l1['TCP_PENDING_STALL_CYCLES_sum'] / l1['TCP_GATE_EN1_sum'],
                   'l2_hit_frac': l1['TCC_HIT_sum'] / l1['TCC_REQ_sum'],
                   'valu_issue_frac': pg['valu_issue_frac'], 'active_lanes': pg['active_lanes'],
                   'valu_insts_per_table_draw': pg['counters']['SQ_INSTS_VALU'] * pg['active_lanes'] / draws,
                   'wave_valu_insts_per_table_draw': pg['counters']['SQ_INSTS_VALU'] / draws,
                   'salu_insts_per_valu_inst': pg['counters']['SQ_INSTS_SALU'] / pg['counters']['SQ_INSTS_VALU']}}
json.dump(out, open('profiles/%s_pmc_perm_general_final.json' % R, 'w'), indent=1)
print(json.dumps(out['derived'], indent=1))
