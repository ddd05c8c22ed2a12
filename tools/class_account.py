#!/usr/bin/env python3
"""Summary of tools/class_account.sh: per kernel the dynamic VALU instruction counts by class, priced with measured
issue costs, against the SIMD cycles of the launch.   python tools/class_account.py WORKLOAD PASS_DIR OUT.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import counters, previous, short  # noqa: E402

# SIMD cycles one wave-instruction of the class occupies the vector issue port for, with >= 2 waves per SIMD competing
# (profiles/r03_ubench_isa.txt at 4 waves per SIMD, halved: its figures are per wave; r03_ubench_mix.txt agrees: 8 v_xor_b32
# 16.05 cycles, 8 v_fma_f64 26.08).  One wave per SIMD: 4.0 for every non-transcendental class.
COST = {'f64_add_mul_fma': 3.2, 'f64_trans': 11.7, 'int64': 3.7, 'int32': 2.6, 'f32_add_mul_fma': 2.6, 'f32_trans': 6.0,
        'cvt': 3.1, 'other_valu': 2.6}
COST_NOTE = ('int32 / f32 / other: between 2.0 (VOP2: v_xor_b32, v_mul_f32) and 3.1 (VOP3: v_mul_hi_u32, v_fma_f32, v_bfe_u32), 2.6 taken; '
             'f64 add / mul / fma 3.2; v_rcp_f64 11.7; v_mad_u64_u32 3.7; conversions 3.1; a scalar instruction beside vector ones '
             'adds 1.2 (8 v_fma_f64 + 8 s_add_u32: 35.8 against 26.1 cycles)')
SALU_COST = 1.2
SIMDS, CLK = 1024, 2.4e9


def main():
    wl, d, out = sys.argv[1:4]
    va, ta = counters(os.path.join(d, 'a'))
    vb, _ = counters(os.path.join(d, 'b'))
    vc, _ = counters(os.path.join(d, 'c'))
    entry = {}
    for k in va:
        n = max(ta[k][0], 1)
        ms = ta[k][1] / n / 1e6
        a = {c: v / n for c, v in va[k].items()}
        b = {c: v / n for c, v in vb.get(k, {}).items()}
        c = {c_: v / n for c_, v in vc.get(k, {}).items()}
        valu = a.get('SQ_INSTS_VALU', 0.0)
        if ms < 0.3 or not valu:
            continue
        cls = {'f64_add_mul_fma': a.get('SQ_INSTS_VALU_ADD_F64', 0) + a.get('SQ_INSTS_VALU_MUL_F64', 0) + a.get('SQ_INSTS_VALU_FMA_F64', 0),
               'f64_trans': a.get('SQ_INSTS_VALU_TRANS_F64', 0), 'int64': a.get('SQ_INSTS_VALU_INT64', 0),
               'int32': a.get('SQ_INSTS_VALU_INT32', 0), 'cvt': a.get('SQ_INSTS_VALU_CVT', 0),
               'f32_add_mul_fma': b.get('SQ_INSTS_VALU_ADD_F32', 0) + b.get('SQ_INSTS_VALU_MUL_F32', 0) + b.get('SQ_INSTS_VALU_FMA_F32', 0),
               'f32_trans': b.get('SQ_INSTS_VALU_TRANS_F32', 0)}
        cls['other_valu'] = max(valu - sum(cls.values()), 0.0)            # moves, selects, compares, shuffles (DPP), ...
        simd_cycles = SIMDS * CLK * ms * 1e-3
        cyc = {k2: v * COST[k2] for k2, v in cls.items()}
        salu = b.get('SQ_INSTS_SALU', 0.0)
        vec, sca = sum(cyc.values()), salu * SALU_COST
        e = {'ms': ms, 'wave_valu_insts': valu, 'wave_insts_by_class': {k2: round(v) for k2, v in cls.items()},
             'share_by_class': {k2: round(v / valu, 4) for k2, v in cls.items()},
             'issue_cycles_by_class_frac_of_simd_cycles': {k2: round(v / simd_cycles, 4) for k2, v in cyc.items()},
             'attributed_vector_issue_frac': vec / simd_cycles, 'attributed_scalar_issue_frac': sca / simd_cycles,
             'unattributed_frac': 1.0 - (vec + sca) / simd_cycles,
             'valu_issue_frac_at_4_cycles': valu * 4.0 / simd_cycles,
             'salu_insts': salu, 'smem_insts': b.get('SQ_INSTS_SMEM', 0.0), 'lds_insts': b.get('SQ_INSTS_LDS', 0.0),
             'vmem_rd_insts': b.get('SQ_INSTS_VMEM_RD', 0.0), 'vmem_wr_insts': c.get('SQ_INSTS_VMEM_WR', 0.0)}
        if c.get('SQ_ACTIVE_INST_VALU'):
            e['active_lanes'] = c.get('SQ_THREAD_CYCLES_VALU', 0.0) / c['SQ_ACTIVE_INST_VALU']
        if c.get('SQ_WAVE_CYCLES'):
            e['wave_cycles_waiting_frac'] = c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES']
            e['waves_resident_per_simd'] = c['SQ_WAVE_CYCLES'] / max(c.get('SQ_BUSY_CYCLES', 1.0), 1.0) / 4.0 / (SIMDS / 4 / 8) if False else None
        entry[short(k)] = e
    res = dict(previous(out))
    res['_how'] = ('tools/class_account.sh: three rocprofv3 --pmc passes (VALU class counters; f32 classes + scalar / LDS / VMEM; busy / wait '
                   'cycles) of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host`; per launch.  issue cycles = '
                   'wave-instructions x the class cost below / (1024 SIMDs x 2.4 GHz x duration)')
    res['_cost_simd_cycles_per_wave_instruction'] = COST
    res['_cost_note'] = COST_NOTE
    res[wl] = entry
    json.dump(res, open(out, 'w'), indent=1)
    for k, e in sorted(entry.items(), key=lambda kv: -kv[1]['ms']):
        print('%-18s %7.2f ms  vector %.2f scalar %.2f unattributed %.2f (4-cycle pricing %.2f)  lanes %.1f  f64 %.2f trans %.3f int32 %.2f int64 %.2f other %.2f'
              % (k, e['ms'], e['attributed_vector_issue_frac'], e['attributed_scalar_issue_frac'], e['unattributed_frac'],
                 e['valu_issue_frac_at_4_cycles'], e.get('active_lanes', 0), e['share_by_class']['f64_add_mul_fma'],
                 e['share_by_class']['f64_trans'], e['share_by_class']['int32'], e['share_by_class']['int64'], e['share_by_class']['other_valu']))


if __name__ == '__main__':
    main()
