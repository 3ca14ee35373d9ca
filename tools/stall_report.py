"""Print the wave-cycle breakdown of the kernels of interest from a summarize_counters.py JSON (tools/gpu_visit.sh stall).
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md §PMC):
  parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (s_waitcnt / barrier: memory or LDS latency the wave sits out)
  issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (dependency / pipe conflicts at issue)
  active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  waves_per_simd = SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES-equivalent) — reported as resident waves = WAVE_CYCLES / (GRBM/8 * 1024 / 4)"""
import json
import sys

d = json.load(open(sys.argv[1]))
want = sys.argv[2:] or ['k_trip_fwd', 'k_trip_bwd', 'k_chainr_fwd', 'k_chainr_bwd', 'k_featconv', 'k_front', 'k_wide', 'k_basis', 'k_wgrad_many',
                        'k_linear_fwd', 'k_linear_bwd']
for k, v in d.items():
    if not any(k.startswith(w) for w in want):
        continue
    c = v['counters']
    wc = c.get('SQ_WAVE_CYCLES')
    row = {'n': v['dispatches']}
    if wc:
        for nm, key in (('parked', 'SQ_WAIT_ANY'), ('issue_stall', 'SQ_WAIT_INST_ANY'), ('active', 'SQ_ACTIVE_INST_ANY'),
                        ('valu_active', 'SQ_ACTIVE_INST_VALU'), ('lds_stall', 'SQ_WAIT_INST_LDS'), ('vmem_active', 'SQ_ACTIVE_INST_VMEM'),
                        ('lds_active', 'SQ_ACTIVE_INST_LDS'), ('sca_active', 'SQ_ACTIVE_INST_SCA')):
            if key in c:
                row[nm] = round(c[key] / wc, 3)
        if c.get('GRBM_GUI_ACTIVE'):
            # quad-cycles of all waves / quad-cycles one SIMD offers during the kernel = average resident waves per SIMD
            row['waves_per_simd'] = round(wc / (c['GRBM_GUI_ACTIVE'] / 8.0 / 4.0 * 1024), 2)
            row['us'] = v.get('kernel_us_at_2.4GHz', round(c['GRBM_GUI_ACTIVE'] / 8.0 / 2400.0, 2))
    if c.get('SQ_WAVES'):
        row['waves'] = int(c['SQ_WAVES'])
        for nm, key in (('valu/wave', 'SQ_INSTS_VALU'), ('vmem_rd/wave', 'SQ_INSTS_VMEM_RD'), ('lds/wave', 'SQ_INSTS_LDS'),
                        ('salu/wave', 'SQ_INSTS_SALU'), ('smem/wave', 'SQ_INSTS_SMEM'), ('mfma/wave', 'SQ_INSTS_MFMA'),
                        ('fma_f32/wave', 'SQ_INSTS_VALU_FMA_F32'), ('mul_f32/wave', 'SQ_INSTS_VALU_MUL_F32')):
            if key in c:
                row[nm] = round(c[key] / c['SQ_WAVES'], 1)
    if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
        row['lds_conflict_frac'] = round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 3)
    if 'mfma_busy_frac' in v:
        row['mfma_busy'] = v['mfma_busy_frac']
    print(k[:60], json.dumps(row))
