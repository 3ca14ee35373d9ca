"""Turn rocprofv3 ``--pmc`` counter CSVs (gpurun_out/pmc_*.csv, one counter group per pass) into a compact per-kernel
summary for profiles/: mean counter values per dispatch and the derived busy fractions

    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
                (GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed)

next to rocprofv3's own MfmaUtil / VALUBusy derived metrics (gfx94x formulas on gfx950, MI355X_MICROARCH.md §PMC).

    python tools/summarize_counters.py gpurun_out/pmc_A.csv gpurun_out/pmc_B.csv ... > profiles/r02_mfma_valu_counters.json
"""
import collections
import csv
import json
import sys

KEEP = ('k_linear_fwd', 'k_linear_bwd', 'k_linear_pw', 'k_linear_dd', 'k_chain_fwd', 'k_chain_bwd', 'k_chain_wgrad', 'k_chainr', 'k_front', 'k_featconv', 'k_basis_project', 'k_basis_wgrad', 'k_trip_fwd', 'k_trip_bwd',
        'k_seg_fused', 'k_segsum', 'k_wide', 'k_wgrad_many', 'k_radial', 'k_smallk', 'k_reduce_many', 'k_graphnorm', 'k_tripgeom', 'k_bessel_d', 'k_harm_d')


def short(name):
    return name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').strip()


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k.startswith(KEEP):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = dict(dispatches=len(next(iter(d.values()))), counters={c: round(x, 2) for c, x in m.items()})
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m.get('GRBM_GUI_ACTIVE'):
        per_xcd = m['GRBM_GUI_ACTIVE'] / 8.0
        e['mfma_busy_frac'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (per_xcd * 256 * 4), 4)
        e['kernel_us_at_2.4GHz'] = round(per_xcd / 2400.0, 2)
    if 'SQ_ACTIVE_INST_VALU' in m and m.get('SQ_WAVE_CYCLES'):
        e['valu_active_per_wave_cycle'] = round(m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES'], 4)
    out[k] = e
json.dump(out, sys.stdout, indent=1)
