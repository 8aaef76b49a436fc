"""summary of tests/gpu_pmc_sq.sh's per-dispatch counter rows (gpurun_out/pmc_sq_{fwd,wgrad}_{1,2,3}.csv) -> one json: MFMA-busy
fraction of the SIMD cycles, wait / issue fractions of the wave cycles, LDS bank-conflict share. Test infrastructure.
    python tests/gpu_pmc_sq_summary.py gpurun_out profiles/r06_pmc_sq_conv3.json"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
rows = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values
times = defaultdict(list)
for f in sorted(src.glob('pmc_sq_*_[0-9].csv')):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r['Kernel_Name'].replace('void ', '').split('(')[0]
            rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
            times[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = dict(source='gpurun_out/pmc_sq_{fwd,wgrad}_{1,2,3}.csv (rocprofv3 --pmc, three counter groups in their own passes with --kernel-trace '
                  'only; tests/gpu_pmc_sq.sh over tests/gpu_kernel_probe.py, D stage-4 second conv: 256 images x 16x16, 512 -> 512, 3x3)',
           units='SQ_VALU_MFMA_BUSY_CYCLES = 32 per v_mfma_f32_32x32x16_bf16 summed over all SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs; '
                 'SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles summed over waves', kernels={})
for k, c in rows.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8
    us = sorted(times[k])[len(times[k]) // 2]
    e = dict(launches=len(c.get('GRBM_GUI_ACTIVE', [])), kernel_cycles=round(cyc), launch_us_under_counters=round(us, 1),
             sclk_ghz=round(cyc / us / 1e3, 2) if us else None, mean_counters={n: round(v) for n, v in m.items()})
    if cyc and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        e['mfma_busy_frac_of_simd_cycles'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), 3)      # 256 CUs x 4 SIMDs
    w = m.get('SQ_WAVE_CYCLES')
    for name, key in (('SQ_WAIT_ANY', 'wait_any'), ('SQ_WAIT_INST_ANY', 'wait_inst_any'), ('SQ_ACTIVE_INST_ANY', 'active_inst_any'),
                      ('SQ_WAIT_INST_LDS', 'wait_inst_lds')):
        if w and name in m:
            e[f'{key}_frac_of_wave_cycles'] = round(m[name] / w, 3)
    if m.get('SQ_LDS_IDX_ACTIVE'):
        e['lds_bank_conflict_frac_of_lds_active'] = round(m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE'], 4)
    out['kernels'][k] = e
dst.write_text(json.dumps(out, indent=1))
print(json.dumps({k: {a: b for a, b in v.items() if a != 'mean_counters'} for k, v in out['kernels'].items()}, indent=1))
