#!/bin/bash
# PMC evidence for the adaptive-conv forward (the north star's explicit kernel target): HBM-side bytes (FETCH_SIZE, WRITE_SIZE) and
# matrix-pipe busy cycles of every kernel of a no-grad generator forward at config 2 / batch 32. Counters in their own passes with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass). Summary: gpurun_out/pmc_modconv.json
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  ( cd /tmp && rm -rf /tmp/pmcm_$i && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcm_$i -o p -- python $R/tests/gpu_gforward_profile.py 3 > /tmp/pmcm_$i.log 2>&1 )
  f=$(find /tmp/pmcm_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|gg_" "$f" > gpurun_out/pmc_modconv_$i.csv
  echo "pass $i ($grp): $(wc -l < gpurun_out/pmc_modconv_$i.csv 2>/dev/null) rows"
done
python - <<'PY'
import csv, json, collections
def rows(i):
    try:
        return list(csv.DictReader(open(f'gpurun_out/pmc_modconv_{i}.csv')))
    except FileNotFoundError:
        return []
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for i in (1, 2, 3):
    for r in rows(i):
        name = r.get('Kernel_Name', '')
        if not any(k in name for k in ('gg_sconv', 'gg_spair', 'gg_aconv', 'gg_lrconv', 'gg_conv3', 'gg_modw', 'gg_splitk', 'gg_gemm', 'gg_modulate')):
            continue
        key = (name.split('(')[0].replace('void ', ''), r.get('Grid_Size') or r.get('Grid_Size_X') or '')
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if i == 1:
            dur[key].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
out = []
for key, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    rec = dict(kernel=key[0], grid=key[1], dispatches=max(len(v) for v in c.values()))
    if 'FETCH_SIZE' in m:
        rec['fetch_MB_x2'] = round(m['FETCH_SIZE'] * 2 * 1024 / 1e6, 2)       # KiB, doubled per the gfx950 correction
    if 'WRITE_SIZE' in m:
        rec['write_MB'] = round(m['WRITE_SIZE'] * 1024 / 1e6, 2)
    if 'SQ_BUSY_CYCLES' in m and m['SQ_BUSY_CYCLES'] > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        rec['mfma_busy_frac_of_sq_busy'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES'], 4)
        rec['SQ_VALU_MFMA_BUSY_CYCLES'] = m['SQ_VALU_MFMA_BUSY_CYCLES']
        rec['GRBM_GUI_ACTIVE'] = m.get('GRBM_GUI_ACTIVE')
    if key in dur:
        rec['us_under_pmc'] = round(sum(dur[key]) / len(dur[key]), 1)
        if 'fetch_MB_x2' in rec and 'write_MB' in rec:
            rec['hbm_side_TBps'] = round((rec['fetch_MB_x2'] + rec['write_MB']) / rec['us_under_pmc'], 3)       # MB / us = TB/s
    out.append(rec)
out.sort(key=lambda r: -(r.get('us_under_pmc') or 0))
json.dump(out, open('gpurun_out/pmc_modconv.json', 'w'), indent=1)
for r in out[:28]:
    print(r)
PY
