#!/bin/bash
# HBM-side traffic and matrix-pipe activity of the round-4 streaming kernels (gg_wgrads, gg_sfwd): FETCH_SIZE / WRITE_SIZE / SQ counters
# in separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950 for wide coalesced reads).
mkdir -p gpurun_out; cd "$(dirname "$0")/.."; export TMPDIR=/tmp; R=$PWD
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && rm -rf /tmp/pmcw_$i && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcw_$i -o p -- python $R/tests/gpu_stream_layers.py 3 > /tmp/pmcw_$i.log 2>&1 )
  f=$(find /tmp/pmcw_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|gg_wgrads|gg_sfwd" "$f" > gpurun_out/r04_pmc_stream_$i.csv
  echo "pass $i: $(wc -l < gpurun_out/r04_pmc_stream_$i.csv 2>/dev/null) rows"; tail -2 /tmp/pmcw_$i.log | cut -c1-200
done
python - <<'PY'
import csv, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for i in (1, 2, 3):
    try:
        seq = collections.Counter()
        for r in csv.DictReader(open(f'gpurun_out/r04_pmc_stream_{i}.csv')):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            # the script launches 7 distinct shapes per repetition in a fixed order: dispatch order identifies the shape
            did = int(r['Dispatch_Id'])
            agg[(name, did)][r['Counter_Name']].append(float(r['Counter_Value']))
            agg[(name, did)]['us'] = [(float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3]
    except FileNotFoundError:
        pass
SHAPES = ['wgrads 32->32 3x3 @256 b64 (537 MB)', 'wgrads 8->32 3x3 @256 b64 (336 MB)', 'wgrads 64->64 3x3 @128 b64 (268 MB)',
          'wgrads 32->32 2x2s2 @256 b64 (336 MB)', 'sfwd 32->32 @256 b64 (537 MB)', 'sfwd 8->32 @256 b64 (336 MB)', 'sfwd 64->64 @128 b64 (268 MB)']
ALG = [537e6, 336e6, 268e6, 336e6, 537e6, 336e6, 268e6]
keys = sorted(agg, key=lambda k: k[1])
# dispatch ids differ between passes; group by order of appearance per pass instead
per_pass = {}
for i in (1, 2, 3):
    try:
        rows = list(csv.DictReader(open(f'gpurun_out/r04_pmc_stream_{i}.csv')))
    except FileNotFoundError:
        continue
    seen, seq = {}, []
    for r in rows:
        did = int(r['Dispatch_Id'])
        if did not in seen:
            seen[did] = len(seq); seq.append(did)
    per_pass[i] = (rows, seen)
out = []
for s, (shape, alg) in enumerate(zip(SHAPES, ALG)):
    rec = dict(shape=shape, algorithmic_bytes=alg)
    for i, (rows, seen) in per_pass.items():
        vals = collections.defaultdict(list)
        for r in rows:
            k = seen[int(r['Dispatch_Id'])]
            if k % 7 == s:
                vals[r['Counter_Name']].append(float(r['Counter_Value']))
                vals['us'].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
        for c, v in vals.items():
            rec[c if c != 'us' else f'us_pass{i}'] = round(sum(v) / len(v), 1)
    if 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
        rec['hbm_bytes'] = (2 * rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024        # KiB counters; FETCH_SIZE x 2 (gfx950 correction)
        rec['traffic_over_algorithmic'] = round(rec['hbm_bytes'] / alg, 3)
        rec['TBps'] = round(rec['hbm_bytes'] / rec.get('us_pass1', 1) / 1e6, 2)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in rec and 'us_pass3' in rec:
        rec['mfma_busy_frac'] = round(rec['SQ_VALU_MFMA_BUSY_CYCLES'] / (rec['us_pass3'] * 1e-6 * 2.0e9 * 1024), 3)
    out.append(rec)
    print(rec)
json.dump(out, open('gpurun_out/r04_pmc_stream_summary.json', 'w'), indent=1)
PY
