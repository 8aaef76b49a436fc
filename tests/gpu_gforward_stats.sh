#!/bin/bash
# rocprofv3 kernel stats of no-grad generator forwards at config 2 / batch 32 (the adaptive-conv forward's kernels with exact durations)
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rm -rf /tmp/gfs && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gfs -o g -- python $R/tests/gpu_gforward_profile.py 8 > /tmp/gfs.log 2>&1 )
find /tmp/gfs -name '*kernel_stats.csv' -exec cp {} gpurun_out/gforward_kernel_stats.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/gforward_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('8 forwards: kernel time', round(tot / 1e6, 2), 'ms =', round(tot / 8e3, 1), 'us per forward;', sum(int(r['Calls']) for r in rows) // 8, 'launches per forward')
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}% x{int(r['Calls']) // 8:3d}/fwd {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:100]}")
PY
