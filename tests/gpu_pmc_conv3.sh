#!/bin/bash
# HBM-side traffic of the halo-staged 3x3 kernel on D's stage-4 second conv: FETCH_SIZE and WRITE_SIZE in separate counter passes
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pmc_$c && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $R/tests/gpu_kernel_probe.py fwd 256 16 512 512 3 7 5 > /tmp/pmc_$c.log 2>&1 )
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python - "$f" $c <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gg_conv3' in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == sys.argv[2]]
print(sys.argv[2], 'launches', len(vals), 'mean', sum(vals) / max(1, len(vals)), 'min', min(vals) if vals else None, 'max', max(vals) if vals else None)
PY
done 2>&1 | tee gpurun_out/pmc_conv3.log
