#!/bin/bash
# same-box A/B of the 16-byte staged write-back of the 8-wave tiles (GG_WB_NARROW=1 = the 8-byte lanes of rounds 2-3), two alternations
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for i in 1 2; do
  for v in 1 0; do
    export GG_WB_NARROW=$v
    timeout 300 python bench.py --steps 24 --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' | python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print('GG_WB_NARROW=' + os.environ.get('GG_WB_NARROW', '0'), round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms')"
  done
done
