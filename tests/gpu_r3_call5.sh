#!/bin/bash
# -m gpu suite + same-box A/B of the input leg: resident batches vs DataLoader + DevicePrefetcher
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r3c5}
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > $O/${T}_pytest.log; tail -8 $O/${T}_pytest.log
for mode in resident loader resident loader; do
  step "bench --data $mode"
  timeout 600 python bench.py --no-cpu-baseline --no-profile-cycle --data $mode 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$mode', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms', d['data'])" | tee -a $O/${T}_data_ab.log
done
step done
