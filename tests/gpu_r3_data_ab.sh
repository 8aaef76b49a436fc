#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for mode in "resident" "loader --loader-workers 0" "loader --loader-workers 2" "resident" "loader --loader-workers 2" "loader --loader-workers 4"; do
  timeout 600 python bench.py --no-cpu-baseline --no-profile-cycle --data $mode 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$mode', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')" | tee -a gpurun_out/r3_data_ab2.log
done
