#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|Error" | tail -5
step bench; timeout 400 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r2c28_bench.log; grep '^{' $O/r2c28_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
"
step done
