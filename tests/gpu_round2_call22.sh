#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|Error|MEASURED" | tail -8
step text; timeout 500 python bench.py --workload text --steps 8 --warmup 4 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r2c22_bench_text.log; grep -E "capture|eagerly" $O/r2c22_bench_text.log | head -3; grep '^{' $O/r2c22_bench_text.log | cut -c1-400
step text_eager; timeout 500 python bench.py --workload text --steps 8 --warmup 4 --no-graphs --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r2c22_bench_text_eager.log; grep '^{' $O/r2c22_bench_text_eager.log | cut -c1-300
step upsampler; timeout 500 python bench.py --workload upsampler --steps 8 --warmup 4 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r2c22_bench_upsampler.log; grep '^{' $O/r2c22_bench_upsampler.log | cut -c1-300
step done
