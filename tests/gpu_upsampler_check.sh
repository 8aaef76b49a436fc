#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1500 python -m pytest tests/test_unet_upsampler.py tests/test_gpu_passes.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|Error" | tail -5
step upsampler; timeout 500 python bench.py --workload upsampler --steps 8 --warmup 8 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/run_bench_upsampler.log; grep '^{' $O/run_bench_upsampler.log | cut -c1-300
step census; timeout 300 python tests/gpu_op_census.py upsampler 2>&1 | grep -v amdgpu.ids > $O/run_census_ups.log; sed -n 5,22p $O/run_census_ups.log | cut -c1-160
step done
