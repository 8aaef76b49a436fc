#!/bin/bash
# Round 2, GPU call 4: plan sweep (tuning cache) + adaptive-conv forward sweep.
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step plan_sweep; timeout 900 python tests/gpu_plan_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/r2c4_plan_sweep.log | tail -80
step modconv_sweep; timeout 400 python tests/gpu_modconv_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/r2c4_modconv_sweep.log
step done
