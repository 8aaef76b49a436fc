#!/bin/bash
# Round 2, GPU call 3: replica probe (why copies inside a batch differ), staggered ring kernel race screen + timing.
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step replica_probe; timeout 200 python tests/gpu_replica_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r2c3_replica_probe.log
step ring_probe; timeout 300 python tests/gpu_ring_gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r2c3_ring_probe.log
step adamw; timeout 200 python -m pytest tests/test_gpu_passes.py -m gpu -q 2>&1 | tail -3
step done
