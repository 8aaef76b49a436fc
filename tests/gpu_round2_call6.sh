#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 600 python -m pytest tests/test_gpu_passes.py tests/test_hip_parity.py tests/test_config2_parity.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -8
step bench; timeout 400 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r2c6_bench.log; grep '^{' $O/r2c6_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
print('modconv fwd', round(m['achieved'], 1), 'TF frac', round(m['frac'], 4), round(m['kernel_ms'], 3), 'ms kernel', round(m['call_ms'], 3), 'ms calls')
for l in m['layers']: print('   ', l['layer'], round(l['kernel_us'], 1), 'us', round(l['kernel_tflops'], 1), 'TF', l['launches'])
"
step done
