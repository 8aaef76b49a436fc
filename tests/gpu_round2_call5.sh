#!/bin/bash
# Round 2, GPU call 5: new no-grad adaptive-conv path (gg_modw / gg_sconv), plan table installed, RCCL world-1, full suite + bench.
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1200 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v amdgpu.ids > $O/r2c5_pytest.log; tail -25 $O/r2c5_pytest.log
step bench; timeout 400 python bench.py 2>&1 | grep -v amdgpu.ids > $O/r2c5_bench.log; grep '^{' $O/r2c5_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
print('modconv fwd', round(m['achieved'], 1), 'TF frac', round(m['frac'], 4), round(m['kernel_ms'], 3), 'ms kernel', round(m['call_ms'], 3), 'ms calls')
for l in m['layers']: print('   ', l['layer'], round(l['kernel_us'], 1), 'us', round(l['kernel_tflops'], 1), 'TF', l['launches'])
print('cpu', d['cpu_baseline'])
"
step bench_noplan; GG_NO_PLAN_TABLE=1 timeout 300 python bench.py --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' | cut -c1-330
step done
