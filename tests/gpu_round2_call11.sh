#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 600 python -m pytest tests/test_gpu_passes.py tests/test_hip_parity.py -m gpu -q -k "adaptive or modconv or models or ops_match or trained" 2>&1 | grep -v amdgpu.ids | tail -4
step gfwd
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python $GRAFT_REPO_ROOT/tests/gpu_gforward_profile.py 6 > $GRAFT_REPO_ROOT/$O/r2c11_gfwd.log 2>&1 )
find /tmp/prof_g -name '*kernel_stats.csv' -exec cp {} $O/r2c11_gfwd_kernel_stats.csv \; ; head -16 $O/r2c11_gfwd_kernel_stats.csv | cut -c1-160
find /tmp/prof_g -name '*kernel_trace.csv' -exec cp {} $O/r2c11_gfwd_kernel_trace.csv \;
step bench; timeout 400 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r2c11_bench.log; grep '^{' $O/r2c11_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
print('modconv fwd', round(m['achieved'], 1), 'TF frac', round(m['frac'], 4), 'graph', round(m['graph_ms'], 3), 'kernel', round(m['kernel_ms'], 3), 'ms calls', round(m['call_ms'], 3))
for l in m['layers']: print('   ', l['layer'], round(l['kernel_us'], 1), 'us', round(l['kernel_tflops'], 1), 'TF', l['launches'])
"
step done
