#!/bin/bash
# round 4: -m gpu suite + default bench on the current tree (records under gpurun_out/r04_<tag>_*)
mkdir -p gpurun_out; cd "$(dirname "$0")/.."; export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
T=${1:-check}; O=gpurun_out
echo "=== pytest"; timeout 1200 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids > $O/r04_${T}_pytest.log; tail -3 $O/r04_${T}_pytest.log
echo "=== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r04_${T}_bench.log
cp $O/bench_gemm_shapes.json $O/r04_${T}_gemm_shapes.json 2>/dev/null
grep '^{' $O/r04_${T}_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; m = r['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms; dominant', r['kernel'], round(r['achieved'], 1), 'TF', round(r['avg_launch_us'], 1), 'us; all gemm', round(r['all_gemm_kernels']['tflops'], 1), 'TF', round(r['all_gemm_kernels']['ms_per_step'], 2), 'ms/step; modconv graph', round(m['graph_ms'], 4), 'frac', round(m['frac'], 4), 'finite', d['finite'])
for k, v in list(r['gemm_kernel_table'].items())[:12]: print('   ', v, k)
"
echo "=== done"
