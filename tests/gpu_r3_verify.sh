#!/bin/bash
# what the driver runs at round end, in its order: the -m gpu suite, smoke(), the default bench
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
O=gpurun_out
echo "=== pytest"; timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids > $O/verify_pytest.log; grep -n "passed\|failed\|Fatal\|Abort" $O/verify_pytest.log | tail -3; grep "^FAILED\|^ERROR" $O/verify_pytest.log | head
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
echo "=== bench"; timeout 500 python bench.py 2>&1 | grep -v amdgpu.ids > $O/verify_bench.log; grep '^{' $O/verify_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms; dominant', round(d['roofline']['achieved'], 1), 'TF frac', round(d['roofline']['frac'], 3), '; all gemm', round(d['roofline']['all_gemm_kernels']['tflops'], 1), 'TF; modconv graph', round(m['graph_ms'], 4), 'frac', round(m['frac'], 4), 'finite', d['finite'], 'graphs', d['config']['hip_graphs'])
print('cpu', {k: v for k, v in (d.get('cpu_baseline') or {}).items() if k != 'sample'})
"
grep -o '"shape_fallbacks": {[^}]*}' gpurun_out/c2_parity.json gpurun_out/c4_parity.json gpurun_out/c5_parity.json 2>/dev/null
echo "=== done"
