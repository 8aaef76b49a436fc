#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6
step prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/run_prof_bench.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/run_bench_kernel_stats.csv \;
head -12 $O/run_bench_kernel_stats.csv | cut -c1-150
step bench; timeout 400 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/run_bench.log; grep '^{' $O/run_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
print('modconv fwd', round(m['achieved'], 1), 'TF frac', round(m['frac'], 4), 'graph', round(m['graph_ms'], 3), 'kernel', round(m['kernel_ms'], 3))
"
step done
