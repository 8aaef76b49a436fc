#!/bin/bash
# HEAD evidence in one call: -m gpu suite, rocprofv3 kernel stats of the bench command, the default bench line, adaptive-conv layer sweep
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r3head}
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > $O/${T}_pytest.log; tail -5 $O/${T}_pytest.log
step prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${T}_bench_under_rocprof.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/${T}_kernel_stats.csv \;
head -14 $O/${T}_kernel_stats.csv | cut -c1-150
step bench; timeout 500 python bench.py 2>&1 | grep -v amdgpu.ids > $O/${T}_bench.log; grep '^{' $O/${T}_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
print('modconv fwd', round(m['achieved'], 1), 'TF frac', round(m['frac'], 4), 'graph', round(m['graph_ms'], 3), 'kernel', round(m['kernel_ms'], 3))
for L in m['layers']: print('  ', L['layer'], round(L['kernel_us'], 1), L['launches'])
print('cpu', d.get('cpu_baseline'))
"
step layers; timeout 400 python tests/gpu_modconv_layers.py --json $O/${T}_modconv_layers.json 2>&1 | grep -v amdgpu.ids > $O/${T}_modconv_layers.log; grep -n "best\|modulation of" $O/${T}_modconv_layers.log | cut -c1-160
step done
