#!/bin/bash
# round 3, GPU call 1: full -m gpu suite (new: config-4/5 parity at BASELINE dims, graph-vs-eager, in-graph gradient exchange),
# the per-layer adaptive-conv sweep, then the default bench line
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > $O/r3c3_pytest.log; tail -25 $O/r3c3_pytest.log
step sweep; timeout 900 python tests/gpu_modconv_layers.py --json $O/r3c3_modconv_layers.json 2>&1 | grep -v amdgpu.ids > $O/r3c3_modconv_layers.log; grep -E "best|modulation" $O/r3c3_modconv_layers.log
step bench; timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids > $O/r3c3_bench.log; grep '^{' $O/r3c3_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
print('modconv fwd', m.get('error') or (round(m['achieved'], 1), 'TF frac', round(m['frac'], 4), 'graph', m['graph_ms'], 'kernel', round(m['kernel_ms'], 3)))
for l in m.get('layers', []): print('   ', l['layer'], round(l['kernel_us'], 1), l['launches'])
" 2>&1 | tail -30
tail -5 $O/r3c3_bench.log | cut -c1-300
step done
