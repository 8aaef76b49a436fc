#!/bin/bash
# round 3, GPU call 2: -m gpu suite again (tolerances fixed, new conv3 / poolhf / multi-modw GPU tests), bench with the Gram cache
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r3c2}
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > $O/${T}_pytest.log; tail -12 $O/${T}_pytest.log
step bench; timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/${T}_bench.log; grep '^{' $O/${T}_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:50], round(d['roofline']['achieved'], 1), 'TF; all gemm', d['roofline']['all_gemm_kernels'])
print('modconv fwd', m.get('error') or (round(m['achieved'], 1), 'TF frac', round(m['frac'], 4), 'graph', m['graph_ms'], 'kernel', round(m['kernel_ms'], 3)))
for l in m.get('layers', []): print('   ', l['layer'], round(l['kernel_us'], 1), l['launches'])
" 2>&1 | tail -30
step done
