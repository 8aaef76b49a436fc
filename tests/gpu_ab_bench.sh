#!/bin/bash
# same-box A/B of two builds of the library: gpurun_ab/libgigagan_amd_prev.so (built from an older commit) vs the in-tree one
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
run() { python -c "
import sys; sys.path.insert(0, '.')
from gigagan_pytorch_amd import _C
if '$1' != 'new': _C.bind('gpurun_ab/libgigagan_amd_prev.so')
import bench, json, io, contextlib
sys.argv = ['bench.py', '--no-cpu-baseline', '--steps', '24', '--warmup', '8']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][-1])
print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms; dominant', round(d['roofline']['achieved'], 1), 'TF; all gemm', round(d['roofline']['all_gemm_kernels']['ms_per_step'], 2), 'ms')
" 2>&1 | grep -v amdgpu.ids | tail -1; }
run prev; run new; run prev; run new
