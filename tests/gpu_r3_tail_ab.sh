#!/bin/bash
# same-box A/B of the small-launch clean-up: nn.Linear on LinearFn (pack-table operand) + fp32 style -> modulation projection ("new")
# against the previous host code ("old": the same library, the two host paths patched back)
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
run() { python -c "
import sys; sys.path.insert(0, '.')
import gigagan_pytorch_amd.ops as o, gigagan_pytorch_amd.modules as m
if '$1' == 'old':
    o.HipOps.linear = lambda self, x, w, b=None, act=None: o.matmul_nt(x, w, None if b is None else b.float().contiguous(), act)
    m.Linear.forward = lambda self, x: o.impl.linear(x, self.weight, self.bias)
import bench, json, io, contextlib
sys.argv = ['bench.py', '--no-cpu-baseline', '--steps', '24', '--warmup', '8']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith('{')][-1])
print('$1', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms; dominant', round(d['roofline']['achieved'], 1), 'TF; all gemm', round(d['roofline']['all_gemm_kernels']['ms_per_step'], 2), 'ms')
" 2>&1 | grep -v amdgpu.ids | tail -1; }
run old; run new; run old; run new
echo "=== pytest"; timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:xdist 2>&1 | grep -v amdgpu.ids > gpurun_out/tail_pytest.log; grep -n "passed\|failed" gpurun_out/tail_pytest.log | tail -2; grep "^FAILED\|^ERROR" gpurun_out/tail_pytest.log | head
