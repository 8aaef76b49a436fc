#!/bin/bash
# default bench (adaptive-conv table), sconv strip-height sweep, 256x64 conv3 tile on the 16x16 / 32x32 layers
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
summ='
import sys, json
d = json.loads(sys.stdin.read()); m = d["roofline"]["modconv_forward"]
print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms; dominant", round(d["roofline"]["achieved"], 1), "TF; all gemm", round(d["roofline"]["all_gemm_kernels"]["tflops"], 1), "TF", round(d["roofline"]["all_gemm_kernels"]["ms_per_step"], 2), "ms; modconv graph", round(m["graph_ms"], 4), "kernel", round(m["kernel_ms"], 4), "frac", round(m["frac"], 4))
for L in m["layers"]: print("   ", L["layer"], round(L["kernel_us"], 1), L["launches"])
'
step bench; timeout 500 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/b_bench.log; grep '^{' $O/b_bench.log | python -c "$summ"
for r in 8 16 32; do
  step "sconv rows $r"; GG_SCONV_ROWS=$r timeout 300 python tests/gpu_modconv_layers.py --only sconv --json $O/b_sconv_$r.json 2>&1 | grep "best" | cut -c1-120
done
step "pimg tile 12"; timeout 300 python tests/gpu_modconv_layers.py --only pimg --json $O/b_pimg.json 2>&1 | grep -v amdgpu.ids > $O/b_pimg.log; grep "best\|planner\|tile 12\|tile 8 sk 1" $O/b_pimg.log | cut -c1-170
step "bank tile 12"; timeout 300 python tests/gpu_modconv_layers.py --only bank --json $O/b_bank.json 2>&1 | grep -v amdgpu.ids > $O/b_bank.log; grep "best\|planner\|tile 12" $O/b_bank.log | cut -c1-170
step "gpu tests (kernels)"; timeout 600 python -m pytest tests/test_gpu_passes.py tests/test_hip_parity.py -m gpu -q 2>&1 | tail -2
step done
