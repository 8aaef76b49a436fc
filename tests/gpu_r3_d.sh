#!/bin/bash
# per-image bank mix on the 16x16 layers, default bench, full -m gpu suite
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
O=gpurun_out
step() { echo "=== $1"; }
step "bank sweep"; timeout 300 python tests/gpu_modconv_layers.py --only bank --quick --json $O/d_bank.json 2>&1 | grep -v amdgpu.ids > $O/d_bank.log; grep "best\|mix\|planner" $O/d_bank.log | cut -c1-170
summ='
import sys, json
d = json.loads(sys.stdin.read()); m = d["roofline"]["modconv_forward"]
print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms; dominant", round(d["roofline"]["achieved"], 1), "TF; all gemm", round(d["roofline"]["all_gemm_kernels"]["tflops"], 1), "TF", round(d["roofline"]["all_gemm_kernels"]["ms_per_step"], 2), "ms; modconv graph", round(m["graph_ms"], 4), "kernel", round(m["kernel_ms"], 4), "frac", round(m["frac"], 4))
for L in m["layers"]: print("   ", L["layer"], round(L["kernel_us"], 1), L["launches"])
print("cpu", d.get("cpu_baseline"))
'
step bench; timeout 500 python bench.py 2>&1 | grep -v amdgpu.ids > $O/d_bench.log; grep '^{' $O/d_bench.log | python -c "$summ"
step pytest; timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:xdist 2>&1 | grep -v amdgpu.ids > $O/d_pytest.log; tail -4 $O/d_pytest.log | cut -c1-200
step done
