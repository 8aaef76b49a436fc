#!/bin/bash
# (historical) modulation-launch sample split sweep - the GG_MODW_BC knob it drove was removed after this measurement (profiles/r03_modw_bc_sweep.log) - then the default bench
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
O=gpurun_out
step() { echo "=== $1"; }
for bc in 32 16 8 4 2; do
  step "GG_MODW_BC=$bc"; GG_MODW_BC=$bc timeout 300 python tests/gpu_modconv_layers.py --only modulation --json $O/c_modw_$bc.json 2>&1 | grep "modulation of"
done
summ='
import sys, json
d = json.loads(sys.stdin.read()); m = d["roofline"]["modconv_forward"]
print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms; dominant", round(d["roofline"]["achieved"], 1), "TF; all gemm", round(d["roofline"]["all_gemm_kernels"]["tflops"], 1), "TF", round(d["roofline"]["all_gemm_kernels"]["ms_per_step"], 2), "ms; modconv graph", round(m["graph_ms"], 4), "kernel", round(m["kernel_ms"], 4), "frac", round(m["frac"], 4))
for L in m["layers"]: print("   ", L["layer"], round(L["kernel_us"], 1), L["launches"])
'
for bc in 32 8; do
step "bench GG_MODW_BC=$bc"; GG_MODW_BC=$bc timeout 500 python bench.py --no-cpu-baseline --steps 8 2>&1 | grep '^{' | python -c "$summ" | head -3
done
step done
