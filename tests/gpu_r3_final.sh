#!/bin/bash
# round-3 closing evidence: -m gpu suite, rocprofv3 kernel stats of the bench command, default bench line, PMC traffic of the
# dominant kernel, op census of a plain and a gradient-penalty step
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
O=gpurun_out
T=${1:-r3final}
step() { echo "=== $1"; }
step pytest; timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:xdist 2>&1 | grep -v amdgpu.ids > $O/${T}_pytest.log; grep -n "passed\|failed" $O/${T}_pytest.log | tail -2; grep "^FAILED\|^ERROR" $O/${T}_pytest.log | head
step prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${T}_bench_under_rocprof.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/${T}_kernel_stats.csv \;
head -8 $O/${T}_kernel_stats.csv | cut -c1-150
summ='
import sys, json
d = json.loads(sys.stdin.read()); m = d["roofline"]["modconv_forward"]
print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms; dominant", round(d["roofline"]["achieved"], 1), "TF", round(d["roofline"]["avg_launch_us"], 1), "us; all gemm", round(d["roofline"]["all_gemm_kernels"]["tflops"], 1), "TF", round(d["roofline"]["all_gemm_kernels"]["ms_per_step"], 2), "ms; modconv graph", round(m["graph_ms"], 4), "kernel", round(m["kernel_ms"], 4), "frac", round(m["frac"], 4))
print("cpu", d.get("cpu_baseline"))
'
step bench; timeout 500 python bench.py 2>&1 | grep -v amdgpu.ids > $O/${T}_bench.log; grep '^{' $O/${T}_bench.log | python -c "$summ"
step pmc; timeout 400 bash tests/gpu_pmc_conv3.sh 2>&1 | tail -3
step census; timeout 400 python tests/gpu_op_census.py > $O/${T}_census.log 2>&1; grep -n "non-view torch ops" $O/${T}_census.log | head -4
step done
