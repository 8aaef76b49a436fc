#!/bin/bash
# Round 2, GPU call 2: replica bisect, the whole -m gpu suite, rocprofv3 kernel stats + SQ counters (csv output this time).
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step replica_bisect; timeout 200 python tests/gpu_replica_bisect.py 2>&1 | grep -v amdgpu.ids | tee $O/r2c2_replica_bisect.log | tail -40
step pytest; timeout 1200 python -m pytest tests -m gpu -q --durations=15 2>&1 | grep -v amdgpu.ids > $O/r2c2_pytest.log; tail -40 $O/r2c2_pytest.log
step bench; timeout 400 python bench.py 2>&1 | grep -v amdgpu.ids > $O/r2c2_bench.log; grep '^{' $O/r2c2_bench.log | cut -c1-400
step rocprof_stats
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 4 > $GRAFT_REPO_ROOT/$O/r2c2_bench_under_rocprof.log 2>&1 )
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $O/r2c2_kernel_stats.csv \; ; head -12 $O/r2c2_kernel_stats.csv | cut -c1-200; ls /tmp/prof_stats | head
step pmc
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof_pmc -o r2 -- python $GRAFT_REPO_ROOT/tests/gpu_gemm_bench.py --pmc-shapes > $GRAFT_REPO_ROOT/$O/r2c2_pmc_run.log 2>&1 )
find /tmp/prof_pmc -name '*counter_collection.csv' -exec cp {} $O/r2c2_pmc_sq_counter_collection.csv \; ; wc -l $O/r2c2_pmc_sq_counter_collection.csv; tail -5 $O/r2c2_pmc_run.log
step done
