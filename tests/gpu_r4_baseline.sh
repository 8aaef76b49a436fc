#!/bin/bash
# round-4 baseline on the round-3 tree: default bench (writes the per-shape contraction breakdown), op census, rocprofv3 kernel stats
mkdir -p gpurun_out; cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out
echo "=== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/r04_base_bench.log; grep '^{' $O/r04_base_bench.log | cut -c1-400
cp $O/bench_gemm_shapes.json $O/r04_base_gemm_shapes.json; cp $O/bench_gemm_breakdown.json $O/r04_base_gemm_breakdown.json
echo "=== census"; timeout 400 python tests/gpu_op_census.py 2>&1 | grep -v amdgpu.ids > $O/r04_base_op_census.log; head -8 $O/r04_base_op_census.log
echo "=== census gp"; timeout 400 python tests/gpu_op_census.py gp 2>&1 | grep -v amdgpu.ids > $O/r04_base_op_census_gp.log; head -8 $O/r04_base_op_census_gp.log
echo "=== prof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-profile-cycle > $GRAFT_REPO_ROOT/$O/r04_base_bench_under_rocprof.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/r04_base_kernel_stats.csv \;
echo "=== done"
