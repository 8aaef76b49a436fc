#!/bin/bash
# rocprofv3 kernel stats of the bench command (+ the op census of a plain and a GP step)
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r3prof}
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${T}_bench_under_rocprof.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/${T}_kernel_stats.csv \;
head -40 $O/${T}_kernel_stats.csv | cut -c1-160
grep '^{' $O/${T}_bench_under_rocprof.log | cut -c1-200
timeout 600 python tests/gpu_op_census.py > $O/${T}_census.log 2>&1; tail -60 $O/${T}_census.log | cut -c1-200
