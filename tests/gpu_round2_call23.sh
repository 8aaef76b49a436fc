#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step prof_text
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $GRAFT_REPO_ROOT/bench.py --workload text --steps 8 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r2c23_prof_text.log 2>&1 )
find /tmp/prof_t -name '*kernel_stats.csv' -exec cp {} $O/r2c23_text_kernel_stats.csv \;
grep '^{' $O/r2c23_prof_text.log | cut -c1-200
step prof_ups
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_u -o u -- python $GRAFT_REPO_ROOT/bench.py --workload upsampler --steps 8 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r2c23_prof_ups.log 2>&1 )
find /tmp/prof_u -name '*kernel_stats.csv' -exec cp {} $O/r2c23_ups_kernel_stats.csv \;
grep '^{' $O/r2c23_prof_ups.log | cut -c1-200
step done
