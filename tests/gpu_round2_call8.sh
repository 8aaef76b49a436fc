#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python $GRAFT_REPO_ROOT/tests/gpu_gforward_profile.py 6 > $GRAFT_REPO_ROOT/$O/r2c8_gfwd.log 2>&1 )
find /tmp/prof_g -name '*kernel_stats.csv' -exec cp {} $O/r2c8_gfwd_kernel_stats.csv \; ; head -40 $O/r2c8_gfwd_kernel_stats.csv | cut -c1-220
find /tmp/prof_g -name '*kernel_trace.csv' -exec cp {} $O/r2c8_gfwd_kernel_trace.csv \; ; wc -l $O/r2c8_gfwd_kernel_trace.csv
