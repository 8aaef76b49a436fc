#!/bin/bash
# round 5: rocprofv3 kernel statistics of (a) the bench's timed steps + eager cycle + adaptive-conv forward graph, (b) no-grad generator
# forwards alone (tests/gpu_gforward_profile.py); summaries land in gpurun_out/ (copied to profiles/r05_* by hand).
#   bash tests/gpu_r5_prof.sh <tag>
cd "$(dirname "$0")/.." || exit 1
tag=${1:-a}
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench_$tag -o bench -- python bench.py --steps 8 --warmup 8 --no-cpu-baseline > gpurun_out/r5_prof_bench_$tag.json 2> gpurun_out/r5_prof_bench_$tag.err
f=$(find /tmp/prof_bench_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r5_kernel_stats_$tag.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gf_$tag -o gf -- python tests/gpu_gforward_profile.py > gpurun_out/r5_gforward_$tag.log 2>&1
f=$(find /tmp/prof_gf_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r5_gforward_kernel_stats_$tag.csv
head -40 gpurun_out/r5_gforward_kernel_stats_$tag.csv | cut -c1-160
