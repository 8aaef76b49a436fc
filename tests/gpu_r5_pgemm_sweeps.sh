#!/bin/bash
# the three plan sweeps of plan tile 15 (gg_pgemm) against the installed table, GG_PGEMM=0 so that the baseline is the plan without it
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for wl in uncond upsampler text; do
    GG_PGEMM=0 timeout 900 python tests/gpu_plan_sweep.py --workload $wl --tiles 15 --keep-table > gpurun_out/r05_plan_sweep_pgemm_$wl.log 2>&1
    tail -2 gpurun_out/r05_plan_sweep_pgemm_$wl.log
done
