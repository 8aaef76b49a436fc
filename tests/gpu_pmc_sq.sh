#!/bin/bash
# SQ counters (own passes, kernel-trace only) of the halo-staged forward kernel and the nine-tap weight gradient on D's stage-4
# second conv (256 images x 16x16, 512 -> 512 channels); raw per-dispatch rows land in gpurun_out/pmc_sq_<kernel>_<group>.csv
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_LDS\|SQ_WAIT_INST_LDS\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" | sort -u > gpurun_out/pmc_lds_counter_names.txt
g=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  g=$((g+1))
  for k in "fwd 7" "wgrad 10"; do
    set -- $k
    ( cd /tmp && rm -rf /tmp/pmcq && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcq -o p -- python $R/tests/gpu_kernel_probe.py $1 256 16 512 512 3 $2 5 > /tmp/pmcq.log 2>&1 )
    f=$(find /tmp/pmcq -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|gg_conv3|gg_wgrad9" "$f" | cut -d, -f1-40 > gpurun_out/pmc_sq_$1_$g.csv
    echo "$1 group $g: $(wc -l < gpurun_out/pmc_sq_$1_$g.csv 2>/dev/null) rows"
  done
done
