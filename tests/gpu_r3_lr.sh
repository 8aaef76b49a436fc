#!/bin/bash
# low-resolution kernel + narrow conv3 tile sweeps; repeated runs of the upsampler GPU tests (an intermittent abort was seen once)
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step "bank layers: lowres kernel"; timeout 300 python tests/gpu_modconv_layers.py --only bank --json $O/lr_bank.json 2>&1 | grep -v amdgpu.ids > $O/lr_bank.log; grep -n "best\|lowres\|planner" $O/lr_bank.log | cut -c1-170
for nv in 0 1; do
  step "pimg layers GG_CONV3_NARROW=$nv"; GG_CONV3_NARROW=$nv timeout 300 python tests/gpu_modconv_layers.py --only pimg --json $O/lr_pimg_$nv.json 2>&1 | grep -v amdgpu.ids > $O/lr_pimg_$nv.log; grep -n "best\|planner\|tile 8 sk 1" $O/lr_pimg_$nv.log | cut -c1-170
done
step "new-kernel GPU tests"; timeout 300 python -m pytest tests/test_gpu_passes.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3 4 5 6; do
  step "unet loop $i"; timeout 300 python -X faulthandler -m pytest tests/test_unet_upsampler.py tests/test_text_conditional.py -m gpu -q -p no:xdist > $O/lr_loop_$i.log 2>&1; rc=$?; tail -1 $O/lr_loop_$i.log | cut -c1-150
  if [ $rc -ne 0 ]; then
    grep -v "^  File" $O/lr_loop_$i.log | head -30 | cut -c1-200
    step "rocgdb"; timeout 600 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args python -m pytest tests/test_unet_upsampler.py -m gpu -q -p no:xdist > $O/lr_gdb.log 2>&1; tail -60 $O/lr_gdb.log | cut -c1-200
    break
  fi
done
step done
