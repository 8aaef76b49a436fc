#!/bin/sh
# First GPU call of the next round: race-screen the experimental kernels written after round 1's GPU budget was spent, then
# A/B every default-off switch against the measured configuration with the same bench command on the same box.
#   gpurun --timeout 600 -- 'sh tests/gpu_next_round_ab.sh'
# Results land in gpurun_out/next_round_ab.log (copy the interesting parts into profiles/).
mkdir -p gpurun_out
{
  echo "== ring GEMM race screen + timing (gg_gemm3.h, force_tile 7)"
  timeout 120 python tests/gpu_ring_gemm_probe.py 2>&1 | grep -v amdgpu.ids
  for cfg in "" "GG_WGRAD_FUSED=1" "GG_MODCONV_NARROW=1 GG_MODCONV_PREMOD=1" "GG_GEMM_V3=1" "GG_GEMM_V3=1 GG_WGRAD_FUSED=1 GG_MODCONV_NARROW=1 GG_MODCONV_PREMOD=1"; do
    echo "== bench.py [$cfg]"
    env $cfg timeout 150 python bench.py --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' | cut -c1-260
  done
  echo "== modulated-conv forward roofline with and without the narrow direct path"
  for cfg in "" "GG_MODCONV_NARROW=1" "GG_MODCONV_PREMOD=1" "GG_MODCONV_NARROW=1 GG_MODCONV_PREMOD=1"; do
    env $cfg timeout 150 python bench.py --no-cpu-baseline --steps 4 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print('[$cfg]', d['value'], 'img/s; modconv fwd', round(m['achieved'], 1), 'TF', round(m['kernel_ms'], 3), 'ms kernel', round(m['call_ms'], 3), 'ms calls')
for l in m['layers']: print('   ', l['layer'], round(l['kernel_us'], 1), 'us', round(l['kernel_tflops'], 1), 'TF')
"
  done
} 2>&1 | tee gpurun_out/next_round_ab.log
