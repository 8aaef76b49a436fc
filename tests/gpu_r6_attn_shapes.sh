#!/bin/bash
# per-stage table of the fused self-attention kernels at the config-2 step's shapes (B = images x scales as the trainer batches them, n = pixels,
# 8 heads of 64): forward, backward (dq + dkv), second-order backward; ms and TFLOP/s. -> gpurun_out/r06_attn_shapes.log
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
for cfg in "32 1024 8" "64 1024 8" "128 1024 8" "256 1024 8" "32 256 8" "64 256 8" "128 256 8" "256 256 8" "512 256 8"; do
    echo "== B n heads = $cfg"
    timeout 120 python tests/gpu_attn_probe.py $cfg 2>&1 | grep -E "fwd|bwd"
done
} | tee gpurun_out/r06_attn_shapes.log
