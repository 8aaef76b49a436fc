"""GPU probe (round 6): would the 64x64 adaptive convs run faster on gg_aconv with ONE kernel per image (per-sample weights in
fragment order) than on the per-image implicit GEMM? Times gg_aconv_fwd with a single shared kernel (NB = 1: the same kernel work as a
per-image bank, the weight stream's L2 locality aside) on 128 -> 64 and 64 -> 64 at 64x64, batch 32, every tile shape.
    python tests/gpu_r6_aconv64_probe.py          (test infrastructure)"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gpu_r5_aconv_probe import time_us   # noqa: E402

dev = torch.device('cuda', 0)
b, R = 32, 64
for I, O in ((128, 64), (64, 64)):
    x = torch.randn(b, R, R, I, device=dev).to(torch.bfloat16)
    W = torch.randn(1, O, I, 3, 3, device=dev) / (3 * I ** 0.5)
    s = torch.ones(b, I, device=dev)
    nz, nw = torch.randn(b * R * R, device=dev), torch.randn(O, device=dev) * 0.3
    wf = K.frag_pack(W)
    print(f'{I}->{O}@64 b32: library plan {K.aconv_plan(b, R, R, I, O, 1)}', flush=True)
    for tm in (1, 2, 4):
        for nwn in (1, 2):
            try:
                us = time_us(lambda: K.aconv(x, wf, s, None, None, O, nz, nw, 'lrelu', 0.2, force_tm=tm, force_nwn=nwn))
                print(f'   TM {tm} NWN {nwn}: {us:.1f} us  ({2.0 * b * O * I * 9 * R * R / us / 1e6:.0f} TF/s)', flush=True)
            except Exception as e:
                print(f'   TM {tm} NWN {nwn}: {str(e)[:100]}')
