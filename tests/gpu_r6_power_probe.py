"""GPU probe (round 6): is gg_conv3<256> bound by its structure or by the chip's power management? The same launch (D's stage-4 second
conv: 256 images x 16x16, 512 -> 512, 309 GFLOP) timed on operands of different bit activity: N(0,1) values as in training, one constant,
zeros. Same instruction stream, same bytes moved.    python tests/gpu_r6_power_probe.py     (test infrastructure)"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gpu_r5_aconv_probe import time_us   # noqa: E402

dev = torch.device('cuda', 0)
n, R, ci, co = 256, 16, 512, 512
fl = 2.0 * n * R * R * co * ci * 9
torch.manual_seed(0)
cases = {
    'N(0,1) x, N(0, 0.05) w (training-like)': (torch.randn(n, R, R, ci, device=dev), torch.randn(co, 9 * ci, device=dev) * 0.05),
    'x = 1.0, w = 1.0 (constant operands)': (torch.ones(n, R, R, ci, device=dev), torch.ones(co, 9 * ci, device=dev)),
    'x = 0, w = 0': (torch.zeros(n, R, R, ci, device=dev), torch.zeros(co, 9 * ci, device=dev)),
    'N(0,1) x, w = 0': (torch.randn(n, R, R, ci, device=dev), torch.zeros(co, 9 * ci, device=dev)),
}
for tile in (7, 10):
    for name, (x, w) in cases.items():
        xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
        if tile == 7:
            fn = lambda: K.conv2d_nhwc(xb, wb, ksize=3, force_tile=7)       # noqa: E731
        else:
            dy = (torch.randn(n, R, R, co, device=dev) if 'N(0,1) x, N' in name else xb[..., :co] * 1.0).to(torch.bfloat16).contiguous()
            fn = lambda: K.conv2d_wgrad_nhwc(xb, dy, ksize=3, force_tile=10)    # noqa: E731
        us = time_us(fn, iters=20)
        print(f"{'gg_conv3<256>' if tile == 7 else 'gg_wgrad9   '}  {name:42s} {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s", flush=True)
