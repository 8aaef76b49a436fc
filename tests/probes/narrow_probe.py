"""narrow high-resolution layers (N <= 64 channels at 128x128 / 256x256): the 4-wave implicit-GEMM tile vs the direct convolution (tile 9), forward and
weight gradient, HIP-event timed (test infrastructure)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from gigagan_pytorch_amd import kernels as K
dev = torch.device('cuda', 0)

def time_us(fn, iters=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for n, R, ci, co in ((64, 256, 32, 32), (32, 256, 32, 32), (64, 128, 64, 64), (32, 128, 64, 64), (64, 256, 32, 8), (32, 256, 64, 16), (64, 256, 16, 32)):
    x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, 9 * ci, device=dev) * 0.05).to(torch.bfloat16)
    dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
    bias = torch.randn(co, device=dev)
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    base = 3 if co <= 32 else 2
    for tile in (base, 9):
        tf = time_us(lambda: K.conv2d_nhwc(x, w, ksize=3, bias=bias, act='lrelu', force_tile=tile))
        tw = time_us(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=tile if tile != 9 else 0))
        print(f'n={n} R={R} ci={ci} co={co} tile {tile}: fwd {tf:7.1f} us ({gb/tf*1e3:5.2f} TB/s)  wgrad {tw:7.1f} us ({gb/tw*1e3:5.2f} TB/s)', flush=True)
