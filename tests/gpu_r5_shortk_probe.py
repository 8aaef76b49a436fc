"""GPU probe (round 5): where does a short-K 1x1 contraction spend its time? Time against K at fixed M x N (the intercept is the per-tile
cost outside the k-loop: prologue, first-tile latency, epilogue), plain GEMM and 1x1 conv gather, the 256x256 and 128x128 tiles, bf16
and no output (N tiny) - hipGraph-timed.   python tests/gpu_r5_shortk_probe.py
(test infrastructure: not part of the product path)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gpu_r5_aconv_probe import time_us   # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    M, N = 131072, 1024
    for tile in (4, 6):
        for conv in (False, True):
            row = [f'tile {tile} {"1x1 conv" if conv else "gemm    "} M={M} N={N}']
            for Kd in (64, 128, 256, 512, 1024, 2048):
                a = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
                w = (torch.randn(N, Kd, device=dev) * 0.05).to(torch.bfloat16)
                if conv:
                    x = a.view(128, 32, 32, Kd)
                    fn = lambda: K.conv2d_nhwc(x, w, ksize=1, force_tile=tile)
                else:
                    fn = lambda: K.gemm(a, w, force_tile=tile)
                us = time_us(fn, iters=5)
                row.append(f'K={Kd}: {us:6.1f} us')
            print(' | '.join(row), flush=True)
    # the output alone: the same bytes written by a copy kernel
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    src = torch.randn(M, N, device=dev).to(torch.bfloat16)
    print(f'copy of the {M}x{N} bf16 output: {time_us(lambda: out.copy_(src), iters=5):6.1f} us; fill: {time_us(lambda: out.zero_(), iters=5):6.1f} us', flush=True)


if __name__ == '__main__':
    main()
