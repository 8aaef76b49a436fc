"""GPU A/B of the nine-tap weight-gradient kernel (force_tile 10) against the planned implicit-GEMM weight gradient on the
config-2 discriminator shapes (batch 32), GEMM + split-K reduce timed together, results cross-checked.
Run on the GPU box:  python tests/gpu_wgrad9_ab.py   (test infrastructure: not part of the product path)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402


def time_us(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    b = 32
    shapes = [('D4.conv2', 8 * b, 16, 512, 512), ('D4.conv1', 8 * b, 16, 256, 512), ('D3.conv2', 4 * b, 32, 256, 256),
              ('D3.conv1', 4 * b, 32, 128, 256), ('D5.conv', 16 * b, 8, 512, 512), ('D4.pred', 4 * b, 16, 512, 512),
              ('D2.conv2', 2 * b, 64, 128, 128), ('D2.conv1', 2 * b, 64, 64, 128), ('D5.pred', 8 * b, 8, 512, 512)]
    for name, n, R, ci, co in shapes:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
        fl = 2.0 * n * R * R * 9 * ci * co
        K.plan_log = []
        ref = K.conv2d_wgrad_nhwc(x, dy, ksize=3)
        plan = K.plan_log[-1]
        us0 = time_us(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=3))
        row = [f'{name:9s} planned {plan}: {us0:7.1f} us {fl / us0 / 1e6:5.0f} TF']
        for sk in (0, 4, 8, 16):
            K.plan_log = []
            got = K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=10, force_splitk=sk)
            pl = K.plan_log[-1]
            err = ((got - ref).norm() / ref.norm()).item()
            us = time_us(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=10, force_splitk=sk))
            row.append(f'{pl}: {us:7.1f} us {fl / us / 1e6:5.0f} TF err {err:.0e}')
        K.plan_log = None
        print(' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
