"""gg_wgrads (plan tile 13, streaming thin weight gradient) against the 4-wave implicit-GEMM weight gradient it replaces, at the thin
layer shapes of the config-2 step (test infrastructure). Run with GG_WGRADS=0 so that the default plan is the old path; the new kernel
is selected with force_tile=13. Prints us per call (contraction + split-K finish), algorithmic TB/s, and the relative error of both
against fp32 math on the same bf16 operands.   GG_WGRADS=0 python tests/gpu_wgrads_ab.py"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402

dev = torch.device('cuda', 0)
SHAPES = [  # n, H, W, ci, co, ksize
    (64, 256, 256, 32, 32, 3), (64, 256, 256, 8, 32, 3), (64, 128, 128, 32, 64, 3), (64, 128, 128, 64, 64, 3),
    (32, 256, 256, 32, 32, 3), (32, 256, 256, 16, 32, 3), (32, 256, 256, 16, 16, 3), (32, 128, 128, 32, 64, 3), (32, 128, 128, 64, 64, 3),
    (32, 128, 128, 64, 32, 3), (64, 128, 128, 8, 32, 1), (32, 256, 256, 16, 8, 1), (64, 64, 64, 8, 64, 1), (32, 128, 128, 32, 8, 1),
    (64, 256, 256, 32, 32, 2), (64, 256, 256, 8, 32, -1), (64, 128, 128, 32, 64, -1),      # 2x2 / stride 2, 1x1 / stride 2 (ksize -1)
]


def timed(fn, iters=20):
    for _ in range(3):
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
  for n, H, W, ci, co, ks in SHAPES:
      stride, pad = (2, 0) if ks in (2, -1) else (1, ks // 2)
      ks = abs(ks)
      kw = dict(ksize=ks, stride=stride, pad=pad)
      torch.manual_seed(0)
      x = torch.randn(n, H, W, ci, device=dev).bfloat16()
      dy = torch.randn(n, H // stride, W // stride, co, device=dev).bfloat16()
      K.plan_log = []
      old = K.conv2d_wgrad_nhwc(x, dy, **kw)
      new = K.conv2d_wgrad_nhwc(x, dy, force_tile=13, **kw)
      plans = list(K.plan_log)
      K.plan_log = None
      # fp32 reference on a subset of images (full fp32 conv backward of 64 x 256 x 256 is slow): linearity lets us check a slice
      m = min(n, 4)
      w = torch.zeros(co, ci, ks, ks, device=dev, requires_grad=True)
      F.conv2d(x[:m].float().permute(0, 3, 1, 2), w, stride=stride, padding=pad).backward(dy[:m].float().permute(0, 3, 1, 2))
      want = w.grad.permute(2, 3, 1, 0).reshape(-1, co)
      got_m = K.conv2d_wgrad_nhwc(x[:m].contiguous(), dy[:m].contiguous(), force_tile=13, **kw)
      err_ref = float((got_m - want).norm() / want.norm())
      err_old = float((new - old).norm() / old.norm())
      t_old = timed(lambda: K.conv2d_wgrad_nhwc(x, dy, **kw))
      t_new = timed(lambda: K.conv2d_wgrad_nhwc(x, dy, force_tile=13, **kw))
      by = n * (H * W * ci + (H // stride) * (W // stride) * co) * 2
      print(f'{ci:3d}->{co:3d} k{ks}s{stride} @{H}x{W} b={n:2d}  old {plans[0]} {t_old:7.1f} us {by / t_old / 1e6:5.2f} TB/s   new {plans[1]} {t_new:7.1f} us '
            f'{by / t_new / 1e6:5.2f} TB/s   x{t_old / t_new:4.2f}   err vs fp32 {err_ref:.1e}  new vs old {err_old:.1e}', flush=True)


if __name__ == '__main__':
    main()
