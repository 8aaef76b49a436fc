"""What the existing contraction kernels can do on the generator's 15 adaptive-conv forward shapes (config 2, batch 32) when tile
and split-K are chosen by exhaustive sweep instead of by the planner: the stacked-bank (shared weights, 2x flops) form on
pre-modulated activations, and the per-sample-weight form (batched GEMM, algorithmic flops) where the bank is small.
Test infrastructure.   python tests/gpu_modconv_sweep.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402


def time_us(fn, iters=8, warmup=2):
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
    b, N = 32, 2
    layers = [(512, 512, 4), (512, 512, 8), (512, 256, 16), (256, 256, 16), (256, 128, 32), (128, 128, 32), (128, 64, 64),
              (64, 64, 64), (64, 32, 128), (32, 32, 128), (32, 16, 256), (16, 16, 256)]
    for I, O, R in layers:
        alg = 2.0 * b * O * I * 9 * R * R
        x2 = torch.randn(b, R, R, N * I, device=dev).to(torch.bfloat16)
        wk = (torch.randn(max(O, 8), 9 * N * I, device=dev) * 0.05).to(torch.bfloat16)
        d = torch.rand(b, max(O, 8), device=dev) + 0.5
        nz = torch.randn(b * R * R, device=dev)
        nw = torch.randn(max(O, 8), device=dev) * 0.1
        res = []
        for tile in (0, 1, 2, 3, 4, 5, 6):
            for sk in (0, 1, 2, 4, 8, 16, 32, 64):
                if sk == 0 and tile != 0:
                    continue
                try:
                    K.plan_log = []
                    fn = lambda: K.conv2d_nhwc(x2, wk, ksize=3, out_scale=d, noise=nz, noise_w=nw, act='lrelu', force_tile=tile,
                                               force_splitk=sk)
                    fn()
                    plan = K.plan_log[0]
                    K.plan_log = None
                    if tile and plan[0] != tile:
                        continue
                    res.append((time_us(fn), tile, sk, plan))
                except RuntimeError:
                    K.plan_log = None
        res.sort()
        auto = [r for r in res if r[1] == 0 and r[2] == 0][0]
        best = res[0]
        print(f'{I:3d}->{O:3d}@{R:3d}  stacked: planner {auto[0]:7.1f} us plan {auto[3]}  best {best[0]:7.1f} us (tile {best[1]} sk {best[2]} -> {best[3]}) '
              f'{alg / best[0] / 1e6:6.1f} alg-TF; runners-up {[(round(t, 1), tl, s) for t, tl, s, _ in res[1:4]]}', flush=True)
        # per-sample weights: batched GEMM over images (conv gather takes batch == 1, so loop-free form = dense GEMM on im2col-free
        # 1x1 proxy is not available); estimate with the plain conv of ONE kernel of the bank at the same M (algorithmic flops)
        x1 = x2[..., :I].contiguous()
        w1 = wk[:, :9 * I].contiguous()
        res = []
        for tile in (0, 1, 2, 3, 4, 5, 6, 9):
            try:
                K.plan_log = []
                fn = lambda: K.conv2d_nhwc(x1, w1, ksize=3, out_scale=d, noise=nz, noise_w=nw, act='lrelu', force_tile=tile)
                fn()
                plan = K.plan_log[0]
                K.plan_log = None
                if tile and plan[0] != tile:
                    continue
                res.append((time_us(fn), tile, plan))
            except RuntimeError:
                K.plan_log = None
        res.sort()
        print(f'            single-kernel conv (algorithmic flops, shared weights): best {res[0][0]:7.1f} us (tile {res[0][1]} -> {res[0][2]}) '
              f'{alg / res[0][0] / 1e6:6.1f} alg-TF; {[(round(t, 1), tl) for t, tl, _ in res[1:4]]}', flush=True)
        del x2, wk, x1, w1


if __name__ == '__main__':
    main()
