"""GPU probe of the HBM-bound helper kernels around the contractions at config-2 sizes: achieved bytes/s of bias/activation
backward, split-K reduce (inside conv2d_wgrad_nhwc), weight-gradient finish and the weight re-pack.
Run on the GPU box:  python tests/gpu_stream_probe.py [other_library.so]   (test infrastructure: not part of the product path)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import _C   # noqa: E402
if len(sys.argv) > 1:
    _C.bind(sys.argv[1])           # another build of the library (same-box A/B)
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
    print('--- bias_act_bwd (dy, y -> dz, db): 6 B per element')
    for R, Cc in [(256, 32), (128, 64), (64, 128), (32, 256), (16, 512), (8, 512)]:
        for n in (b, 4 * b):
            dy = torch.randn(n * R * R, Cc, device=dev).to(torch.bfloat16)
            y = torch.randn(n * R * R, Cc, device=dev).to(torch.bfloat16)
            if dy.numel() > 2 ** 28:
                continue
            us = time_us(lambda: K.bias_act_bwd(dy, y, True, partials=True))
            print(f'rows {n * R * R:8d} C {Cc:4d}: {us:7.1f} us  {dy.numel() * 6 / us / 1e6:6.2f} TB/s', flush=True)
    print('--- exact GELU of the attention feed-forwards: forward 4 B, backward 6 B, second order 10 B per element')
    for rows, Cc in [(128 * 1024, 1024), (256 * 256, 2048), (32 * 1024, 1024)]:
        x = torch.randn(rows, Cc, device=dev).to(torch.bfloat16)
        dy = torch.randn(rows, Cc, device=dev).to(torch.bfloat16)
        g = torch.randn(rows, Cc, device=dev).to(torch.bfloat16)
        us = [time_us(lambda: K.gelu(x)), time_us(lambda: K.gelu(x, dy)), time_us(lambda: K.gelu(x, dy, g))]
        print(f'{rows:7d} x {Cc:5d}: ' + '  '.join(f'{nm} {u:7.1f} us {x.numel() * bpe / u / 1e6:5.2f} TB/s'
                                                    for nm, u, bpe in zip(('fwd', 'bwd', 'bwd2'), us, (4, 6, 10))), flush=True)
    print('--- weight gradient: GEMM + split-K reduce, then finish (fp32 (9C, O) -> (O, C, 9) accumulate)')
    for n, R, ci, co in [(8 * b, 16, 512, 512), (4 * b, 32, 256, 256), (16 * b, 8, 512, 512), (2 * b, 64, 128, 128), (b, 128, 64, 64)]:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
        K.plan_log = []
        g = K.conv2d_wgrad_nhwc(x, dy, ksize=3)
        plan, K.plan_log = K.plan_log[-1], None
        us_g = time_us(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=3))
        out = torch.zeros(co, ci, 9, device=dev)
        us_f = time_us(lambda: K.wgrad_finish(g, co, ci, 9, out=out, accumulate=True))
        fl = 2.0 * n * R * R * 9 * ci * co
        print(f'{ci:4d}->{co:4d} @{R:3d} plan {plan}: gemm+reduce {us_g:7.1f} us ({fl / us_g / 1e6:5.0f} TF)  finish {us_f:6.1f} us '
              f'({g.numel() * 12 / us_f / 1e6:5.2f} TB/s of 12 B/elem)', flush=True)
    print('--- weight re-pack (fp32 (O, I, 9) -> bf16 fwd + bwd operands): 12 B per weight')
    tab = K.PackTable(dev, capacity=256)
    tot = 0
    for co, ci, reps in [(512, 512, 10), (256, 256, 6), (512, 256, 4), (128, 128, 6), (64, 64, 6), (1024, 512, 4)]:
        for _ in range(reps):
            w = torch.randn(co, ci, 9, device=dev)
            tab.register(w, co, ci, 9, 'fwd'); tab.register(w, co, ci, 9, 'bwd')
            tot += w.numel()
    us = time_us(tab.refresh)
    print(f'{tot / 1e6:.1f} M weights, {tab.n} entries: {us:7.1f} us  {tot * 12 / us / 1e6:5.2f} TB/s', flush=True)
    tab, tot = K.PackTable(dev, capacity=256), 0           # the 1x1 part of the discriminator: attention projections, feed-forwards
    for co, ci, reps in [(2048, 512, 4), (512, 2048, 4), (512, 512, 16), (1024, 256, 4), (256, 1024, 4), (256, 256, 16)]:
        for _ in range(reps):
            w = torch.randn(co, ci, 1, device=dev)
            tab.register(w, co, ci, 1, 'fwd'); tab.register(w, co, ci, 1, 'bwd')
            tot += w.numel()
    us = time_us(tab.refresh)
    print(f'{tot / 1e6:.1f} M weights, {tab.n} entries: {us:7.1f} us  {tot * 12 / us / 1e6:5.2f} TB/s', flush=True)


if __name__ == '__main__':
    main()
