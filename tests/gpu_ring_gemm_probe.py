"""GPU race screen + timing of the experimental LDS-DMA ring GEMM (gg_gemm3.h, force_tile 7) against the planned 8-wave
kernel (tile 4) and fp32 torch: repeated runs on random data at sizes where stages stay in flight across many barriers.
Not part of the pytest suites: the kernel has not run on hardware yet (written after round 1's GPU budget was spent).
usage: python tests/gpu_ring_gemm_probe.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402


def time_ms(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    bad = 0
    for M, N, Kd in [(256, 256, 64), (512, 512, 4096), (4096, 4096, 4096), (131072, 512, 4608), (65536, 256, 2304),
                     (300, 520, 96), (8192, 8192, 8192)]:
        a = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
        b = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
        ref = K.gemm(a, b, out_dtype=torch.float32, force_tile=1)           # the validated 4-wave kernel
        worst = 0.
        for rep in range(20):                                               # race screen: same launch, many times
            out = K.gemm(a, b, out_dtype=torch.float32, force_tile=7)
            worst = max(worst, ((out - ref).norm() / ref.norm()).item())
        flops = 2.0 * M * N * Kd
        for rep in range(20):                                               # the staggered (ping-pong) variant
            out = K.gemm(a, b, out_dtype=torch.float32, force_tile=8)
            worst = max(worst, ((out - ref).norm() / ref.norm()).item())
        t7 = time_ms(lambda: K.gemm(a, b, force_tile=7))
        t8 = time_ms(lambda: K.gemm(a, b, force_tile=8))
        t4 = time_ms(lambda: K.gemm(a, b, force_tile=4)) if N >= 192 else float('nan')
        ok = worst < 1e-5
        bad += not ok
        print(f'M={M:6d} N={N:5d} K={Kd:5d}  ring {t7*1e3:8.1f} us {flops/t7/1e9:7.1f} TF   staggered {t8*1e3:8.1f} us {flops/t8/1e9:7.1f} TF   '
              f'tile4 {t4*1e3:8.1f} us {flops/t4/1e9:7.1f} TF   worst rel err over 40 runs {worst:.2e} {"ok" if ok else "MISMATCH"}', flush=True)
    # conv gather (forward / data gradient of the discriminator's 3x3 layers), same screen against the planned kernel
    for name, n, R, ci, co in [('D3.conv2', 256, 32, 256, 256), ('D4.conv2', 512, 16, 512, 512), ('D5.conv', 1024, 8, 512, 512),
                               ('D2.conv2', 128, 64, 128, 128), ('edge', 3, 20, 32, 40)]:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, 9 * ci, device=dev) * 0.05).to(torch.bfloat16)
        ref = K.conv2d_nhwc(x, w, ksize=3, out_dtype=torch.float32, force_tile=1)
        worst = 0.
        for rep in range(20):
            out = K.conv2d_nhwc(x, w, ksize=3, out_dtype=torch.float32, force_tile=7)
            worst = max(worst, ((out - ref).norm() / ref.norm()).item())
        flops = 2.0 * n * R * R * ci * co * 9
        for rep in range(20):
            out = K.conv2d_nhwc(x, w, ksize=3, out_dtype=torch.float32, force_tile=8)
            worst = max(worst, ((out - ref).norm() / ref.norm()).item())
        t7 = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=3, force_tile=7))
        t8 = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=3, force_tile=8))
        t0 = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=3))
        ok = worst < 1e-5
        bad += not ok
        print(f'{name:9s} M={n*R*R:7d} N={co:4d} K={9*ci:5d}  ring {t7*1e3:8.1f} us {flops/t7/1e9:7.1f} TF   staggered {t8*1e3:8.1f} us {flops/t8/1e9:7.1f} TF   planned '
              f'{t0*1e3:8.1f} us {flops/t0/1e9:7.1f} TF   worst rel err over 20 runs {worst:.2e} {"ok" if ok else "MISMATCH"}',
              flush=True)
    # weight gradients (reduction-major operands, transpose reads): same screen
    for name, n, R, ci, co in [('D3.conv2', 256, 32, 256, 256), ('D4.conv2', 512, 16, 512, 512), ('D5.conv', 1024, 8, 512, 512),
                               ('D2.conv2', 128, 64, 128, 128)]:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
        ref = K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=1)
        worst = 0.
        for rep in range(20):
            out = K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=7)
            worst = max(worst, ((out - ref).norm() / ref.norm()).item())
        flops = 2.0 * n * R * R * ci * co * 9
        t7 = time_ms(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=7))
        t0 = time_ms(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=3))
        ok = worst < 1e-5
        bad += not ok
        print(f'wgrad {name:9s} pixels={n*R*R:7d} {ci}->{co}  ring {t7*1e3:8.1f} us {flops/t7/1e9:7.1f} TF   planned '
              f'{t0*1e3:8.1f} us {flops/t0/1e9:7.1f} TF   worst rel err over 20 runs {worst:.2e} {"ok" if ok else "MISMATCH"}',
              flush=True)
    print('RACE SCREEN', 'PASSED' if not bad else f'FAILED ({bad} shapes)')


if __name__ == '__main__':
    main()
