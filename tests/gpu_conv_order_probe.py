"""A/B probe of the 8-wave conv kernel's k-tile order on config-2 discriminator shapes (forward and data gradient run the
same kernel): run once as is (channel-chunk-major) and once with GG_CONV_TAP_MAJOR=1. Test infrastructure.
usage: python tests/gpu_conv_order_probe.py [tag]"""
import json
import os
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
    tag = sys.argv[1] if len(sys.argv) > 1 else ('tap' if os.environ.get('GG_CONV_TAP_MAJOR') else 'chunk')
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    b = 64      # merged D(fake)+D(real) batch of a D-step
    convs = [('D2.conv2', 2 * b, 64, 128, 128), ('D3.conv1', 4 * b, 32, 128, 256), ('D3.conv2', 4 * b, 32, 256, 256),
             ('D4.conv1', 8 * b, 16, 256, 512), ('D4.conv2', 8 * b, 16, 512, 512), ('D5.conv', 16 * b, 8, 512, 512),
             ('D6.conv', 16 * b, 4, 512, 512), ('D4.pred', 4 * b, 16, 512, 512)]
    rows = []
    for name, n, R, ci, co in convs:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, 9 * ci, device=dev) * 0.05).to(torch.bfloat16)
        flops = 2.0 * n * R * R * ci * co * 9
        y = K.conv2d_nhwc(x, w, ksize=3)
        t = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=3))
        rows.append(dict(layer=name, M=n * R * R, N=co, K=9 * ci, us=t * 1e3, TF=flops / t / 1e9,
                         checksum=float(y.float().abs().mean())))
        print(f"{tag:6s} {name:10s} M={n*R*R:7d} N={co:4d} K={9*ci:5d} {t*1e3:8.1f} us {flops/t/1e9:7.1f} TF", flush=True)
    out = ROOT / 'gpurun_out'
    out.mkdir(exist_ok=True)
    (out / f'conv_order_{tag}.json').write_text(json.dumps(rows, indent=1))


if __name__ == '__main__':
    main()
