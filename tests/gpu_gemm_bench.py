"""GPU micro-benchmark of the contraction kernels on the GigaGAN config-2 layer shapes (batch 32): every tile variant
(4-wave 128x{128,64,32}, 8-wave 256x{256,128}) timed with HIP events on the launch stream, cross-checked against the
128x128 variant. Run on the GPU box:  python tests/gpu_gemm_bench.py [--quick]  -> gpurun_out/gemm_bench.json
(test infrastructure: not part of the product path)."""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402


def time_ms(fn, iters=6, warmup=2):
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


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-12)).item()


def pmc_shapes():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    b = 32
    for name, n, R, ci, co in [('D4.conv2', 8 * b, 16, 512, 512), ('D3.conv2', 4 * b, 32, 256, 256), ('D5.conv', 16 * b, 8, 512, 512)]:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, 9 * ci, device=dev) * 0.05).to(torch.bfloat16)
        dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
        for _ in range(3):
            K.conv2d_nhwc(x, w, ksize=3)
            K.conv2d_wgrad_nhwc(x, dy, ksize=3)
        torch.cuda.synchronize()
        print(name, 'done', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--pmc-shapes', action='store_true',
                    help='a handful of launches of the dominant layers with the planned tiles (for rocprofv3 --pmc passes)')
    args = ap.parse_args()
    if args.pmc_shapes:
        return pmc_shapes()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    b = 32
    # (name, n_img, res, Cin, Cout, ksize, in_scale?)
    convs = [
        ('D3.conv1', 4 * b, 32, 128, 256, 3, False), ('D3.conv2', 4 * b, 32, 256, 256, 3, False),
        ('D3.pred', 2 * b, 32, 256, 256, 3, False),
        ('D4.conv1', 8 * b, 16, 256, 512, 3, False), ('D4.conv2', 8 * b, 16, 512, 512, 3, False),
        ('D4.pred', 4 * b, 16, 512, 512, 3, False),
        ('D5.conv', 16 * b, 8, 512, 512, 3, False), ('D5.pred', 8 * b, 8, 512, 512, 3, False),
        ('D6.conv', 16 * b, 4, 512, 512, 3, False),
        ('D2.conv1', 2 * b, 64, 64, 128, 3, False), ('D2.conv2', 2 * b, 64, 128, 128, 3, False),
        ('D1.conv2', b, 128, 64, 64, 3, False),
        ('G0.train', b, 4, 512, 1024, 3, True), ('G1.train', b, 8, 512, 1024, 3, True),
        ('G2.train', b, 16, 256, 512, 3, True), ('G3.train', b, 32, 128, 256, 3, True),
        ('G4.train', b, 64, 64, 128, 3, True),
        ('D3.ff1', 4 * b, 32, 256, 1024, 1, False), ('D4.ff2', 8 * b, 16, 2048, 512, 1, False),
    ]
    if args.quick:
        convs = convs[:6]
    rows = []
    for name, n, R, ci, co, ks, scaled in convs:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, ks * ks * ci, device=dev) * 0.05).to(torch.bfloat16)
        dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
        insc = (torch.rand(n, ci, device=dev) + 0.5) if scaled else None
        flops = 2.0 * n * R * R * ci * co * ks * ks
        ref_f = ref_w = None
        for tile in (1, 2, 4, 5, 6):
            if tile == 2 and co > 64:
                continue
            try:
                of = K.conv2d_nhwc(x, w, ksize=ks, in_scale=insc, force_tile=tile)
                ow = K.conv2d_wgrad_nhwc(x, dy, ksize=ks, in_scale=insc, force_tile=tile)
            except RuntimeError as e:
                print(name, tile, 'ERR', e)
                continue
            if ref_f is None:
                ref_f, ref_w = of, ow
            ef, ew = rel(of, ref_f), rel(ow, ref_w)
            tf = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=ks, in_scale=insc, force_tile=tile))
            tw = time_ms(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=ks, in_scale=insc, force_tile=tile))
            row = dict(layer=name, M=n * R * R, N=co, K=ks * ks * ci, tile=tile, fwd_us=tf * 1e3, fwd_TF=flops / tf / 1e9,
                       wgrad_us=tw * 1e3, wgrad_TF=flops / tw / 1e9, err_fwd=ef, err_wgrad=ew)
            rows.append(row)
            print(f"{name:10s} M={row['M']:7d} N={co:5d} K={row['K']:5d} tile={tile}  fwd {tf*1e3:8.1f} us {row['fwd_TF']:7.1f} TF"
                  f"   wgrad {tw*1e3:8.1f} us {row['wgrad_TF']:7.1f} TF   err {ef:.1e} {ew:.1e}", flush=True)
        # heuristic choice
        tf = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=ks, in_scale=insc))
        tw = time_ms(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=ks, in_scale=insc))
        K.plan_log = []
        K.conv2d_nhwc(x, w, ksize=ks, in_scale=insc); K.conv2d_wgrad_nhwc(x, dy, ksize=ks, in_scale=insc)
        plans, K.plan_log = K.plan_log, None
        print(f"{name:10s} heuristic: fwd {tf*1e3:8.1f} us {flops/tf/1e9:7.1f} TF   wgrad {tw*1e3:8.1f} us {flops/tw/1e9:7.1f} TF  plans {plans}", flush=True)
        rows.append(dict(layer=name, tile=0, fwd_us=tf * 1e3, fwd_TF=flops / tf / 1e9, wgrad_us=tw * 1e3, wgrad_TF=flops / tw / 1e9))
        del x, w, dy
    out = ROOT / 'gpurun_out'
    out.mkdir(exist_ok=True)
    (out / 'gemm_bench.json').write_text(json.dumps(rows, indent=1))


if __name__ == '__main__':
    main()
