"""Round-3 tuning / evidence script for the generator's no-grad adaptive convolutions (config 2, batch 32, N = 2 kernels): every
candidate formulation of every layer shape, each timed as a hipGraph replay (no host gaps) and checked against a reference
variant on the same operands. Test infrastructure (run on the GPU box):   python tests/gpu_modconv_layers.py [--json out.json]

  bank layers (4x4 .. 16x16: weights >> activations)
      old   gg_modulate_bank_fwd + shared-bank conv as planned
      insc  shared-bank conv with the modulation on the operand staging (in_scale, CV = 2C): planner's choice, and forced
            tiles 7 / 8 (gg_conv3 SCALED) x split-K ladder, tiles 4 / 5 / 6 (implicit GEMM, scale per tap)
  per-image-weight layers (32x32, 64x64)
      pimg  per-image weights: planner's choice, forced tiles 7 / 8 (gg_conv3) x split-K, 5 / 6 (implicit GEMM)
      insc  the stacked shared bank with in_scale (2x flops, no per-sample weights)
  streaming layers (128x128, 256x256): gg_sconv_fwd
  modulation: gg_modw_multi_fwd over the 12 non-excited layers vs one gg_modw_fwd per layer
"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K, ops   # noqa: E402

DEV = torch.device('cuda', 0)
B, N = 32, 2
# (I, O, resolution, excited): init conv + (conv1, conv2) of the 7 blocks of config 2's generator
LAYERS = [(512, 512, 4, False), (512, 512, 4, False), (512, 512, 4, False), (512, 512, 8, False), (512, 512, 8, False),
          (512, 256, 16, False), (256, 256, 16, False), (256, 128, 32, False), (128, 128, 32, False), (128, 64, 64, True),
          (64, 64, 64, False), (64, 32, 128, True), (32, 32, 128, False), (32, 16, 256, True), (16, 16, 256, False)]


def graph_us(fn, replays=10):
    """GPU time of fn()'s launches as one hipGraph replay (mean over `replays`)."""
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        out = fn()
    torch.cuda.current_stream(DEV).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / replays * 1e3, out


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-12))


def try_variant(name, fn, ref, results, flops):
    try:
        K.plan_log = []
        us, out = graph_us(fn)
        plan = K.plan_log[-1] if K.plan_log else None
        K.plan_log = None
        err = rel(out, ref) if ref is not None else 0.
        results.append(dict(variant=name, us=round(us, 1), tflops=round(flops / us / 1e6, 1), plan=plan, rel_err=round(err, 5)))
        return out
    except Exception as e:      # noqa: BLE001
        K.plan_log = None
        results.append(dict(variant=name, error=f'{type(e).__name__}: {str(e)[:120]}'))
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default=str(ROOT / 'gpurun_out' / 'modconv_layers.json'))
    ap.add_argument('--only', default='', help='comma-separated paths to run (bank, pimg, sconv); default: all')
    ap.add_argument('--quick', action='store_true', help='bank layers: reference + low-resolution kernel variants only')
    args = ap.parse_args()
    only = set(filter(None, args.only.split(',')))
    torch.manual_seed(0)
    report = {}
    seen = set()
    for I, O, R, excited in LAYERS:
        key = f'{I}->{O}@{R}x{R}'
        if key in seen:
            continue
        seen.add(key)
        flops = 2.0 * B * O * I * 9 * R * R
        x = torch.randn(B, R, R, I, device=DEV).to(torch.bfloat16)
        w = torch.randn(N, O, I, 3, 3, device=DEV) * (2.0 / (9 * I)) ** 0.5
        mod, kmod = torch.randn(B, I, device=DEV) * 0.3, torch.randn(B, N, device=DEV)
        nz, nw = torch.randn(B * R * R, device=DEV), torch.randn(O, device=DEV) * 0.1
        path = ops.HipOps._modconv_path(B, N, O, I, R, R)
        if only and path not in only:
            continue
        res = []
        if path == 'bank':
            s, a, d = K.modw_fwd(w, mod, kmod, True, 1e-8, I, O)
            insc = (a[:, :, None] * s[:, None, :]).reshape(B, N * I).contiguous()
            wk = w.permute(1, 3, 4, 0, 2).reshape(O, 9 * N * I).to(torch.bfloat16).contiguous()
            epi = dict(out_scale=d, noise=nz, noise_w=nw, act='lrelu')
            ref = try_variant('old: modulate_bank + conv (planner)',
                              lambda: K.conv2d_nhwc(K.modulate_bank(x, s, a), wk, ksize=3, **epi), None, res, flops)
            try_variant('old: conv only (planner), pre-modulated input', (lambda x2: (lambda: K.conv2d_nhwc(x2, wk, ksize=3, **epi)))(
                K.modulate_bank(x, s, a)), ref, res, flops)
            if R == 16:
                for sk in (1, 2, 4):
                    try_variant(f'lowres per-image bank mix sk {sk}',
                                (lambda k: (lambda: K.conv2d_nhwc(x, wk, ksize=3, cv=N * I, in_scale=s, bank_mix=a, force_splitk=k,
                                                                 **epi)))(sk), ref, res, flops)
            for sk in (2, 4, 8, 16):
                try_variant(f'lowres tile 11 sk {sk}',
                            (lambda k: (lambda: K.conv2d_nhwc(x, wk, ksize=3, cv=N * I, in_scale=insc, force_tile=11, force_splitk=k,
                                                             **epi)))(sk), ref, res, flops)
            if args.quick:
                pass
            elif R >= 8:
                try_variant('insc (planner)', lambda: K.conv2d_nhwc(x, wk, ksize=3, cv=N * I, in_scale=insc, **epi), ref, res, flops)
                for tile in (7, 8, 12, 4, 5, 6):
                    for sk in ((1, 2, 4, 8, 16, 32) if tile >= 7 else (0,)):
                        if tile == 7 and O < 192:
                            continue
                        try_variant(f'insc tile {tile} sk {sk}',
                                    (lambda t, k: (lambda: K.conv2d_nhwc(x, wk, ksize=3, cv=N * I, in_scale=insc, force_tile=t,
                                                                        force_splitk=k, **epi)))(tile, sk), ref, res, flops)
            else:
                for tile in (4, 5, 6):
                    for sk in (2, 4, 8, 16, 32):
                        try_variant(f'old conv tile {tile} sk {sk}',
                                    (lambda t, k, x2: (lambda: K.conv2d_nhwc(x2, wk, ksize=3, force_tile=t, force_splitk=k, **epi)))(
                                        tile, sk, K.modulate_bank(x, s, a)), ref, res, flops)
        elif path == 'pimg':
            wm = torch.empty(B, O, 9 * I, dtype=torch.bfloat16, device=DEV)
            K.modw_fwd(w, mod, kmod, True, 1e-8, I, O, coef=False, wmix=wm, layout=1)
            epi = dict(noise=nz, noise_w=nw, act='lrelu')
            ref = try_variant('pimg (planner)', lambda: K.conv2d_nhwc(x, wm, ksize=3, per_image_weights=True, **epi), None, res, flops)
            for tile in (7, 8, 12, 5, 6):
                for sk in ((1, 2, 4) if tile >= 7 else (0,)):
                    if tile == 7 and O < 192:
                        continue
                    try_variant(f'pimg tile {tile} sk {sk}',
                                (lambda t, k: (lambda: K.conv2d_nhwc(x, wm, ksize=3, per_image_weights=True, force_tile=t,
                                                                    force_splitk=k, **epi)))(tile, sk), ref, res, flops)
            if I % 64 == 0:
                s, a, d = K.modw_fwd(w, mod, kmod, True, 1e-8, I, O)
                insc = (a[:, :, None] * s[:, None, :]).reshape(B, N * I).contiguous()
                wk = w.permute(1, 3, 4, 0, 2).reshape(O, 9 * N * I).to(torch.bfloat16).contiguous()
                for tile in (8,):
                    for sk in (1, 2):
                        try_variant(f'insc stacked bank tile {tile} sk {sk}',
                                    (lambda t, k: (lambda: K.conv2d_nhwc(x, wk, ksize=3, cv=N * I, in_scale=insc, out_scale=d,
                                                                        force_tile=t, force_splitk=k, **epi)))(tile, sk), ref, res, flops)
        else:
            wm = torch.zeros(B, 9, I // 16, 32, 16, dtype=torch.bfloat16, device=DEV)
            K.modw_fwd(w, mod, kmod, True, 1e-8, I, O, coef=False, wmix=wm, layout=2)
            try_variant('sconv', lambda: K.sconv(x, wm, O, nz, nw, 'lrelu', 0.2), None, res, flops)
        try_variant('modw (this layer alone)', lambda: K.modw_fwd(w, mod, kmod, True, 1e-8, I, O)[2] if path == 'bank' else
                    K.modw_fwd(w, mod, kmod, True, 1e-8, I, O, coef=False,
                               wmix=(torch.empty(B, O, 9 * I, dtype=torch.bfloat16, device=DEV) if path == 'pimg' else
                                     torch.zeros(B, 9, I // 16, 32, 16, dtype=torch.bfloat16, device=DEV)),
                               layout=1 if path == 'pimg' else 2) or torch.zeros(1, device=DEV), None, res, flops)
        report[key] = dict(path=path, gflop=flops / 1e9, variants=res)
        best = min((r for r in res if 'us' in r and not r['variant'].startswith('modw')), key=lambda r: r['us'])
        print(f'{key:16s} {path:5s} best {best["us"]:7.1f} us ({best["variant"]}, plan {best.get("plan")}, {best["tflops"]} TF)', flush=True)
        for r in res:
            print('      ', r, flush=True)
        del x, w
    if only and 'modulation' not in only:
        Path(args.json).parent.mkdir(exist_ok=True)
        Path(args.json).write_text(json.dumps(report, indent=1))
        return
    # the batched modulation launch over the non-excited layers vs one launch per layer
    layers, keep = [], []
    for I, O, R, excited in LAYERS:
        if excited:
            continue
        w = torch.randn(N, O, I, 3, 3, device=DEV) * 0.1
        mod, kmod = torch.randn(B, I, device=DEV) * 0.3, torch.randn(B, N, device=DEV)
        ly = dict(w=w, mod=mod, kmod=kmod, demod=True, eps=1e-8, Ip=I, Op=O)
        path = ops.HipOps._modconv_path(B, N, O, I, R, R)
        if path == 'pimg':
            ly.update(coef=False, wmix=torch.empty(B, O, 9 * I, dtype=torch.bfloat16, device=DEV), layout=1)
        elif path == 'sconv':
            ly.update(coef=False, wmix=torch.zeros(B, 9, I // 16, 32, 16, dtype=torch.bfloat16, device=DEV), layout=2)
        layers.append(ly)
    us_multi, _ = graph_us(lambda: K.modw_multi(layers)[0]['d'] if layers[0].get('coef', True) else None)

    def per_layer():
        for ly in layers:
            K.modw_fwd(ly['w'], ly['mod'], ly['kmod'], True, 1e-8, ly['Ip'], ly['Op'], coef=ly.get('coef', True), wmix=ly.get('wmix'),
                       layout=ly.get('layout', 0))
    us_single, _ = graph_us(per_layer)
    report['modulation'] = dict(layers=len(layers), multi_us=round(us_multi, 1), per_layer_us=round(us_single, 1))
    print('modulation of', len(layers), 'layers: one launch', round(us_multi, 1), 'us; one launch per layer', round(us_single, 1), 'us', flush=True)
    Path(args.json).parent.mkdir(exist_ok=True)
    Path(args.json).write_text(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
