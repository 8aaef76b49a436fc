"""GPU diagnostic sweep (run by hand through gpurun, not collected by pytest): checks every kernel family
against the oracle on the device and times the C2 (256x256, batch 32) layer shapes. Writes
gpurun_out/diag.json so a cut-off call still leaves evidence."""
from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from gigagan_pytorch_amd import kernels as K, ops  # noqa: E402
from oracle.torch_ops import OracleOps  # noqa: E402

dev = torch.device('cuda', 0)
OUT = ROOT / 'gpurun_out'
OUT.mkdir(exist_ok=True)
results = {'checks': [], 'timings': []}


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-12)).item()


def bf(t):
    return t.to(torch.bfloat16)


def record(name, err, tol=1e-2):
    ok = err == err and err < tol
    results['checks'].append(dict(name=name, err=err, ok=bool(ok)))
    print(f'[{"ok" if ok else "FAIL"}] {name}: {err:.3e}', flush=True)
    (OUT / 'diag.json').write_text(json.dumps(results, indent=1))


def timeit(name, fn, flops=None, bytes_=None, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    row = dict(name=name, ms=ms)
    if flops:
        row['tflops'] = flops / ms / 1e9
    if bytes_:
        row['gbps'] = bytes_ / ms / 1e6
    results['timings'].append(row)
    print(f'  time {name}: {ms*1e3:.1f} us' + (f' {row["tflops"]:.1f} TF/s' if flops else '') +
          (f' {row["gbps"]:.0f} GB/s' if bytes_ else ''), flush=True)
    (OUT / 'diag.json').write_text(json.dumps(results, indent=1))
    return ms


def main():
    torch.manual_seed(0)
    print(torch.cuda.get_device_name(0), flush=True)

    # ---- GEMM: transposes / tiles / split-K --------------------------------------------------------------
    for (M, N, Kd, batch) in [(130, 70, 104, 2), (257, 129, 40, 3), (1024, 1032, 64, 8)]:
        A = bf(torch.randn(batch, M, Kd, device=dev))
        B = bf(torch.randn(batch, N, Kd, device=dev))
        ref = torch.einsum('bmk,bnk->bmn', A.float(), B.float())
        for tile in (0, 1, 2, 3):
            out = K.gemm(A, B, out_dtype=torch.float32, force_tile=tile)
            record(f'gemm NT {M}x{N}x{Kd} b{batch} tile{tile}', rel(out, ref), 1e-5)
        out = K.gemm(A, B, out_dtype=torch.float32, force_splitk=2)
        record(f'gemm NT splitk {M}x{N}x{Kd}', rel(out, ref), 1e-5)
        if M % 8 == 0 and N % 8 == 0:
            At, Bt = A.transpose(1, 2).contiguous(), B.transpose(1, 2).contiguous()
            for ta, tb, a_, b_ in ((True, True, At, B), (False, False, A, Bt), (True, False, At, Bt)):
                out = K.gemm(a_, b_, trans_a=ta, trans_b=tb, out_dtype=torch.float32)
                record(f'gemm ta={ta} tb={tb} {M}x{N}x{Kd}', rel(out, ref), 1e-5)
    # k tail (1025 valid of 1032)
    A = bf(torch.randn(4, 256, 1032, device=dev)); B = bf(torch.randn(4, 64, 1032, device=dev))
    ref = torch.einsum('bmk,bnk->bmn', A.float()[..., :1025], B.float()[..., :1025])
    record('gemm k_valid=1025', rel(K.gemm(A, B, k_valid=1025, out_dtype=torch.float32), ref), 1e-5)
    bias = torch.randn(64, device=dev)
    ref2 = F.leaky_relu(ref * 0.5 + bias, 0.2)
    record('gemm epilogue', rel(K.gemm(A, B, k_valid=1025, alpha=0.5, bias=bias, act='lrelu'), ref2), 5e-3)

    # ---- conv fwd / dgrad / wgrad ----------------------------------------------------------------------------
    for (n, H, W, Ci, Co, ks) in [(2, 8, 8, 16, 24, 3), (3, 5, 7, 8, 40, 7), (2, 16, 16, 32, 136, 1), (4, 32, 32, 64, 64, 3),
                                  (2, 64, 64, 128, 128, 3)]:
        x = bf(torch.randn(n, Ci, H, W, device=dev)); w = bf(torch.randn(Co, Ci, ks, ks, device=dev) * 0.1)
        dy = bf(torch.randn(n, Co, H, W, device=dev))
        xf = x.float().requires_grad_(); wf = w.float().requires_grad_()
        ref = F.conv2d(xf, wf, padding=ks // 2)
        ref.backward(dy.float())
        xh = x.permute(0, 2, 3, 1).contiguous(); dyh = dy.permute(0, 2, 3, 1).contiguous()
        wh = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()
        tag = f'{n}x{H}x{W} {Ci}->{Co} k{ks}'
        record(f'conv fwd {tag}', rel(K.conv2d_nhwc(xh, wh, ksize=ks, out_dtype=torch.float32).permute(0, 3, 1, 2), ref), 1e-4)
        record(f'conv wgrad {tag}', rel(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks), wf.grad.permute(2, 3, 1, 0).reshape(-1, Co)), 1e-4)
        wT = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, -1).contiguous()
        record(f'conv dgrad {tag}', rel(K.conv2d_nhwc(dyh, wT, ksize=ks, out_dtype=torch.float32).permute(0, 3, 1, 2), xf.grad), 1e-4)

    # ---- ops vs oracle (fwd + grads) ---------------------------------------------------------------------
    H_, O_ = ops.HipOps(), OracleOps(bf16_operands=True)

    def check(name, fn, inputs, tol=2e-2):
        ih = [t.clone().requires_grad_() for t in inputs]
        io = [t.clone().requires_grad_() for t in inputs]
        yh, yo = fn(H_, *ih), fn(O_, *io)
        g = torch.randn_like(yo)
        gh = torch.autograd.grad(yh.float(), ih, g)
        go = torch.autograd.grad(yo, io, bf(g).float())
        record(f'{name} fwd', rel(yh, yo), tol)
        for i, (a, b) in enumerate(zip(gh, go)):
            record(f'{name} grad{i}', rel(a, b), 5 * tol)

    x = torch.randn(2, 16, 8, 8, device=dev); w = torch.randn(24, 16, 3, 3, device=dev) * 0.1; b = torch.randn(24, device=dev)
    check('op conv3x3', lambda I, x, w, b: I.conv2d(x, w, b), [x, w, b])
    check('op conv3x3+lrelu', lambda I, x, w, b: I.conv2d(x, w, b, act='lrelu'), [x, w, b])
    check('op conv7x7 c3', lambda I, x, w: I.conv2d(x, w, None), [torch.randn(2, 3, 8, 8, device=dev), torch.randn(16, 3, 7, 7, device=dev) * 0.1])
    check('op linear', lambda I, x, w, b: I.linear(x, w, b), [torch.randn(6, 20, device=dev), torch.randn(5, 20, device=dev), torch.randn(5, device=dev)])
    q, k, v = (torch.randn(2, 2, n_, 16, device=dev) for n_ in (16, 17, 17))
    check('op attn dot', lambda I, q, k, v: I.attention(q, k, v, scale=0.25), [q, k, v])
    check('op attn l2', lambda I, q, k, v: I.attention(q, k, v, scale=0.25, l2=True), [q, k, v])
    wm = torch.randn(2, 24, 16, 3, 3, device=dev) * 0.1; mod = torch.randn(2, 16, device=dev) * 0.5; km = torch.randn(2, 2, device=dev)
    check('op modconv', lambda I, x, w, m, k: I.modconv2d(x, w, m, k), [x, wm, mod, km])
    with torch.no_grad():
        nz = torch.randn(2, 1, 8, 8, device=dev); nw = torch.randn(24, 1, 1, device=dev)
        record('op modconv fused fwd', rel(H_.modconv2d(x, wm, mod, km, noise=nz, noise_weight=nw, act='lrelu'),
                                           O_.modconv2d(x, wm, mod, km, noise=nz, noise_weight=nw, act='lrelu')), 2e-2)
    check('op rmsnorm', lambda I, x, g: I.channel_rmsnorm(x, g), [x, torch.randn(16, 1, 1, device=dev)])
    check('op upsample_blur', lambda I, x: I.upsample_blur(x), [x])
    check('op resize', lambda I, x: I.resize_bilinear(x, 4), [torch.rand(2, 3, 16, 16, device=dev)])

    # ---- timings: C2 generator adaptive convs (fused forward), batch 32 -----------------------------------------
    b = 32
    total_ms, total_fl = 0., 0.
    for (I, O, R) in [(512, 512, 4), (512, 512, 4), (512, 512, 4), (512, 512, 8), (512, 512, 8), (512, 256, 16), (256, 256, 16),
                      (256, 128, 32), (128, 128, 32), (128, 64, 64), (64, 64, 64), (64, 32, 128), (32, 32, 128), (32, 16, 256),
                      (16, 16, 256)]:
        x = bf(torch.randn(b, I, R, R, device=dev)).contiguous(memory_format=torch.channels_last)
        wts = torch.randn(2, O, I, 3, 3, device=dev) * 0.05
        mod = torch.randn(b, I, device=dev) * 0.1; km = torch.randn(b, 2, device=dev)
        nz = torch.randn(b, 1, R, R, device=dev); nw = torch.randn(O, 1, 1, device=dev) * 0.1
        fl = 2. * b * O * I * 9 * R * R
        with torch.no_grad():
            ms = timeit(f'modconv fused {I}->{O} @{R}', lambda: H_.modconv2d(x, wts, mod, km, noise=nz, noise_weight=nw, act='lrelu'), flops=fl)
        total_ms += ms; total_fl += fl
    results['modconv_fwd_total_us'] = total_ms * 1e3
    results['modconv_fwd_tflops'] = total_fl / total_ms / 1e9
    print(f'modconv forward total {total_ms*1e3:.0f} us = {total_fl/total_ms/1e9:.1f} TF/s algorithmic', flush=True)

    # ---- timings: C2 discriminator convs (fwd / dgrad / wgrad), batch 32 -----------------------------------------
    for (nb, Ci, Co, R, ks) in [(32, 32, 32, 256, 3), (32, 64, 64, 128, 3), (64, 128, 128, 64, 3), (128, 256, 256, 32, 3),
                                (256, 512, 512, 16, 3), (512, 512, 512, 8, 3), (512, 512, 512, 4, 3), (128, 256, 1024, 32, 1),
                                (256, 2048, 512, 16, 1)]:
        x = bf(torch.randn(nb, R, R, Ci, device=dev)); dy = bf(torch.randn(nb, R, R, Co, device=dev))
        wh = bf(torch.randn(Co, ks * ks * Ci, device=dev) * 0.05)
        fl = 2. * nb * R * R * Co * Ci * ks * ks
        byt = 2. * nb * R * R * (Ci + Co)
        timeit(f'D conv fwd {nb}x{R}^2 {Ci}->{Co} k{ks}', lambda: K.conv2d_nhwc(x, wh, ksize=ks), flops=fl, bytes_=byt)
        timeit(f'D conv wgrad {nb}x{R}^2 {Ci}->{Co} k{ks}', lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=ks), flops=fl, bytes_=byt)

    # attention contractions at D's 32^2 stage (batch 4b = 128, 8 heads)
    BH, n, m, d = 1024, 1024, 1032, 64
    q = bf(torch.randn(BH, n, d, device=dev)); kk = bf(torch.randn(BH, m, d, device=dev)); vv = bf(torch.randn(BH, m, d, device=dev))
    timeit('attn QK^T 1024x1024x1032x64 fp32 out', lambda: K.gemm(q, kk, out_dtype=torch.float32), flops=2. * BH * n * m * d, bytes_=BH * n * m * 4.)
    at = bf(torch.rand(BH, n, m, device=dev))
    timeit('attn AV', lambda: K.gemm(at, vv, trans_b=False), flops=2. * BH * n * m * d, bytes_=BH * n * m * 2.)
    del q, kk, vv, at

    # resample + adamw
    x = bf(torch.randn(32, 128, 128, 64, device=dev))
    spec = K.ResampleSpec.upsample_blur(128, 128)
    timeit('upsample_blur 32x128^2x64', lambda: K.resample_nhwc(x, spec), bytes_=x.numel() * 2 * 5.)
    from gigagan_pytorch_amd.optimizer import FlatAdamW
    ps = [torch.nn.Parameter(torch.randn(61_000_000 // 4, device=dev)) for _ in range(4)]
    opt = FlatAdamW(ps, lr=2e-4, betas=(0.5, 0.9))
    opt.flat_g.normal_()
    timeit('adamw 61M', lambda: opt.step(), bytes_=opt.total * 28.)
    # adamw correctness vs torch
    p0 = torch.nn.Parameter(torch.randn(1000, 33, device=dev)); p1 = torch.nn.Parameter(torch.randn(77, device=dev))
    r0, r1 = torch.nn.Parameter(p0.detach().clone()), torch.nn.Parameter(p1.detach().clone())
    fo = FlatAdamW([p0, p1], lr=2e-4, betas=(0.5, 0.9)); to = torch.optim.AdamW([{'params': [r0]}, {'params': [r1], 'weight_decay': 0.}], lr=2e-4, betas=(0.5, 0.9), weight_decay=1e-2)
    for _ in range(3):
        g0, g1 = torch.randn_like(p0), torch.randn_like(p1)
        p0.grad.copy_(g0); p1.grad.copy_(g1); r0.grad = g0.clone(); r1.grad = g1.clone()
        fo.step(); to.step()
    record('adamw vs torch p0', rel(p0, r0), 1e-6); record('adamw vs torch p1', rel(p1, r1), 1e-6)

    nfail = sum(not c['ok'] for c in results['checks'])
    print(f'DONE: {len(results["checks"])} checks, {nfail} failed', flush=True)
    (OUT / 'diag.json').write_text(json.dumps(results, indent=1))


if __name__ == '__main__':
    main()
