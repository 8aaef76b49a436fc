"""GPU probe (round 5): gg_aconv_fwd on the generator's eleven 4x4 .. 64x64 adaptive-conv layer shapes at batch 32 - every eligible
(TM, NWN) tile shape against the library's own choice, checked against fp32 math on the same rounded operands, next to what the same
layer costs on the round-3/4 kernels (ops.modconv2d with GG_ACONV off: modulation launch + convolution + split-K finish).

    python tests/gpu_r5_aconv_probe.py [--quick]
(test infrastructure: not part of the product path)."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K, ops   # noqa: E402
from gigagan_pytorch_amd.modules import AdaptiveConv2DMod   # noqa: E402

LAYERS = [(512, 512, 4), (512, 512, 8), (512, 256, 16), (256, 256, 16), (256, 128, 32), (128, 128, 32), (128, 64, 64), (64, 64, 64)]
# the UnetUpsampler's (config 5, batch 16) shapes that take the same kernel: checked against fp32 math with the library's own plan
C5_LAYERS = [(64, 64, 64), (128, 64, 64), (256, 64, 64), (128, 128, 32), (256, 128, 32), (512, 128, 32), (256, 256, 16), (512, 256, 16),
             (512, 512, 8)]


def time_us(fn, iters=10, warmup=2, replays=5):
    """GPU time per call: `iters` calls captured into ONE hipGraph and replayed (eager launches of a < 20 us kernel measure the host:
    a ctypes call + torch.empty is ~15-20 us); a kernel boundary (~1.5 us) is part of every call, as in the training step's graphs."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * replays) * 1e3


def main():
    dev = torch.device('cuda', 0)
    b = 32
    quick = '--quick' in sys.argv
    for I, O, R in C5_LAYERS:
        torch.manual_seed(1)
        b5 = 16
        x = torch.randn(b5, R, R, I, device=dev).to(torch.bfloat16)
        W = torch.randn(2, O, I, 3, 3, device=dev) * (1.0 / (3 * I ** 0.5))
        s = torch.rand(b5, I, device=dev) + 0.5
        a = torch.softmax(torch.randn(b5, 2, device=dev), -1)
        d = torch.rand(b5, O, device=dev) + 0.5
        xm = (x.float() * s[:, None, None, :]).to(torch.bfloat16).float().permute(0, 3, 1, 2)
        ref = sum(a[:, n].view(b5, 1, 1, 1) * F.conv2d(xm, W[n].to(torch.bfloat16).float(), padding=1) for n in range(2))
        ref = (ref * d.view(b5, O, 1, 1)).permute(0, 2, 3, 1)
        y = K.aconv(x, K.frag_pack(W), s, a, d, O).float()
        print(f'c5 {I}->{O}@{R} b16 plan {K.aconv_plan(b5, R, R, I, O, 2)}: err {((y - ref).norm() / ref.norm()).item():.2e} '
              f'max|diff| {(y - ref).abs().max().item():.3e} (ref max {ref.abs().max().item():.2f})', flush=True)
    for I, O, R in LAYERS:
        torch.manual_seed(0)
        x = torch.randn(b, R, R, I, device=dev).to(torch.bfloat16)
        W = torch.randn(2, O, I, 3, 3, device=dev) * (1.0 / (3 * I ** 0.5))
        s = torch.rand(b, I, device=dev) + 0.5
        a = torch.softmax(torch.randn(b, 2, device=dev), -1)
        d = torch.rand(b, O, device=dev) + 0.5
        nz, nw = torch.randn(b * R * R, device=dev), torch.randn(O, device=dev) * 0.3
        wf = K.frag_pack(W)
        xm = (x.float() * s[:, None, None, :]).to(torch.bfloat16).float().permute(0, 3, 1, 2)
        ref = sum(a[:, n].view(b, 1, 1, 1) * F.conv2d(xm, W[n].to(torch.bfloat16).float(), padding=1) for n in range(2))
        ref = F.leaky_relu(ref * d.view(b, O, 1, 1) + nz.view(b, 1, R, R) * nw.view(1, O, 1, 1), 0.2).permute(0, 2, 3, 1)
        flops = 2.0 * b * O * I * 9 * R * R
        row = [f'{I:3d}->{O:3d}@{R:2d}']
        plan = K.aconv_plan(b, R, R, I, O, 2)
        cands = [(0, 0)] + ([] if quick else [(tm, nwn) for tm in (1, 2, 4) for nwn in (1, 2, 4)])
        for tm, nwn in cands:
            pl = K.aconv_plan(b, R, R, I, O, 2, tm, nwn)
            if pl is None:
                continue
            fn = lambda: K.aconv(x, wf, s, a, d, O, nz, nw, 'lrelu', force_tm=tm, force_nwn=nwn)
            y = fn().float()
            err = ((y - ref).norm() / ref.norm()).item()
            us = time_us(fn)
            print(f'   {I}->{O}@{R} tm{pl[0]} n{pl[1]}: {us:.1f} us err {err:.2e}', flush=True)
            tag = 'plan' if (tm, nwn) == (0, 0) else ''
            row.append(f'{tag}(tm{pl[0]} n{pl[1]} k{pl[2]} lds{pl[3] >> 10}K g{pl[4]}): {us:6.1f} us {flops / us / 1e6:5.0f} TF' + ('' if err < 5e-3 else f' ERR {err:.1e}'))
        if '--phases' in sys.argv:       # where a launch's time goes: phases switched off one at a time (results meaningless)
            for dbg, what in ((1, 'no reduction loop'), (2, 'no halo staging'), (3, 'neither'), (0, 'no noise / demod / act operands')):
                if dbg:
                    fn = lambda: K.aconv(x, wf, s, a, d, O, nz, nw, 'lrelu', _dbg=dbg)
                else:
                    fn = lambda: K.aconv(x, wf, s, a, None, O, None, None, None)
                row.append(f'[{what}: {time_us(fn):6.1f} us]')
        # the same layer through ops.modconv2d: new path (modulation launch + gg_aconv) and the round-3/4 kernels
        conv = AdaptiveConv2DMod(I, O, 3, num_conv_kernels=2).to(dev)
        xc = x.permute(0, 3, 1, 2)
        mod, km = torch.randn(b, I, device=dev) * 0.3, torch.randn(b, 2, device=dev)
        nzc, nwc = torch.randn(b, 1, R, R, device=dev), torch.randn(O, 1, 1, device=dev) * 0.1
        outs = {}
        for flag in (True, False):
            ops._ACONV = flag
            with torch.no_grad(), ops.use_impl(ops.HipOps()):
                fn = lambda: conv(xc, mod, km, noise=nzc, noise_weight=nwc, act='lrelu')
                outs[flag] = fn().float()
                row.append(f"modconv2d[{'aconv' if flag else 'r4'}]: {time_us(fn):6.1f} us")
        ops._ACONV = True
        row.append(f'new-vs-r4 rel {((outs[True] - outs[False]).norm() / outs[False].norm()).item():.1e}')
        print(' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
