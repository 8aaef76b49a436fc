"""GPU probe (round 5): does the ROW PITCH of the weight operand bound the convolution kernels?

Hypothesis: a weight matrix [co][9 C] has a row pitch of 9 C * 2 bytes - for C = 256 / 512 a multiple of 512 B / 1 KB - so the 64-channel
(128-byte) pieces that all workgroups of a launch fetch for one tap at the same time sit a power-of-two-ish stride apart and fall on a
few of an XCD's 16 L2 channels (gg_conv3<256>: 32 KB per workgroup and tap from ~4 channels = the ~3900 cycles per tap measured in
round 2 against 2048 cycles of MFMA work; gg_lrconv: 5 us per 32-channel chunk). The same launch with the rows padded by 64 / 128 / 192
elements spreads the pieces over all channels; if the hypothesis holds it runs faster by a large factor.

    python tests/gpu_r5_pitch_probe.py
(test infrastructure: not part of the product path)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402


def time_us(fn, iters=20, warmup=3):
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


def padded(w, pad):
    """the same matrix with `pad` unused elements behind every row (last dim)"""
    if pad == 0:
        return w
    big = torch.zeros(*w.shape[:-1], w.shape[-1] + pad, dtype=w.dtype, device=w.device)
    big[..., :w.shape[-1]] = w
    return big[..., :w.shape[-1]]


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    b = 32
    print('## shared weights: gg_conv3 (tiles 7 / 8 / 12), gg_gemm2 implicit GEMM (tile 4), planner (0)')
    convs = [('D4.conv2 512->512@16', 8 * b, 16, 512, 512, 3, (0, 7, 4)), ('D3.conv2 256->256@32', 4 * b, 32, 256, 256, 3, (0, 7, 4)),
             ('D5.conv 512->512@8', 16 * b, 8, 512, 512, 3, (0, 7, 4)), ('D2.conv2 128->128@64', 2 * b, 64, 128, 128, 3, (0, 8)),
             ('D3.ff1 1x1 256->1024@32', 4 * b, 32, 256, 1024, 1, (0, 4)), ('D4.ff1 1x1 512->2048@16', 8 * b, 16, 512, 2048, 1, (0, 4)),
             ('D4.ff2 1x1 2048->512@16', 8 * b, 16, 2048, 512, 1, (0, 4))]
    for name, n, R, ci, co, ks, tiles in convs:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, ks * ks * ci, device=dev) * 0.05).to(torch.bfloat16)
        ref = None
        for tile in tiles:
            row = [f'{name:26s} tile {tile}']
            for pad in (0, 64, 128, 192):
                wp = padded(w, pad)
                out = K.conv2d_nhwc(x, wp, ksize=ks, force_tile=tile).float()
                if ref is None:
                    ref = out
                err = ((out - ref).norm() / ref.norm()).item()
                us = time_us(lambda: K.conv2d_nhwc(x, wp, ksize=ks, force_tile=tile))
                tf = 2.0 * n * R * R * co * ks * ks * ci / us / 1e6
                row.append(f'pad {pad:3d}: {us:7.1f} us {tf:6.0f} TF' + ('' if err < 1e-2 else f' ERR {err:.1e}'))
            print(' | '.join(row), flush=True)

    print('## per-image weights (the generator\'s 32x32 / 64x64 adaptive convs): gg_conv3 64-column tile')
    for name, R, ci, co in (('256->128@32', 32, 256, 128), ('128->128@32', 32, 128, 128), ('128->64@64', 64, 128, 64), ('64->64@64', 64, 64, 64)):
        x = torch.randn(b, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(b, co, 9 * ci, device=dev) * 0.05).to(torch.bfloat16)
        ref = None
        row = [f'{name:26s} planner']
        for pad in (0, 64, 128, 192):
            wp = padded(w, pad)
            out = K.conv2d_nhwc(x, wp, ksize=3, per_image_weights=True).float()
            if ref is None:
                ref = out
            err = ((out - ref).norm() / ref.norm()).item()
            us = time_us(lambda: K.conv2d_nhwc(x, wp, ksize=3, per_image_weights=True))
            tf = 2.0 * b * R * R * co * 9 * ci / us / 1e6
            row.append(f'pad {pad:3d}: {us:7.1f} us {tf:6.0f} TF' + ('' if err < 1e-2 else f' ERR {err:.1e}'))
        print(' | '.join(row), flush=True)

    print('## stacked / mixed banks on low-resolution maps: gg_lrconv (tile 11), gg_conv3 SCALED (tile 8)')
    for name, R, ci, co, mix in (('512->512@4 stacked', 4, 512, 512, False), ('512->512@8 stacked', 8, 512, 512, False),
                                 ('512->256@16 mix', 16, 512, 256, True), ('256->256@16 mix', 16, 256, 256, True)):
        x = torch.randn(b, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, 9 * 2 * ci, device=dev) * 0.05).to(torch.bfloat16)
        s = torch.rand(b, ci, device=dev) + 0.5
        a = torch.softmax(torch.randn(b, 2, device=dev), -1)
        insc = (a[:, :, None] * s[:, None, :]).reshape(b, 2 * ci).contiguous()
        d = torch.rand(b, co, device=dev) + 0.5
        for tile in ((11,) if (mix or R == 4) else (11, 8)):
            ref = None
            row = [f'{name:26s} tile {tile}']
            for pad in (0, 64, 128, 192):
                wp = padded(w, pad)
                if mix:
                    fn = lambda: K.conv2d_nhwc(x, wp, ksize=3, cv=2 * ci, in_scale=s.contiguous(), bank_mix=a.contiguous(), out_scale=d, act='lrelu')
                else:
                    fn = lambda: K.conv2d_nhwc(x, wp, ksize=3, cv=2 * ci, in_scale=insc, out_scale=d, act='lrelu', force_tile=tile)
                out = fn().float()
                if ref is None:
                    ref = out
                err = ((out - ref).norm() / ref.norm()).item()
                us = time_us(fn)
                tf = 2.0 * b * R * R * co * 9 * ci / us / 1e6
                row.append(f'pad {pad:3d}: {us:7.1f} us {tf:6.0f} TF(alg)' + ('' if err < 1e-2 else f' ERR {err:.1e}'))
            print(' | '.join(row), flush=True)

    print('## plain GEMM (gg_gemm2 256x256): activation pitch lda and weight pitch ldb')
    for name, M, N, Kd in (('FF up   M131072 N2048 K512', 131072, 2048, 512), ('FF down M131072 N512 K2048', 131072, 512, 2048),
                           ('proj    M262144 N256 K512', 262144, 256, 512)):
        a0 = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
        b0 = (torch.randn(N, Kd, device=dev) * 0.05).to(torch.bfloat16)
        row = [f'{name:28s}']
        for pa, pb in ((0, 0), (0, 64), (64, 0), (64, 64)):
            ap, bp = padded(a0, pa), padded(b0, pb)
            us = time_us(lambda: K.gemm(ap, bp), iters=10)
            row.append(f'lda+{pa} ldb+{pb}: {us:7.1f} us {2.0 * M * N * Kd / us / 1e6:6.0f} TF')
        print(' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
