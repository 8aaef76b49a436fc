"""Narrow down why replicas inside a batch differ (follow-up of gpu_replica_bisect.py): the first adaptive conv of the generator
(512 -> 512 at 4x4, batch 32 = 16 copies of 2 samples), piece by piece: coefficient kernel, modulate pass, the contraction with
every tile / split-K choice (fp32 output), the fused epilogue. Test infrastructure.   python tests/gpu_replica_probe.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import kernels as K, ops   # noqa: E402

B, R = 32, 2


def spread(t):
    r = t.reshape(B // R, R, -1).float()
    d = (r - r[:1]).abs()
    return float(d.max()), int((d > 0).sum())


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    N, O, I, k, H = 2, 512, 512, 3, 4
    w = (torch.randn(N, O, I, k, k, device=dev) * 0.03)
    mod = (torch.randn(R, I, device=dev) * 0.5).repeat(B // R, 1).contiguous()
    kmod = torch.randn(R, N, device=dev).repeat(B // R, 1).contiguous()
    x = torch.randn(R, H, H, I, device=dev).repeat(B // R, 1, 1, 1).to(torch.bfloat16).contiguous()
    s, a, d = K.modcoef_fwd(w, mod, kmod, True, 1e-8, I, O)
    print('modcoef s a d spread', spread(s), spread(a), spread(d))
    insc = (a[:, :, None] * s[:, None, :]).reshape(B, N * I).contiguous()
    print('insc spread', spread(insc))
    x2 = torch.cat([K.modulate(x, insc[:, n * I:(n + 1) * I].contiguous()) for n in range(N)], dim=-1)
    print('modulate spread', spread(x2))
    wk = w.permute(1, 3, 4, 0, 2).reshape(O, k * k * N * I).to(torch.bfloat16).contiguous()
    for tile in (0, 1, 4, 5, 6):
        for sk in (0, 1, 2, 8):
            try:
                K.plan_log = []
                y = K.conv2d_nhwc(x2, wk, ksize=3, out_dtype=torch.float32, force_tile=tile, force_splitk=sk)
                plan = K.plan_log
            except RuntimeError as e:
                print('tile', tile, 'sk', sk, 'ERR', str(e)[:80]); continue
            finally:
                K.plan_log = None
            print(f'conv fp32 tile {tile} force_splitk {sk} plan {plan}: spread {spread(y)}')
    y = K.conv2d_nhwc(x2, wk, ksize=3, out_scale=d.contiguous(), act='lrelu')
    print('conv bf16 + demod + lrelu spread', spread(y))
    y = K.conv2d_nhwc(x, wk, ksize=3, cv=N * I, in_scale=insc, out_dtype=torch.float32)
    print('in-gather scale conv fp32 spread', spread(y))
    # dense GEMM with replicated rows
    a2 = torch.randn(R * 16, 4608, device=dev).repeat(B // R, 1).to(torch.bfloat16).contiguous()
    b2 = torch.randn(512, 4608, device=dev).to(torch.bfloat16)
    for tile in (1, 4, 6):
        y = K.gemm(a2, b2, out_dtype=torch.float32, force_tile=tile)[0]
        print('dense gemm tile', tile, 'spread', spread(y))


if __name__ == '__main__':
    main()
