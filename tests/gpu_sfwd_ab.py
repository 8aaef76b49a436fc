"""gg_sfwd (plan tile 14, streaming forward / data-gradient convolution) against what it replaces at the thin layer shapes of the config-2
step: gg_dconv / the 4-wave implicit GEMM (shared weights, the discriminator) and gg_sconv (per-image weights, the generator's no-grad
pass). Test infrastructure. Run with GG_SFWD=0 so that the default plan is the old path:   GG_SFWD=0 python tests/gpu_sfwd_ab.py"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gpu_wgrads_ab import timed   # noqa: E402

dev = torch.device('cuda', 0)
SHARED = [  # n, H, W, ci, co, epilogue
    (64, 256, 256, 32, 32, 'bias_act'), (64, 256, 256, 8, 32, 'bias_act'), (64, 256, 256, 32, 8, 'plain'), (64, 128, 128, 64, 64, 'bias_act'),
    (64, 128, 128, 32, 64, 'plain'), (64, 128, 128, 64, 32, 'plain'), (32, 256, 256, 16, 16, 'plain'), (32, 128, 128, 64, 64, 'plain'),
    (32, 256, 256, 32, 32, 'residual'),
]
PERIMG = [(32, 128, 128, 64, 32), (32, 128, 128, 32, 32), (32, 256, 256, 32, 16), (32, 256, 256, 16, 16)]


def main():
    for n, H, W, ci, co, epi in SHARED:
        torch.manual_seed(0)
        x = torch.randn(n, H, W, ci, device=dev).bfloat16()
        w = (torch.randn(co, 9 * ci, device=dev) * 0.1).bfloat16()
        kw = {}
        if epi == 'bias_act':
            kw = dict(bias=torch.randn(co, device=dev), act='lrelu')
        elif epi == 'residual':
            kw = dict(residual=torch.randn(n, H, W, co, device=dev).bfloat16())
        K.plan_log = []
        old = K.conv2d_nhwc(x, w, ksize=3, **kw)
        new = K.conv2d_nhwc(x, w, ksize=3, force_tile=14, **kw)
        plans = list(K.plan_log)
        K.plan_log = None
        want = F.conv2d(x[:2].float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        if epi == 'bias_act':
            want = F.leaky_relu(want + kw['bias'], 0.2)
        elif epi == 'residual':
            want = want + kw['residual'][:2].float()
        err = float((new[:2].float() - want).norm() / want.norm())
        same = float((new.float() - old.float()).abs().max())
        t_old = timed(lambda: K.conv2d_nhwc(x, w, ksize=3, **kw))
        t_new = timed(lambda: K.conv2d_nhwc(x, w, ksize=3, force_tile=14, **kw))
        by = n * H * W * (ci + co + (co if epi == 'residual' else 0)) * 2
        print(f'shared {ci:3d}->{co:3d} @{H}x{W} b={n:2d} {epi:9s} old {plans[0]} {t_old:7.1f} us {by / t_old / 1e6:5.2f} TB/s   new {plans[1]} '
              f'{t_new:7.1f} us {by / t_new / 1e6:5.2f} TB/s  x{t_old / t_new:4.2f}  err vs fp32 {err:.1e}  max|new-old| {same:.3f}', flush=True)
    for n, H, W, ci, co in PERIMG:      # per-image weights + noise + leaky-relu: the generator's last adaptive convolutions
        torch.manual_seed(0)
        x = torch.randn(n, H, W, ci, device=dev).bfloat16()
        w = (torch.randn(n, co, 9 * ci, device=dev) * 0.1).bfloat16()
        nz, nw = torch.randn(n * H * W, device=dev), torch.randn(co, device=dev)
        # gg_sconv's bank layout [b][tap][ci/16][32][16]
        wl2 = torch.zeros(n, 9, ci // 16, 32, 16, device=dev, dtype=torch.bfloat16)
        wl2[:, :, :, :co] = w.view(n, co, 9, ci // 16, 16).permute(0, 2, 3, 1, 4)
        old = K.sconv(x, wl2, co, noise=nz, noise_w=nw, act='lrelu')
        K.plan_log = []
        new = K.conv2d_nhwc(x, w, ksize=3, per_image_weights=True, noise=nz, noise_w=nw, act='lrelu', force_tile=14)
        plan = K.plan_log[-1]
        K.plan_log = None
        same = float((new.float() - old.float()).abs().max())
        t_old = timed(lambda: K.sconv(x, wl2, co, noise=nz, noise_w=nw, act='lrelu'))
        t_new = timed(lambda: K.conv2d_nhwc(x, w, ksize=3, per_image_weights=True, noise=nz, noise_w=nw, act='lrelu', force_tile=14))
        by = n * H * W * ((ci + co) * 2 + 4)
        print(f'per-image {ci:3d}->{co:3d} @{H}x{W} b={n:2d}  gg_sconv {t_old:7.1f} us {by / t_old / 1e6:5.2f} TB/s   new {plan} {t_new:7.1f} us '
              f'{by / t_new / 1e6:5.2f} TB/s  x{t_old / t_new:4.2f}  max|new-old| {same:.3f}', flush=True)


if __name__ == '__main__':
    main()
