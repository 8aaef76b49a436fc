"""GPU probe (round 5): the persistent short-K contraction (gg_pgemm.h, plan tile 15) against the kernel the planner picks without it,
on the step's own short-K shapes (profiles/r05_final_gemm_shapes.json) and epilogues - hipGraph-timed, results compared bit for bit.
Run with GG_PGEMM=0 so that the default plan is the one without the substitution:
    GG_PGEMM=0 python tests/gpu_r5_pgemm_probe.py
(test infrastructure: not part of the product path)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gpu_r5_aconv_probe import time_us   # noqa: E402

# (M, N, K, epilogue): 'plain' alpha only; 'bias'; 'res' bias + residual; 'aux1' bias + gelu aux (FF up); 'aux2' gelu' aux (FF down dgrad)
SHAPES = [
    (262144, 1024, 256, 'aux1'), (262144, 1024, 256, 'aux2'), (262144, 256, 1024, 'res'), (262144, 256, 1024, 'plain'),
    (131072, 2048, 512, 'aux1'), (131072, 2048, 512, 'aux2'), (131072, 512, 2048, 'res'),
    (262144, 512, 256, 'bias'), (262144, 256, 512, 'res'), (262144, 256, 512, 'plain'),
    (131072, 512, 512, 'bias'), (131072, 512, 512, 'res'), (131072, 512, 512, 'plain'),
    (65536, 2048, 512, 'aux1'), (65536, 512, 512, 'bias'), (65536, 512, 512, 'plain'),
    (131072, 1024, 256, 'bias'), (131072, 256, 256, 'plain'), (131072, 256, 1024, 'res'),
    (32768, 512, 128, 'bias'), (32768, 128, 512, 'res'), (32768, 512, 512, 'plain'),
    (524288, 128, 576, 'plain'), (262144, 64, 256, 'bias'), (131072, 128, 64, 'plain'),
]


def phases():
    """GG_PGEMM_DBG switches phases of the kernel off (results are garbage): what is left tells where a tile's time goes. Needs a PROBE
    build of the library (hipcc ... -DGG_PROBE): the product build compiles the mask out (round 6)."""
    import os
    dev = torch.device('cuda', 0)
    names = {0: 'all', 1: 'no MFMA', 2: 'no epilogue', 3: 'transfers only', 4: 'no transfers', 5: 'epilogue only', 6: 'k-loop only', 7: 'barriers only', 8: 'no stores', 12: 'no stores, no transfers'}
    for M, N, Kd, epi in [(262144, 1024, 256, 'bias'), (262144, 1024, 256, 'aux1'), (131072, 512, 512, 'plain'), (262144, 256, 1024, 'plain')]:
        hw = 32
        n_img = M // (hw * hw)
        x = torch.randn(n_img, hw, hw, Kd, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) if epi != 'plain' else None
        aux = torch.randn(n_img, hw, hw, N, device=dev).to(torch.bfloat16) if epi == 'aux1' else None
        kw = dict(ksize=1, pad=0, bias=bias, gelu_aux=aux, gelu_mode=1 if aux is not None else 0, force_tile=15)
        tiles = (M // 128) * (N // 128)
        row = [f'M={M} N={N} K={Kd} {epi} ({tiles / 256:.0f} tiles per CU)']
        for dbg in list(range(8)) + [8, 12]:
            os.environ['GG_PGEMM_DBG'] = str(dbg)
            t = time_us(lambda: K.conv2d_nhwc(x, w, **kw), iters=4)
            row.append(f'{names[dbg]}: {t:6.1f} us = {t / (tiles / 256):5.2f} us/tile')
        os.environ.pop('GG_PGEMM_ORDER', None); os.environ.pop('GG_PGEMM_DBG', None)
        print('\n   '.join(row), flush=True)


def main():
    if '--phases' in sys.argv:
        return phases()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    tot_old = tot_new = 0.0
    for M, N, Kd, epi in SHAPES:
        hw = 32
        n_img = M // (hw * hw)
        x = torch.randn(n_img, hw, hw, Kd, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) if epi in ('bias', 'res', 'aux1') else None
        res = torch.randn(n_img, hw, hw, N, device=dev).to(torch.bfloat16) if epi == 'res' else None
        aux = torch.randn(n_img, hw, hw, N, device=dev).to(torch.bfloat16) if epi in ('aux1', 'aux2') else None
        kw = dict(ksize=1, pad=0, bias=bias, residual=res, gelu_aux=aux, gelu_mode={'aux1': 1, 'aux2': 2}.get(epi, 0))
        old_plan = K.conv2d_nhwc(x, w, plan_only=True, **kw)
        new_plan = K.conv2d_nhwc(x, w, plan_only=True, force_tile=15, **kw)
        y_old = K.conv2d_nhwc(x, w, **kw)
        a_old = aux.clone() if epi == 'aux1' else None
        y_new = K.conv2d_nhwc(x, w, force_tile=15, **kw)
        same = torch.equal(y_old, y_new) and (a_old is None or torch.equal(a_old, aux))
        t_old = time_us(lambda: K.conv2d_nhwc(x, w, **kw), iters=4)
        t_new = time_us(lambda: K.conv2d_nhwc(x, w, force_tile=15, **kw), iters=4)
        if '--orders' in sys.argv:            # the kernel's other tile order (GG_PGEMM_ORDER=flip takes the one the host did not choose)
            import os
            os.environ['GG_PGEMM_ORDER'] = 'flip'
            t_flip = time_us(lambda: K.conv2d_nhwc(x, w, force_tile=15, **kw), iters=4)
            os.environ.pop('GG_PGEMM_ORDER', None); os.environ.pop('GG_PGEMM_DBG', None)
            tiles_n = (N + 127) // 128
            rr_default = tiles_n <= 2 or (tiles_n <= 4 and Kd >= 512) or epi == 'aux2'
            print(f'   orders: round-robin {t_new if rr_default else t_flip:7.1f} us | contiguous runs {t_flip if rr_default else t_new:7.1f} us', flush=True)
        nbytes = 2 * M * (Kd + N) + 2 * N * Kd + (2 * M * N if epi in ('res', 'aux1', 'aux2') else 0)
        floor = nbytes / 6.3e12 * 1e6
        tot_old += t_old
        tot_new += t_new
        print(f'M={M:6d} N={N:4d} K={Kd:4d} {epi:5s} | tile {old_plan[0]:2d}: {t_old:7.1f} us | tile {new_plan[0]:2d}: {t_new:7.1f} us | '
              f'x{t_old / t_new:4.2f} | HBM floor {floor:6.1f} us | {2.0 * M * N * Kd / t_new / 1e6:6.0f} TF/s | bits {"same" if same else "DIFFER"}', flush=True)
    print(f'sum: {tot_old:.0f} -> {tot_new:.0f} us', flush=True)


if __name__ == '__main__':
    main()
