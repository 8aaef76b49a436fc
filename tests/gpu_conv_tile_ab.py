"""GPU micro A/B of the row-major convolution kernels on the config-2 shapes (batch 32): the planned kernel (tile 0) against forced
tiles (7 / 8: the halo-staged 3x3 convolution; 4 / 5 / 6: the 8-wave implicit GEMM), each checked against the 4-wave 128x128 tile.
An optional argument binds another build of the library, for same-box comparisons of two builds:
    python tests/gpu_conv_tile_ab.py [other_library.so]
(test infrastructure: not part of the product path)."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import _C   # noqa: E402
if len(sys.argv) > 1:
    _C.bind(sys.argv[1])           # another build of the library (same-box A/B)
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
    b = 32
    convs = [('D3.conv1', 4 * b, 32, 128, 256, 3), ('D3.conv2', 4 * b, 32, 256, 256, 3), ('D4.conv1', 8 * b, 16, 256, 512, 3),
             ('D4.conv2', 8 * b, 16, 512, 512, 3), ('D4.pred', 4 * b, 16, 512, 512, 3), ('D5.conv', 16 * b, 8, 512, 512, 3),
             ('D6.conv', 16 * b, 4, 512, 512, 3), ('D2.conv2', 2 * b, 64, 128, 128, 3), ('D3.ff1', 4 * b, 32, 256, 1024, 1),
             ('D4.ff2', 8 * b, 16, 2048, 512, 1), ('D4.conv1.dgrad', 8 * b, 16, 512, 256, 3),
             ('D3.conv2.g', 2 * b, 32, 256, 256, 3), ('D5.pred', 8 * b, 8, 512, 512, 3), ('D2.conv1', 2 * b, 64, 64, 128, 3), ('D2.conv2.n256', 2 * b, 64, 128, 256, 3)]
    tot = 0.0
    for name, n, R, ci, co, ks in convs:
        x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(co, ks * ks * ci, device=dev) * 0.05).to(torch.bfloat16)
        ref = K.conv2d_nhwc(x, w, ksize=ks, force_tile=1).float()
        row = [name]
        for tile in (0, 7, 8):
            out = K.conv2d_nhwc(x, w, ksize=ks, force_tile=tile).float()
            err = ((out - ref).norm() / ref.norm()).item()
            ms = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=ks, force_tile=tile))
            tf = 2.0 * n * R * R * co * ks * ks * ci / ms / 1e9
            row.append(f'tile{tile}: {ms * 1e3:7.1f} us {tf:6.0f} TF err {err:.1e}')
            if tile == 0:
                tot += ms
        print(' | '.join(row), flush=True)
    print('planned total', round(tot, 3), 'ms', flush=True)


if __name__ == '__main__':
    main()
