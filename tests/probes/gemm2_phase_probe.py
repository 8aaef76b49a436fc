"""Which phase of the 8-wave kernel's k-loop bounds it? Times one conv shape with phases of the loop switched off in a
probe build (tests/probes/build_gemm2_probe.sh; GG2_DBG bit 0: no LDS stores, 1: no global loads, 2: no LDS reads,
3: no MFMAs). Results of the masked runs are garbage by design. usage: GG2_DBG=<mask> python tests/probes/gemm2_phase_probe.py"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import _C, kernels as K   # noqa: E402

_C.bind(ROOT / 'tests' / 'probes' / 'libgg_gemm2_probe.so')
mask = int(os.environ.get('GG2_DBG', '0'))
dev = torch.device('cuda', 0)
torch.manual_seed(0)


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


for name, n, R, ci, co in [('D4.conv2', 512, 16, 512, 512), ('D3.conv2', 256, 32, 256, 256)]:
    x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(co, 9 * ci, device=dev) * 0.05).to(torch.bfloat16)
    dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
    flops = 2.0 * n * R * R * ci * co * 9
    tf = time_ms(lambda: K.conv2d_nhwc(x, w, ksize=3, force_tile=4))
    tw = time_ms(lambda: K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=4))
    a = torch.randn(n * R * R, 9 * ci, device=dev).to(torch.bfloat16) if n * R * R * 9 * ci < 2 ** 31 else None
    td = time_ms(lambda: K.gemm(a, w, force_tile=4)) if a is not None else float('nan')
    print(f'mask {mask:2d} {name}: conv fwd {tf*1e3:7.1f} us ({flops/tf/1e9:6.0f} TF)   wgrad {tw*1e3:7.1f} us ({flops/tw/1e9:6.0f} TF)'
          f'   dense gemm {td*1e3:7.1f} us ({flops/td/1e9:6.0f} TF)', flush=True)
