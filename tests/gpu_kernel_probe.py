"""runs ONE contraction-kernel configuration a few times (for rocprofv3 --pmc / --kernel-trace passes).
usage: python tests/gpu_kernel_probe.py {fwd|wgrad|dense_wgrad} n_img res cin cout ksize tile [iters]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402

kind, n, R, ci, co, ks, tile = sys.argv[1], *map(int, sys.argv[2:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
dev = torch.device('cuda', 0)
torch.manual_seed(0)
x = torch.randn(n, R, R, ci, device=dev).to(torch.bfloat16)
w = (torch.randn(co, ks * ks * ci, device=dev) * 0.05).to(torch.bfloat16)
dy = torch.randn(n, R, R, co, device=dev).to(torch.bfloat16)
for _ in range(iters):
    if kind == 'fwd':
        K.conv2d_nhwc(x, w, ksize=ks, force_tile=tile)
    else:
        K.conv2d_wgrad_nhwc(x, dy, ksize=ks, force_tile=tile)
torch.cuda.synchronize()
print('ok')
