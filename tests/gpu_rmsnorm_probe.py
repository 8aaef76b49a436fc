"""gg_rmsnorm fwd / bwd: achieved HBM bandwidth per shape (fwd: x in, y out; bwd: x, g in, dx out)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigagan_pytorch_amd import kernels as K
dev = torch.device('cuda', 0)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for rows, C in ((64 * 1024, 256), (64 * 256, 512), (32 * 1024, 256), (32 * 256, 512), (32 * 4096, 128), (64 * 4096, 128), (32 * 16384, 64), (16 * 4096, 256),
                (32 * 65536, 32)):
    x = torch.randn(rows, C, device=dev).bfloat16()
    g = torch.randn(rows, C, device=dev).bfloat16()
    gamma = torch.randn(C, device=dev)
    us = timeit(lambda: K.rmsnorm_fwd(x, gamma))
    ub = timeit(lambda: K.rmsnorm_bwd(x, g, gamma, True))
    uc = timeit(lambda: K.rmsnorm_bwd(x, g, gamma, True, carry=g))
    print('rows %8d C %4d  fwd %7.1f us %5.2f TB/s   bwd %7.1f us %5.2f TB/s   bwd+carry %7.1f us %5.2f TB/s' % (
        rows, C, us, rows * C * 4 / us / 1e6, ub, rows * C * 6 / ub / 1e6, uc, rows * C * 8 / uc / 1e6), flush=True)
