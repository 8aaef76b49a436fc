"""gg_bias_act_bwd: achieved HBM bandwidth per shape (3 passes of 2 B per element: dy, y in; dz out)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigagan_pytorch_amd import kernels as K
dev = torch.device('cuda', 0)
for rows, C in ((64 * 65536, 32), (64 * 16384, 64), (64 * 4096, 128), (64 * 1024, 256), (64 * 256, 512), (64 * 64, 512), (64 * 16, 512),
                (32 * 65536, 16), (32 * 16384, 32), (32 * 4096, 64), (32 * 1024, 128), (32 * 65536, 8)):
    dy = torch.randn(rows, C, device=dev).bfloat16()
    y = torch.randn(rows, C, device=dev).bfloat16()
    for want_db in (True, False):
        for _ in range(3):
            K.bias_act_bwd(dy, y, want_db, partials=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.bias_act_bwd(dy, y, want_db, partials=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('rows %8d C %4d db %d  %7.1f us  %6.2f TB/s' % (rows, C, want_db, us, rows * C * 6 / us / 1e6), flush=True)
