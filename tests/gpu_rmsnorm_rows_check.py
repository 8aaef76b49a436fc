"""GPU: gg_rmsnorm_rows_kernel against gg_rmsnorm_kernel (GG_RMS_ROWS=0) and the fp32 formulas on the trainer's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigagan_pytorch_amd import kernels as K
dev = torch.device('cuda', 0)
torch.manual_seed(0)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for rows, C in ((65536, 256), (16384, 512), (131072, 128), (524288, 64), (2097152, 32), (1201, 256), (37, 64), (4099, 512)):
    x = (torch.randn(rows, C, device=dev) * 1.7).bfloat16()
    g = torch.randn(rows, C, device=dev).bfloat16()
    carry = torch.randn(rows, C, device=dev).bfloat16()
    gamma = torch.rand(C, device=dev) + 0.5
    res = {}
    for tag, env in (('rows', '1'), ('wave', '0')):
        os.environ['GG_RMS_ROWS'] = env
        y = K.rmsnorm_fwd(x, gamma)
        dx, dg = K.rmsnorm_bwd(x, g, gamma, True)
        dxc, _ = K.rmsnorm_bwd(x, g, gamma, False, carry)
        res[tag] = (y, dx, dxc, dg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gx2, gg2, dg2 = K.rmsnorm_bwd2(x, g, carry, gamma, True)
        e1.record()
        torch.cuda.synchronize()
        res[tag + '2'] = (gx2, gg2, dg2, e0.elapsed_time(e1) / 10 * 1e3)
    xf = x.float().requires_grad_()
    gm = gamma.clone().requires_grad_()
    yr = xf / xf.norm(dim=-1, keepdim=True).clamp(min=K.RMS_EPS) * C ** 0.5 * gm
    gx, gg = torch.autograd.grad((yr * g.float()).sum(), [xf, gm])
    a2, b2 = res['rows2'], res['wave2']
    print('   bwd2 rows vs wave: gx %.2e gg %.2e dgamma %.2e | %.1f us vs %.1f us (%.2f vs %.2f TB/s)' % (
        rel(a2[0], b2[0]), rel(a2[1], b2[1]), rel(a2[2], b2[2]), a2[3], b2[3], rows * C * 10 / a2[3] / 1e6, rows * C * 10 / b2[3] / 1e6), flush=True)
    a, b = res['rows'], res['wave']
    print((rows, C), 'rows vs wave: y %.2e dx %.2e dx+carry %.2e dgamma %.2e | vs fp32: y %.2e dx %.2e dgamma rows %.2e wave %.2e' % (
        rel(a[0], b[0]), rel(a[1], b[1]), rel(a[2], b[2]), rel(a[3], b[3]), rel(a[0], yr), rel(a[1], gx), rel(a[3], gg), rel(b[3], gg)), flush=True)
