"""multi-stream hipGraph capture probe: does work launched on a second stream (forked from / joined to the capturing stream) get
replayed, forward AND autograd backward (the engine runs each node on its forward's stream)? (diagnostic)"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K, ops   # noqa: E402
from gigagan_pytorch_amd.modules import Linear   # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
x = torch.randn(4, 64, 32, 32, device=dev).to(torch.bfloat16).requires_grad_()
l1, l2 = Linear(64, 32).to(dev), Linear(32, 64).to(dev)
params = [*l1.parameters(), *l2.parameters()]
for p in params:
    p.grad = torch.zeros_like(p)
H = ops.HipOps()


def body(use_side):
    H.side_streams = use_side
    with ops.use_impl(H):
        m, xt = H.global_mean(H.prepare(x), fork=True)

        def mlp(m):
            return torch.sigmoid(l2(torch.nn.functional.silu(l1(m))))[:, :, None, None]
        e = ops.run_on_side_stream(mlp, m)
        z = H.conv2d(xt, wconv, None, act='lrelu')
        e = ops.ready(e)
        y = H.channel_scale(z, e)
        loss = y.float().pow(2).mean()
        x.grad = None
        for p in params:
            p.grad.zero_()
        wconv.grad = None
        loss.backward()
    return loss.detach().clone()


wconv = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).requires_grad_()
res = {}
for use_side in (False, True):
    for _ in range(2):
        body(use_side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = body(use_side)
    with torch.no_grad():
        x.copy_(torch.randn_like(x) * 1.5)
    g.replay(); torch.cuda.synchronize()
    res[use_side] = (float(loss), x.grad.float().norm().item(), [p.grad.norm().item() for p in params], wconv.grad.norm().item())
    with torch.no_grad():
        torch.manual_seed(1)
print(res[False]); print(res[True])
