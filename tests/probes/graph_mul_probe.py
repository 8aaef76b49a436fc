"""does a broadcast multiply + its autograd reduction behave under hipGraph replay? (fault localisation)"""
import torch
dev = torch.device('cuda', 0)
torch.manual_seed(0)
x0 = torch.randn(2, 16, 64, 64, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
e0 = torch.rand(2, 16, device=dev)
xs = x0.clone().requires_grad_(); es = e0.clone().requires_grad_()

def step():
    xs.grad = None; es.grad = None
    y = xs * es[:, :, None, None].to(xs.dtype)
    (y.float().square().sum() * 1e-3).backward()
    return xs.grad.float().abs().max().clone(), es.grad.abs().max().clone()

ref = [float(v) for v in step()]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = step()
for i in range(4):
    junk = torch.full((64 << 20,), float('nan'), device=dev); del junk
    g.replay(); torch.cuda.synchronize()
    print('replay', i, [float(v) for v in outs], 'eager ref', ref)
