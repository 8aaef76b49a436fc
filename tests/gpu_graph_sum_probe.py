"""does a captured torch reduction replay correctly on this ROCm / PyTorch build? (per-block partials -> dgamma fold is `part.sum(0)`)
Replays with fresh inputs; prints the relative error of every replay."""
import sys
import torch
dev = torch.device('cuda', 0)
torch.manual_seed(0)
print(torch.__version__, torch.version.hip)


def probe(name, shape, fn, ref, n=5):
    x = torch.randn(*shape, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fn(x)
    errs = []
    for i in range(n):
        x.copy_(torch.randn(*shape, device=dev) * (3. ** i))
        g.replay()
        torch.cuda.synchronize()
        want = ref(x.double())
        errs.append(float((y.double() - want).abs().max() / want.abs().max()))
    e = fn(x)
    want = ref(x.double())
    print('%-14s %-16s replays %s   eager %.2g' % (name, shape, ' '.join('%.2g' % v for v in errs), float((e.double() - want).abs().max() / want.abs().max())),
          flush=True)


for shape in ((1024, 256), (2048, 512), (2048, 1024), (4096, 1024), (2048, 2048), (1024, 1024), (512, 4096), (8192, 256), (65536, 64)):
    probe('sum(0)', shape, lambda t: t.sum(0), lambda t: t.sum(0))
for shape in ((1 << 20,), (1 << 24,), (4096, 4096)):
    probe('sum()', shape, lambda t: t.sum(), lambda t: t.sum())
    probe('square.mean()', shape, lambda t: t.square().mean(), lambda t: t.square().mean())
for shape in ((16, 3, 256, 256), (32, 512, 16, 16)):
    probe('sum((1,2,3))', shape, lambda t: t.sum((1, 2, 3)), lambda t: t.sum((1, 2, 3)))
    probe('sum((0,2,3))', shape, lambda t: t.sum((0, 2, 3)), lambda t: t.sum((0, 2, 3)))
