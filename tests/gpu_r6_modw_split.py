"""how long the forward's batched modulation launch takes when split: the coefficient-only items (the 4x4 .. 32x32 layers: needed first)
against the per-sample-weight items (64x64 .. 256x256: needed ~250 us later). Test infrastructure.   python tests/gpu_r6_modw_split.py"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench                                             # noqa: E402
from gigagan_pytorch_amd import kernels as K, ops        # noqa: E402

dev = torch.device('cuda', 0)
gan = bench.build_gan(256, dev, use_hip_graphs=False)
real = K.modw_multi
captured = []
K.modw_multi = lambda layers, **kw: (captured.append(layers), real(layers, **kw))[1]
with torch.no_grad():
    gan.G(noise=torch.randn(32, gan.G.style_network_dim, device=dev))
K.modw_multi = real
layers = captured[0]
coef = [ly for ly in layers if ly.get('coef', True)]
wmix = [ly for ly in layers if not ly.get('coef', True)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


print(f'all {len(layers)} items: {timed(lambda: real(layers)):.1f} us')
print(f'{len(coef)} coefficient-only items: {timed(lambda: real(coef)):.1f} us')
print(f'{len(wmix)} per-sample-weight items: {timed(lambda: real(wmix)):.1f} us')
for ly in wmix:
    print(f"   item {tuple(ly['w'].shape)} layout {ly.get('layout')}: {timed(lambda: real([ly])):.1f} us")
for ly in coef:
    print(f"   coefficient item {tuple(ly['w'].shape)}: {timed(lambda: real([ly])):.1f} us")
