"""timing of gg_pack_weights / gg_wgrad_finish at discriminator-like sizes (test infrastructure)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from gigagan_pytorch_amd import kernels as K
dev = torch.device('cuda', 0)
tab = K.PackTable(dev, capacity=512)
ws = []
for _ in range(24):
    w = torch.randn(512, 512, 9, device=dev); ws.append(w)
    tab.register(w, 512, 512, 9, 'fwd'); tab.register(w, 512, 512, 9, 'bwd')
for _ in range(16):
    w = torch.randn(512, 512, 1, device=dev); ws.append(w)
    tab.register(w, 512, 512, 1, 'fwd'); tab.register(w, 512, 512, 1, 'bwd')
n = sum(w.numel() for w in ws)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timeit(tab.refresh)
print(f'pack refresh: {n/1e6:.1f}M params x 2 kinds: {ms:.3f} ms  ({n*2*6/ms/1e6:.0f} GB/s)')
g = torch.randn(9 * 512, 512, device=dev); dst = torch.zeros(512, 512, 9, device=dev)
ms = timeit(lambda: K.wgrad_finish(g, 512, 512, 9, 1.0, out=dst, accumulate=True), 50)
print(f'wgrad_finish 512x512x9 accumulate: {ms*1e3:.1f} us ({g.numel()*12/ms/1e6:.0f} GB/s)')
ms = timeit(lambda: (g.view(3, 3, 512, 512).permute(3, 2, 0, 1).contiguous()), 50)
print(f'torch permute-contiguous same: {ms*1e3:.1f} us')
