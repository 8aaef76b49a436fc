"""gg_modw_fwd on replicated samples: which of its outputs (s, a, d, per-sample weights) depends on the sample's slot in the batch?
(the config-5 replica spread starts inside modconv2d on a 64 -> 64 bank at 64x64, batch 16). Test infrastructure."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402
dev = torch.device('cuda', 0)
torch.manual_seed(0)
for b, N, O, I, layout in [(16, 2, 64, 64, 1), (32, 2, 64, 64, 1), (16, 2, 32, 32, 2), (16, 2, 128, 128, 1), (16, 4, 64, 64, 1), (16, 2, 64, 64, 0)]:
    w = torch.randn(N, O, I, 3, 3, device=dev)
    mod = torch.randn(2, I, device=dev).repeat(b // 2, 1).contiguous()
    kmod = torch.randn(2, N, device=dev).repeat(b // 2, 1).contiguous()
    wmix = None
    if layout == 1:
        wmix = torch.zeros(b, O, 9 * I, device=dev, dtype=torch.bfloat16)
    elif layout == 2:
        wmix = torch.zeros(b, 9, I // 16, 32, 16, device=dev, dtype=torch.bfloat16)
    s, a, d = K.modw_fwd(w, mod, kmod, True, 1e-8, I, O, wmix=wmix, layout=layout)
    sp = lambda t: float((t.float().reshape(b // 2, 2, -1) - t.float().reshape(b // 2, 2, -1)[:1]).abs().max())
    print(f'b={b} N={N} O={O} I={I} layout={layout}: spread s {sp(s):.2e} a {sp(a):.2e} d {sp(d):.2e}' + (f' wmix {sp(wmix):.2e}' if wmix is not None else ''), flush=True)
    if sp(d) > 0:
        r = d.reshape(b // 2, 2, -1)
        bad = (r != r[:1]).any(-1).any(-1)
        print('   replicas whose d differs from replica 0:', bad.nonzero().flatten().tolist())
