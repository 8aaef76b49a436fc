"""times the fused attention kernels at the discriminator's 32x32 shape (test infrastructure; also the target of PMC runs).
usage: python tests/gpu_attn_probe.py [B n heads]"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from gigagan_pytorch_amd import kernels as K   # noqa: E402

B, n, h = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 1024, 8)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
mk = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
q, v, d_o, aq, av = mk(B, n, h * 64), mk(B, n, h * 64), mk(B, n, h * 64), mk(B, n, h * 64), mk(B, n, h * 64)
k0, v0, ak0, av0 = mk(h, 64), mk(h, 64), mk(h, 64), mk(h, 64)
alpha, beta = 0.25, -0.125
unit = 2.0 * n * n * 64 * B * h      # one n x n x 64 contraction over all heads


def timed(fn, iters=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


t, (o, lse) = timed(lambda: K.attn_fwd(q, q, v, k0, v0, h, alpha, beta))
print(f'fwd   {t:7.3f} ms  {2 * unit / t / 1e9:7.1f} TF')
t, outs = timed(lambda: K.attn_bwd(q, q, v, k0, v0, o, lse, d_o, h, alpha, beta, return_dvec=True))
print(f'bwd   {t:7.3f} ms  {7 * unit / t / 1e9:7.1f} TF  (dq 3 + dkv 4 contractions)')
dvec = outs[-1]
t, _ = timed(lambda: K.attn_bwd2(q, q, v, k0, v0, d_o, lse, dvec, aq, aq, av, ak0, av0, h, alpha, beta))
print(f'bwd2  {t:7.3f} ms  {22 * unit / t / 1e9:7.1f} TF  (stats 5 + q 9 + kv 8 contractions)')
