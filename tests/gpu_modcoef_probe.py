"""gg_modcoef_fwd / gg_modcoef_bwd launch times at the generator's bank shapes (batch 32, N = 2 kernels, 3x3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigagan_pytorch_amd import kernels as K
dev = torch.device('cuda', 0)
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


b = 32
for O, I in ((512, 512), (256, 512), (256, 256), (128, 256), (128, 128), (64, 128), (64, 64), (32, 64)):
    w = torch.randn(2, O, I, 3, 3, device=dev) * 0.05
    mod = torch.randn(b, I, device=dev) * 0.3
    km = torch.randn(b, 2, device=dev)
    Ip, Op = (I + 7) // 8 * 8, (O + 7) // 8 * 8
    s, a, d = K.modcoef_fwd(w, mod, km, True, 1e-8, Ip, Op)
    gs, ga, gd = torch.randn_like(s), torch.randn_like(a), torch.randn_like(d)
    gw = torch.zeros_like(w)
    tf = timeit(lambda: K.modcoef_fwd(w, mod, km, True, 1e-8, Ip, Op))
    tb = timeit(lambda: K.modcoef_bwd(w, km, s, d, gs, ga, gd, gw, 1e-8))
    gram = K.modgram(w)
    s1, a1, d1, tsum = K.modcoef_gram_fwd(gram, 2, mod, km, 1e-8, Ip, Op)
    tg = timeit(lambda: K.modgram(w))
    tf2 = timeit(lambda: K.modcoef_gram_fwd(gram, 2, mod, km, 1e-8, Ip, Op))
    tb2 = timeit(lambda: K.modcoef_gram_bwd(w, gram, km, s1, d1, tsum, gs, ga, gd, gw, 1e-8))
    print('O %4d I %4d  direct: fwd %6.1f us  bwd (2 kernels + zeros) %6.1f us | through Gram rows: gram %5.1f us  fwd %5.1f us  bwd (2 kernels) %6.1f us   d rel %.1e'
          % (O, I, tf, tb, tg, tf2, tb2, float((d1 - d).norm() / d.norm())), flush=True)
