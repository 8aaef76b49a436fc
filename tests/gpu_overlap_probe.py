"""Debug aid (GPU box): test_in_backward_gradient_exchange_is_captured_with_the_step's runs compared entry by entry - is a difference
between the overlapped and the post-backward exchange run-to-run noise (two identical runs differ too) or tied to the overlap?"""
import gc
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import GigaGAN, distributed as gdist   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402
from helpers import C1_G, C1_D   # noqa: E402

d = torch.device('cuda', 0)
comm = gdist.enable_native_comm(d)
tmp = Path(tempfile.mkdtemp())
runs, names = {}, None
for tag, overlap, graphs in (('post_a', False, True), ('post_b', False, True), ('overlap', True, True), ('overlap_eager', True, False),
                             ('post_eager', False, False)):
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=d,
                  create_ema_generator_at_init=False, use_hip_graphs=graphs, model_folder=str(tmp / f'm{tag}'),
                  results_folder=str(tmp / f'r{tag}'))
    gan.overlap_grad_reduce = overlap
    torch.manual_seed(10)
    it = cycle(SyntheticImages(2, 64, device=d, seed=3))
    for _ in range(4):
        gan.train_step(it, 2)
    torch.cuda.synchronize()
    runs[tag] = (gan.D_opt.flat_p.clone(), gan.G_opt.flat_p.clone())
    if names is None:
        names = {}
        for which, opt, net in (('D', gan.D_opt, gan.D), ('G', gan.G_opt, gan.G)):
            id2name = {id(p): n for n, p in net.named_parameters()}
            names[which] = [(off, id2name.get(id(p), '?'), p.numel()) for p, off in zip(opt._all, opt.offsets)]
    del gan, it
    gc.collect()
ref = runs['post_a']
for tag, (dp, gp) in runs.items():
    for which, a, b in (('D', dp, ref[0]), ('G', gp, ref[1])):
        diff = (a - b).abs()
        nz = int((diff > 0).sum())
        print(f'{tag:14s} {which}: differing entries {nz:8d} / {a.numel()}  max abs {float(diff.max()):.3e}', flush=True)
        if nz and tag != 'post_a':
            idx = torch.nonzero(diff > 0).flatten()
            lo = int(idx.min()); hi = int(idx.max())
            hit = [(n, off, k) for off, n, k in names[which] if off <= hi and off + k > lo]
            cnt = []
            for n, off, k in hit:
                c = int(((idx >= off) & (idx < off + k)).sum())
                if c:
                    cnt.append((n, c, float(diff[off:off + k].max())))
            print('      ', cnt[:12], flush=True)
gdist.shutdown()
