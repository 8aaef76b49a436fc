"""Which gg_bias_act_bwd launches does a config-2 cycle issue (rows, channels, with / without the leaky-relu mask, the autograd node
that asked), and what does each cost? HIP events around every call of one eager D step (plain / with the gradient penalty) and G step.
usage: python tests/gpu_bias_act_census.py"""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd import kernels as K   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

dev = torch.device('cuda', 0)
B = 32
gan = bench.build_gan(256, dev, use_hip_graphs=False, workload='uncond')
it = cycle(SyntheticImages(B, 256, device=dev))
for _ in range(2):
    gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=False)
    gan.train_generator_step(batch_size=B, dl_iter=it)
rec = []
orig = K.bias_act_bwd


def logged(dy, y, want_db, slope=0.2, partials=False):
    nd = torch._C._current_autograd_node()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(dy, y, want_db, slope, partials)
    e1.record()
    rec.append((phase, nd.name() if nd is not None else '-', dy.numel() // dy.shape[-1], dy.shape[-1], y is not None, bool(want_db), e0, e1))
    return out


K.bias_act_bwd = logged
for phase, fn in (('D', lambda: gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=False)),
                  ('Dgp', lambda: gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=True)),
                  ('G', lambda: gan.train_generator_step(batch_size=B, dl_iter=it))):
    fn()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for ph, node, rows, C, mask, db, e0, e1 in rec:
    a = agg.setdefault((ph, node, rows, C, mask, db), [0, 0.])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
for ph in ('D', 'Dgp', 'G'):
    tot = sum(v[1] for k, v in agg.items() if k[0] == ph)
    print(f'== {ph}: {sum(v[0] for k, v in agg.items() if k[0] == ph)} launches, {tot:.0f} us (events around eager calls)')
    for (p, node, rows, C, mask, db), (n, us) in sorted(((k, v) for k, v in agg.items() if k[0] == ph), key=lambda kv: -kv[1][1]):
        print(f'  x{n:3d} {us / n:8.1f} us each  rows {rows:8d} C {C:5d} mask {int(mask)} db {int(db)}  {node}')
