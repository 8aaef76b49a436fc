"""Which PyTorch (non-gg) device ops does one eager training step launch, from where, and how many bytes do they touch?
TorchDispatchMode census of one plain D+G step at config 2 / batch 32 (test infrastructure; feeds the glue-fusion work).
usage: python tests/gpu_op_census.py [gp] [uncond|text|upsampler]"""
import collections
import sys
import traceback
from pathlib import Path

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

dev = torch.device('cuda', 0)
workload = next((a for a in sys.argv[1:] if a in ('uncond', 'text', 'upsampler')), 'uncond')
B = 32 if workload == 'uncond' else 16
gan = bench.build_gan(256, dev, use_hip_graphs=False, workload=workload)
it = iter(bench.SyntheticTextImages(B, 256, dev)) if workload == 'text' else cycle(SyntheticImages(B, 256, device=dev))
gp = 'gp' in sys.argv[1:]
STRIDES = 'strides' in sys.argv[1:]
for _ in range(2):
    gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=False)
    gan.train_generator_step(batch_size=B, dl_iter=it)
SKIP = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'detach', 'alias',
        'as_strided', 't.default', 'unbind', 'split', '_unsafe_view', 'empty', 'lift_fresh', '_local_scalar', 'chunk', 'narrow',
        'unflatten', 'new_empty', 'is_same_size', 'record_function', 'profiler')
cnt = collections.Counter()
byt = collections.Counter()


class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(s in name for s in SKIP):
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if isinstance(o, torch.Tensor)]
            nbytes = sum(t.numel() * t.element_size() for t in ts + outs)
            st = traceback.extract_stack()
            fr = [f for f in st if 'gigagan_pytorch_amd' in f.filename and 'autograd' not in f.filename]
            loc = ' < '.join(f'{f.filename.split("/")[-1]}:{f.lineno}' for f in fr[-2:][::-1]) if fr else 'engine'
            nd = torch._C._current_autograd_node()
            if nd is not None:
                loc += ' @' + nd.name()
            dt = str(ts[0].dtype).replace('torch.', '') if ts else ''
            shp = tuple(ts[0].shape) if (ts and nbytes > (32 << 20)) else ()
            if STRIDES and nbytes > (16 << 20) and ('add' in name or 'copy' in name or 'clone' in name or 'contiguous' in name):
                shp = (tuple(ts[0].shape), tuple(tuple(t.stride()) for t in ts[:2]), tuple(o.stride() for o in outs[:1]))
            key = (name.replace('aten.', ''), loc, dt, shp)
            cnt[key] += 1
            byt[key] += nbytes
        return out


with M():
    gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
    gan.train_generator_step(batch_size=B, dl_iter=it)
torch.cuda.synchronize()
print('non-view torch ops in the step:', sum(cnt.values()), ' total bytes touched: %.1f GB' % (sum(byt.values()) / 1e9))
print('--- by bytes')
for k, v in sorted(byt.items(), key=lambda kv: -kv[1])[:110]:
    print(f'{v / 1e6:10.1f} MB  x{cnt[k]:4d}  {k}')
print('--- by count')
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:80]:
    print(f'x{v:4d}  {byt[k] / 1e6:10.1f} MB  {k}')
