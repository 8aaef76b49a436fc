"""Which PyTorch (non-gg) ops does one training step issue, and from where? TorchDispatchMode census of the D step (plain and with
the gradient penalty) and the G step of a small model on the host-side emulator build: the SITES and COUNTS are those of the
config-2 step's code paths (bytes are not: see tests/gpu_op_census.py for the byte census on the GPU). Test infrastructure.
usage: python tests/cpu_op_census.py [d|dgp|g ...]"""
import collections
import sys
import traceback
from pathlib import Path

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
from gigagan_pytorch_amd import _C, GigaGAN   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

_C.bind(ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so')
G = dict(image_size=32, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2), unconditional=True,
         num_skip_layers_excite=2, self_attn_resolutions=(8,), self_attn_heads=2, self_attn_dim_head=16)
D = dict(image_size=32, dim_capacity=8, dim_max=32, unconditional=True, num_skip_layers_excite=2, attn_resolutions=(8,),
         attn_heads=2, attn_dim_head=16, multiscale_input_resolutions=(16,))
torch.manual_seed(0)
gan = GigaGAN(generator=G, discriminator=D, device='cpu', create_ema_generator_at_init=False, model_folder='/tmp/census_m',
              results_folder='/tmp/census_r')
it = cycle(SyntheticImages(2, 32))
gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=False)
gan.train_generator_step(batch_size=2, dl_iter=it)
SKIP = ('view', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'detach', 'alias',
        'as_strided', '::t.default', 'unbind', 'split', '_unsafe_view', 'empty', 'lift_fresh', '_local_scalar', 'chunk', 'narrow',
        'unflatten', 'new_empty', 'is_same_size', 'record_function', 'profiler')


class M(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(s in name for s in SKIP):
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            st = traceback.extract_stack()
            fr = [f for f in st if 'gigagan_pytorch_amd' in f.filename and 'autograd' not in f.filename]
            loc = ' < '.join(f'{f.filename.split("/")[-1]}:{f.lineno}' for f in fr[-2:][::-1]) if fr else 'engine'
            nd = torch._C._current_autograd_node()
            if nd is not None:
                loc += ' @' + nd.name()
            dt = str(ts[0].dtype).replace('torch.', '') if ts else ''
            shp = tuple(ts[0].shape) if ts else ()
            self.cnt[(name.replace('aten.', ''), loc, dt, len(shp))] += 1
        return out


for what in (sys.argv[1:] or ['d', 'dgp', 'g']):
    m = M()
    with m:
        if what == 'g':
            gan.train_generator_step(batch_size=2, dl_iter=it)
        else:
            gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=(what == 'dgp'))
    print(f'=== {what}: {sum(m.cnt.values())} non-view torch ops')
    for k, n in m.cnt.most_common():
        print(f'x{n:4d}  {k}')
