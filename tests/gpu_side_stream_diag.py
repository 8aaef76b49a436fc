"""does the side-stream branch keep the training state equal to the single-stream run under hipGraph replay? (diagnostic)"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd import ops, modules, discriminator, generator   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402

dev = torch.device('cuda', 0)
orig = ops.run_on_side_stream


def plain(fn, *inputs):
    return fn(*inputs)


def only_nograd(fn, *inputs):
    return orig(fn, *inputs) if not torch.is_grad_enabled() else fn(*inputs)


def only_grad(fn, *inputs):
    return orig(fn, *inputs) if torch.is_grad_enabled() else fn(*inputs)


fork_side = modules.squeeze_excite_fork


def fork_plain(se, x):
    m, x = ops.impl.global_mean(x, fork=True)
    for layer in list(se)[1:]:
        m = layer(m)
    return m, x


variants = {'none': (plain, None, None), 'all': (orig, None, None), 'nograd-only': (only_nograd, None, None),
            'grad-only': (only_grad, None, None), 'G-only': (orig, fork_plain, fork_side), 'D-only': (orig, fork_side, fork_plain)}
for name, (runner, dfork, gfork) in variants.items():
    ops.run_on_side_stream = runner
    discriminator.squeeze_excite_fork = dfork or fork_side
    generator.squeeze_excite_fork = gfork or fork_side
    torch.manual_seed(0)
    gan = bench.build_gan(256, dev, use_hip_graphs=True)
    it = cycle(SyntheticImages(32, 256, device=dev))
    out = []
    for step in range(3):
        d, g = gan.train_step(it, 32)
        out.append((float(d.divergence), float(g.divergence)))
    torch.cuda.synchronize()
    print(f'{name:12s}', ' '.join(f'({a:.3f},{b:.3f})' for a, b in out), flush=True)
    del gan
    torch.cuda.empty_cache()
