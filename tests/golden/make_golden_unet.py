"""Generates tests/golden/unet_small.pt by running the UNMODIFIED reference UnetUpsampler and its building blocks
(/root/reference/gigagan_pytorch/unet_upsampler.py, with the dependency stubs of tests/oracle_stubs) on CPU fp32 with
fixed seeds. Run in the build container:

    python tests/golden/make_golden_unet.py

Pins the oracle's unet ops (linear attention, SDPA attention, max-pool + high-frequency skip, RMSNorm + SiLU) and our
UnetUpsampler host logic to the reference's outputs; travels to the GPU box where /root/reference does not exist.
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / 'tests' / 'oracle_stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

import gigagan_pytorch as ref  # noqa: E402
from gigagan_pytorch import unet_upsampler as ru  # noqa: E402
from helpers import UNET_SMALL  # noqa: E402

OUT = Path(__file__).resolve().parent


def clone(sd):
    return {k: v.clone() for k, v in sd.items()}


def main():
    torch.manual_seed(0)
    fx = {}
    x = torch.randn(2, 16, 8, 8)
    for name, mod in (('linattn', ru.LinearAttention(16, heads=2, dim_head=8)),
                      ('attn', ru.Attention(16, heads=2, dim_head=8, flash=False)),
                      ('lintr', ru.LinearTransformer(16, heads=2, dim_head=8)),
                      ('tr', ru.Transformer(16, heads=2, dim_head=8, flash_attn=True))):
        with torch.no_grad():
            for p in mod.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.2)      # gammas / biases off their trivial init
        fx[name] = dict(state=clone(mod.state_dict()), x=x, y=mod(x).detach())
    down = ru.Downsample(16, 24)
    y, hf = down(x)
    fx['down'] = dict(state=clone(down.state_dict()), x=x, y=y.detach(), hf=hf.detach())
    blk = ru.ResnetBlock(24, 16, num_conv_kernels=2, style_dims=[])
    xb = torch.randn(2, 24, 8, 8)
    mods = [torch.randn(2, 24) * 0.3, torch.randn(2, 2), torch.randn(2, 16) * 0.3, torch.randn(2, 2)]
    fx['resnet'] = dict(state=clone(blk.state_dict()), x=xb, mods=mods, y=blk(xb, conv_mods_iter=iter(mods)).detach())

    U = ref.UnetUpsampler(**UNET_SMALL)
    lowres = torch.rand(2, 3, 8, 8)
    z = torch.randn(2, 16)
    img, rgbs = U(lowres, noise=z, return_all_rgbs=True)
    fx['unet'] = dict(state=clone(U.state_dict()), lowres=lowres, z=z, img=img.detach(), rgbs=[r.detach() for r in rgbs],
                      split_dims=list(U.style_embed_split_dims))
    torch.save(fx, OUT / 'unet_small.pt')
    print('unet_small.pt', (OUT / 'unet_small.pt').stat().st_size)


if __name__ == '__main__':
    main()
