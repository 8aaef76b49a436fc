"""Generates tests/golden/*.pt by running the UNMODIFIED reference (/root/reference, with the inert dependency
stubs of tests/oracle_stubs) on CPU fp32 with fixed seeds. Run in the build container:

    python tests/golden/make_golden.py

The fixtures pin (a) our oracle restatement (oracle/torch_ops.py) and (b) our Generator/Discriminator host
logic to the reference's actual outputs, and travel to the GPU box where /root/reference does not exist.
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
from gigagan_pytorch.gigagan_pytorch import (AdaptiveConv2DMod, SelfAttention, Upsample, ChannelRMSNorm,  # noqa: E402
                                             gradient_penalty)
from helpers import SMALL_G, SMALL_D  # noqa: E402

OUT = Path(__file__).resolve().parent


def half(sd):
    return {k: v.clone() for k, v in sd.items()}


def ops_fixture():
    torch.manual_seed(0)
    fx = {}
    conv = AdaptiveConv2DMod(16, 24, 3, num_conv_kernels=2)
    x = torch.randn(2, 16, 8, 8); mod = torch.randn(2, 16) * 0.5; km = torch.randn(2, 2)
    fx['modconv'] = dict(weights=conv.weights.detach().clone(), x=x, mod=mod, kernel_mod=km, y=conv(x, mod, km).detach())
    torgb = AdaptiveConv2DMod(16, 3, 1, num_conv_kernels=1, demod=False)
    fx['torgb'] = dict(weights=torgb.weights.detach().clone(), x=x, mod=mod, y=torgb(x, mod, None).detach())
    for dot in (True, False):
        attn = SelfAttention(16, dim_head=8, heads=2, dot_product=dot)
        xa = torch.randn(2, 16, 6, 6)
        fx[f'selfattn_dot{int(dot)}'] = dict(state=half(attn.state_dict()), x=xa, y=attn(xa).detach())
    up = Upsample()
    xu = torch.randn(2, 5, 6, 7)
    fx['upsample'] = dict(x=xu, y=up(xu).detach())
    norm = ChannelRMSNorm(16)
    with torch.no_grad():
        norm.gamma.normal_()
    fx['rmsnorm'] = dict(gamma=norm.gamma.detach().clone(), x=x, y=norm(x).detach())
    imgs = torch.rand(2, 3, 32, 32)
    fx['resize'] = dict(x=imgs, y8=torch.nn.functional.interpolate(imgs, 8, mode='bilinear'),
                        y16=torch.nn.functional.interpolate(imgs, 16, mode='bilinear'))
    torch.save(fx, OUT / 'ops_small.pt')


def model_fixture():
    torch.manual_seed(0)
    G = ref.Generator(**SMALL_G)
    D = ref.Discriminator(**SMALL_D)
    with torch.no_grad():   # zero-initialised Noise weights would hide the noise path
        for n, p in G.named_parameters():
            if p.abs().sum() == 0:
                p.normal_(std=0.1)
    z = torch.randn(2, 32)
    torch.manual_seed(1)
    img, rgbs = G(noise=z, return_all_rgbs=True)
    D.eval()   # no dropout / random patches in the aux decoder path for the fixture
    real = torch.rand(2, 3, 32, 32).requires_grad_()
    rg = D.real_images_to_rgbs(real)
    logits, ms, _ = D(real, rg, calc_aux_loss=False)
    gp = gradient_penalty(real, [logits, *ms], grad_output_weights=[1., *(0.1,) * len(ms)])
    loss = logits.mean() + 0.1 * sum(m.mean() for m in ms) + gp
    grads = torch.autograd.grad(loss, list(D.parameters()), allow_unused=True)
    names = [n for n, _ in D.named_parameters()]
    torch.save(dict(
        G=half(G.state_dict()), D=half(D.state_dict()), z=z, img=img.detach(), rgbs=[r.detach() for r in rgbs],
        real=real.detach(), logits=logits.detach(), ms=[m.detach() for m in ms], gp=gp.detach(),
        d_grads={n: g.detach() for n, g in zip(names, grads) if g is not None}), OUT / 'model_small.pt')


if __name__ == '__main__':
    ops_fixture()
    model_fixture()
    for f in sorted(OUT.glob('*.pt')):
        print(f.name, f.stat().st_size)
