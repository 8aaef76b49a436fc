"""Generates tests/golden/text_small.pt by running the UNMODIFIED reference text-conditional Generator / Discriminator
(gp.py:596-655 CrossAttention, :659-867 Transformer / TextEncoder, :1459 Predictor with AdaptiveConv2DMod, :1664-1723
text conditioning) on CPU fp32 with fixed seeds and pre-computed token encodings — CLIP itself (open_clip, weights not
available offline) is replaced by a parameter-less adapter object, exactly the case the reference supports through
`text_encodings=` (gp.py:843-852). Run in the build container:

    python tests/golden/make_golden_text.py
"""
import sys
from pathlib import Path

import torch
from torch import nn

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / 'tests' / 'oracle_stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

import gigagan_pytorch as ref  # noqa: E402
from gigagan_pytorch.gigagan_pytorch import gradient_penalty, aux_matching_loss  # noqa: E402
from helpers import TEXT_ENC, TEXT_CLIP_DIM, TEXT_G, TEXT_D, text_encodings  # noqa: E402

OUT = Path(__file__).resolve().parent


class PrecomputedClip(nn.Module):
    dim_latent = TEXT_CLIP_DIM

    def embed_texts(self, texts):
        raise RuntimeError('fixtures use pre-computed text_encodings')


def main():
    torch.manual_seed(0)
    G = ref.Generator(text_encoder=ref.TextEncoder(clip=PrecomputedClip(), **TEXT_ENC), **TEXT_G)
    D = ref.Discriminator(text_encoder=ref.TextEncoder(clip=PrecomputedClip(), **TEXT_ENC), **TEXT_D)
    with torch.no_grad():
        for n, p in G.named_parameters():
            if p.abs().sum() == 0:
                p.normal_(std=0.1)
    enc = text_encodings()
    z = torch.randn(2, 32)
    torch.manual_seed(1)
    img, rgbs = G(noise=z, text_encodings=enc, return_all_rgbs=True)
    g_tokens, fine, mask = G.text_encoder(text_encodings=enc)
    D.eval()
    real = torch.rand(2, 3, 16, 16).requires_grad_()
    logits, ms, _ = D(real, D.real_images_to_rgbs(real), text_encodings=enc, calc_aux_loss=False)
    # mismatched pairs (the matching-aware loss rolls the text batch, gp.py:2432-2475)
    logits_mis, ms_mis, _ = D(real, D.real_images_to_rgbs(real), text_encodings=enc.roll(1, 0), calc_aux_loss=False)
    gp = gradient_penalty(real, [logits, *ms], grad_output_weights=[1., *(0.1,) * len(ms)])
    loss = logits.mean() + 0.1 * sum(m.mean() for m in ms) + gp
    grads = torch.autograd.grad(loss, list(D.parameters()), allow_unused=True)
    names = [n for n, _ in D.named_parameters()]
    xs = torch.tensor([-3., -0.5, 0., 2.]); ys = torch.tensor([1.5, -2., 0.3, 4.])
    torch.save(dict(
        G={k: v.clone() for k, v in G.state_dict().items()}, D={k: v.clone() for k, v in D.state_dict().items()},
        enc=enc, z=z, img=img.detach(), rgbs=[r.detach() for r in rgbs], global_tokens=g_tokens.detach(),
        fine_tokens=fine.detach(), mask=mask, real=real.detach(), logits=logits.detach(), ms=[m.detach() for m in ms],
        logits_mis=logits_mis.detach(), ms_mis=[m.detach() for m in ms_mis], gp=gp.detach(),
        d_grads={n: g.detach() for n, g in zip(names, grads) if g is not None},
        mal=dict(real=xs, fake=ys, loss=aux_matching_loss(xs, ys))), OUT / 'text_small.pt')
    print('text_small.pt', (OUT / 'text_small.pt').stat().st_size)


if __name__ == '__main__':
    main()
