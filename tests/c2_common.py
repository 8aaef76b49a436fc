"""Shared by tests/golden/make_golden_c2.py (runs here, on CPU) and tests/test_config2_parity.py (-m gpu): BASELINE config 2
(unconditional 256x256, G dim_capacity 8 / D dim_capacity 16, dim_max 512; reference README.md:47-67, gp.py:949-973, :1502-1528)
built from a fixed seed, a deterministic replacement for every `torch.randn` draw of a step, and the step-1 quantities that
are compared between the MI355X path, the bf16-operand oracle and the unmodified reference:

  * generator images + the 7 multi-scale rgbs of a no-grad forward (the D-step's generator pass, gp.py:2297-2301)
  * discriminator logits + the 4 multi-scale predictor maps on a real batch (gp.py:1698-1838)
  * step-1 discriminator losses, plain and with the gradient penalty (gp.py:2227-2430), and the parameter gradient they
    leave behind; step-1 generator losses (gp.py:2491-2580) and the generator's parameter gradient.

Samples are independent in G and D (no batch statistics), and every loss is a mean over the batch, so a batch of 32 made of
16 copies of the same 2 samples has the same losses / gradients as the batch of 2: the GPU side runs at the bench's batch 32
(the launch planner's real config-2 choices: 256x256 tiles, split-K weight gradients, the direct convolution, the pack-table
bank operand), the CPU side at batch 2.
"""
from __future__ import annotations

import copy
import hashlib
from contextlib import contextmanager

import torch

C2_G = dict(image_size=256, dim_capacity=8, dim_max=512, style_network=dict(dim=64, depth=4), num_skip_layers_excite=4,
            unconditional=True)
C2_D = dict(image_size=256, dim_capacity=16, dim_max=512, num_skip_layers_excite=4, unconditional=True)
BASE_BATCH = 2
GRAD_STRIDE = 389         # the committed fixture keeps every 389th element of a flat gradient (and per-parameter norms)


def build_models():
    """config-2 G and D with the reference's initialisation from torch.manual_seed(0) on the CPU; the zero-initialised
    per-layer Noise weights (gp.py:928) are given small values so that the noise path takes part."""
    from gigagan_pytorch_amd.generator import Generator
    from gigagan_pytorch_amd.discriminator import Discriminator
    from gigagan_pytorch_amd.modules import Noise
    torch.manual_seed(0)
    G = Generator(**C2_G)
    D = Discriminator(**C2_D)
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for m in G.modules():
            if isinstance(m, Noise):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.1)
    return G, D


def weights_checksum(*modules) -> str:
    h = hashlib.sha256()
    for m in modules:
        for k, v in m.state_dict().items():
            h.update(k.encode())
            h.update(v.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()
                     if v.dtype != torch.bool else bytes(v.cpu().numpy()))
    return h.hexdigest()[:32]


def real_images(batch=BASE_BATCH):
    g = torch.Generator().manual_seed(7)
    x = torch.rand(BASE_BATCH, 3, 256, 256, generator=g)
    return x.repeat(batch // BASE_BATCH, 1, 1, 1)


@contextmanager
def randn_replay():
    """every torch.randn(shape...) inside the block returns a deterministic tensor that depends only on the trailing shape and
    on how many draws of that trailing shape came before: the first BASE_BATCH rows are drawn from a generator seeded by
    (trailing shape, count) and repeated along the batch. Trainer code running at batch 2 on the CPU and at batch 32 on the
    GPU (and the reference's Noise modules, gp.py:938) therefore see the same latents and per-layer noise."""
    counts: dict = {}
    orig = torch.randn

    def fake(*size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        size = tuple(int(s) for s in size)
        if len(size) < 2 or size[0] % BASE_BATCH or kw.get('generator') is not None:
            return orig(*size, **kw)
        tail = size[1:]
        n = counts.get(tail, 0)
        counts[tail] = n + 1
        seed = int.from_bytes(hashlib.sha256(repr((tail, n)).encode()).digest()[:4], 'little')
        base = orig((BASE_BATCH, *tail), generator=torch.Generator().manual_seed(seed))
        out = base.repeat(size[0] // BASE_BATCH, *([1] * len(tail)))
        dt = kw.get('dtype')
        if dt is not None:
            out = out.to(dt)
        dv = kw.get('device')
        return out.to(dv) if dv is not None else out

    torch.randn = fake
    try:
        yield
    finally:
        torch.randn = orig


def latents(batch=BASE_BATCH):
    return torch.randn(BASE_BATCH, 64, generator=torch.Generator().manual_seed(5)).repeat(batch // BASE_BATCH, 1)


def make_trainer(G, D, device, tmp, **kw):
    """our trainer around (copies of) the seeded models; the aux reconstruction loss is weighted 0 (its dropout mask and
    patch choice are the only draws of the step that cannot be replayed across devices), no EMA copy."""
    from gigagan_pytorch_amd import GigaGAN
    return GigaGAN(generator=copy.deepcopy(G), discriminator=copy.deepcopy(D), device=device, use_hip_graphs=False,
                   apply_gradient_penalty_every=4, calc_multiscale_loss_every=1, discr_aux_recon_loss_weight=0.,
                   create_ema_generator_at_init=False, model_folder=f'{tmp}/m', results_folder=f'{tmp}/r', **kw)


def run_step_one(gan, batch):
    """the quantities listed in the module docstring, through OUR trainer (gigagan.py) on whatever op implementation is
    active; returns CPU tensors / floats."""
    dev = gan.device
    out = {}
    gan.G.train()
    gan.D.train()
    with randn_replay(), torch.no_grad():
        img, rgbs = gan.G(noise=latents(batch).to(dev), return_all_rgbs=True)
    out['img'] = img[:BASE_BATCH].float().cpu()
    out['rgbs'] = [r[:BASE_BATCH].float().cpu() for r in rgbs]
    out['img_all'] = img.float().cpu() if batch > BASE_BATCH else None
    real = real_images(batch).to(dev)
    with torch.no_grad():
        logits, ms, _ = gan.D(real, gan.D.real_images_to_rgbs(real), calc_aux_loss=False)
    out['logits'] = logits[:, :BASE_BATCH].float().cpu()
    out['ms'] = [m.reshape(-1, batch, *m.shape[1:])[:, :BASE_BATCH].float().cpu() for m in ms]
    for name, gp in (('d_plain', False), ('d_gp', True)):
        gan.D_opt.zero_grad()
        with randn_replay():
            div, msd, gpl, _ = gan._d_micro(real, None, None, 1, gp, True)
        out[name] = dict(divergence=float(div), multiscale=float(msd), gradient_penalty=float(gpl))
        out[name + '_grad'] = gan.D_opt.flat_g.detach().cpu().clone()
    for p in gan.D.parameters():
        p.requires_grad_(False)
    try:
        gan.G_opt.zero_grad()
        with randn_replay():
            div, msd = gan._g_micro(batch, None, 1, True)
    finally:
        for p in gan.D.parameters():
            p.requires_grad_(True)
    out['g'] = dict(divergence=float(div), multiscale=float(msd))
    out['g_grad'] = gan.G_opt.flat_g.detach().cpu().clone()
    return out


def param_norms(opt, flat):
    """per-parameter L2 norms of a flat gradient (FlatAdamW layout)."""
    return torch.stack([flat[o:o + p.numel()].norm() for p, o in zip(opt._all, opt.offsets)])
