"""Shared by tests/golden/make_golden_c45.py (CPU, build container) and tests/test_config45_parity.py (-m gpu): BASELINE
configs 4 and 5 AT THEIR BASELINE DIMENSIONS, following the config-2 recipe of tests/c2_common.py (batch 2 on the CPU, the
same two samples replicated to the bench batch of 16 on the GPU, every `torch.randn` draw replayed by shape):

  c4  text-conditional GigaGAN 256x256 (reference README.md:68-95): G cap 8 / D cap 16 / dim_max 512, TextEncoder dim 64
      depth 4 over pre-computed (b, 77, 512) CLIP token encodings with ragged zero padding (gp.py:843-853), cross attention
      (gp.py:617-655), text-modulated predictors (gp.py:1459), matching-aware loss ON (gp.py:2432-2475).
  c5  UnetUpsampler 64 -> 256, dim 32 (reference README.md:104-125, unet_upsampler.py:447-898), `train_upsampler=True`,
      D cap 16 with the 128x128 multi-scale input.

Step-1 quantities: generator images + rgbs of a no-grad forward, discriminator logits + multi-scale maps on the real batch,
the losses of a plain discriminator step, a gradient-penalty discriminator step and a generator step THROUGH THE TRAINER
(`train_discriminator_step` / `train_generator_step`, optimizer updates suppressed so that all three start from the same
weights), and the flat parameter gradients they leave behind.
"""
from __future__ import annotations

import copy

import torch

import c2_common as c2

BASE_BATCH = c2.BASE_BATCH
GRAD_STRIDE = 389
TOKENS, CLIP_DIM = 77, 512

C4_TEXT = dict(dim=64, depth=4, clip_dim_latent=CLIP_DIM)
C4_G = dict(image_size=256, dim_capacity=8, dim_max=512, style_network=dict(dim=64, depth=4, dim_text_latent=64),
            num_skip_layers_excite=4, unconditional=False)
C4_D = dict(image_size=256, dim_capacity=16, dim_max=512, num_skip_layers_excite=4, unconditional=False)
C5_G = dict(image_size=256, style_network=dict(dim=64, depth=4), dim=32, input_image_size=64, unconditional=True)
C5_D = dict(image_size=256, dim_capacity=16, dim_max=512, num_skip_layers_excite=4, multiscale_input_resolutions=(128,),
            unconditional=True)

CONFIGS = ('c4', 'c5')


def text_encodings(batch=BASE_BATCH):
    """(b, 77, 512) token encodings of two captions of different length (trailing zero rows are padding)."""
    g = torch.Generator().manual_seed(11)
    enc = torch.randn(BASE_BATCH, TOKENS, CLIP_DIM, generator=g)
    enc[0, 23:] = 0
    enc[1, 51:] = 0
    return enc.repeat(batch // BASE_BATCH, 1, 1)


def real_images(batch=BASE_BATCH):
    return c2.real_images(batch)


def latents(batch=BASE_BATCH):
    return c2.latents(batch)


def _perturb_zero_params(module, seed):
    """zero-initialised parameters (Noise weights gp.py:928, zero-init output projections) would switch their branch off."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
            if p.numel() > 0 and float(p.abs().sum()) == 0.:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


def build_models(cfg):
    """our G and D of the config from torch.manual_seed(0) on the CPU (same initialisation code path as the reference's: the
    state dict is what the reference side loads)."""
    from gigagan_pytorch_amd.generator import Generator
    from gigagan_pytorch_amd.discriminator import Discriminator
    torch.manual_seed(0)
    if cfg == 'c4':
        G = Generator(text_encoder=dict(C4_TEXT), **C4_G)
        D = Discriminator(text_encoder=dict(C4_TEXT), **C4_D)
    else:
        from gigagan_pytorch_amd.unet_upsampler import UnetUpsampler
        G = UnetUpsampler(**C5_G)
        D = Discriminator(**C5_D)
    _perturb_zero_params(G, 123)
    return G, D


class Loader:
    """endless identical batches: images for c5, (images, encodings) for c4, resident on `device`."""

    def __init__(self, cfg, batch, device):
        self.cfg, self.batch_size = cfg, batch
        self.images = real_images(batch).to(device)
        self.enc = text_encodings(batch).to(device) if cfg == 'c4' else None

    def __iter__(self):
        while True:
            yield (self.images, self.enc) if self.cfg == 'c4' else self.images


def make_trainer(cfg, G, D, device, tmp, use_hip_graphs=False, aux_weight=0.):
    from gigagan_pytorch_amd import GigaGAN
    return GigaGAN(generator=copy.deepcopy(G), discriminator=copy.deepcopy(D), device=device, use_hip_graphs=use_hip_graphs,
                   train_upsampler=cfg == 'c5', apply_gradient_penalty_every=4, calc_multiscale_loss_every=1,
                   discr_aux_recon_loss_weight=aux_weight, generator_contrastive_loss_weight=0.,
                   create_ema_generator_at_init=False, model_folder=f'{tmp}/m', results_folder=f'{tmp}/r')


class _NoStep:
    """suppress an optimizer's update (the three steps all start from the fixture's weights); gradients stay in flat_g."""

    def __init__(self, opt):
        self.opt = opt

    def __enter__(self):
        self.saved = self.opt.step
        self.opt.step = lambda *a, **k: None

    def __exit__(self, *exc):
        self.opt.step = self.saved


def run_step_one(gan, cfg, batch):
    dev = gan.device
    out = {}
    gan.G.train()
    gan.D.train()
    text = dict(text_encodings=text_encodings(batch).to(dev)) if cfg == 'c4' else {}
    real = real_images(batch).to(dev)
    with c2.randn_replay(), torch.no_grad():
        if cfg == 'c5':
            from gigagan_pytorch_amd import ops
            lowres = ops.impl.resize_nearest(real, (64, 64))
            img, rgbs = gan.G(lowres_image=lowres, noise=latents(batch).to(dev), return_all_rgbs=True)
        else:
            img, rgbs = gan.G(noise=latents(batch).to(dev), return_all_rgbs=True, **text)
    out['img'] = img[:BASE_BATCH].float().cpu()
    out['rgbs'] = [r[:BASE_BATCH].float().cpu() for r in rgbs]
    out['img_all'] = img.float().cpu() if batch > BASE_BATCH else None
    with torch.no_grad():
        logits, ms, _ = gan.D(real, gan.D.real_images_to_rgbs(real), calc_aux_loss=False, **text)
    out['logits'] = logits[:, :BASE_BATCH].float().cpu() if logits.dim() > 1 else logits[:BASE_BATCH].float().cpu()
    out['ms'] = [m.reshape(-1, batch, *m.shape[1:])[:, :BASE_BATCH].float().cpu() for m in ms]
    it = iter(Loader(cfg, batch, dev))
    for name, gp in (('d_plain', False), ('d_gp', True)):
        with _NoStep(gan.D_opt), c2.randn_replay():
            L = gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
        out[name] = dict(divergence=float(L.divergence), multiscale=float(L.multiscale_divergence),
                         gradient_penalty=float(L.gradient_penalty), matching_aware=float(L.total_matching_aware_loss))
        out[name + '_grad'] = gan.D_opt.flat_g.detach().cpu().clone()
    with _NoStep(gan.G_opt), c2.randn_replay():
        L = gan.train_generator_step(batch_size=batch, dl_iter=it)
    out['g'] = dict(divergence=float(L.divergence), multiscale=float(L.multiscale_divergence))
    out['g_grad'] = gan.G_opt.flat_g.detach().cpu().clone()
    return out


def compress(out, gan):
    o = dict(out)
    for k, opt in (('d_plain_grad', gan.D_opt), ('d_gp_grad', gan.D_opt), ('g_grad', gan.G_opt)):
        flat = o.pop(k)
        o[k + '_sub'] = flat[::GRAD_STRIDE].clone()
        o[k + '_pnorm'] = c2.param_norms(opt, flat)
        o[k + '_norm'] = float(flat.norm())
    o.pop('img_all', None)
    return o
