"""Drop-in surface: constructor options of Generator / Discriminator beyond the benchmark configs, each compared with
the unmodified reference imported live (same weights through the shared state-dict, same RNG stream) on the fp32 oracle
ops. Live only (needs /root/reference)."""
import pytest
import torch

from gigagan_pytorch_amd import ops, Generator, Discriminator
from oracle.torch_ops import OracleOps
from helpers import rel_err

TOL = 1e-5
BASE_G = dict(image_size=32, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2),
              unconditional=True, self_attn_heads=2, self_attn_dim_head=16)
BASE_D = dict(image_size=32, dim_capacity=8, dim_max=32, unconditional=True, attn_heads=2, attn_dim_head=16)

G_OPTS = [dict(pixel_shuffle_upsample=True), dict(num_conv_kernels=1), dict(num_conv_kernels=3),
          dict(self_attn_dot_product=False), dict(channels=1), dict(channels=4), dict(num_skip_layers_excite=3),
          dict(self_attn_resolutions=(8,), self_attn_ff_mult=2), dict(style_network=None, style_network_dim=32),
          dict(image_size=16)]
D_OPTS = [dict(channels=1), dict(predictor_depth=1), dict(self_attn_dot_product=True),
          dict(multiscale_input_resolutions=(16,)),
          dict(aux_recon_resolutions=(16, 8), aux_recon_patch_dims=(4, 2), aux_recon_frac_patches=(0.5, 0.25)),
          dict(num_skip_layers_excite=2), dict(filter_input_resolutions=False, multiscale_input_resolutions=(16, 8)),
          dict(ff_mult=2), dict(image_size=16), dict(num_conv_kernels=1), dict(aux_recon_fmap_dropout=0.)]


@pytest.mark.parametrize('opt', G_OPTS, ids=lambda o: ','.join(f'{k}={v}' for k, v in o.items()))
def test_generator_option_matches_reference(reference, opt):
    cfg = {**BASE_G, **opt}
    torch.manual_seed(0)
    Gr, G = reference.Generator(**cfg), Generator(**cfg)
    assert list(G.state_dict().keys()) == list(Gr.state_dict().keys())
    G.load_state_dict(Gr.state_dict())
    kw = dict(noise=torch.randn(2, 32)) if cfg.get('style_network') else dict(styles=torch.randn(2, 32))
    torch.manual_seed(1)
    img_r, rgbs_r = Gr(**kw, return_all_rgbs=True)
    with ops.use_impl(OracleOps()):
        torch.manual_seed(1)
        img, rgbs = G(**kw, return_all_rgbs=True)
    assert img.shape == img_r.shape and rel_err(img, img_r) < TOL
    assert len(rgbs) == len(rgbs_r) and all(rel_err(a, b) < TOL for a, b in zip(rgbs, rgbs_r))


@pytest.mark.parametrize('opt', D_OPTS, ids=lambda o: ','.join(f'{k}={v}' for k, v in o.items()))
def test_discriminator_option_matches_reference(reference, opt):
    cfg = {**BASE_D, **opt}
    torch.manual_seed(0)
    Dr, D = reference.Discriminator(**cfg), Discriminator(**cfg)
    assert list(D.state_dict().keys()) == list(Dr.state_dict().keys())
    D.load_state_dict(Dr.state_dict())
    x = torch.rand(2, cfg.get('channels', 3), cfg['image_size'], cfg['image_size'])
    torch.manual_seed(2)
    l_r, ms_r, aux_r = Dr(x, Dr.real_images_to_rgbs(x))                 # train mode: dropout + random patches
    with ops.use_impl(OracleOps()):
        torch.manual_seed(2)
        l, ms, aux = D(x, D.real_images_to_rgbs(x))
    assert rel_err(l, l_r) < TOL
    assert len(ms) == len(ms_r) and all(rel_err(a, b) < TOL for a, b in zip(ms, ms_r))
    assert len(aux) == len(aux_r) and all(rel_err(a, b) < TOL for a, b in zip(aux, aux_r))


def test_invalid_multiscale_output_stages_are_refused_like_the_reference(reference):
    cfg = {**BASE_D, 'multiscale_output_skip_stages': 2}
    with pytest.raises(AssertionError):
        reference.Discriminator(**cfg)
    with pytest.raises(AssertionError):
        Discriminator(**cfg)


# ---- UnetUpsampler options (`init_dim != dim` is left out: the reference's own final_res_block then fails, unet.py:629) ----

from helpers import UNET_SMALL, TEXT_ENC, TEXT_CLIP_DIM   # noqa: E402

U_OPTS = [dict(num_conv_kernels=3), dict(skip_connect_scale=1.0), dict(attn_depths=(2, 1, 1), mid_attn_depth=2),
          dict(full_attn=True), dict(full_attn=False), dict(self_attn_ff_mult=2), dict(channels=1), dict(image_size=64),
          'text']


@pytest.mark.parametrize('opt', U_OPTS, ids=lambda o: o if isinstance(o, str) else ','.join(f'{k}={v}' for k, v in o.items()))
def test_unet_upsampler_option_matches_reference(reference, opt):
    from torch import nn
    from gigagan_pytorch_amd.unet_upsampler import UnetUpsampler

    class PrecomputedClip(nn.Module):
        dim_latent = TEXT_CLIP_DIM

    torch.manual_seed(0)
    kw = {}
    if opt == 'text':       # text-conditional upsampler: cross attention in the last stages, tokens handed over directly
        cfg = {**UNET_SMALL, 'unconditional': False, 'style_network': dict(dim=16, depth=2, dim_text_latent=16),
               'cross_attn': (False, True, True)}
        Ur = reference.UnetUpsampler(text_encoder=reference.TextEncoder(clip=PrecomputedClip(), **TEXT_ENC), **cfg)
        U = UnetUpsampler(text_encoder=dict(clip_dim_latent=TEXT_CLIP_DIM, **TEXT_ENC), **cfg)
        mask = torch.ones(2, 5, dtype=torch.bool)
        mask[1, 3:] = False
        kw = dict(global_text_tokens=torch.randn(2, 16), fine_text_tokens=torch.randn(2, 5, 16), text_mask=mask)
    else:
        cfg = {**UNET_SMALL, **opt}
        Ur, U = reference.UnetUpsampler(**cfg), UnetUpsampler(**cfg)
    assert list(U.state_dict().keys()) == list(Ur.state_dict().keys())
    U.load_state_dict(Ur.state_dict())
    x, z = torch.rand(2, cfg.get('channels', 3), 8, 8), torch.randn(2, 16)
    with torch.no_grad():
        img_r, rgbs_r = Ur(x, noise=z, return_all_rgbs=True, **kw)
        with ops.use_impl(OracleOps()):
            img, rgbs = U(x, noise=z, return_all_rgbs=True, **kw)
    assert rel_err(img, img_r) < TOL
    assert len(rgbs) == len(rgbs_r) and all(rel_err(a, b) < TOL for a, b in zip(rgbs, rgbs_r))
