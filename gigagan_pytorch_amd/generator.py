"""Generator — same constructor / forward surface and state-dict as the reference (gp.py:947-1250),
re-expressed over the MI355X op set: every adaptive conv + noise + leaky-relu is one fused call, upsample +
blur is one stencil kernel, attention contractions and all 1x1/linear layers run on the MFMA GEMM.
"""
from __future__ import annotations

from functools import partial
from math import log2

import torch
from torch import nn

from . import ops
from .modules import (AdaptiveConv2DMod, CrossAttentionBlock, LeakyReLU, Linear, Conv2d, Noise, PixelShuffleUpsample,
                      SelfAttentionBlock, SqueezeExcite, StyleNetwork, Upsample, exists, squeeze_excite_fork)
from .text import TextEncoder


class BaseGenerator(nn.Module):
    pass


def _pair_args(conv, batch, mod, kernel_mod, noise, noise_weight, in_excite):
    """one AdaptiveConv2DMod call (modules.AdaptiveConv2DMod.forward) as the keyword dict ops.modconv_pair takes."""
    from .modules import tile_batch
    mod = tile_batch(mod, batch)
    kmod = None
    if conv.adaptive:
        assert exists(kernel_mod) and kernel_mod.numel() > 0
        kmod = tile_batch(kernel_mod, batch)
    return dict(weights=conv.weights, mod=mod, kernel_mod=kmod, demod=conv.demod, eps=conv.eps, noise=noise, noise_weight=noise_weight,
                act='lrelu', in_excite=in_excite)


def is_power_of_two(n):
    return log2(n).is_integer()


class Generator(BaseGenerator):
    def __init__(
        self,
        *,
        image_size,
        dim_capacity=16,
        dim_max=2048,
        channels=3,
        style_network=None,
        style_network_dim=None,
        text_encoder=None,
        dim_latent=512,
        self_attn_resolutions=(32, 16),
        self_attn_dim_head=64,
        self_attn_heads=8,
        self_attn_dot_product=True,
        self_attn_ff_mult=4,
        cross_attn_resolutions=(32, 16),
        cross_attn_dim_head=64,
        cross_attn_heads=8,
        cross_attn_ff_mult=4,
        num_conv_kernels=2,
        num_skip_layers_excite=0,
        unconditional=False,
        pixel_shuffle_upsample=False,
    ):
        super().__init__()
        self.channels = channels

        if isinstance(style_network, dict):
            style_network = StyleNetwork(**style_network)
        self.style_network = style_network
        assert exists(style_network) ^ exists(style_network_dim), \
            'style_network_dim must be given to the generator if StyleNetwork not passed in as style_network'
        if not exists(style_network_dim):
            style_network_dim = style_network.dim
        self.style_network_dim = style_network_dim

        if isinstance(text_encoder, dict):
            text_encoder = TextEncoder(**text_encoder)
        self.text_encoder = text_encoder
        self.unconditional = unconditional

        assert not (unconditional and exists(text_encoder))
        assert not (unconditional and exists(style_network) and style_network.dim_text_latent > 0)
        assert unconditional or (exists(text_encoder) and text_encoder.dim == style_network.dim_text_latent), \
            'the `dim_text_latent` on your StyleNetwork must be equal to the `dim` set for the TextEncoder'

        assert is_power_of_two(image_size)
        num_layers = int(log2(image_size) - 1)
        self.num_layers = num_layers

        is_adaptive = num_conv_kernels > 1
        dim_kernel_mod = num_conv_kernels if is_adaptive else 0
        split_dims = []
        adaptive_conv = partial(AdaptiveConv2DMod, kernel=3, num_conv_kernels=num_conv_kernels)

        self.init_block = nn.Parameter(torch.randn(dim_latent, 4, 4))
        self.init_conv = adaptive_conv(dim_latent, dim_latent)
        split_dims.extend([dim_latent, dim_kernel_mod])

        resolutions = [image_size // (2 ** e) for e in reversed(range(num_layers))]
        dim_layers = [min((2 ** (e + 1)) * dim_capacity, dim_max) for e in reversed(range(num_layers))]
        dim_layers = [dim_latent, *dim_layers]
        dim_pairs = list(zip(dim_layers[:-1], dim_layers[1:]))

        self.num_skip_layers_excite = num_skip_layers_excite
        self.layers = nn.ModuleList([])

        for ind, ((dim_in, dim_out), resolution) in enumerate(zip(dim_pairs, resolutions)):
            is_first, is_last = ind == 0, (ind + 1) == len(dim_pairs)
            should_excite = num_skip_layers_excite > 0 and (ind + num_skip_layers_excite) < len(dim_pairs)
            has_self_attn = resolution in self_attn_resolutions
            has_cross_attn = resolution in cross_attn_resolutions and not unconditional

            squeeze_excite = None
            if should_excite:
                dim_skip_in, _ = dim_pairs[ind + num_skip_layers_excite]
                squeeze_excite = SqueezeExcite(dim_in, dim_skip_in)

            resnet_block = nn.ModuleList([
                adaptive_conv(dim_in, dim_out), Noise(dim_out), LeakyReLU(fused=True),
                adaptive_conv(dim_out, dim_out), Noise(dim_out), LeakyReLU(fused=True),
            ])
            to_rgb = AdaptiveConv2DMod(dim_out, channels, 1, num_conv_kernels=1, demod=False)

            up_klass = Upsample if not pixel_shuffle_upsample else PixelShuffleUpsample
            upsample = up_klass(dim_in) if not is_first else None
            rgb_upsample = up_klass(channels) if not is_last else None

            self_attn = cross_attn = None
            if has_self_attn:
                self_attn = SelfAttentionBlock(dim_out, dim_head=self_attn_dim_head, heads=self_attn_heads,
                                               ff_mult=self_attn_ff_mult, dot_product=self_attn_dot_product)
            if has_cross_attn:
                cross_attn = CrossAttentionBlock(dim_out, dim_context=text_encoder.dim, dim_head=cross_attn_dim_head,
                                                 heads=cross_attn_heads, ff_mult=cross_attn_ff_mult)

            split_dims.extend([dim_in, dim_kernel_mod, dim_out, dim_kernel_mod, dim_out, 0])
            self.layers.append(nn.ModuleList([squeeze_excite, resnet_block, to_rgb, self_attn, cross_attn, upsample,
                                              rgb_upsample]))

        self.style_to_conv_modulations = Linear(style_network_dim, sum(split_dims))
        self.style_to_conv_modulations.out_f32 = True      # its column slices feed the fp32 coefficient kernels directly
        self.style_embed_split_dims = split_dims

        self.apply(self.init_)
        nn.init.normal_(self.init_block, std=0.02)

    def init_(self, m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')

    @property
    def total_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, styles=None, noise=None, texts=None, text_encodings=None, global_text_tokens=None,
                fine_text_tokens=None, text_mask=None, batch_size=1, return_all_rgbs=False):
        if not self.unconditional:
            if exists(texts) or exists(text_encodings):
                assert exists(texts) ^ exists(text_encodings), \
                    'either raw texts as List[str] or text_encodings (from clip) as Tensor is passed in, but not both'
                assert exists(self.text_encoder)
                kw = dict(texts=texts) if exists(texts) else dict(text_encodings=text_encodings)
                global_text_tokens, fine_text_tokens, text_mask = self.text_encoder(**kw)
            else:
                assert all(map(exists, (global_text_tokens, fine_text_tokens, text_mask))), \
                    'raw text or text embeddings were not passed in for conditional training'
        else:
            assert not any(map(exists, (texts, global_text_tokens, fine_text_tokens)))

        if not exists(styles):
            assert exists(self.style_network)
            if not exists(noise):
                noise = torch.randn((batch_size, self.style_network_dim), device=self.device)
            styles = self.style_network(noise, global_text_tokens)

        mods = self.style_to_conv_modulations(styles)
        batch = styles.shape[0]
        device = styles.device
        if torch.is_grad_enabled() and mods.requires_grad and mods.is_contiguous():
            # the differentiable pass hands every layer a dense (b, I) block (the coefficient kernels take dense rows): ONE gather into
            # layer-major order instead of ~30 per-layer `.contiguous()` copies of column slices; its backward is one index_add
            conv_mods = self._layer_major(mods, batch)
        else:
            conv_mods = mods.split(self.style_embed_split_dims, dim=-1)
        prepared = self._announce_adaptive_convs(conv_mods, batch)
        try:
            return self._synthesise(iter(conv_mods), batch, device, fine_text_tokens, text_mask, return_all_rgbs)
        finally:
            if prepared:
                ops.impl.modconv_release()

    def _layer_major(self, mods, batch):
        dims = tuple(self.style_embed_split_dims)
        key = (batch, mods.device)
        cache = self.__dict__.setdefault('_layer_major_idx', {})
        idx = cache.get(key)
        if idx is None:
            total = sum(dims)
            cols = torch.arange(total).split(dims)
            rows = torch.arange(batch)[:, None] * total
            idx = cache[key] = torch.cat([(rows + c[None, :]).reshape(-1) for c in cols]).to(mods.device)
        flat = mods.reshape(-1).index_select(0, idx)
        return tuple(c.view(batch, d) for c, d in zip(flat.split([batch * d for d in dims]), dims))

    def _announce_adaptive_convs(self, conv_mods, batch):
        """no-grad forward on an op set that batches the style-dependent work (ops.HipOps.modconv_prepare): hand over every demodulated
        3x3 adaptive conv with its modulation slices - all of them come out of the one projection above (gp.py:1160-1175) - so that
        their coefficients / per-sample weights are computed by ONE launch before the first convolution."""
        if torch.is_grad_enabled() or not hasattr(ops.impl, 'modconv_prepare'):
            return 0
        res = 4
        specs = [(self.init_conv.weights, conv_mods[0], conv_mods[1], res, res, False, self.init_conv.demod, self.init_conv.eps)]
        for ind, (squeeze_excite, block, *_rest) in enumerate(self.layers):
            upsample = _rest[3]
            if exists(upsample):
                res *= 2
            excited = self.num_skip_layers_excite > 0 and ind >= self.num_skip_layers_excite
            conv1, conv2 = block[0], block[3]
            m = conv_mods[2 + 6 * ind: 8 + 6 * ind]
            specs.append((conv1.weights, m[0], m[1], res, res, excited, conv1.demod, conv1.eps))
            specs.append((conv2.weights, m[2], m[3], res, res, False, conv2.demod, conv2.eps))
        specs = [sp for sp in specs if sp[1].shape[0] == batch]
        return ops.impl.modconv_prepare(specs)

    def _synthesise(self, conv_mods, batch, device, fine_text_tokens, text_mask, return_all_rgbs):

        x = self.init_block[None].expand(batch, -1, -1, -1)
        x = self.init_conv(x, mod=next(conv_mods), kernel_mod=next(conv_mods))

        rgb = None
        excitations = [None] * self.num_skip_layers_excite
        rgbs = []

        for squeeze_excite, block, to_rgb, self_attn, cross_attn, upsample, upsample_rgb in self.layers:
            conv1, noise1, _, conv2, noise2, _ = block
            if exists(upsample):
                x = upsample(x)
            if exists(squeeze_excite):
                excite_new, x = squeeze_excite_fork(squeeze_excite, x)
                excitations.append(excite_new)
            excite = excitations.pop(0) if excitations else None
            # `x = x * excite` (gp.py:1023-1024): x has one consumer, the first conv of the block, which takes the scale along
            # (no-grad: folded into its per-sample weights; otherwise the fused multiply with its one-pass backward)

            h, w = x.shape[-2:]
            # noise draws: same order, shape and device as the reference's Noise modules (gp.py:938)
            mod1, kmod1, mod2, kmod2 = next(conv_mods), next(conv_mods), next(conv_mods), next(conv_mods)
            nz1 = torch.randn(batch, 1, h, w, device=device)
            nz2 = torch.randn(batch, 1, h, w, device=device)
            y = None
            if not torch.is_grad_enabled() and hasattr(ops.impl, 'modconv_pair'):
                # no-grad pass: conv1 -> noise -> leaky-relu -> conv2 -> noise -> leaky-relu as ONE launch where the op set has a fused
                # form for the geometry (the 128x128 / 256x256 blocks: the intermediate map stays on chip); None = run them one by one
                y = ops.impl.modconv_pair(x, _pair_args(conv1, batch, mod1, kmod1, nz1, noise1.weight, excite),
                                          _pair_args(conv2, batch, mod2, kmod2, nz2, noise2.weight, None))
            if y is None:
                x = conv1(x, mod=mod1, kernel_mod=kmod1, in_excite=excite, noise=nz1, noise_weight=noise1.weight, act='lrelu')
                x = conv2(x, mod=mod2, kernel_mod=kmod2, noise=nz2, noise_weight=noise2.weight, act='lrelu')
            else:
                x = y

            if exists(self_attn):
                x = self_attn(x)
            if exists(cross_attn):
                x = cross_attn(x, context=fine_text_tokens, mask=text_mask)

            layer_rgb = to_rgb(x, mod=next(conv_mods), kernel_mod=next(conv_mods))
            rgb = layer_rgb if rgb is None else rgb + layer_rgb
            rgbs.append(rgb)
            if exists(upsample_rgb):
                rgb = upsample_rgb(rgb)

        assert len([*conv_mods]) == 0, 'convolutions were incorrectly modulated'

        if return_all_rgbs:
            return rgb, rgbs
        return rgb
