"""UnetUpsampler (BASELINE config 5: 64 -> 256) — the reference's constructor / forward surface and state-dict
(gigagan_pytorch/unet_upsampler.py, "unet.py": Downsample :82-160, RMSNorm :224-234, Block / ResnetBlock :238-310,
LinearAttention :312-349, Attention :351-380 over attend.py:83-110, FeedForward :384-390, Transformer /
LinearTransformer :394-443, UnetUpsampler :447-898) over the MI355X op set: the 62 style-modulated adaptive convolutions
run on the same modulate -> weight-stationary implicit GEMM -> mix/demodulate path as the generator's, 1x1 projections
and both attention flavours' contractions on the MFMA GEMM, blur / bilinear resizes on the separable stencil kernel.

Image path only: the reference's temporal (video) layers (`has_temporal_layers=True`, unet.py:162-220, :575-611) are out
of scope and raise.  `forward` additionally accepts the `lowres_image=` keyword, which is the name the
reference's own trainer passes (gp.py:2212) although its forward declares `lowres_image_or_video`.
"""
from __future__ import annotations

from functools import partial
from math import log2

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .generator import BaseGenerator, is_power_of_two
from .modules import (AdaptiveConv2DMod, Act, Conv2d, CrossAttentionBlock, Linear, PixelShuffleUpsample, StyleNetwork,
                      Upsample, default, exists)
from .text import TextEncoder


def cast_tuple(t, length=1):
    if isinstance(t, tuple):
        return t
    return (t,) * length


def null_iterator():
    while True:
        yield None


# ---- small modules ---------------------------------------------------------------------------------------

class RMSNorm(nn.Module):
    """F.normalize(x, dim=1) * gamma * sqrt(dim) over channels (unet.py:224-234); gamma is (dim,) here (the generator's
    ChannelRMSNorm keeps (dim,1,1)), same fused pass."""

    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(dim))

    def forward(self, x, act=None):
        return ops.impl.channel_rmsnorm(x, self.gamma, act=act)


class Downsample(nn.Module):
    """3x3 conv, then (unless `skip_downsample`) 2x2 max-pool of the conv output, handing back the high-frequency map
    `x - blur(x)` for the skip connection (unet.py:107-160)."""

    def __init__(self, dim, dim_out=None, skip_downsample=False, has_temporal_layers=False):
        super().__init__()
        assert not has_temporal_layers, 'temporal (video) layers are out of scope for the MI355X build'
        dim_out = default(dim_out, dim)
        self.skip_downsample = skip_downsample
        self.conv2d = Conv2d(dim, dim_out, 3, padding=1)
        self.register_buffer('filter', torch.Tensor([1., 2., 1.]))

    def forward(self, x):
        x = self.conv2d(x)
        if self.skip_downsample:
            return x, x[:, 0:0]
        return ops.impl.maxpool_highfreq(x)


class Block(nn.Module):
    """adaptive conv 3x3 -> RMSNorm -> SiLU (unet.py:238-270)."""

    def __init__(self, dim, dim_out, num_conv_kernels=0, conv_type='2d'):
        super().__init__()
        assert conv_type == '2d', 'temporal (1d) blocks are out of scope for the MI355X build'
        self.proj = AdaptiveConv2DMod(dim, dim_out, kernel=3, num_conv_kernels=num_conv_kernels)
        self.norm = RMSNorm(dim_out)
        self.act = nn.SiLU()

    def forward(self, x, conv_mods_iter=None):
        conv_mods_iter = default(conv_mods_iter, null_iterator())
        x = self.proj(x, mod=next(conv_mods_iter), kernel_mod=next(conv_mods_iter))
        return self.norm(x, act='silu')      # norm + SiLU in one pass


class ResnetBlock(nn.Module):
    def __init__(self, dim, dim_out, *, num_conv_kernels=0, conv_type='2d', style_dims=None):
        super().__init__()
        mod_dims = [dim, num_conv_kernels, dim_out, num_conv_kernels]
        if style_dims is not None:
            style_dims.extend(mod_dims)
        self.num_mods = len(mod_dims)
        self.block1 = Block(dim, dim_out, num_conv_kernels=num_conv_kernels, conv_type=conv_type)
        self.block2 = Block(dim_out, dim_out, num_conv_kernels=num_conv_kernels, conv_type=conv_type)
        self.res_conv = Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward(self, x, conv_mods_iter=None):
        h = self.block1(x, conv_mods_iter=conv_mods_iter)
        h = self.block2(h, conv_mods_iter=conv_mods_iter)
        if isinstance(self.res_conv, nn.Identity):
            return h + x.to(h.dtype)
        return self.res_conv(x, residual=h)          # h + res_conv(x), the add in the 1x1's epilogue


class LinearAttention(nn.Module):
    """softmax(q over features) (softmax(k over positions) v^T) (unet.py:312-349)."""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden_dim = dim_head * heads
        self.norm = RMSNorm(dim)
        self.to_qkv = Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Sequential(Conv2d(hidden_dim, dim, 1), RMSNorm(dim))

    def forward(self, x):
        x = self.norm(x)
        qkv = self.to_qkv(x)
        fused = getattr(ops.impl, 'linear_attention_qkv', None)
        out = fused(qkv, heads=self.heads, scale=self.scale) if fused is not None else None
        if out is None:
            q, k, v = qkv.chunk(3, dim=1)
            out = ops.impl.linear_attention(q, k, v, heads=self.heads, scale=self.scale)
        return self.to_out(out)


class Attend(nn.Module):
    """parameter-less slot of the reference's Attend (attend.py:34-110): softmax(q k^T / sqrt(d)) v, no dropout."""

    def __init__(self, dropout=0., flash=False):
        super().__init__()
        assert dropout == 0.
        self.flash = flash

    def forward(self, q, k, v):
        return ops.impl.attention(q, k, v, scale=q.shape[-1] ** -0.5)


class Attention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32, flash=False):
        super().__init__()
        self.heads = heads
        hidden_dim = dim_head * heads
        self.norm = RMSNorm(dim)
        self.attend = Attend(flash=flash)
        self.to_qkv = Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = Conv2d(hidden_dim, dim, 1)

    def forward(self, x, residual=None):
        b, _, h, w = x.shape
        x = self.norm(x)
        q, k, v = (t.reshape(b, self.heads, -1, h * w).transpose(2, 3) for t in self.to_qkv(x).chunk(3, dim=1))
        out = self.attend(q, k, v)                                      # (b, heads, n, d)
        out = out.transpose(2, 3).reshape(b, -1, h, w)
        return self.to_out(out, residual=residual)


def _gelu(x):
    return ops.impl.gelu(x)


def FeedForward(dim, mult=4):
    return nn.Sequential(RMSNorm(dim), Conv2d(dim, dim * mult, 1), Act(_gelu), Conv2d(dim * mult, dim, 1))


def _ff_residual(ff, x):
    norm, conv_in, act, conv_out = ff
    return conv_out(act(conv_in(norm(x))), residual=x)


class Transformer(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, depth=1, flash_attn=True, ff_mult=4):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([Attention(dim=dim, dim_head=dim_head, heads=heads, flash=flash_attn),
                           FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x, residual=x)
            x = _ff_residual(ff, x)
        return x


class LinearTransformer(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, depth=1, ff_mult=4):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([LinearAttention(dim=dim, dim_head=dim_head, heads=heads), FeedForward(dim=dim, mult=ff_mult)])
            for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = _ff_residual(ff, x)
        return x


# ---- the model ---------------------------------------------------------------------------------------------

class UnetUpsampler(BaseGenerator):
    def __init__(
        self,
        dim,
        *,
        image_size,
        input_image_size,
        init_dim=None,
        out_dim=None,
        text_encoder=None,
        style_network=None,
        style_network_dim=None,
        dim_mults=(1, 2, 4, 8, 16),
        channels=3,
        full_attn=(False, False, False, True, True),
        cross_attn=(False, False, False, True, True),
        flash_attn=True,
        self_attn_dim_head=64,
        self_attn_heads=8,
        self_attn_dot_product=True,
        self_attn_ff_mult=4,
        attn_depths=(1, 1, 1, 1, 1),
        temporal_attn_depths=(1, 1, 1, 1, 1),
        cross_attn_dim_head=64,
        cross_attn_heads=8,
        cross_ff_mult=4,
        has_temporal_layers=False,
        mid_attn_depth=1,
        num_conv_kernels=2,
        unconditional=True,
        skip_connect_scale=None,
    ):
        super().__init__()
        if has_temporal_layers:
            raise NotImplementedError('video / temporal layers are out of scope for the MI355X build (SURVEY.md §2)')
        self.can_upsample_video = False

        if isinstance(text_encoder, dict):
            text_encoder = TextEncoder(**text_encoder)
        self.text_encoder = text_encoder
        if isinstance(style_network, dict):
            style_network = StyleNetwork(**style_network)
        self.style_network = style_network
        assert exists(style_network) ^ exists(style_network_dim), \
            'either style_network or style_network_dim must be passed in'

        self.unconditional = unconditional
        assert unconditional ^ exists(text_encoder), \
            'if unconditional, text encoder should not be given, and vice versa'
        assert not (unconditional and exists(style_network) and style_network.dim_text_latent > 0)
        assert unconditional or text_encoder.dim == style_network.dim_text_latent, \
            'the `dim_text_latent` on your StyleNetwork must be equal to the `dim` set for the TextEncoder'

        assert is_power_of_two(image_size) and is_power_of_two(input_image_size), \
            'both output image size and input image size must be power of 2'
        assert input_image_size < image_size, 'input image size must be smaller than the output image size, thus upsampling'
        num_layer_no_downsample = int(log2(image_size) - log2(input_image_size))
        assert num_layer_no_downsample <= len(dim_mults), 'you need more stages in this unet for the level of upsampling'

        self.image_size = image_size
        self.input_image_size = input_image_size

        style_embed_split_dims = []
        self.channels = channels
        init_dim = default(init_dim, dim)
        self.init_conv = Conv2d(channels, init_dim, 7, padding=3)

        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        *_, mid_dim = dims
        in_out = list(zip(dims[:-1], dims[1:]))

        block_klass = partial(ResnetBlock, num_conv_kernels=num_conv_kernels, style_dims=style_embed_split_dims)

        full_attn = cast_tuple(full_attn, length=len(dim_mults))
        assert len(full_attn) == len(dim_mults)
        FullAttention = partial(Transformer, flash_attn=flash_attn)
        cross_attn = cast_tuple(cross_attn, length=len(dim_mults))

        self.skip_connect_scale = default(skip_connect_scale, 2 ** -0.5)

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        skip_connect_dims = []

        for ind, ((dim_in, dim_out), layer_full_attn, layer_cross_attn, layer_attn_depth) in enumerate(
                zip(in_out, full_attn, cross_attn, attn_depths)):
            should_not_downsample = ind < num_layer_no_downsample
            has_cross_attn = not unconditional and layer_cross_attn
            attn_klass = FullAttention if layer_full_attn else LinearTransformer

            skip_connect_dims.append(dim_in)
            skip_connect_dims.append(dim_in + (dim_out if not should_not_downsample else 0))

            self.downs.append(nn.ModuleList([
                block_klass(dim_in, dim_in),
                block_klass(dim_in, dim_in),
                CrossAttentionBlock(dim_in, dim_context=text_encoder.dim, dim_head=self_attn_dim_head,
                                    heads=self_attn_heads, ff_mult=self_attn_ff_mult) if has_cross_attn else None,
                attn_klass(dim_in, dim_head=self_attn_dim_head, heads=self_attn_heads, depth=layer_attn_depth),
                None,
                None,
                Downsample(dim_in, dim_out, skip_downsample=should_not_downsample),
            ]))

        self.mid_block1 = block_klass(mid_dim, mid_dim)
        self.mid_attn = FullAttention(mid_dim, dim_head=self_attn_dim_head, heads=self_attn_heads, depth=mid_attn_depth)
        self.mid_block2 = block_klass(mid_dim, mid_dim)
        self.mid_to_rgb = Conv2d(mid_dim, channels, 1)

        for ind, ((dim_in, dim_out), layer_cross_attn, layer_full_attn, layer_attn_depth) in enumerate(
                zip(reversed(in_out), reversed(full_attn), reversed(cross_attn), reversed(attn_depths))):
            # NB the reference zips (cross_attn, full_attn) in swapped order here (unet.py:596): kept, it decides the classes
            attn_klass = FullAttention if layer_full_attn else LinearTransformer
            has_cross_attn = not unconditional and layer_cross_attn

            self.ups.append(nn.ModuleList([
                PixelShuffleUpsample(dim_out, dim_in),
                Upsample(),
                None,
                None,
                Conv2d(dim_in, channels, 1),
                block_klass(dim_in + skip_connect_dims.pop(), dim_in),
                block_klass(dim_in + skip_connect_dims.pop(), dim_in),
                CrossAttentionBlock(dim_in, dim_context=text_encoder.dim, dim_head=self_attn_dim_head,
                                    heads=self_attn_heads, ff_mult=cross_ff_mult) if has_cross_attn else None,
                attn_klass(dim_in, dim_head=cross_attn_dim_head, heads=self_attn_heads, depth=layer_attn_depth),
                None,
                None,
            ]))

        self.out_dim = default(out_dim, channels)
        self.final_res_block = block_klass(dim, dim)
        self.final_to_rgb = Conv2d(dim, channels, 1)

        style_dim = style_network.dim if exists(style_network) else style_network_dim
        self.style_to_conv_modulations = Linear(style_dim, sum(style_embed_split_dims))
        self.style_to_conv_modulations.out_f32 = True      # its column slices feed the fp32 coefficient kernels directly
        self.style_embed_split_dims = style_embed_split_dims

    @property
    def allowable_rgb_resolutions(self):
        input_res_base = int(log2(self.input_image_size))
        output_res_base = int(log2(self.image_size))
        return [2 ** p for p in range(input_res_base, output_res_base)]

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def total_params(self):
        return sum(p.numel() for p in self.parameters())

    def resize_to_same_dimensions(self, x, size):
        return ops.impl.resize_bilinear(x, tuple(size))

    def forward(self, lowres_image_or_video=None, styles=None, noise=None, texts=None, global_text_tokens=None,
                fine_text_tokens=None, text_mask=None, return_all_rgbs=False, replace_rgb_with_input_lowres_image=True,
                lowres_image=None, text_encodings=None):
        x = default(lowres_image_or_video, lowres_image)
        assert exists(x), 'a low resolution image must be passed in'
        assert x.dim() == 4, 'video inputs need temporal layers, which are out of scope for the MI355X build'
        shape = x.shape
        batch_size = shape[0]
        assert tuple(shape[-2:]) == ((self.input_image_size,) * 2)

        if not self.unconditional:
            if exists(texts) or exists(text_encodings):
                assert exists(self.text_encoder)
                kw = dict(texts=texts) if exists(texts) else dict(text_encodings=text_encodings)
                global_text_tokens, fine_text_tokens, text_mask = self.text_encoder(**kw)
            else:
                assert all(map(exists, (global_text_tokens, fine_text_tokens, text_mask)))
        else:
            assert not any(map(exists, (texts, global_text_tokens, fine_text_tokens)))

        if not exists(styles):
            assert exists(self.style_network)
            noise = default(noise, torch.randn((batch_size, self.style_network.dim), device=self.device))
            styles = self.style_network(noise, global_text_tokens)

        conv_mods_list = self.style_to_conv_modulations(styles).split(self.style_embed_split_dims, dim=-1)
        prepared = self._announce_adaptive_convs(conv_mods_list, batch_size)
        try:
            return self._synthesise(x, iter(conv_mods_list), shape, fine_text_tokens, text_mask, return_all_rgbs)
        finally:
            if prepared:
                ops.impl.modconv_release()

    def _announce_adaptive_convs(self, conv_mods, batch):
        """no-grad forward on an op set that batches the style-dependent work (ops.HipOps.modconv_prepare): every adaptive conv of
        the unet (two per Block pair of each ResnetBlock, consumed in forward order from the one style projection,
        unet_upsampler.py:700-706) is announced with its modulation slices and the resolution it runs at, so that coefficients /
        per-sample weights are computed by a few batched launches (16 layers each) instead of one per layer (62 at config 5)."""
        if torch.is_grad_enabled() or not hasattr(ops.impl, 'modconv_prepare'):
            return 0
        specs, k = [], 0

        def add(resblock, res):
            nonlocal k
            for blk in (resblock.block1, resblock.block2):
                conv = blk.proj
                specs.append((conv.weights, conv_mods[k], conv_mods[k + 1], res, res, False, conv.demod, conv.eps))
                k += 2
        res = self.input_image_size
        for block1, block2, _, _, _, _, downsample in self.downs:
            add(block1, res)
            add(block2, res)
            if not downsample.skip_downsample:
                res //= 2
        add(self.mid_block1, res)
        add(self.mid_block2, res)
        for stage in self.ups:
            res *= 2
            add(stage[5], res)
            add(stage[6], res)
        add(self.final_res_block, res)
        if k != len(conv_mods):
            return 0                # (a structure this walk does not know: leave every layer to its own launch)
        specs = [sp for sp in specs if sp[1].shape[0] == batch]
        return ops.impl.modconv_prepare(specs)

    def _synthesise(self, x, conv_mods, shape, fine_text_tokens, text_mask, return_all_rgbs):
        lowres_images = x
        x = self.init_conv(ops.impl.prepare(x))

        h = []
        for block1, block2, cross_attn, attn, _, _, downsample in self.downs:
            x = block1(x, conv_mods_iter=conv_mods)
            h.append(x)
            x = block2(x, conv_mods_iter=conv_mods)
            x = attn(x)
            if exists(cross_attn):
                x = cross_attn(x, context=fine_text_tokens, mask=text_mask)
            skip_connect = x
            x, hf_fmap = downsample(x)
            if hf_fmap.shape[1] > 0:    # high-frequency map rides along the skip connection (videogigagan)
                skip_connect = torch.cat((skip_connect, hf_fmap.to(skip_connect.dtype)), dim=1)
            h.append(skip_connect)

        x = self.mid_block1(x, conv_mods_iter=conv_mods)
        x = self.mid_attn(x)
        x = self.mid_block2(x, conv_mods_iter=conv_mods)

        rgbs = []
        rgb = self.mid_to_rgb(x)
        rgbs.append(rgb)

        for upsample, upsample_rgb, _, _, to_rgb, block1, block2, cross_attn, attn, _, _ in self.ups:
            x = upsample(x)
            rgb = upsample_rgb(rgb)

            res1 = h.pop() * self.skip_connect_scale
            res2 = h.pop() * self.skip_connect_scale
            if x.shape[0] != res1.shape[0] or x.shape[2:] != res1.shape[2:]:
                res1 = self.resize_to_same_dimensions(res1, x.shape[2:])
                res2 = self.resize_to_same_dimensions(res2, x.shape[2:])

            x = torch.cat((x, res1.to(x.dtype)), dim=1)
            x = block1(x, conv_mods_iter=conv_mods)
            x = torch.cat((x, res2.to(x.dtype)), dim=1)
            x = block2(x, conv_mods_iter=conv_mods)

            if exists(cross_attn):
                x = cross_attn(x, context=fine_text_tokens, mask=text_mask)
            x = attn(x)

            rgb = to_rgb(x, residual=rgb)        # rgb + to_rgb(x)
            rgbs.append(rgb)

        x = self.final_res_block(x, conv_mods_iter=conv_mods)
        assert len([*conv_mods]) == 0

        rgb = self.final_to_rgb(x, residual=rgb)

        if not return_all_rgbs:
            return rgb

        # only the rgbs larger than the input, with the input image itself as the smallest (unet.py:887-893)
        rgbs = [t for t in rgbs if t.shape[-1] > shape[-1]]
        rgbs = [lowres_images, *rgbs]
        return rgb, rgbs
