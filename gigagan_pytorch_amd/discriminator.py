"""Discriminator (+ Predictor, SimpleDecoder) — reference surface and state-dict (gp.py:1255-1317, :1444-1838),
built on the MI355X op set: every 3x3 / 7x7 / 1x1 conv (+bias +leaky-relu) is one implicit-GEMM MFMA launch,
space-to-depth and stride-2 subsampling are layout glue in front of a GEMM, L2-distance attention uses the
dot-product-with-key-bias identity.
"""
from __future__ import annotations

from functools import partial
from math import log2

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .modules import (AdaptiveConv2DMod, Conv2d, Downsample, LeakyReLU, Linear, Placeholder, SelfAttentionBlock,
                      SqueezeExcite, Upsample, conv_lrelu, exists, default, squeeze_excite_fork, tile_batch)
from .text import TextEncoder


def is_power_of_two(n):
    return log2(n).is_integer()


class SimpleDecoder(nn.Module):
    """auxiliary reconstruction decoder (gp.py:1255-1317)."""

    def __init__(self, dim, *, dims, patch_dim=1, frac_patches=1., dropout=0.5):
        super().__init__()
        assert 0 < frac_patches <= 1.
        self.patch_dim = patch_dim
        self.frac_patches = frac_patches
        self.dropout = nn.Dropout(dropout)
        dims = [dim, *dims]
        layers = [Conv2d(dim, dim, 3, padding=1)]
        for dim_in, dim_out in zip(dims[:-1], dims[1:]):
            layers.append(nn.Sequential(Upsample(dim_in), *conv_lrelu(dim_in, dim_out)))
        self.net = nn.Sequential(*layers)

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, fmap, orig_image):
        fmap = self.dropout(fmap)
        if self.frac_patches < 1.:
            batch, p = fmap.shape[0], self.patch_dim
            assert fmap.shape[-1] % p == 0 and orig_image.shape[-1] % p == 0

            def to_patches(t):   # 'b c (p1 h) (p2 w) -> b (p1 p2) c h w'
                b, c, hh, ww = t.shape
                t = t.reshape(b, c, p, hh // p, p, ww // p).permute(0, 2, 4, 1, 3, 5)
                return t.reshape(b, p * p, c, hh // p, ww // p)

            fmap, orig_image = to_patches(fmap), to_patches(orig_image)
            total = p * p
            num = max(int(self.frac_patches * total), 1)
            batch_arange = torch.arange(batch, device=fmap.device)[..., None]
            # the reference draws this on the CPU generator whatever the device (gp.py:1310): a host sync + H2D copy
            # per call, and not hipGraph-capturable. On the GPU the same distribution is drawn on the device
            # generator; `reference_rng = True` restores the reference's CPU stream (parity tests).
            if fmap.device.type == 'cuda' and not getattr(self, 'reference_rng', False):
                perm = torch.randn((batch, total), device=fmap.device).sort(dim=-1).indices[..., :num]
            else:
                perm = torch.randn((batch, total)).sort(dim=-1).indices[..., :num].to(fmap.device)
            fmap, orig_image = (t[batch_arange, perm].flatten(0, 1) for t in (fmap, orig_image))
        recon = self.net(fmap)
        return F.mse_loss(recon.float(), orig_image.float())


class Predictor(nn.Module):
    """multi-scale logit head (gp.py:1444-1498)."""

    def __init__(self, dim, depth=4, num_conv_kernels=2, unconditional=False):
        super().__init__()
        self.unconditional = unconditional
        self.residual_fn = Conv2d(dim, dim, 1)
        self.residual_scale = 2 ** -0.5
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            if unconditional:
                c1, a1 = conv_lrelu(dim, dim)
                c2, a2 = conv_lrelu(dim, dim)
            else:
                c1 = AdaptiveConv2DMod(dim, dim, 3, num_conv_kernels=num_conv_kernels)
                c2 = AdaptiveConv2DMod(dim, dim, 3, num_conv_kernels=num_conv_kernels)
                a1, a2 = LeakyReLU(fused=True), LeakyReLU(fused=True)
            self.layers.append(nn.ModuleList([c1, a1, c2, a2]))
        self.to_logits = Conv2d(dim, 1, 1)

    def forward(self, x, mod=None, kernel_mod=None):
        """reference gp.py:1482-1498: residual = conv1x1(x); per layer x = (conv2(conv1(x)) + x) * c; x + residual; to_logits.
        Unconditional convs: each layer's first conv hands its input on to the skip (Conv2d.forward `fork`: the skip's gradient
        joins inside that conv's data-gradient pass) and the merges are one pass each (ops.scaled_add, the last one including
        `+ residual`). (A merge cannot ride on conv2's epilogue: its leaky-relu mask is recovered from the sign of the stored
        output, which a residual added after the activation would destroy.)"""
        c = self.residual_scale
        if not self.unconditional:
            residual = self.residual_fn(x)
            for conv1, _, conv2, _ in self.layers:
                inner = x
                x = conv1(x, mod=mod, kernel_mod=kernel_mod, act='lrelu')
                x = conv2(x, mod=mod, kernel_mod=kernel_mod, act='lrelu')
                x = (x + inner) * c
            return self.to_logits(x + residual)
        # x -> first conv (fork) -> residual conv (fork) -> first skip: a chain, every gradient merge inside a conv's backward
        residual = None
        last = len(self.layers) - 1
        for i, (conv1, _, conv2, _) in enumerate(self.layers):
            h, inner = conv1(x, fork=True)
            if i == 0:
                residual, inner = self.residual_fn(inner, fork=True)
            x = ops.impl.scaled_add(conv2(h), inner, c, residual if i == last else None)
        return self.to_logits(x)


class Discriminator(nn.Module):
    def __init__(
        self,
        *,
        dim_capacity=16,
        image_size,
        dim_max=2048,
        channels=3,
        attn_resolutions=(32, 16),
        attn_dim_head=64,
        attn_heads=8,
        self_attn_dot_product=False,
        ff_mult=4,
        text_encoder=None,
        text_dim=None,
        filter_input_resolutions=True,
        multiscale_input_resolutions=(64, 32, 16, 8),
        multiscale_output_skip_stages=1,
        aux_recon_resolutions=(8,),
        aux_recon_patch_dims=(2,),
        aux_recon_frac_patches=(0.25,),
        aux_recon_fmap_dropout=0.5,
        resize_mode='bilinear',
        num_conv_kernels=2,
        num_skip_layers_excite=0,
        unconditional=False,
        predictor_depth=2,
    ):
        super().__init__()
        self.unconditional = unconditional
        assert not (unconditional and exists(text_encoder))
        assert is_power_of_two(image_size)
        assert all(map(is_power_of_two, attn_resolutions))

        if filter_input_resolutions:
            multiscale_input_resolutions = [r for r in multiscale_input_resolutions if r < image_size]
        assert len(set(multiscale_input_resolutions)) == len(multiscale_input_resolutions)
        assert all(is_power_of_two(r) and r < image_size for r in multiscale_input_resolutions)
        self.multiscale_input_resolutions = list(multiscale_input_resolutions)

        assert multiscale_output_skip_stages > 0
        multiscale_output_resolutions = [r // (2 ** multiscale_output_skip_stages) for r in multiscale_input_resolutions]
        assert all(4 <= r < image_size for r in multiscale_output_resolutions)
        if multiscale_input_resolutions and multiscale_output_resolutions:
            assert max(multiscale_input_resolutions) > max(multiscale_output_resolutions)
            assert min(multiscale_input_resolutions) > min(multiscale_output_resolutions)
        self.multiscale_output_resolutions = multiscale_output_resolutions

        assert all(map(is_power_of_two, aux_recon_resolutions))
        assert len(aux_recon_resolutions) == len(aux_recon_patch_dims) == len(aux_recon_frac_patches)
        self.aux_recon_resolutions_to_patches = {
            r: (p, f) for r, p, f in zip(aux_recon_resolutions, aux_recon_patch_dims, aux_recon_frac_patches)}
        assert resize_mode == 'bilinear', 'only the bilinear resize of the reference default is implemented'
        self.resize_mode = resize_mode

        num_layers = int(log2(image_size) - 1)
        self.num_layers = num_layers
        self.image_size = image_size

        resolutions = [image_size // (2 ** e) for e in range(num_layers)]
        dim_layers = [min(d, dim_max) for d in [channels, *[(2 ** (e + 1)) * dim_capacity for e in range(num_layers)]]]
        dim_last = dim_layers[-1]
        dim_pairs = list(zip(dim_layers[:-1], dim_layers[1:]))

        self.num_skip_layers_excite = num_skip_layers_excite
        self.residual_scale = 2 ** -0.5
        self.layers = nn.ModuleList([])

        upsample_dims = []
        predictor_dims = []
        dim_kernel_attn = num_conv_kernels if num_conv_kernels > 1 else 0

        for ind, ((dim_in, dim_out), resolution) in enumerate(zip(dim_pairs, resolutions)):
            is_first, is_last = ind == 0, (ind + 1) == len(dim_pairs)
            should_downsample = not is_last
            should_excite = (not is_first and num_skip_layers_excite > 0
                             and (ind + num_skip_layers_excite) < len(dim_pairs))
            has_attn = resolution in attn_resolutions
            has_multiscale_output = resolution in multiscale_output_resolutions
            has_aux = resolution in aux_recon_resolutions
            upsample_dims.insert(0, dim_in)

            squeeze_excite = None
            if should_excite:
                dim_skip_in, _ = dim_pairs[ind + num_skip_layers_excite]
                squeeze_excite = SqueezeExcite(dim_in, dim_skip_in)

            from_rgb = Conv2d(channels, dim_in, 7, padding=3)
            residual_conv = Conv2d(dim_in, dim_out, 1, stride=(2 if should_downsample else 1))
            if should_downsample:
                # (downsample(x) + residual) * c  ==  c * downsample(x) + [c * residual]: the residual conv carries c
                residual_conv.out_scale = self.residual_scale
            resnet_block = nn.Sequential(*conv_lrelu(dim_in, dim_out), *conv_lrelu(dim_out, dim_out))

            predictor = None
            if has_multiscale_output:
                predictor = Predictor(dim_out, num_conv_kernels=num_conv_kernels, depth=2, unconditional=unconditional)
                predictor_dims.extend([dim_out, dim_kernel_attn])

            aux_decoder = None
            if has_aux:
                patch_dim, frac = self.aux_recon_resolutions_to_patches[resolution]
                aux_decoder = SimpleDecoder(dim_out, dims=tuple(upsample_dims), patch_dim=patch_dim, frac_patches=frac,
                                            dropout=aux_recon_fmap_dropout)

            self.layers.append(nn.ModuleList([
                squeeze_excite,
                from_rgb,
                resnet_block,
                residual_conv,
                SelfAttentionBlock(dim_out, heads=attn_heads, dim_head=attn_dim_head, ff_mult=ff_mult,
                                   dot_product=self_attn_dot_product) if has_attn else None,
                predictor,
                aux_decoder,
                Downsample(dim_out) if should_downsample else None,
            ]))

        self.to_logits = nn.Sequential(
            Conv2d(dim_last, dim_last, 3, padding=1),
            Placeholder(lambda t: t.flatten(1)),           # 'b c h w -> b (c h w)'
            Linear(dim_last * (4 ** 2), 1),
            Placeholder(lambda t: t.squeeze(-1)),          # 'b 1 -> b'
        )

        assert unconditional or (exists(text_dim) ^ exists(text_encoder))
        if not unconditional:
            if isinstance(text_encoder, dict):
                text_encoder = TextEncoder(**text_encoder)
            self.text_dim = default(text_dim, text_encoder.dim if exists(text_encoder) else None)
            self.predictor_dims = predictor_dims
            self.text_to_conv_conditioning = Linear(self.text_dim, sum(predictor_dims)) if exists(self.text_dim) else None
        self.text_encoder = text_encoder

        self.apply(self.init_)

    def init_(self, m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in', nonlinearity='leaky_relu')

    def unused_parameters(self):
        """from_rgb convs of stages that take no multi-scale input never run: their
        grads stay None in the reference, so the fused optimizer must skip them rather than decay them."""
        out = []
        res = self.image_size
        for layer in self.layers:
            if res not in self.multiscale_input_resolutions:
                out.extend(layer[1].parameters())
            res //= 2
        return out

    def multiscale_parameters(self):
        """parameters that only receive a gradient when the multi-scale outputs are requested (the predictors and the text
        conditioning that modulates them): with `calc_multiscale_loss_every > 1` their grads stay None in the other steps."""
        out = []
        for layer in self.layers:
            if layer[5] is not None:
                out.extend(layer[5].parameters())
        t2c = getattr(self, 'text_to_conv_conditioning', None)
        if t2c is not None:
            out.extend(t2c.parameters())
        # the text encoder's tokens only reach the logits through that conditioning (gp.py:1717-1723): without the multi-scale
        # outputs the whole encoder (transformer, learned_global_token, project_in) is left without gradients in the reference
        te = getattr(self, 'text_encoder', None)
        if te is not None and not self.unconditional:
            out.extend(p for p in te.parameters() if p.requires_grad)
        return out

    def aux_parameters(self):
        """the auxiliary reconstruction decoders: trained only through D(real) with a positive aux loss weight (gp.py:2327-2335)."""
        out = []
        for layer in self.layers:
            if layer[6] is not None:
                out.extend(layer[6].parameters())
        return out

    def resize_image_to(self, images, resolution):
        return ops.impl.resize_bilinear(images, resolution)

    def real_images_to_rgbs(self, images):
        return [self.resize_image_to(images, r) for r in self.multiscale_input_resolutions]

    @property
    def total_params(self):
        return sum(p.numel() for p in self.parameters())

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, images, rgbs, texts=None, text_encodings=None, text_embeds=None, real_images=None,
                return_multiscale_outputs=True, calc_aux_loss=True, aux_rows=None):
        """`aux_rows=(a, b)` (an extension over the reference signature) restricts the auxiliary reconstruction loss to
        samples a..b of the batch — used when fake and real images share one forward pass."""
        if not self.unconditional:
            assert (exists(texts) ^ exists(text_encodings)) ^ exists(text_embeds), \
                'either texts as List[str] is passed in, or clip text_encodings as Tensor'
            if exists(texts):
                assert exists(self.text_encoder)
                text_embeds, *_ = self.text_encoder(texts=texts)
            elif exists(text_encodings):
                assert exists(self.text_encoder)
                text_embeds, *_ = self.text_encoder(text_encodings=text_encodings)
            assert exists(text_embeds), 'raw text or text embeddings were not passed into discriminator'
            conv_mods = iter(self.text_to_conv_conditioning(text_embeds).split(self.predictor_dims, dim=-1))
        else:
            assert not any(map(exists, (texts, text_embeds)))
            conv_mods = iter(())

        x = images
        assert tuple(x.shape[-2:]) == (self.image_size, self.image_size)
        batch = x.shape[0]

        rgbs_index = {t.shape[-1]: t for t in rgbs} if exists(rgbs) else {}
        missing = set(self.multiscale_input_resolutions) - set(rgbs_index.keys())
        assert not missing, f'rgbs of necessary resolution {self.multiscale_input_resolutions} were not passed in'

        multiscale_outputs = []
        aux_recon_losses = []
        excitations = [None] * (self.num_skip_layers_excite + 1)

        x = ops.impl.prepare(x)

        for squeeze_excite, from_rgb, block, residual_fn, attn, predictor, recon_decoder, downsample in self.layers:
            resolution = x.shape[-1]

            if exists(squeeze_excite):
                excite_new, x = squeeze_excite_fork(squeeze_excite, x)
                excitations.append(excite_new)
            excite = excitations.pop(0) if excitations else None
            if exists(excite):
                x = ops.impl.channel_scale(x, tile_batch(excite, x.shape[0]))

            batch_prev_stage = x.shape[0]
            if resolution in self.multiscale_input_resolutions:
                rgb = rgbs_index[resolution]
                feats = from_rgb(ops.impl.prepare(rgb))
                x = ops.impl.add_cat(x, feats)          # cat((x + feats, feats)) with feats tiled over the scale-major batch

            # x feeds the residual conv and the block: the block's first conv hands x on (fork), so that the residual branch's
            # gradient is added inside that conv's data-gradient pass
            y, x = block[0](x, fork=True)
            residual = residual_fn(x)
            for m in list(block)[1:]:
                y = m(y)
            x = y

            if exists(attn):
                x = attn(x)

            if exists(predictor):
                pred_kwargs = dict()
                if not self.unconditional:
                    pred_kwargs = dict(mod=next(conv_mods), kernel_mod=next(conv_mods))
                if return_multiscale_outputs:
                    rows, x = ops.impl.take_rows(x, batch_prev_stage, fork=True)
                    multiscale_outputs.append(predictor(rows, **pred_kwargs))

            if exists(downsample):
                x = downsample(x, residual=residual, scale=self.residual_scale)   # merge fused into the conv epilogue
            else:
                x = (x + residual) * self.residual_scale

            if exists(recon_decoder) and calc_aux_loss:
                # reference behaviour (Appendix B.5): first `batch` rows of the post-downsample tensor
                if aux_rows is None:
                    aux_recon_losses.append(recon_decoder(x[:batch], images))
                else:
                    aux_recon_losses.append(recon_decoder(x[aux_rows[0]:aux_rows[1]], images[aux_rows[0]:aux_rows[1]]))

        assert self.unconditional or len([*conv_mods]) == 0, 'convolutions were incorrectly modulated'

        logits = self.to_logits(x)
        logits = logits.reshape(-1, batch)     # '(s b) -> s b'
        return logits, multiscale_outputs, aux_recon_losses
