"""Building blocks of the generator / discriminator with the reference's parameter names and shapes
(state-dict compatible), every forward routed through `ops.impl` (HIP kernels on MI355X).

Reference: gigagan_pytorch/gigagan_pytorch.py ("gp.py") lines 224-307 (norms, blur, up/down-sample,
squeeze-excite), 315-409 (AdaptiveConv2DMod), 513-594 (SelfAttention), 596-655 (CrossAttention),
726-778 (FeedForward and blocks), 871-940 (EqualLinear, StyleNetwork, Noise).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def exists(v):
    return v is not None


def default(*vals):
    for v in vals:
        if exists(v):
            return v
    return None


def tile_batch(t, batch):
    """reference `repeat(t, 'b ... -> (s b) ...')`: the multi-scale batch is scale-major (gp.py:365-366)."""
    if t.shape[0] == batch:
        return t
    if t.dim() == 4 and t.shape[2] * t.shape[3] > 1:    # cat keeps channels_last storage; repeat() would hand back an NCHW tensor and
        return torch.cat([t] * (batch // t.shape[0]), dim=0)        # every consumer (add, cat, conv) would fall onto strided kernels
    # ((b, C, 1, 1) rows - a skip-layer excitation - have no storage order to keep: repeat's backward is ONE sum, cat's was a slice and an
    # accumulation per copy)
    return t.repeat(batch // t.shape[0], *((1,) * (t.dim() - 1)))


# ---- parameter holders with kernel-backed forwards ------------------------------------------------------

class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters (weight (O,I,k,k), bias) — forward on the implicit-GEMM kernel.
    stride 2 is only used with 1x1 kernels (gp.py:1612) and is a pixel sub-sampling in front of a GEMM."""

    act = None
    out_scale = 1.0     # y = out_scale * (conv + bias): lets a following "(a + b) * c" merge be folded into its producers

    def forward(self, x, residual=None, fork=False, res_scale=1.0):
        """`residual` (same shape as the output) is added in the kernel's epilogue: conv(x) + bias + residual.
        `fork=True` returns (y, x'): x' is x for its second consumer; where the ops implementation fuses forks, that
        consumer's gradient is added inside this conv's data-gradient pass instead of by a separate accumulation."""
        k = self.kernel_size[0]
        if self.stride[0] != 1:
            assert k == 1 and self.padding[0] == 0 and self.stride[0] == 2
        else:
            assert self.padding[0] == k // 2
        kw = dict(act=self.act, stride=self.stride[0], scale=self.out_scale, residual=residual, res_scale=res_scale)
        if fork:
            if getattr(ops.impl, 'fuses_forks', False):
                return ops.impl.conv2d(x, self.weight, self.bias, fork=True, **kw)
            return ops.impl.conv2d(x, self.weight, self.bias, **kw), x
        return ops.impl.conv2d(x, self.weight, self.bias, **kw)


class Linear(nn.Linear):
    out_f32 = False     # set on an instance whose consumers read fp32 (the style -> modulation projection)

    def forward(self, x):
        if self.out_f32 and hasattr(ops.impl, 'linear_f32'):
            return ops.impl.linear_f32(x, self.weight, self.bias)
        return ops.impl.linear(x, self.weight, self.bias)


class LeakyReLU(nn.Module):
    """leaky_relu(0.2) (gp.py:109). `fused=True` marks an activation already applied in the producing
    conv's epilogue (the module stays in the container so Sequential indices match the reference)."""

    def __init__(self, fused=False):
        super().__init__()
        self.fused = fused

    def forward(self, x):
        return x if self.fused else F.leaky_relu(x, 0.2)


def conv_lrelu(dim_in, dim_out, k=3):
    """conv -> leaky_relu pair occupying two container slots, activation fused into the conv epilogue."""
    conv = Conv2d(dim_in, dim_out, k, padding=k // 2)
    conv.act = 'lrelu'
    return conv, LeakyReLU(fused=True)


class Act(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)


class Placeholder(nn.Module):
    """parameter-less slot (the reference has einops Reduce/Rearrange layers at these indices)."""

    def __init__(self, fn=None):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return x if self.fn is None else self.fn(x)


# ---- norms ----------------------------------------------------------------------------------------------

class ChannelRMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim, 1, 1))

    def forward(self, x, fork=False):
        """`fork=True` returns (norm(x), x'): x' feeds the skip connection around the normalised branch (see Conv2d.forward)."""
        if fork:
            if getattr(ops.impl, 'fuses_forks', False):
                return ops.impl.channel_rmsnorm(x, self.gamma, fork=True)
            return ops.impl.channel_rmsnorm(x, self.gamma), x
        return ops.impl.channel_rmsnorm(x, self.gamma)


class RMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return (F.normalize(x.float(), dim=-1) * self.scale * self.gamma).to(x.dtype)


# ---- resampling ---------------------------------------------------------------------------------------------

class Blur(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('f', torch.Tensor([1, 2, 1]))

    def forward(self, x):
        return ops.impl.blur(x)


class Upsample(nn.Sequential):
    """bilinear x2 + blur (gp.py:257-261); executed as ONE fused stencil kernel."""

    def __init__(self, *args):
        super().__init__(Placeholder(), Blur())

    def forward(self, x):
        return ops.impl.upsample_blur(x)


def space_to_depth(x):
    """'b c (h s1) (w s2) -> b (c s1 s2) h w' (gp.py:291)."""
    b, c, h, w = x.shape
    x = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4)
    return x.reshape(b, c * 4, h // 2, w // 2)


class Downsample(nn.Sequential):
    """space-to-depth 2x2 -> 1x1 conv 4C -> C (gp.py:289-293), same container layout as the reference
    (index 0: rearrange, index 1: conv); executed as ONE 2x2 / stride-2 gather conv, optionally with the
    discriminator's residual merge `(x + residual) * scale` (gp.py:1826) in its epilogue."""

    def __init__(self, dim):
        super().__init__(Placeholder(space_to_depth), Conv2d(dim * 4, dim, 1))

    def forward(self, x, residual=None, scale=1.0):
        conv = self[1]
        return ops.impl.downsample(x, conv.weight, conv.bias, residual=residual, scale=scale)


class PixelShuffleUpsample(nn.Module):
    """1x1 conv -> SiLU -> PixelShuffle(2) (gp.py:263-287)."""

    def __init__(self, dim, dim_out=None):
        super().__init__()
        dim_out = default(dim_out, dim)
        conv = Conv2d(dim, dim_out * 4, 1)
        self.net = nn.Sequential(conv, Act(F.silu), nn.PixelShuffle(2))
        self.init_conv_(conv)

    def init_conv_(self, conv):
        o, i, h, w = conv.weight.shape
        cw = torch.empty(o // 4, i, h, w)
        nn.init.kaiming_uniform_(cw)
        conv.weight.data.copy_(cw.repeat_interleave(4, dim=0))
        nn.init.zeros_(conv.bias.data)

    def forward(self, x):
        return self.net(x)


# ---- skip-layer excitation (gp.py:297-307) ----------------------------------------------------------------

class SqueezeExciteNet(nn.Sequential):
    """the reference's six-slot Sequential (same state-dict keys: `1.weight`, `3.weight`, ...); behind the pool the four middle
    modules run as one fused op where the op set has one (ops.HipOps.squeeze_excite_mlp), module by module otherwise."""

    def excite(self, m):
        """pooled (b, C) rows -> (b, O, 1, 1) excitation"""
        fused = getattr(ops.impl, 'squeeze_excite_mlp', None)
        e = fused(m, self[1], self[3]) if (fused is not None and len(self) == 6) else None
        if e is not None:
            return e[:, :, None, None]
        for layer in list(self)[1:]:
            m = layer(m)
        return m

    def forward(self, x):
        return self.excite(self[0](x))


def SqueezeExcite(dim, dim_out, reduction=4, dim_min=32):
    dim_hidden = max(dim_out // reduction, dim_min)
    return SqueezeExciteNet(
        Placeholder(lambda x: ops.impl.global_mean(x)),
        Linear(dim, dim_hidden),
        Act(F.silu),
        Linear(dim_hidden, dim_out),
        Act(torch.sigmoid),
        Placeholder(lambda x: x[:, :, None, None]),
    )


def squeeze_excite_fork(se, x):
    """(se(x), x'): the excitation of a SqueezeExcite stack and x for the trunk (its gradient and the pool's meet in one pass)."""
    m, x = ops.impl.global_mean(x, fork=True)
    return se.excite(m), x


# ---- adaptive conv (gp.py:315-409) ---------------------------------------------------------------------------

class AdaptiveConv2DMod(nn.Module):
    def __init__(self, dim, dim_out, kernel, *, demod=True, stride=1, dilation=1, eps=1e-8, num_conv_kernels=1):
        super().__init__()
        assert stride == 1 and dilation == 1, 'the GigaGAN models only use stride 1 / dilation 1'
        self.eps = eps
        self.dim_out = dim_out
        self.kernel = kernel
        self.stride = stride
        self.dilation = dilation
        self.adaptive = num_conv_kernels > 1
        self.weights = nn.Parameter(torch.randn((num_conv_kernels, dim_out, dim, kernel, kernel)))
        self.demod = demod
        nn.init.kaiming_normal_(self.weights, a=0, mode='fan_in', nonlinearity='leaky_relu')

    def forward(self, fmap, mod, kernel_mod=None, noise=None, noise_weight=None, act=None, in_excite=None):
        b = fmap.shape[0]
        mod = tile_batch(mod, b)
        if exists(kernel_mod):
            has_el = kernel_mod.numel() > 0
            assert self.adaptive or not has_el
            kernel_mod = tile_batch(kernel_mod, b) if has_el else None
        if self.adaptive:
            assert exists(kernel_mod)
        return ops.impl.modconv2d(fmap, self.weights, mod, kernel_mod if self.adaptive else None, demod=self.demod,
                                  eps=self.eps, noise=noise, noise_weight=noise_weight, act=act, in_excite=in_excite)


# ---- attention ------------------------------------------------------------------------------------------------

def _heads(t, h):
    """'b (h d) x y -> b h (x y) d'"""
    b, c, x, y = t.shape
    return t.reshape(b, h, c // h, x * y).transpose(2, 3)


def self_attention_unfused(impl, q, k, v, null_kv, heads, scale, l2):
    """reference data flow of SelfAttention.forward (gp.py:562-592): split heads, prepend the null key / value,
    similarity -> softmax -> aggregate (through `impl.attention`), merge heads."""
    b, _, x, y = q.shape
    q, k, v = (_heads(t, heads) for t in (q, k, v))
    nk, nv = (t[None, :, None, :].expand(b, -1, -1, -1).to(q.dtype) for t in null_kv)
    k = torch.cat((nk, k), dim=2)
    v = torch.cat((nv, v), dim=2)
    out = impl.attention(q, k, v, scale=scale, l2=l2)
    return out.transpose(2, 3).reshape(b, -1, x, y)


class SelfAttention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, dot_product=False):
        super().__init__()
        self.heads = heads
        self.scale = dim_head ** -0.5
        dim_inner = dim_head * heads
        self.dot_product = dot_product
        self.norm = ChannelRMSNorm(dim)
        self.to_q = Conv2d(dim, dim_inner, 1, bias=False)
        self.to_k = Conv2d(dim, dim_inner, 1, bias=False) if dot_product else None
        self.to_v = Conv2d(dim, dim_inner, 1, bias=False)
        self.null_kv = nn.Parameter(torch.randn(2, heads, dim_head))
        self.to_out = Conv2d(dim_inner, dim, 1, bias=False)

    def forward(self, fmap, skip=False):
        """`skip=True`: attn(fmap) + fmap (the block's residual, gp.py:757-758), the sum in to_out's epilogue and the skip
        gradient joining inside the norm's backward pass."""
        b, _, x, y = fmap.shape
        h = self.heads
        residual = None
        if skip:
            fmap, residual = self.norm(fmap, fork=True)
        else:
            fmap = self.norm(fmap)
        # the normalised map has two (dot product: three) consumers: chained forks, one gradient pass each
        if exists(self.to_k):
            q, fmap = self.to_q(fmap, fork=True)
            k, fmap = self.to_k(fmap, fork=True)
        else:
            q, fmap = self.to_q(fmap, fork=True)
            k = q
        v = self.to_v(fmap)
        out = ops.impl.self_attention(q, k, v, self.null_kv, heads=h, scale=self.scale, l2=not self.dot_product)
        return self.to_out(out, residual=residual)


class CrossAttention(nn.Module):
    def __init__(self, dim, dim_context, dim_head=64, heads=8):
        super().__init__()
        self.heads = heads
        self.scale = dim_head ** -0.5
        dim_inner = dim_head * heads
        kv_input_dim = default(dim_context, dim)
        self.norm = ChannelRMSNorm(dim)
        self.norm_context = RMSNorm(kv_input_dim)
        self.to_q = Conv2d(dim, dim_inner, 1, bias=False)
        self.to_kv = Linear(kv_input_dim, dim_inner * 2, bias=False)
        self.to_out = Conv2d(dim_inner, dim, 1, bias=False)

    def forward(self, fmap, context, mask=None):
        b, _, x, y = fmap.shape
        h = self.heads
        fmap = self.norm(fmap)
        context = self.norm_context(context)
        q = _heads(self.to_q(fmap), h)
        k, v = self.to_kv(context).chunk(2, dim=-1)
        k, v = (t.reshape(b, -1, h, t.shape[-1] // h).transpose(1, 2) for t in (k, v))
        out = ops.impl.attention(q, k, v, scale=self.scale, l2=False, key_mask=mask)
        out = out.transpose(2, 3).reshape(b, -1, x, y)
        return self.to_out(out)


def _gelu(x):
    return ops.impl.gelu(x)


def FeedForward(dim, mult=4, channel_first=False):
    dim_hidden = int(dim * mult)
    if channel_first:
        return nn.Sequential(ChannelRMSNorm(dim), Conv2d(dim, dim_hidden, 1), Act(_gelu), Conv2d(dim_hidden, dim, 1))
    return nn.Sequential(RMSNorm(dim), Linear(dim, dim_hidden), Act(_gelu), Linear(dim_hidden, dim))


def _ff_residual(ff, x):
    """ff(x) + x for the channel-first FeedForward (norm, 1x1, gelu, 1x1) with the skip added in the last 1x1's epilogue."""
    norm, conv_in, act, conv_out = ff
    n, x = norm(x, fork=True)
    fused = getattr(ops.impl, 'ff_tail', None)
    if fused is not None and conv_in.out_scale == 1.0 and conv_out.out_scale == 1.0 and conv_in.act is None and conv_out.act is None:
        y = fused(n, conv_in.weight, conv_in.bias, conv_out.weight, conv_out.bias, x)      # GELU on the GEMM epilogues (ops.FFTailFn)
        if y is not None:
            return y
    return conv_out(act(conv_in(n)), residual=x)


class SelfAttentionBlock(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, ff_mult=4, dot_product=False):
        super().__init__()
        self.attn = SelfAttention(dim=dim, dim_head=dim_head, heads=heads, dot_product=dot_product)
        self.ff = FeedForward(dim=dim, mult=ff_mult, channel_first=True)

    def forward(self, x):
        x = self.attn(x, skip=True)        # attn(x) + x, the skip added in to_out's epilogue (gp.py:757-758)
        x = _ff_residual(self.ff, x)
        return x


class CrossAttentionBlock(nn.Module):
    def __init__(self, dim, dim_context, dim_head=64, heads=8, ff_mult=4):
        super().__init__()
        self.attn = CrossAttention(dim=dim, dim_context=dim_context, dim_head=dim_head, heads=heads)
        self.ff = FeedForward(dim=dim, mult=ff_mult, channel_first=True)

    def forward(self, x, context, mask=None):
        x = self.attn(x, context=context, mask=mask) + x
        x = _ff_residual(self.ff, x)
        return x


# ---- text transformer (gp.py:659-867) — small (77 tokens); config 4 ----------------------------------------------

class TextAttention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8):
        super().__init__()
        self.heads = heads
        self.scale = dim_head ** -0.5
        dim_inner = dim_head * heads
        self.norm = RMSNorm(dim)
        self.to_qkv = Linear(dim, dim_inner * 3, bias=False)
        self.null_kv = nn.Parameter(torch.randn(2, heads, dim_head))
        self.to_out = Linear(dim_inner, dim, bias=False)

    def forward(self, encodings, mask=None):
        b, n, _ = encodings.shape
        h = self.heads
        x = self.norm(encodings)
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)
        q, k, v = (t.reshape(b, n, h, -1).transpose(1, 2) for t in (q, k, v))
        nk, nv = (t[None, :, None, :].expand(b, -1, -1, -1).to(q.dtype) for t in self.null_kv)
        k = torch.cat((nk, k), dim=2)
        v = torch.cat((nv, v), dim=2)
        if exists(mask):
            mask = F.pad(mask, (1, 0), value=True)
        out = ops.impl.attention(q, k, v, scale=self.scale, l2=False, key_mask=mask)
        out = out.transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class Transformer(nn.Module):
    def __init__(self, dim, depth, dim_head=64, heads=8, ff_mult=4):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([TextAttention(dim=dim, dim_head=dim_head, heads=heads), FeedForward(dim=dim, mult=ff_mult)])
            for _ in range(depth)])
        self.norm = RMSNorm(dim)

    def forward(self, x, mask=None):
        for attn, ff in self.layers:
            x = attn(x, mask=mask) + x
            x = ff(x) + x
        return self.norm(x)


# ---- style mapping network (gp.py:871-921) ------------------------------------------------------------------------

class EqualLinear(nn.Module):
    def __init__(self, dim, dim_out, lr_mul=1, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(dim_out, dim))
        if bias:
            self.bias = nn.Parameter(torch.zeros(dim_out))
        self.lr_mul = lr_mul

    def forward(self, x, act=None):
        return ops.impl.equal_linear(x, self.weight, getattr(self, 'bias', None), self.lr_mul, act)


class StyleNetwork(nn.Module):
    def __init__(self, dim, depth, lr_mul=0.1, dim_text_latent=0):
        super().__init__()
        self.dim = dim
        self.dim_text_latent = dim_text_latent
        layers = []
        for i in range(depth):
            dim_in = (dim + dim_text_latent) if i == 0 else dim
            layers.extend([EqualLinear(dim_in, dim, lr_mul), LeakyReLU()])
        self.net = nn.Sequential(*layers)

    def forward(self, x, text_latent=None):
        x = F.normalize(x.float(), dim=1)
        if self.dim_text_latent > 0:
            assert exists(text_latent)
            x = torch.cat((x, text_latent.float()), dim=-1)
        mods = list(self.net)           # [EqualLinear, LeakyReLU] pairs: the activation rides on the linear layer's epilogue
        for lin in mods[0::2]:
            x = lin(x, act='lrelu')
        return x


class Noise(nn.Module):
    """x + weight * noise (gp.py:925-940). In the generator the add (and the following leaky-relu) is
    executed inside the adaptive conv's epilogue; this module holds the parameter and the stand-alone path."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(dim, 1, 1))

    def forward(self, x, noise=None):
        b, _, h, w = x.shape
        if not exists(noise):
            noise = torch.randn(b, 1, h, w, device=x.device)
        return (x.float() + self.weight * noise).to(x.dtype)
