"""Functional op set of the GigaGAN G+D step on MI355X.

Every function here takes / returns *logical NCHW* tensors (the reference's module-boundary convention);
activations are stored channels_last (= NHWC in memory) bf16, parameters stay fp32.  All dense
contractions run in the hand-written HIP kernels behind the C ABI (`kernels.py`), wrapped in
`torch.autograd.Function`s that are closed under differentiation (conv <-> wgrad <-> conv, gemm <-> gemm),
so the gradient penalty's double backward (reference gp.py:120-155) needs no special casing.  PyTorch is
used for storage, autograd bookkeeping and the small pointwise glue listed in DESIGN.md §coverage.

The active implementation is the module-level `impl` object.  The product default is `HipOps`; it raises
if the native library or a GPU tensor is missing.  Tests install `oracle.torch_ops.OracleOps` (a plain
fp32 restatement) through `use_impl()` to check the host-side model assembly on CPU.
"""
from __future__ import annotations

import math
from contextlib import contextmanager

import os
import weakref

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import kernels as K

ACT_DTYPE = torch.bfloat16
LRELU_SLOPE = 0.2
second_order = False   # set (by the trainer) while a graph that will be differentiated twice is being built
shape_fallbacks: dict = {}     # name -> count of calls that took a SHAPE-conditional tensor-algebra form instead of the HIP kernel (ragged
                               # channel counts, non-dense views ...). The BASELINE-dimension parity tests assert it stays empty: at the
                               # benchmark's configurations every such op runs on its kernel


def _shape_fallback(name):
    shape_fallbacks[name] = shape_fallbacks.get(name, 0) + 1
inputs_only = False    # set while a backward pass is run only for gradients w.r.t. activations (the gradient penalty's
                       # d out / d images): custom Functions then skip their parameter gradients, which autograd would
                       # discard anyway but cannot prune inside a Function


# --------------------------------------------------------------------------------------------------
# layout glue
# --------------------------------------------------------------------------------------------------

def to_act(x: torch.Tensor) -> torch.Tensor:
    """logical NCHW tensor -> bf16, channels_last storage (no-op when already so)."""
    if x.dim() == 4:
        if x.dtype != ACT_DTYPE:
            return x.to(ACT_DTYPE, memory_format=torch.channels_last)      # cast and re-layout in ONE pass
        return x.contiguous(memory_format=torch.channels_last)
    if x.dtype != ACT_DTYPE:
        x = x.to(ACT_DTYPE)
    return x


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """(b, C, H, W) channels_last -> the (b, H, W, C) contiguous view the kernels consume."""
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x: torch.Tensor) -> torch.Tensor:
    """(b, H, W, C) contiguous -> logical (b, C, H, W) with channels_last strides (a view)."""
    return x.permute(0, 3, 1, 2)


def _round8(n: int) -> int:
    return (n + 7) // 8 * 8


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """(O, I, kh, kw) any float dtype -> bf16 (O, kh*kw*I) with the reduction ordered [kh][kw][ci]."""
    o = w.shape[0]
    return w.permute(0, 2, 3, 1).reshape(o, -1).to(ACT_DTYPE).contiguous()


def flip_transpose(w: torch.Tensor) -> torch.Tensor:
    """weights of the adjoint (backward-data) convolution: spatial flip, in/out swapped."""
    return w.flip(2, 3).transpose(0, 1)


# ---- packed-weight cache -----------------------------------------------------------------------------------------
# A parameter is packed to the kernels' bf16 layouts at most once per optimizer step: the forward layout
# [co][kh][kw][ci], the data-gradient layout [ci][kh'][kw'][co] (flipped), and for space-to-depth convs
# [co][s1][s2][c]. Keyed by the Parameter object; invalidated by `bump_weight_epoch()` (FlatAdamW.step, state-dict
# loads) and bypassed for anything that is not a leaf Parameter (second-order "weights" in gradient-penalty steps).
_weight_epoch = 0


_pack_tables = weakref.WeakSet()     # every FlatAdamW's PackTable (persistent operands, refreshed by one launch)


def pack_cache_clear():
    """drop the per-parameter cached packs (start/end of a hipGraph capture: a graph must contain its own packing
    launches and nothing outside may keep tensors of its private pool). The persistent pack tables are not affected."""
    global _weight_epoch
    _weight_epoch += 1


def bump_weight_epoch():
    """parameters were modified behind the optimizers' back (state-dict load, broadcast, manual edits): drop the cached
    packs and have every pack table re-packed on its next use."""
    pack_cache_clear()
    for tab in _pack_tables:
        tab.dirty = True


# bisect aids (module attributes a debugging session flips by hand; no environment switch since round 6): weights packed per call instead
# of through the models' pack tables / gradients through autograd's AccumulateGrad instead of the sinking finishes
_DEBUG_NO_TABLE = False
_DEBUG_NO_SINK = False


def _refresh_table(tab):
    """one gg_pack_weights launch re-packs every operand of the table from the live parameters; afterwards every registered
    parameter's recorded version is current (a state-dict load bumps them all: one re-pack, not one per parameter)."""
    tab.refresh()
    tab.dirty = False
    for ref in list(getattr(tab, 'owners', {}).values()):
        w = ref()
        slot = None if w is None else w.__dict__.get('_gg_tpacks')
        if slot:
            v = w._version
            for k, ent in slot.items():
                if ent[2] != v:
                    slot[k] = (ent[0], ent[1], v)


def _table_pack(w, kind: str):
    """parameters owned by a FlatAdamW: persistent operands, all re-packed by ONE gg_pack_weights launch right after
    each optimizer step (FlatAdamW.step) or, when something else touched the weights (`bump_weight_epoch`), on the next
    use. None -> caller falls back to the per-weight path."""
    tab = w._gg_pack_table
    slot = w.__dict__.get('_gg_tpacks')
    ent = slot.get(kind) if slot else None
    if ent is not None and ent[1] == w.data_ptr() and ent[2] != w._version:
        # the parameter was written in place behind the optimizer's back (nn.Module.load_state_dict, a manual edit, another
        # optimizer): its version counter moved, so the persistent operands are stale -> re-pack the table before use
        slot[kind] = ent = (ent[0], ent[1], w._version)
        tab.dirty = True
    if ent is None or ent[1] != w.data_ptr():
        if w.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return None                     # table appends are host->device copies: not while a graph is being captured
        shp = tuple(w.shape)
        if kind in ('modk', 'gram', 'frag'):        # (N, O, I, k, k) bank -> [co][tap][n][ci] (fused no-grad adaptive conv) / its Gram rows /
            src = w.detach()                         # its MFMA-fragment order (gg_aconv)
            if len(shp) != 5 or not src.is_contiguous():
                return None
            reg = {'modk': tab.register_bank, 'gram': tab.register_gram, 'frag': tab.register_frag}[kind]
            dst = reg(src, shp[0], shp[1], shp[2], shp[3] * shp[4])
            if slot is None:
                slot = w.__dict__.setdefault('_gg_tpacks', {})
            ent = slot[kind] = (dst, w.data_ptr(), w._version)
            _owners(tab)[id(w)] = weakref.ref(w)
            _refresh_table(tab)
            return ent[0]
        if len(shp) == 5:
            assert shp[0] == 1 or shp[1] % 8 == 0, 'stacked kernel banks need O % 8 == 0'
            shp = (shp[0] * shp[1],) + shp[2:]
        if kind == 's2d':                   # (O, 4C, 1, 1) over channels (c, s1, s2) == (O, C, 2*2) taps
            O, I, T, tk = shp[0], shp[1] // 4, 4, 'fwd'
        elif len(shp) == 2:                 # linear weight (O, I): a one-tap kernel
            O, I, T, tk = shp[0], shp[1], 1, kind
        else:
            O, I, T, tk = shp[0], shp[1], shp[2] * shp[3], kind
        src = w.detach()
        if not src.is_contiguous():
            return None
        dst = tab.register(src, O, I, T, tk)
        if slot is None:
            slot = w.__dict__.setdefault('_gg_tpacks', {})
        ent = slot[kind] = (dst, w.data_ptr(), w._version)
        _owners(tab)[id(w)] = weakref.ref(w)
        tab.dirty = True
    if tab.dirty:
        _refresh_table(tab)
    return ent[0]


def _owners(tab):
    o = getattr(tab, 'owners', None)
    if o is None:
        o = tab.owners = {}          # id -> weak reference (tensors compare element-wise: no sets of them)
    return o


def _pad_oi(w: torch.Tensor, o_to: int, i_to: int) -> torch.Tensor:
    o, i = w.shape[0], w.shape[1]
    if o_to != o or i_to != i:
        w = F.pad(w, (0, 0, 0, 0, 0, i_to - i, 0, o_to - o))
    return w


def packed_weight(w: torch.Tensor, kind: str) -> torch.Tensor:
    """bf16 GEMM operand for conv weights w (O, I, k, k):
         'fwd'  -> (O8, k*k*I8)  [co][kh][kw][ci]           (forward B operand / depth-to-space data gradient)
         'bwd'  -> (I8, k*k*O8)  [ci][kh'][kw'][co] flipped (stride-1 data gradient B operand)
         's2d'  -> w is the reference's (O, 4C, 1, 1) over channels (c, s1, s2): (O8, 4*C8) [co][s1][s2][c]
       channel counts are zero-padded to multiples of 8 (the kernels' 16-byte vectors)."""
    cacheable = isinstance(w, torch.nn.Parameter)
    if cacheable and getattr(w, '_gg_pack_table', None) is not None and not _DEBUG_NO_TABLE:
        out = _table_pack(w, kind)
        if out is not None:
            return out
    if cacheable:       # the cache lives on the Parameter object itself: (epoch, {kind: packed}, tensor version)
        slot = getattr(w, '_gg_packed', None)
        if slot is not None and slot[0] == _weight_epoch and slot[2] == w._version and kind in slot[1]:
            return slot[1][kind]
    with torch.no_grad():
        wd = w.detach()
        if wd.dim() == 5:       # AdaptiveConv2DMod bank (N, O, I, k, k): the N kernels stacked along output channels
            n, o = wd.shape[0], wd.shape[1]
            assert n == 1 or o % 8 == 0, 'stacked kernel banks need O % 8 == 0'
            wd = wd.reshape(n * o, *wd.shape[2:])
        if kind == 's2d':
            o, c4 = wd.shape[0], wd.shape[1]
            c = c4 // 4
            wr = wd.reshape(o, c, 2, 2)
            wr = _pad_oi(wr, _round8(o), _round8(c))
            out = wr.permute(0, 2, 3, 1).reshape(wr.shape[0], -1).to(ACT_DTYPE).contiguous()
        else:
            wp = _pad_oi(wd, _round8(wd.shape[0]), _round8(wd.shape[1]))
            if kind == 'fwd':
                out = wp.permute(0, 2, 3, 1).reshape(wp.shape[0], -1).to(ACT_DTYPE).contiguous()
            elif kind == 'bwd':
                out = wp.flip(2, 3).permute(1, 2, 3, 0).reshape(wp.shape[1], -1).to(ACT_DTYPE).contiguous()
            else:
                raise ValueError(kind)
    if cacheable:
        slot = getattr(w, '_gg_packed', None)
        if slot is None or slot[0] != _weight_epoch or slot[2] != w._version:
            slot = (_weight_epoch, {}, w._version)
            w._gg_packed = slot
        slot[1][kind] = out
    return out


# --------------------------------------------------------------------------------------------------
# autograd Functions over the HIP kernels (each one's backward is built from the others)
# --------------------------------------------------------------------------------------------------
# Geometry of a conv: (ksize, stride, pad, wkind). wkind 'oihw': w is (O, I, k, k); 's2d': w is the reference's
# (O, 4C, 1, 1) 1x1 conv over space-to-depth channels, executed as a 2x2 / stride-2 window over the un-rearranged
# input. Activations are (b, H, W, C8) bf16 with C8 = channels rounded up to 8 (zero padded).

def _geom_k(geom):
    return geom[0]


# Inside `with ops.sinking():` (the trainer's .backward() calls), a conv's weight gradient is accumulated by the finish kernel
# straight into the parameter's fp32 .grad (the flat gradient buffer) and autograd is handed None: no separate
# transpose / scale / AccumulateGrad passes. Only when no graph of the backward is being recorded. The finishes are QUEUED and run
# when the context exits, so the flag is private to that context: setting a module attribute by hand (the pre-round-4 pattern) would
# sink gradients into a queue nobody flushes.
_grad_sink = False


def _grad_sink_of(w):
    if not _grad_sink or _DEBUG_NO_SINK or torch.is_grad_enabled() or not isinstance(w, torch.nn.Parameter):
        return None
    g = w.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != w.shape:
        return None
    return g


@contextmanager
def sinking(on: bool = True):
    """`with ops.sinking(): loss.backward()` - the trainer's backward passes: weight / bias gradients go straight into the flat
    gradient buffer (`grad_sink`), their finish passes are queued and the queue is flushed when the pass is over (dropped if it
    raised)."""
    global _grad_sink
    prev, _grad_sink = _grad_sink, bool(on)
    try:
        yield
        K.finish_queue.flush()
    finally:
        _grad_sink = prev
        K.finish_queue.clear()


def flush_finishes():
    """run the queued weight- / bias-gradient finishes (end of a backward pass; before anything reads the flat gradient buffer)."""
    K.finish_queue.flush()


def grad_ready(w):
    """a finish kernel has just been enqueued that writes w's gradient into the flat buffer behind autograd's back (no
    AccumulateGrad node runs for it): tell the model's in-backward gradient exchange (distributed.GradReducer), if any."""
    r = w.__dict__.get('_gg_reducer')
    if r is not None:
        r.fired(w)


def _check_act_residual(act, residual, *tensors):
    if act and residual is not None and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError('conv2d: an activation and a residual cannot share one epilogue under autograd (the backward recovers '
                           'the leaky-relu mask from the sign of the stored output); apply the residual in a separate op')


def _bias8(bias, o8):
    """a bias of O % 8 != 0 channels (rgb / logit convolutions) as the o8 floats the kernels read: a FlatAdamW-owned parameter sits in a
    zero-padded 256-float slot of the flat buffer (optimizer.FlatAdamW._build: the tail is never written - zero gradient, zero moments,
    zero decay), so the longer view IS the zero-padded bias; anything else is padded by a copy."""
    slot = bias.__dict__.get('_gg_slot') if isinstance(bias, torch.nn.Parameter) else None
    if (slot is not None and slot[0] == bias.data_ptr() and slot[1] >= o8 and bias.dim() == 1 and bias.dtype == torch.float32
            and bias.untyped_storage().nbytes() >= (bias.storage_offset() + o8) * 4):
        return torch.as_strided(bias.detach(), (o8,), (1,), bias.storage_offset())
    return F.pad(bias, (0, o8 - bias.shape[0]))


class ConvFn(Function):
    """y = act(alpha * (conv(x * in_scale, w) + bias)) + res_scale * residual ;  x: (b,H,W,C8) bf16, w: float parameter layout."""

    @staticmethod
    def forward(ctx, x, w, bias, in_scale, act, geom, alpha, residual, fork=False, res_scale=1.0):
        """`fork=True` returns (y, x): x's OTHER consumer takes the returned alias; its gradient then arrives here and is
        added in the data-gradient GEMM's epilogue (no separate accumulation pass over the activation gradient)."""
        ksize, stride, pad, wkind = geom
        # (activation + residual in one epilogue is forward-only - the backward recovers the leaky-relu mask from the sign of the
        # stored output: checked in `_check_act_residual` by the callers, where the grad mode is still the caller's)
        wmat = packed_weight(w, 's2d' if wkind == 's2d' else 'fwd')
        o8 = wmat.shape[0]
        b8 = bias
        if bias is not None and bias.shape[0] != o8:
            b8 = _bias8(bias, o8)
        y = K.conv2d_nhwc(x, wmat, ksize=ksize, stride=stride, pad=pad, in_scale=in_scale, bias=b8, bias_scale=alpha,
                          alpha=alpha, act=act, act_slope=LRELU_SLOPE, residual=residual, res_scale=res_scale)
        ctx.act, ctx.geom, ctx.alpha, ctx.res_scale = act, geom, alpha, res_scale
        ctx.save_for_backward(x, w, in_scale, y if act else None, bias)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.n_bias = bias.shape[0] if bias is not None else 0
        if fork:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, g_alias=None):
        x, w, in_scale, y, bias = ctx.saved_tensors
        geom, alpha = ctx.geom, ctx.alpha
        if dy is None:          # only the alias was used downstream
            return g_alias, None, None, None, None, None, None, None, None, None
        dy = dy.contiguous()
        want_db = ctx.has_bias and ctx.needs_input_grad[2] and not inputs_only
        db = None
        bsink = _grad_sink_of(bias) if want_db else None
        if bsink is not None:       # bias gradient: partial column sums -> a queued finish into the flat .grad (K.FinishQueue)
            dz, part = K.bias_act_bwd(dy, y if ctx.act == 'lrelu' else None, True, LRELU_SLOPE, partials=True)
            K.finish_queue.add_colsum(part, ctx.n_bias, alpha, bsink, notify=lambda b=bias: grad_ready(b))
        elif ctx.act == 'lrelu' or want_db:
            dz, db = BiasActBwdFn.apply(dy, y if ctx.act == 'lrelu' else None, want_db)
            if want_db:
                db = db[:ctx.n_bias]
                if alpha != 1.0:
                    db = db * alpha
            else:
                db = None
        else:
            dz = dy
        dx = dw = ds = None
        if ctx.needs_input_grad[0] or (in_scale is not None and ctx.needs_input_grad[3]):
            carry = g_alias if (g_alias is not None and in_scale is None) else None
            dxs = DgradFn.apply(dz, w, geom, alpha, x.shape[1], x.shape[2], carry)
            if in_scale is None:
                dx = dxs
            else:
                if ctx.needs_input_grad[3]:
                    ds = (x.float() * dxs.float()).sum(dim=(1, 2))
                dx = (dxs.float() * in_scale[:, None, None, :]).to(dxs.dtype)
                if g_alias is not None:
                    dx = dx + g_alias
        elif g_alias is not None:
            dx = g_alias
        if ctx.needs_input_grad[1] and not inputs_only:
            sink = _grad_sink_of(w)
            if sink is not None:
                WgradFn.compute(x, dz, in_scale, geom, alpha, tuple(w.shape), sink, notify=lambda w=w: grad_ready(w))
            else:
                dw = WgradFn.apply(x, dz, in_scale, geom, alpha, tuple(w.shape)).to(w.dtype)
        dres = None
        if ctx.has_res and ctx.needs_input_grad[7]:
            dres = dy if ctx.res_scale == 1.0 else dy * ctx.res_scale
        return dx, dw, db, ds, None, None, None, dres, None, None


_ff_plan_cache: dict = {}
_ones_cache: dict = {}


def _ones_col(b, device):
    """a shared read-only (b, 1) fp32 column of ones (the kernel-selection weights of a one-kernel bank: one fill launch per call before)"""
    key = (b, device)
    t = _ones_cache.get(key)
    if t is None:
        t = torch.ones((b, 1), device=device, dtype=torch.float32)
        if not (device.type == 'cuda' and torch.cuda.is_current_stream_capturing()):      # (a graph's private pool is not ours to keep)
            _ones_cache[key] = t
    return t


_NO_FF_FUSE = False      # (round 6: the GG_NO_FF_FUSE A/B switch is gone - GELU on the 1x1 pair's epilogues, profiles/r04_ff_fuse_ab.log)


class HingeFn(Function):
    """the trainer's hinge losses over one logit tensor (gp.py:157-163) in one launch, their backward in one more (kernels.hinge).
    mode 1: x is (outer, 2b, ...) with the fake half first (the merged discriminator pass); mode 0: mean(x). Piecewise linear: the
    gradient penalty differentiates the logits themselves, never this loss, so first order is all there is."""

    @staticmethod
    def forward(ctx, x, nb, split, mode):
        ctx.cfg = (nb, split, mode)
        ctx.save_for_backward(x)
        return K.hinge(x, nb, split, mode)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, = ctx.saved_tensors
        nb, split, mode = ctx.cfg
        return K.hinge(x, nb, split, mode, gscale=g.float().reshape(1).contiguous()), None, None, None


class FFTailFn(Function):
    """conv1x1(gelu(conv1x1(n) + b_in)) + b_out + residual - everything of the channel-first FeedForward behind its norm (gp.py:726-740)
    as ONE autograd node, with the GELU riding on GEMM epilogues: forward, the up-projection stores its pre-activation h AND gelu(h)
    (gelu_mode 1: no separate GELU pass over the 4x-wide hidden); backward, the down-projection's data gradient comes out already
    multiplied by gelu'(h) (gelu_mode 2: no dg tensor, no GELU-backward pass). Parameter gradients go through the grad sink / the
    queued finishes like ConvFn's. First order only: gradient-penalty graphs keep the separate, twice-differentiable Functions."""

    @staticmethod
    def forward(ctx, n, w_in, b_in, w_out, b_out, residual):
        wi, wo = packed_weight(w_in, 'fwd'), packed_weight(w_out, 'fwd')
        h = torch.empty(n.shape[:3] + (wi.shape[0],), dtype=n.dtype, device=n.device)
        g = K.conv2d_nhwc(n, wi, ksize=1, bias=b_in, gelu_aux=h, gelu_mode=1)
        y = K.conv2d_nhwc(g, wo, ksize=1, bias=b_out, residual=residual)
        ctx.save_for_backward(n, h, g, w_in, b_in, w_out, b_out)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        n, h, g, w_in, b_in, w_out, b_out = ctx.saved_tensors
        dy = dy.contiguous()
        geom = (1, 1, 0, 'oihw')

        def bias_grad(dz, bias, idx):
            if bias is None or not ctx.needs_input_grad[idx] or inputs_only:
                return None
            _, part = K.bias_act_bwd(dz, None, True, LRELU_SLOPE, partials=True)
            sink = _grad_sink_of(bias)
            if sink is not None:
                K.finish_queue.add_colsum(part, bias.shape[0], 1.0, sink, notify=lambda b=bias: grad_ready(b))
                return None
            return K.colsum_finish(part, bias.shape[0])

        def weight_grad(x, dz, w, idx):
            if not ctx.needs_input_grad[idx] or inputs_only:
                return None
            sink = _grad_sink_of(w)
            if sink is not None:
                WgradFn.compute(x, dz, None, geom, 1.0, tuple(w.shape), sink, notify=lambda w=w: grad_ready(w))
                return None
            return WgradFn.compute(x, dz, None, geom, 1.0, tuple(w.shape), None).to(w.dtype)

        db_out = bias_grad(dy, b_out, 4)
        dw_out = weight_grad(g, dy, w_out, 3)
        dh = K.conv2d_nhwc(dy, packed_weight(w_out, 'bwd'), ksize=1, gelu_aux=h, gelu_mode=2)      # dgrad * gelu'(h) in one launch
        db_in = bias_grad(dh, b_in, 2)
        dw_in = weight_grad(n, dh, w_in, 1)
        dn = K.conv2d_nhwc(dh, packed_weight(w_in, 'bwd'), ksize=1) if ctx.needs_input_grad[0] else None
        return dn, dw_in, db_in, dw_out, db_out, (dy if ctx.has_res and ctx.needs_input_grad[5] else None)


class DgradFn(Function):
    """data gradient of ConvFn: dx = alpha * conv^T(dz, w). Stride-1 'same' convs run the forward kernel on the
    flipped/transposed weights; the non-overlapping stride-2 windows (1x1 stride 2, space-to-depth) run a dense GEMM
    with the depth-to-space scatter store."""

    @staticmethod
    def forward(ctx, dz, w, geom, alpha, H, W, carry=None):
        """`carry` (dx's shape): a gradient that reached the conv's input over another branch, added in the epilogue."""
        ksize, stride, pad, wkind = geom
        if stride == 1:
            wmat = packed_weight(w, 'bwd')
            dx = K.conv2d_nhwc(dz, wmat, ksize=ksize, stride=1, pad=ksize - 1 - pad, alpha=alpha,
                               residual=None if carry is None else carry.contiguous())
        else:
            assert pad == 0 and ksize <= stride
            wmat = packed_weight(w, 's2d' if wkind == 's2d' else 'fwd')
            dx = K.conv2d_dgrad_d2s(dz, wmat, cell=stride, taps=ksize, alpha=alpha)
            assert dx.shape[1] == H and dx.shape[2] == W
            if carry is not None:       # the scatter store has no residual operand
                dx = dx + carry
        ctx.geom, ctx.alpha = geom, alpha
        ctx.save_for_backward(dz, w)
        return dx

    @staticmethod
    def backward(ctx, g):
        dz, w = ctx.saved_tensors
        geom, alpha = ctx.geom, ctx.alpha
        g = g.contiguous()
        ddz = dw = None
        if ctx.needs_input_grad[0]:     # linear in dz: the adjoint of the adjoint is the forward conv
            ddz = ConvFn.apply(g, w, None, None, None, geom, alpha, None)
        if ctx.needs_input_grad[1] and not inputs_only:     # dL/dw[co][tap][ci] = alpha * sum_p dz[p][co] * g[p + tap][ci]
            sink = _grad_sink_of(w)
            if sink is not None:
                WgradFn.compute(g, dz, None, geom, alpha, tuple(w.shape), sink, notify=lambda w=w: grad_ready(w))
            else:
                dw = WgradFn.apply(g, dz, None, geom, alpha, tuple(w.shape)).to(w.dtype)
        return ddz, dw, None, None, None, None, (g if ctx.needs_input_grad[6] else None)


class BiasActBwdFn(Function):
    """(dz, db) = (dy * lrelu'(y), column sums of dz) in one HIP pass. Linear in dy with a piecewise-constant mask,
    so its own backward (gradient penalty) is the same kernel applied to the incoming gradient."""

    @staticmethod
    def forward(ctx, dy, y, want_db):
        ctx.set_materialize_grads(False)
        dz, db = K.bias_act_bwd(dy, y, want_db, LRELU_SLOPE)
        ctx.save_for_backward(y)
        ctx.like = (dy.shape, dy.dtype)
        if db is None:
            db = dy.new_empty(0, dtype=torch.float32)      # placeholder (no launch); never differentiated
            ctx.mark_non_differentiable(db)
        return dz, db

    @staticmethod
    def backward(ctx, g_dz, g_db):
        y, = ctx.saved_tensors
        g = g_dz
        if g_db is not None and g_db.numel() > 0:
            shape, dtype = ctx.like
            gb = g_db.to(dtype).view(1, 1, 1, -1)
            g = gb + g if g is not None else gb.expand(shape)
        if g is None:
            return None, None, None
        if y is not None:
            g, _ = BiasActBwdFn.apply(g.contiguous(), y, False)
        return g, None, None


class WgradFn(Function):
    """dw[o][i][kh][kw] = alpha * sum_pixels dy[p][o] * (x*in_scale)[window(p) + (kh,kw)][i]  (fp32, parameter layout)."""

    @staticmethod
    def forward(ctx, x, dy, in_scale, geom, alpha, wshape):
        ksize, stride, pad, wkind = geom
        ctx.geom, ctx.alpha, ctx.wshape = geom, alpha, wshape
        ctx.save_for_backward(x, dy, in_scale)
        return WgradFn.compute(x, dy, in_scale, geom, alpha, wshape, None)

    @staticmethod
    def compute(x, dy, in_scale, geom, alpha, wshape, sink, notify=None):
        """the GEMM ([tap][ci][co] fp32, pixels reduced) + the transpose/scale pass into the parameter layout; with
        `sink` (a parameter's fp32 .grad) the result is accumulated there instead of being returned - by a QUEUED finish
        (K.finish_queue: executed in batches, at the latest by `flush_finishes()` at the end of the backward pass)."""
        ksize, stride, pad, wkind = geom
        # (k*k*C8, O8) fp32; with a sink and few split-K slices: the slice stack, summed by the queued finish (no reduce launch)
        g, nsplit = K.conv2d_wgrad_nhwc(x, dy, ksize=ksize, stride=stride, pad=pad, in_scale=in_scale, keep_slices=True) \
            if sink is not None else (K.conv2d_wgrad_nhwc(x, dy, ksize=ksize, stride=stride, pad=pad, in_scale=in_scale), 1)
        if wkind == 's2d':              # (O, C, s1, s2) == (O, 4C, 1, 1)
            O, I = wshape[0], wshape[1] // 4
        elif len(wshape) == 5:          # kernel bank (N, O, I, k, k) stacked along output channels
            O, I = wshape[0] * wshape[1], wshape[2]
        else:
            O, I = wshape[0], wshape[1]
        if sink is not None:
            K.finish_queue.add_wgrad(g, O, I, ksize * ksize, alpha, sink.view(-1), notify=notify, nsplit=nsplit)
            return None
        return K.wgrad_finish(g, O, I, ksize * ksize, alpha).view(wshape)

    @staticmethod
    def backward(ctx, ddw):
        x, dy, in_scale = ctx.saved_tensors
        geom, alpha = ctx.geom, ctx.alpha
        dx = ddy = None
        if ctx.needs_input_grad[0]:
            dx = DgradFn.apply(dy, ddw, geom, alpha, x.shape[1], x.shape[2])
            if in_scale is not None:
                dx = (dx.float() * in_scale[:, None, None, :]).to(dx.dtype)
        if ctx.needs_input_grad[1]:
            ddy = ConvFn.apply(x, ddw, None, in_scale, None, geom, alpha, None)
        return dx, ddy, None, None, None, None


class ModulateFn(Function):
    """xs = x * s[b, :] (the weight modulation `weights * (mod + 1)` of gp.py:394-396 moved onto the activation);
    backward in one pass: dx = g * s and ds = sum over pixels of g * x. First order only (generator path)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return K.modulate(x, s)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        dx, ds = K.modulate_bwd(g.contiguous(), x, s)
        return (dx if ctx.needs_input_grad[0] else None), (ds if ctx.needs_input_grad[1] else None)


class ModCoefFn(Function):
    """(s, a, d) of the adaptive convolution (gp.py:378-400) in ONE launch; backward in two (gg_modcoef.h). The
    weights' gradient through the demodulation is accumulated into the flat gradient buffer when `grad_sink` is on."""

    @staticmethod
    def forward(ctx, mod, kmod, weights, eps, Ip, Op):
        ctx.set_materialize_grads(False)
        ctx.eps = eps
        ctx.gram = mod.shape[0] <= K.MODGRAM_MAX_B
        if ctx.gram:       # through the bank's Gram rows: 17x fewer operations (gg_modcoef.h, second half)
            # the rows only change when the weights do: a FlatAdamW-owned bank keeps them in its pack table (kind 'gram', refreshed by
            # the one gg_pack_weights launch after the optimizer step - the same sums in the same order as gg_modgram)
            gram = None
            if (isinstance(weights, torch.nn.Parameter) and getattr(weights, '_gg_pack_table', None) is not None and not _DEBUG_NO_TABLE
                    and weights.dtype == torch.float32 and weights.is_contiguous()):
                gram = _table_pack(weights, 'gram')
            if gram is None:
                gram = K.modgram(weights.detach())
            s, a, d, tsum = K.modcoef_gram_fwd(gram, weights.shape[0], mod, kmod, eps, Ip, Op)
            ctx.save_for_backward(kmod, weights, s, d, gram, tsum)
        else:
            s, a, d = K.modcoef_fwd(weights.detach(), mod, kmod, True, eps, Ip, Op)
            ctx.save_for_backward(kmod, weights, s, d)
        return s, a, d

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gs, ga, gd):
        kmod, weights, s, d = ctx.saved_tensors[:4]

        def f32(t):
            return None if t is None else t.float().contiguous()
        gs, ga, gd = f32(gs), f32(ga), f32(gd)
        gw = sink = None
        if ctx.needs_input_grad[2] and not inputs_only and gd is not None:
            sink = _grad_sink_of(weights)
            gw = sink if sink is not None else torch.zeros_like(weights, dtype=torch.float32)
        if gd is None:      # d was not used downstream: only the linear parts remain
            gmod = None if gs is None else gs[:, :weights.shape[2]]
            gk = None
            if ga is not None and kmod is not None:
                a = kmod.softmax(dim=-1)
                gk = a * (ga - (a * ga).sum(-1, keepdim=True))
            return gmod, gk, None, None, None, None
        if ctx.gram:
            gram, tsum = ctx.saved_tensors[4:]
            gmod, gk = K.modcoef_gram_bwd(weights.detach(), gram, kmod, s, d, tsum, gs, ga, gd, gw, ctx.eps)
        else:
            gmod, gk = K.modcoef_bwd(weights.detach(), kmod, s, d, gs, ga, gd, gw, ctx.eps)
        if sink is not None:
            grad_ready(weights)
        return gmod, gk, (gw if (gw is not None and sink is None) else None), None, None, None


class ModMixFn(Function):
    """y = act(d[b,o] * sum_n a[b,n] * Y[..., n*Os + o] + noise_w[o] * noise[b,p]) — the per-sample kernel mix,
    demodulation, noise and leaky-relu after the stacked conv — one bf16 pass forward, one backward."""

    @staticmethod
    def forward(ctx, Y, a, d, noise, noise_w, O, N, act):
        y = K.modmix_fwd(Y, a, d, noise, noise_w, O, N, act)
        ctx.cfg = (O, N, act)
        ctx.save_for_backward(Y, a, d, noise, y if act else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        Y, a, d, noise, y = ctx.saved_tensors
        O, N, act = ctx.cfg
        dY, da, dd, dnw = K.modmix_bwd(dy.contiguous(), y, Y, a, d, noise, O, N, act)
        return dY, (da if N > 1 else None), dd, None, dnw, None, None, None


class RmsNormFn(Function):
    """ChannelRMSNorm over the last (channel) axis of an NHWC bf16 tensor, one fused pass (gg_rmsnorm_kernel).
    `fork=True` returns (y, x): x's second consumer (the skip connection around the normalised branch) takes the returned
    alias, and its gradient is added inside this op's backward pass instead of by autograd's accumulation.
    `silu=True`: y = silu(norm(x)) in the same pass (the unet Block, unet.py:268-269); first-order backward only."""

    @staticmethod
    def forward(ctx, x, gamma, fork=False, silu=False, gamma_param=None):
        """gamma: (C,) fp32; gamma_param: the nn.Parameter it is a view of, if any - under `ops.sinking()` its gradient then goes
        through the finish queue straight into the flat gradient buffer (no partial-sum fold, no AccumulateGrad launch)."""
        ctx.save_for_backward(x, gamma)
        ctx.set_materialize_grads(False)
        ctx.silu = silu
        ctx.gamma_param = gamma_param
        y = K.rmsnorm_fwd(x, gamma, silu)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, g, g_alias=None):
        x, gamma = ctx.saved_tensors
        want_dgamma = ctx.needs_input_grad[1] and not inputs_only
        if g is None:
            return g_alias, None, None, None, None
        carry = None if g_alias is None else g_alias.contiguous()
        if ctx.silu or not torch.is_grad_enabled():
            assert not (ctx.silu and torch.is_grad_enabled()), 'RmsNormFn(silu=True) is first-order only'
            gp = ctx.gamma_param
            sink = _grad_sink_of(gp) if (want_dgamma and gp is not None) else None
            dx, dgamma = K.rmsnorm_bwd(x, g.contiguous(), gamma, want_dgamma, carry, silu=ctx.silu, partials=sink is not None)
            if sink is not None:
                K.finish_queue.add_colsum(dgamma, gamma.numel(), 1.0, sink, notify=lambda q=gp: grad_ready(q))
                dgamma = None
        else:
            dx, dgamma = RmsNormBwdFn.apply(x, g.contiguous(), gamma, want_dgamma, carry)
        return dx, (dgamma if want_dgamma else None), None, None, None


class RmsNormBwdFn(Function):
    """(dx, dgamma) of RmsNormFn in one pass; differentiable once more (gradient penalty) through gg_rmsnorm bwd2.
    The second-order pass ignores gradients flowing into `dgamma` (nothing in the GigaGAN losses produces them)."""

    @staticmethod
    def forward(ctx, x, g, gamma, want_dgamma, carry=None):
        ctx.set_materialize_grads(False)
        dx, dgamma = K.rmsnorm_bwd(x, g, gamma, want_dgamma, carry)
        ctx.save_for_backward(x, g, gamma)
        if dgamma is None:
            dgamma = x.new_empty(0, dtype=torch.float32)     # placeholder (no launch)
            ctx.mark_non_differentiable(dgamma)
        return dx, dgamma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v, v_dgamma):
        x, g, gamma = ctx.saved_tensors
        if v is None:
            return None, None, None, None, None
        want = ctx.needs_input_grad[2] and not inputs_only
        gx, gg, dgamma = K.rmsnorm_bwd2(x, g, v.contiguous(), gamma, want)
        return gx, gg, dgamma, None, (v if ctx.needs_input_grad[4] else None)     # dx = ... + carry: d/d(carry) is v itself


class FlashAttnFn(Function):
    """fused self-attention over [B][n][heads*64] projections with the learned null key / value (gg_attention.h):
    forward saves only o and the per-query log-sum-exp; backward = two kernels (dq; dk/dv) that recompute P. When the
    backward itself is recorded (gradient penalty, `create_graph=True`) it runs as FlashAttnBwdFn, whose own backward is
    the fused second-order pass (gg_attention2.h)."""

    @staticmethod
    def forward(ctx, q, k, v, k0, v0, heads, alpha, beta):
        """k None: the keys ARE the queries (tied projections of the L2 attention, gp.py:566-569); the backward then stores
        dq + dk in one buffer instead of handing autograd two tensors to add."""
        if (k0.is_contiguous() and v0.is_contiguous() and k0.dtype == v0.dtype and k0.shape == v0.shape
                and v0.data_ptr() == k0.data_ptr() + k0.numel() * k0.element_size()
                and k0.untyped_storage().data_ptr() == v0.untyped_storage().data_ptr()):
            # the two halves of one (2, heads, 64) null_kv parameter (gp.py:541): one cast launch for both
            kvb = torch.as_strided(k0, (2, *k0.shape), (k0.numel(), *k0.stride())).to(ACT_DTYPE)
            k0b, v0b = kvb[0], kvb[1]
        else:
            k0b, v0b = k0.to(ACT_DTYPE).contiguous(), v0.to(ACT_DTYPE).contiguous()
        ctx.tied = k is None
        if k is None:
            k = q
        o, lse = K.attn_fwd(q, k, v, k0b, v0b, heads, alpha, beta)
        ctx.cfg = (heads, alpha, beta)
        ctx.save_for_backward(q, k, v, k0, v0, o, lse, k0b, v0b)      # (the bf16 null key / value: two launches fewer per backward)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, k0, v0, o, lse, k0b, v0b = ctx.saved_tensors
        heads, alpha, beta = ctx.cfg
        if torch.is_grad_enabled():       # this backward is being differentiated: keep it on the autograd tape
            dq, dk, dv, dk0, dv0 = FlashAttnBwdFn.apply(q, k, v, k0, v0, o, lse, d_o.contiguous(), heads, alpha, beta)
            if ctx.tied:
                dq, dk = dq + dk, None
        else:
            dq, dk, dv, dk0q, dv0, dbias0 = K.attn_bwd(q, k, v, k0b, v0b, o, lse, d_o.contiguous(), heads, alpha, beta,
                                                       tied=ctx.tied)
            dk0 = torch.addcmul(dk0q, dbias0[:, None], k0b.float(), value=2.0 * beta)
            if ctx.tied:
                dk = None
        return dq, dk, dv, dk0.to(k0.dtype), dv0.to(v0.dtype), None, None, None


class FlashAttnBwdFn(Function):
    """(dq, dk, dv, dk0, dv0) of FlashAttnFn as a differentiable function of (q, k, v, k0, v0, dO). `o` and `lse` are the
    forward's saved outputs; the second-order pass returns TOTAL derivatives w.r.t. q, k, v, k0, v0 (it differentiates
    through P and D = dO.O itself), so no gradient is routed back through them."""

    @staticmethod
    def forward(ctx, q, k, v, k0, v0, o, lse, d_o, heads, alpha, beta):
        k0b, v0b = k0.to(ACT_DTYPE).contiguous(), v0.to(ACT_DTYPE).contiguous()
        dq, dk, dv, dk0q, dv0, dbias0, dvec = K.attn_bwd(q, k, v, k0b, v0b, o, lse, d_o, heads, alpha, beta, return_dvec=True)
        dk0 = dk0q + (2.0 * beta) * dbias0[:, None] * k0b.float()
        ctx.cfg = (heads, alpha, beta)
        ctx.save_for_backward(q, k, v, k0b, v0b, lse, dvec, d_o)
        return dq, dk, dv, dk0, dv0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, a_q, a_k, a_v, a_k0, a_v0):
        q, k, v, k0b, v0b, lse, dvec, d_o = ctx.saved_tensors
        heads, alpha, beta = ctx.cfg

        def like(t, ref):
            return torch.zeros_like(ref) if t is None else t.to(ACT_DTYPE).contiguous()

        gq, gk, gv, gdo, gk0, gv0 = K.attn_bwd2(q, k, v, k0b, v0b, d_o, lse, dvec, like(a_q, q), like(a_k, q), like(a_v, q),
                                                like(a_k0, k0b), like(a_v0, v0b), heads, alpha, beta)
        return gq, gk, gv, gk0, gv0, None, None, gdo, None, None, None


class FlashAttnGenFn(Function):
    """the general fused attention (gg_attention.h GEN: the unet's Attend attend.py:64-110, CrossAttention gp.py:617-655, the text
    transformer's attention gp.py:659-722): q (B, n, heads*64), k / v (B, m, heads*64) bf16 views with a free row pitch, optional
    null key / value (heads, 64), optional per-key bias (B, m) fp32 (key-padding mask as -1e30; a constant: no gradient). No
    (B*heads, n, m) probability tensor exists in either direction. First order."""

    @staticmethod
    def forward(ctx, q, k, v, k0, v0, kbias, heads, alpha):
        k0b = None if k0 is None else k0.to(ACT_DTYPE).contiguous()
        v0b = None if v0 is None else v0.to(ACT_DTYPE).contiguous()
        o, lse = K.attn_gen_fwd(q, k, v, k0b, v0b, kbias, heads, alpha)
        ctx.cfg = (heads, alpha)
        ctx.save_for_backward(q, k, v, k0, v0, kbias, o, lse)
        return o

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_o):
        q, k, v, k0, v0, kbias, o, lse = ctx.saved_tensors
        heads, alpha = ctx.cfg
        k0b = None if k0 is None else k0.to(ACT_DTYPE).contiguous()
        v0b = None if v0 is None else v0.to(ACT_DTYPE).contiguous()
        dq, dk, dv, dk0, dv0, _ = K.attn_gen_bwd(q, k, v, k0b, v0b, kbias, o, lse, d_o.to(ACT_DTYPE).contiguous(), heads, alpha)
        if k0 is not None:
            dk0, dv0 = dk0.to(k0.dtype), dv0.to(v0.dtype)
        return dq, dk, dv, dk0, dv0, None, None, None


def _rows_view(t):
    """(B, h, len, 64) logical attention operand -> (B, len, h*64) bf16 view with unit feature stride and a batch stride of
    len * row pitch, without a copy when the operand already is a channel slice of a token-major projection; else a dense copy."""
    B, h, n, d = t.shape
    if (t.dtype == ACT_DTYPE and t.stride(3) == 1 and t.stride(1) == d and t.stride(2) % 8 == 0 and t.stride(2) >= h * d
            and t.stride(0) == n * t.stride(2) and t.data_ptr() % 16 == 0):
        return t.as_strided((B, n, h * d), (t.stride(0), t.stride(2), 1), t.storage_offset())
    return t.to(ACT_DTYPE).transpose(1, 2).reshape(B, n, h * d).contiguous()


class GemmFn(Function):
    """out[b][r][c] = act(alpha * sum_t X[b](r,t) * Y[b](c,t) + bias[c]).

    `x_red_last`: X stored (..., R, T) (reduction contiguous) else (..., T, R); same for Y.
    bf16 operands, bf16 or fp32 result.  Logical extents may be smaller than storage (pitches are
    multiples of 8); `dims = (R, C, T)`.
    """

    @staticmethod
    def forward(ctx, x, y, x_red_last, y_red_last, dims, bias, act, alpha, out_f32):
        R, Cc, T = dims
        out = K.gemm(x, y, trans_a=not x_red_last, trans_b=y_red_last, m_valid=R, n_valid=Cc, k_valid=T,
                     bias=bias, act=act, act_slope=LRELU_SLOPE, alpha=alpha,
                     out_dtype=torch.float32 if out_f32 else ACT_DTYPE)
        if x.dim() == 2 and y.dim() == 2:
            out = out[0]
        ctx.cfg = (x_red_last, y_red_last, dims, act, alpha, out_f32)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, y, out if act else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, out = ctx.saved_tensors
        x_red_last, y_red_last, (R, Cc, T), act, alpha, out_f32 = ctx.cfg
        if act == 'lrelu':
            dout = dout * torch.where(out > 0, 1.0, LRELU_SLOPE).to(dout.dtype)
        elif act is not None:
            raise RuntimeError('GemmFn: only the leaky-relu epilogue is differentiable; apply other activations outside')
        db = None
        if ctx.has_bias and ctx.needs_input_grad[5]:
            db = dout.float().reshape(-1, dout.shape[-1]).sum(0)
        g = dout.to(ACT_DTYPE)
        if g.stride(-1) != 1 or g.stride(-2) % 8:
            g = _pad_last(g.contiguous())
        dx = dy = None
        if ctx.needs_input_grad[0]:
            if x_red_last:   # dX(r,t) = sum_c g(r,c) Y(c,t)
                dx = GemmFn.apply(g, y, True, not y_red_last, (R, T, Cc), None, None, alpha, False)
            else:            # stored (T,R): dX(t,r) = sum_c Y(c,t) g(r,c)
                dx = GemmFn.apply(y, g, not y_red_last, True, (T, R, Cc), None, None, alpha, False)
            dx = _fit(dx, x)
        if ctx.needs_input_grad[1]:
            if y_red_last:   # dY(c,t) = sum_r g(r,c) X(r,t)
                dy = GemmFn.apply(g, x, False, not x_red_last, (Cc, T, R), None, None, alpha, False)
            else:            # stored (T,C): dY(t,c) = sum_r X(r,t) g(r,c)
                dy = GemmFn.apply(x, g, not x_red_last, False, (T, Cc, R), None, None, alpha, False)
            dy = _fit(dy, y)
        return dx, dy, None, None, None, db, None, None, None


class AttnProbsFn(Function):
    """attn = softmax_j(alpha * q k^T + bias) as bf16 (BH, n, mp): MFMA GEMM with fp32 logits + ONE fused
    softmax pass; the fp32 logits never leave this Function. Backward = fused softmax-backward pass + two GEMMs,
    all of them differentiable again (SoftmaxBwdFn / GemmFn), so the gradient penalty can pass through."""

    @staticmethod
    def forward(ctx, q, k, bias, alpha, m_valid):
        n, mp, dh = q.shape[1], k.shape[1], q.shape[2]
        qk = K.gemm(q, k, trans_a=False, trans_b=True, out_dtype=torch.float32)
        attn = K.softmax_fwd(qk, bias, alpha, m_valid)
        ctx.cfg = (alpha, m_valid, (n, mp, dh))
        ctx.has_bias = bias is not None
        ctx.save_for_backward(q, k, attn)
        return attn

    @staticmethod
    def backward(ctx, dattn):
        q, k, attn = ctx.saved_tensors
        alpha, m_valid, (n, mp, dh) = ctx.cfg
        want_dbias = ctx.has_bias and ctx.needs_input_grad[2]      # key-padding masks are constants: no column sums
        dx, dbias = SoftmaxBwdFn.apply(attn, dattn.contiguous(), alpha, m_valid, want_dbias)
        dq = dk = None
        if ctx.needs_input_grad[0]:   # dq(i,d) = sum_j dx(i,j) k(j,d)
            dq = GemmFn.apply(dx, k, True, False, (n, dh, mp), None, None, 1.0, False)
        if ctx.needs_input_grad[1]:   # dk(j,d) = sum_i dx(i,j) q(i,d)
            dk = GemmFn.apply(dx, q, False, False, (mp, dh, n), None, None, 1.0, False)
        return dq, dk, (dbias if want_dbias else None), None, None


class SoftmaxBwdFn(Function):
    """(dx, dbias) = (alpha*u, column sums of u), u = S*(dS - rowsum(S*dS)); its own backward (second order,
    gradient-penalty steps only) is one fused HIP pass too (gg_softmax_bwd2); third order is not provided."""

    @staticmethod
    def forward(ctx, S, dS, alpha, m_valid, want_dbias):
        ctx.set_materialize_grads(False)
        dx, dbias = K.softmax_bwd(S, dS, alpha, m_valid, want_dbias)
        ctx.alpha, ctx.m_valid = alpha, m_valid
        ctx.save_for_backward(S, dS)
        if dbias is None:
            dbias = S.new_empty(0, dtype=torch.float32)      # placeholder (no launch)
            ctx.mark_non_differentiable(dbias)
        return dx, dbias

    @staticmethod
    def backward(ctx, g_dx, g_dbias):
        S, dS = ctx.saved_tensors
        gb = g_dbias if (g_dbias is not None and g_dbias.dim() == 2) else None
        if g_dx is None and gb is None:
            return None, None, None, None, None
        g_S, g_dS = K.softmax_bwd2(S, dS, None if g_dx is None else g_dx.contiguous(),
                                   None if gb is None else gb.float().contiguous(), ctx.alpha, ctx.m_valid)
        return g_S, g_dS, None, None, None


def _pad_last(t: torch.Tensor) -> torch.Tensor:
    p = (-t.shape[-1]) % 8
    return F.pad(t, (0, p)) if p else t


def _fit(g: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """pad a logical-extent gradient with zeros up to the (aligned) storage shape of the operand; reduce over
    a broadcast batch dimension."""
    if g.dim() == 3 and like.dim() == 2:
        g = g.sum(0) if g.shape[0] > 1 else g[0]
    elif g.dim() == 3 and like.dim() == 3 and like.shape[0] == 1 and g.shape[0] > 1:
        g = g.sum(0, keepdim=True)
    pr, pc = like.shape[-2] - g.shape[-2], like.shape[-1] - g.shape[-1]
    if pr or pc:
        g = F.pad(g, (0, pc, 0, pr))
    return g


def matmul_nt(x: torch.Tensor, w: torch.Tensor, bias=None, act=None, out_f32=False) -> torch.Tensor:
    """x (..., K) @ w(N, K)^T  — the nn.Linear contraction, on the HIP GEMM. K and N multiples of 8 are
    native; other sizes are zero-padded by differentiable glue."""
    lead = x.shape[:-1]
    k, n = x.shape[-1], w.shape[0]
    x2 = _pad_last(x.reshape(-1, k).to(ACT_DTYPE))
    w2 = _pad_last(w.to(ACT_DTYPE))
    out = GemmFn.apply(x2.contiguous(), w2.contiguous(), True, True, (x2.shape[0], n, k), bias, act, 1.0, out_f32)
    return out.reshape(*lead, n)


class LinearFn(Function):
    """y = act(scale * (x @ W^T + bias)) for a 2-D weight parameter owned by a FlatAdamW (EqualLinear of the style network,
    gp.py:871-888, :909-921): the bf16 operand comes from the pack table (no per-call scale / cast / pad launches), the learning-rate
    multiplier rides on the GEMM's alpha / bias_scale and the leaky-relu on its epilogue. Backward: one mask + column-sum pass,
    two GEMMs, and the parameter gradients go straight into the flat .grad views when the trainer's grad sink is on. First
    order only (the style network is never under the gradient penalty)."""

    @staticmethod
    def forward(ctx, x, w, bias, scale, act):
        O, I = w.shape
        wb = packed_weight(w, 'fwd')                              # (O, I) bf16 (O, I multiples of 8: checked by the caller)
        xb = x.reshape(-1, I)
        xb = xb.to(ACT_DTYPE) if xb.dtype != ACT_DTYPE else xb
        xb = xb.contiguous()
        out = K.gemm(xb, wb, trans_b=True, alpha=scale, bias=bias, bias_scale=scale, act=act, act_slope=LRELU_SLOPE)[0]
        ctx.cfg = (scale, act, tuple(x.shape), x.dtype)
        ctx.save_for_backward(xb, w, bias, out if act else None)
        return out.reshape(*x.shape[:-1], O)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        xb, w, bias, out = ctx.saved_tensors
        scale, act, xshape, xdtype = ctx.cfg
        O, I = w.shape
        wb = packed_weight(w, 'fwd')
        g = dout.reshape(-1, O)
        g = (g.to(ACT_DTYPE) if g.dtype != ACT_DTYPE else g).contiguous()
        want_db = bias is not None and ctx.needs_input_grad[2]
        db = dw = dx = None
        if act == 'lrelu' or want_db:
            g, part = K.bias_act_bwd(g, out if act == 'lrelu' else None, want_db, LRELU_SLOPE, partials=True)
            if want_db:
                bsink = _grad_sink_of(bias)
                if bsink is not None:
                    K.finish_queue.add_colsum(part, O, scale, bsink, notify=lambda b=bias: grad_ready(b))
                else:
                    db = K.colsum_finish(part, O, scale)
        if ctx.needs_input_grad[0]:     # dx(r, i) = scale * sum_o g(r, o) W(o, i)
            dx = K.gemm(g, wb, trans_b=False, alpha=scale)[0].reshape(xshape).to(xdtype)
        if ctx.needs_input_grad[1]:     # dW(o, i) = scale * sum_r g(r, o) x(r, i)
            gw = K.gemm(g, xb, trans_a=True, trans_b=False, alpha=scale, out_dtype=torch.float32)[0]
            sink = _grad_sink_of(w)
            if sink is not None:
                K.finish_queue.add_axpy(gw, 1.0, sink.view(-1), notify=lambda w=w: grad_ready(w))
            else:
                dw = gw
        return dx, dw, db, None, None


class GlobalMeanFn(Function):
    """(b, C, H, W) bf16 channels_last -> (b, C) fp32 mean over the pixels (SqueezeExcite's pool, gp.py:300): a two-stage HIP
    reduction over the bf16 tensor (gg_pool_mean_fwd). `fork=True` returns (mean, x): the trunk continues with the alias and
    its gradient meets the pool's broadcast gradient inside ONE pass (gg_pool_mean_bwd, in place on the trunk's gradient)
    instead of expand + cast + strided add. First order only (the gradient-penalty graphs use the tensor-algebra form)."""

    @staticmethod
    def forward(ctx, x, fork=False):
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)
        C = x.shape[1]
        if C % 8 or C > 512:
            m = x.mean(dim=(2, 3), dtype=torch.float32)
        else:
            m = K.pool_mean(nhwc(x))
        return (m, x.view_as(x)) if fork else m

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, g_alias=None):
        b, C, H, W = ctx.shape
        if g is None:
            return g_alias, None
        gs = (g.float() * (1.0 / (H * W))).contiguous()
        if C % 8:
            out = gs.to(ACT_DTYPE)[:, None, None, :].expand(b, H, W, C).contiguous().permute(0, 3, 1, 2)
            return (out if g_alias is None else out + g_alias), None
        ga = None
        if g_alias is not None:
            ga = nhwc(g_alias.contiguous(memory_format=torch.channels_last))
        return nchw(K.pool_mean_bwd(gs, (b, H, W, C), ga, inplace=True)), None


class SeMlpFn(Function):
    """SqueezeExcite's excitation MLP (gp.py:297-307) behind its pool: sigmoid(W2 silu(W1 m + b1) + b2) on the pooled (b, C) fp32 rows
    as ONE launch on the fp32 parameters where they lie (no operand packing, casts or pointwise passes: ten launches before), the
    backward as two (per-sample chain, then the parameter gradients summed over the samples), parameter gradients through the queued
    finishes into the flat .grad views like every other layer's. First order only (the gradient-penalty graphs keep the modules)."""

    @staticmethod
    def forward(ctx, m, w1, b1, w2, b2):
        m = m.contiguous()
        h, hs, e = K.se_mlp_fwd(m, w1, b1, w2, b2)
        ctx.save_for_backward(m, w1, b1, w2, b2, h, hs, e)
        return e

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, de):
        m, w1, b1, w2, b2, h, hs, e = ctx.saved_tensors
        params = (w1, b1, w2, b2)
        want = [p_ is not None and ctx.needs_input_grad[1 + i] and not inputs_only for i, p_ in enumerate(params)]
        de = de.float().contiguous()
        dm, gw = K.se_mlp_bwd(de, e, h, hs, m, w1, w2, want_dm=ctx.needs_input_grad[0], want_gw=any(want))
        grads = [None, None, None, None]
        for i, p_ in enumerate(params):
            if not want[i]:
                continue
            sink = _grad_sink_of(p_)
            if sink is not None:
                K.finish_queue.add_axpy(gw[i].reshape(-1), 1.0, sink.view(-1), notify=lambda q=p_: grad_ready(q))
            else:
                grads[i] = gw[i].reshape(p_.shape).clone()
        return (dm, *grads)


class GeluFn(Function):
    """exact GELU on a dense bf16 buffer (any shape; the storage order is irrelevant to a pointwise op), one HIP pass;
    backward and the backward of the backward are one pass each (gg_gelu modes 1 / 2)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.gelu(x)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        return GeluBwdFn.apply(x, dy.contiguous())


class GeluBwdFn(Function):
    @staticmethod
    def forward(ctx, x, dy):
        ctx.save_for_backward(x, dy)
        return K.gelu(x, dy)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, dy = ctx.saved_tensors
        g_dy, g_x = K.gelu(x, dy, g.contiguous())
        return g_x, g_dy


def _dense_view(x: torch.Tensor):
    """the flat storage-order view of a dense tensor (contiguous in SOME dimension order), or None."""
    if x.is_contiguous():
        return x.view(-1)
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return x.permute(0, 2, 3, 1).reshape(-1)
    return None


class TakeRowsFn(Function):
    """x[:n] along the batch axis (the discriminator's predictors see the stage's first rows only, gp.py:1789). The stock
    slice backward fills an NCHW zeros tensor and copies into it, after which the accumulation with the main path's
    channels_last gradient runs on the strided add kernel; here the gradient is built channels_last in one cat."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.rest = (x.shape[0] - n,) + tuple(x.shape[1:])
        return x[:n]

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous(memory_format=torch.channels_last)
        pad = torch.zeros(ctx.rest, dtype=g.dtype, device=g.device).contiguous(memory_format=torch.channels_last)
        return torch.cat((g, pad), dim=0), None


def _heads_view(t: torch.Tensor, heads: int) -> torch.Tensor:
    """(n, C) rows of one image (row pitch free, unit channel stride) -> (heads, n, 64) strided view, no copy."""
    n, C = t.shape
    return t.as_strided((heads, n, C // heads), (C // heads, t.stride(0), 1), t.storage_offset())


class LinearAttnFn(Function):
    """LinearAttention core (unet.py:338-348) on the fused to_qkv output qkv (b, n, 3C) NHWC bf16, heads of 64 features:
    out = (scale * softmax_features(q)) @ (softmax_positions(k)^T v). Two HIP softmax passes (gg_linattn_q / _k) and, per
    image, two batched MFMA GEMMs over strided head views — q, k, v, out and all their gradients stay in the (b, n, C)
    channel-slice layout the 1x1 projections read and write (the stock formulation moves each operand through fp32 and two
    transposing copies: 195 GB of elementwise traffic per upsampler step). First order (generator side)."""

    @staticmethod
    def forward(ctx_, qkv, heads, scale):
        b, n, C3 = qkv.shape
        C = C3 // 3
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        qs = K.linattn_q_fwd(q, scale)
        eks = K.linattn_k_fwd(k)
        d = C // heads
        ctx = torch.empty((b, heads, d, d), dtype=ACT_DTYPE, device=qkv.device)
        out = torch.empty((b, n, C), dtype=ACT_DTYPE, device=qkv.device)
        # one batched launch per contraction and per HEAD (batch = the images: a head's 64 channels of (b, n, C) are a (b, n, 64) view
        # with batch stride n C) when there are more images than heads, per IMAGE (batch = the heads, stride 64) otherwise: the
        # GEMM descriptor carries one batch stride, and (image, head) is a two-level index
        if b > heads:
            for h in range(heads):
                sl = slice(h * d, (h + 1) * d)
                K.gemm(eks[..., sl], v[..., sl], trans_a=True, trans_b=False, out=ctx[:, h])                           # (b, d, e)
                K.gemm(qs[..., sl], ctx[:, h], trans_a=False, trans_b=False, out=out[..., sl])
        else:
            for i in range(b):
                K.gemm(_heads_view(eks[i], heads), _heads_view(v[i], heads), trans_a=True, trans_b=False, out=ctx[i])   # (h, d, e)
                K.gemm(_heads_view(qs[i], heads), ctx[i], trans_a=False, trans_b=False, out=_heads_view(out[i], heads))
        ctx_.cfg = (heads, scale)
        ctx_.save_for_backward(qkv, qs, eks, ctx)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx_, g):
        qkv, qs, eks, ctx = ctx_.saved_tensors
        heads, scale = ctx_.cfg
        b, n, C3 = qkv.shape
        C = C3 // 3
        d = C // heads
        g = g.contiguous()
        v = qkv[..., 2 * C:]
        dqkv = torch.empty_like(qkv)
        dqs = torch.empty((b, n, C), dtype=ACT_DTYPE, device=qkv.device)
        deks = torch.empty((b, n, C), dtype=ACT_DTYPE, device=qkv.device)
        if b > heads:        # (as in the forward: batch = the images of one head)
            dctx = torch.empty((b, d, d), dtype=ACT_DTYPE, device=qkv.device)
            dv = dqkv[..., 2 * C:]
            for h in range(heads):
                sl = slice(h * d, (h + 1) * d)
                K.gemm(qs[..., sl], g[..., sl], trans_a=True, trans_b=False, out=dctx)                     # dctx[d, e] = sum_n qs g
                K.gemm(g[..., sl], ctx[:, h], trans_a=False, trans_b=True, out=dqs[..., sl])               # dqs = g ctx^T
                K.gemm(v[..., sl], dctx, trans_a=False, trans_b=True, out=deks[..., sl])                   # deks = v dctx^T
                K.gemm(eks[..., sl], dctx, trans_a=False, trans_b=False, out=dv[..., sl])                  # dv = eks dctx
        else:
            dctx = torch.empty((heads, d, d), dtype=ACT_DTYPE, device=qkv.device)
            for i in range(b):
                gi, qi, ei, vi = (_heads_view(t[i], heads) for t in (g, qs, eks, v))
                K.gemm(qi, gi, trans_a=True, trans_b=False, out=dctx)                                          # dctx[d, e] = sum_n qs g
                K.gemm(gi, ctx[i], trans_a=False, trans_b=True, out=_heads_view(dqs[i], heads))               # dqs = g ctx^T
                K.gemm(vi, dctx, trans_a=False, trans_b=True, out=_heads_view(deks[i], heads))                # deks = v dctx^T
                K.gemm(ei, dctx, trans_a=False, trans_b=False, out=_heads_view(dqkv[i, :, 2 * C:], heads))    # dv = eks dctx
        K.linattn_q_bwd(qs, dqs, dqkv[..., :C], scale)
        K.linattn_k_bwd(eks, deks, dqkv[..., C:2 * C])
        return dqkv, None, None


class ScaledAddFn(Function):
    """(a + b) * c + d in one pass over dense bf16 tensors of one layout (b, d may be None); linear, so every derivative is the
    op itself: a and b receive g * c — ONE tensor, computed once — and d receives g."""

    @staticmethod
    def forward(ctx, a, b, c, d=None):
        ctx.c, ctx.has = c, (b is not None, d is not None)
        return K.scaled_add(a, b, c, d)

    @staticmethod
    def backward(ctx, g):
        if _dense_view(g) is None:
            g = g.contiguous(memory_format=torch.channels_last) if g.dim() == 4 else g.contiguous()
        gs = ScaledAddFn.apply(g, None, ctx.c)
        return gs, (gs if ctx.has[0] else None), None, (g if ctx.has[1] else None)


class AddCatFn(Function):
    """cat((x + tile(feats), tile(feats))) over the batch axis — the discriminator's multi-scale input merge (gp.py:1797-1803)
    — in one pass over NHWC bf16 tensors (the stock form is a tile copy, an add and a cat); backward: the x gradient is the
    first half of g (a view), the feats gradient one reduction pass. First order (penalty graphs keep the tensor algebra)."""

    @staticmethod
    def forward(ctx, x, feats):
        ctx.f = feats.shape[0]
        return K.addcat(x, feats)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        g = g.contiguous()
        gx = g[:g.shape[0] // 2] if ctx.needs_input_grad[0] else None
        gf = K.addcat_bwd(g, ctx.f) if ctx.needs_input_grad[1] else None
        return gx, gf


class RowsForkFn(Function):
    """(x[:n], x): the first n batch rows for one consumer (a predictor) and x itself for the other (the rest of the
    trunk). The backward adds the rows' gradient INTO the trunk gradient's first n rows (in place: that buffer is the fresh
    output of the trunk's data-gradient pass and has no other reader) instead of zero-padding it to full size and letting
    autograd add two full-size tensors."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.total = n, x.shape[0]
        ctx.set_materialize_grads(False)
        return x[:n], x.view_as(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rows, g_all):
        n = ctx.n
        if g_all is None:           # the trunk contributed nothing (never the case in the discriminator): zero-padded rows
            if g_rows is None:
                return None, None
            g_rows = g_rows.contiguous(memory_format=torch.channels_last)
            pad = torch.zeros((ctx.total - n,) + tuple(g_rows.shape[1:]), dtype=g_rows.dtype,
                              device=g_rows.device).contiguous(memory_format=torch.channels_last)
            return torch.cat((g_rows, pad), dim=0), None
        if g_rows is None:
            return g_all, None
        if not (g_all.is_contiguous(memory_format=torch.channels_last) or g_all.is_contiguous()):
            g_all = g_all.contiguous(memory_format=torch.channels_last)
        g_all[:n].add_(g_rows)
        return g_all, None


class PoolHighFreqFn(Function):
    """(max_pool2d(x, 2), x - blur(x)) of an NHWC bf16 tensor in one HIP pass; backward in one pass too (the pooled gradient
    goes to each window's first maximum, the high-frequency gradient through the exact adjoint of the reflect-padded blur).
    First order (the unet is the generator: never under the gradient penalty)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return K.poolhf_fwd(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_pool, g_hf):
        x, = ctx.saved_tensors
        if g_pool is None and g_hf is None:
            return None
        gp = None if g_pool is None else g_pool.to(ACT_DTYPE).contiguous()
        gh = None if g_hf is None else g_hf.to(ACT_DTYPE).contiguous()
        return K.poolhf_bwd(x, gp, gh)


class ResampleFn(Function):
    """Separable banded linear resampling of an NHWC tensor (bilinear x2 + binomial blur, bilinear resize,
    and their adjoints), closed under differentiation: backward = the same kernel with the transposed
    tables."""

    @staticmethod
    def forward(ctx, x, spec):
        ctx.spec = spec
        return K.resample_nhwc(x, spec)

    @staticmethod
    def backward(ctx, dy):
        return ResampleFn.apply(dy.contiguous(), ctx.spec.transposed()), None


# --------------------------------------------------------------------------------------------------
# the op set
# --------------------------------------------------------------------------------------------------
_prepared: dict = {}        # id(weights) -> coefficients / per-sample weights computed by HipOps.modconv_prepare for the running forward

class HipOps:
    """MI355X implementation: HIP kernels for contractions/resampling, torch glue for pointwise pieces."""

    name = 'hip'
    act_dtype = ACT_DTYPE

    # -- activations layout ------------------------------------------------------------------------
    def prepare(self, x):
        return to_act(x)

    # -- convolution -------------------------------------------------------------------------------
    fuses_forks = True      # conv2d / channel_rmsnorm accept fork=True (second consumer's gradient joins inside the backward pass)

    def conv2d(self, x, weight, bias=None, act=None, stride=1, scale=1.0, residual=None, fork=False, res_scale=1.0):
        """scale * (conv(x, w) + bias) [-> leaky relu]: stride-1 'same' conv (odd square kernel) — the reference's
        nn.Conv2d(…, padding=k//2) call sites — or the stride-2 1x1 residual conv (gp.py:1612), whose pixel
        sub-sampling is part of the kernel's gather. `fork=True` returns (y, x'): hand x' to x's other consumer."""
        x = to_act(x)
        o, i, k = weight.shape[0], weight.shape[1], weight.shape[-1]
        assert stride == 1 or k == 1
        xh = nhwc(x)
        ip = _round8(i)
        if fork and (ip != i or o % 8 or stride != 1):
            return self.conv2d(x, weight, bias, act, stride, scale, residual, res_scale=res_scale), x      # ragged channels: plain fork
        if ip != i:
            xh = F.pad(xh, (0, ip - i))
        geom = (k, stride, k // 2 if stride == 1 else 0, 'oihw')
        res = None
        if residual is not None:
            if o % 8:        # ragged channel count: add outside the kernel
                return self.conv2d(x, weight, bias, act, stride, scale) + residual.to(ACT_DTYPE) * res_scale
            _check_act_residual(act, residual, x, weight, bias, residual)
            res = nhwc(to_act(residual))
        if fork:
            y, xa = ConvFn.apply(xh, weight, None if bias is None else bias.float().contiguous(), None, act, geom,
                                 float(scale), res, True, float(res_scale))
            return nchw(y), nchw(xa)
        y = ConvFn.apply(xh, weight, None if bias is None else bias.float().contiguous(), None, act, geom, float(scale),
                         res, False, float(res_scale))
        if y.shape[-1] != o:
            y = y[..., :o]
        return nchw(y)

    def hinge(self, x, split=None):
        """discriminator hinge over a merged (outer, 2b, ...) logit tensor whose first `split` batch rows are the fake half, or
        (split None) the generator's hinge mean(x); None when the tensor is not in a form the kernel takes."""
        if x.dtype not in (torch.bfloat16, torch.float32) or x.dim() < 2 or x.numel() == 0:
            return None
        if not x.is_contiguous():       # (a channels_last map viewed as (s, 2b, c, h, w): one dense copy of a tiny tensor, still 2 launches for ~16)
            x = x.contiguous()
        if split is None:
            return HingeFn.apply(x, x.shape[1], 0, 0)
        if x.shape[1] != 2 * split:
            return None
        return HingeFn.apply(x, x.shape[1], split, 1)

    def ff_tail(self, n, w_in, b_in, w_out, b_out, residual):
        """the FeedForward behind its norm with the GELU on the GEMM epilogues (FFTailFn), or None when this call cannot take that
        form (twice-differentiated graphs, ragged channel counts, launches the planner would not put on the staged epilogue)."""
        if second_order or inputs_only or _NO_FF_FUSE:
            return None
        hid, dim = w_in.shape[0], w_in.shape[1]
        if w_in.shape[-1] != 1 or w_out.shape[-1] != 1 or dim % 8 or hid % 8 or w_out.shape[0] != dim or b_in is None:
            return None
        nh = nhwc(to_act(n))
        res = None if residual is None else nhwc(to_act(residual))
        bi = b_in.float().contiguous()
        bo = None if b_out is None else b_out.float().contiguous()
        # both GELU-carrying launches (the up-projection, and the down-projection's data gradient: hid -> dim channels back to hid)
        # must land on a staged epilogue (the 8-wave tiles or the persistent short-K kernel), unsplit: ask the planner with the launches' own descriptors
        staged = lambda plan: (4 <= plan[0] <= 6 or plan[0] == 15) and plan[1] == 1       # (15: gg_pgemm, the same staged epilogue)
        with_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (n, w_in, b_in, w_out, b_out, residual))
        # (the cached answer belongs to one plan table: kernels.Library.load_plan_table clears this cache)
        key = (tuple(nh.shape), hid, nh.device, nh.dtype, with_grad)
        ok = _ff_plan_cache.get(key)
        if ok is None:
            ok = staged(K.conv2d_nhwc(nh, packed_weight(w_in, 'fwd'), ksize=1, bias=bi, plan_only=True))
            if ok and with_grad:        # the data gradient of the down-projection only exists in a differentiated call
                hid_like = nh.new_empty(nh.shape[:3] + (dim,))
                ok = staged(K.conv2d_nhwc(hid_like, packed_weight(w_out, 'bwd'), ksize=1, plan_only=True))
            _ff_plan_cache[key] = ok
        if not ok:
            return None
        if with_grad:
            return nchw(FFTailFn.apply(nh, w_in, bi, w_out, bo, res))
        hgelu = K.conv2d_nhwc(nh, packed_weight(w_in, 'fwd'), ksize=1, bias=bi, act='gelu')      # no-grad: GELU in the epilogue, one output
        return nchw(K.conv2d_nhwc(hgelu, packed_weight(w_out, 'fwd'), ksize=1, bias=bo, residual=res))

    def downsample(self, x, weight, bias=None, residual=None, scale=1.0):
        """scale * (conv1x1(space_to_depth(x), w) + bias) + residual  (gp.py:289-293 and the residual merge
        gp.py:1826): one launch of the 2x2 / stride-2 gather with the merge in its epilogue."""
        x = to_act(x)
        o = weight.shape[0]
        assert weight.shape[1] == 4 * x.shape[1] and x.shape[1] % 8 == 0 and o % 8 == 0
        res = None if residual is None else nhwc(to_act(residual))
        y = ConvFn.apply(nhwc(x), weight, None if bias is None else bias.float().contiguous(), None, None,
                         (2, 2, 0, 's2d'), float(scale), res)
        return nchw(y)

    @staticmethod
    def _table_linear_ok(x, weight, bias, act):
        """a 2-D weight parameter owned by a FlatAdamW with 8-aligned extents: LinearFn takes its bf16 operand from the pack table."""
        return (not second_order and isinstance(weight, torch.nn.Parameter) and getattr(weight, '_gg_pack_table', None) is not None
                and not _DEBUG_NO_TABLE and weight.dim() == 2 and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0
                and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
                and act in (None, 'lrelu') and x.shape[-1] == weight.shape[1]
                and not (weight.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
                         and '_gg_tpacks' not in weight.__dict__))

    def linear(self, x, weight, bias=None, act=None):
        if self._table_linear_ok(x, weight, bias, act):      # (no per-call weight cast / pad, gradients into the flat views)
            return LinearFn.apply(x, weight, bias, 1.0, act)
        return matmul_nt(x, weight, None if bias is None else bias.float().contiguous(), act)

    def linear_f32(self, x, weight, bias=None):
        """nn.Linear with an fp32 result: the generator's style -> modulation projection (gp.py:1160-1175). Every adaptive conv
        reads a column slice of it as fp32 rows (the coefficient kernels' input type), so a bf16 result would cost one cast per
        slice and layer in the forward and one in the backward (~90 launches of a few microseconds per step on config 2)."""
        return matmul_nt(x, weight, None if bias is None else bias.float().contiguous(), None, out_f32=True)

    def equal_linear(self, x, weight, bias, lr_mul, act=None):
        """EqualLinear (gp.py:871-888): act(linear(x, weight * lr_mul, bias * lr_mul)). Parameters of a FlatAdamW-owned model with
        8-aligned extents run as LinearFn (operand from the pack table, multiplier and activation in the GEMM epilogue)."""
        if self._table_linear_ok(x, weight, bias, act):
            return LinearFn.apply(x, weight, bias, float(lr_mul), act)
        y = self.linear(x, weight * lr_mul, None if bias is None else bias * lr_mul)
        return F.leaky_relu(y, LRELU_SLOPE) if act == 'lrelu' else y

    def squeeze_excite_mlp(self, m, lin1, lin2):
        """the four modules behind SqueezeExcite's pool (gp.py:301-304) as one autograd node / one launch: m (b, C) fp32 pooled rows
        -> the (b, O) fp32 excitation; None when the fused form does not apply (gradient-penalty graphs, foreign dtypes, widths
        beyond the kernel's LDS rows): the caller runs the modules."""
        w1, b1, w2, b2 = lin1.weight, lin1.bias, lin2.weight, lin2.bias
        ok = (not second_order and m.dim() == 2 and m.dtype == torch.float32 and w1.shape[1] == m.shape[1] and w2.shape[1] == w1.shape[0]
              and all(t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.device == m.device) for t in (w1, b1, w2, b2))
              and max(m.shape[1], w2.shape[0]) <= K.SEMLP_MAX_C and w1.shape[0] <= K.SEMLP_MAX_H and m.shape[0] <= 65535)
        if not ok:
            return None
        return SeMlpFn.apply(m, w1, b1, w2, b2)

    # -- skip-layer excitation multiply (gp.py:1023-1024, :1812-1813) ----------------------------
    def channel_scale(self, x, s):
        """x (b, C, H, W) * s (b, C, 1, 1): one bf16 pass forward, and ONE backward pass for both dx = g*s and
        ds = sum over pixels of g*x (gg_modulate_fwd / _bwd). Graphs that are differentiated twice keep the tensor-algebra
        form."""
        x = to_act(x)
        b, C = x.shape[0], x.shape[1]
        if second_order or C % 8 or not (torch.is_grad_enabled() and (x.requires_grad or s.requires_grad)):
            return x * s.reshape(b, C, 1, 1).to(x.dtype)
        return nchw(ModulateFn.apply(nhwc(x), s.reshape(b, C).float().contiguous()))

    def gelu(self, x):
        """nn.GELU() (exact) of the attention feed-forward, gp.py:731."""
        if x.dtype != ACT_DTYPE:
            x = x.to(ACT_DTYPE)
        flat = _dense_view(x)
        if flat is None or flat.numel() % 8:
            _shape_fallback('gelu')
            return F.gelu(x)
        y = GeluFn.apply(flat)
        if x.is_contiguous():
            return y.view(x.shape)
        b, c, h, w = x.shape
        return y.view(b, h, w, c).permute(0, 3, 1, 2)

    def scaled_add(self, a, b, c, d=None):
        """(a + b) * c [+ d]: the predictor blocks' residual merges (gp.py:1493, :1495)."""
        a, b = to_act(a), to_act(b)
        d = None if d is None else to_act(d)
        ok = a.numel() % 8 == 0 and a.stride() == b.stride() and _dense_view(a) is not None and (d is None or d.stride() == a.stride())
        if not ok:
            y = (a + b) * c
            return y if d is None else y + d
        return ScaledAddFn.apply(a, b, float(c), d)

    def add_cat(self, x, feats):
        """cat((x + tile(feats), tile(feats)), dim=0), feats (f, C, H, W) tiled scale-major over x's batch (gp.py:1797-1803)."""
        x, feats = to_act(x), to_act(feats)
        if second_order or x.shape[1] % 8:
            from .modules import tile_batch
            ft = tile_batch(feats, x.shape[0])
            return torch.cat((x + ft, ft), dim=0)
        return nchw(AddCatFn.apply(nhwc(x), nhwc(feats)))

    def take_rows(self, x, n, fork=False):
        """x[:n] (batch rows) with a channels_last gradient. `fork=True` returns (x[:n], x'): the caller continues with x'
        and the rows' gradient is added into the trunk's gradient in place (RowsForkFn)."""
        if fork:
            if n >= x.shape[0] or second_order or x.dim() != 4 or not (torch.is_grad_enabled() and x.requires_grad):
                return self.take_rows(x, n), x
            return RowsForkFn.apply(x, n)
        if n >= x.shape[0]:
            return x
        if second_order or x.dim() != 4 or not (torch.is_grad_enabled() and x.requires_grad):
            return x[:n]
        return TakeRowsFn.apply(x, n)

    def global_mean(self, x, fork=False):
        """mean over the pixels in fp32 (the squeeze of SqueezeExcite, gp.py:300). `fork=True` returns (mean, x'): the trunk
        continues with x' (GlobalMeanFn)."""
        x = to_act(x)
        if second_order:
            m = x.mean(dim=(2, 3), dtype=torch.float32)
            return (m, x) if fork else m
        if not (torch.is_grad_enabled() and x.requires_grad):
            C = x.shape[1]
            m = x.mean(dim=(2, 3), dtype=torch.float32) if (C % 8 or C > 512) else K.pool_mean(nhwc(x))
            return (m, x) if fork else m
        return GlobalMeanFn.apply(x, True) if fork else GlobalMeanFn.apply(x)

    # -- adaptive / modulated conv (gp.py:315-409) -----------------------------------------------
    @staticmethod
    def _modconv_path(b, N, O, I, H, W):
        """which no-grad formulation a demodulated 3x3 adaptive conv runs in (see modconv2d)."""
        if I in (16, 32, 64) and O <= 32 and W % 32 == 0 and b * H * W >= 32768:
            return 'sconv'
        pimg_ok = H * W % 128 == 0 and H * W >= 1024 and b * O * I * 9 <= (16 << 20)
        # 32x32 (round 6): per-sample weights + gg_conv3's 64-column tile with two workgroups per CU beats the shared bank's doubled
        # reduction where its grid fills the chip (b * tiles * column tiles >= 256 workgroups: config 2's batch 32; 44.9 -> 39.4 us and
        # 26.5 -> 24.7 us, the modulation launch + 6 us: forward graph 0.506 -> 0.490 ms same-box, profiles/r06_conv3_pair_ab.log)
        if _ACONV and _aconv_ok(b, N, O, I, H, W) and not (W == 32 and pimg_ok and I % 64 == 0 and b * (H * W // 256) * ((O + 63) // 64) >= 256):
            return 'aconv'
        if pimg_ok:
            return 'pimg'
        return 'bank'

    def modconv_prepare(self, specs):
        """ONE launch for the style-dependent part of every announced adaptive conv of a no-grad generator forward (gg_modw_multi_fwd):
        all of a forward's modulations are column slices of one style projection (gp.py:1160-1175), so softmax(kernel_mod), the
        demodulation coefficients and - where the layer runs on per-sample weights - the weights themselves are computed before the
        first convolution instead of by one launch in front of each (15 on config 2: ~330 us of a 1.1 ms forward). `specs`: tuples
        (weights, mod, kernel_mod, H, W, excited, demod, eps). A layer behind a skip-layer excitation (gp.py:1023-1024: its input is
        scaled per sample and channel by a value that only exists once earlier layers have run) gets its per-sample weights WITHOUT
        that scale here; its convolution applies the scale to the weights as they are staged (gg_sconv_fwd `xs`, gg_conv3 SCALED) -
        conv(x * e, w) = conv(x, w * e). The bank's Gram rows come from the pack table (refreshed once per optimizer step), so a
        coefficient-only layer never reads its bank here. The results wait in a registry keyed by the weight tensor until that
        layer's modconv2d call picks them up."""
        _prepared.clear()
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for sp in specs for t in sp[:3]):
            return 0
        layers, metas = [], []
        for weights, mod, kmod, H, W, excited, demod, eps in specs:
            N, O, I, k, _ = weights.shape
            b = mod.shape[0]
            if (k != 3 or not demod or weights.dtype != torch.float32 or not weights.is_contiguous() or I % 8 or O % 8
                    or not K.modw_eligible(b, N, I, k * k) or weights.device.type != mod.device.type):
                continue
            path = self._modconv_path(b, N, O, I, H, W)
            if excited and (path == 'bank' or (path == 'pimg' and I % 64)):
                continue            # (no consumer-side scale on these paths: the layer keeps its own launch)
            # ('aconv': coefficients only - s, a, d; the excitation is a second input scale of the convolution itself)
            ly = dict(w=weights.detach(), mod=_rows_f32(mod), kmod=_rows_f32(kmod) if N > 1 else None, demod=demod, eps=eps,
                      Ip=I, Op=O)
            if (isinstance(weights, torch.nn.Parameter) and getattr(weights, '_gg_pack_table', None) is not None
                    and not _DEBUG_NO_TABLE):
                ly['gram'] = _table_pack(weights, 'gram')
            if path == 'sconv':
                ly.update(coef=False, wmix=_wmix_buffer(weights, b, I), layout=2)
            elif path == 'pimg':
                ly.update(coef=False, wmix=_wmix_rows(weights, b, O, 9 * I), layout=1)
            layers.append(ly)
            metas.append((weights, mod, path, b, excited))
        if not layers:
            return 0
        # heaviest items first: the per-sample-weight builds (one workgroup per output channel walking every sample) start in the
        # launch's first wave of workgroups, the short coefficient-only items fill in behind them
        order = sorted(range(len(layers)), key=lambda i: -(0 if layers[i].get('coef', True) else layers[i]['w'].numel()))
        layers, metas = [layers[i] for i in order], [metas[i] for i in order]
        outs = K.modw_multi(layers)
        for (weights, mod, path, b, excited), o in zip(metas, outs):
            _prepared[id(weights)] = dict(o, mod_ptr=mod.data_ptr(), path=path, b=b, excited=excited)
        # the one-launch layers in the order the forward runs them (`specs` order): each one's launch requests the NEXT one's bank
        # into the L2 (kernels.aconv `next_bank`) - the banks are the traffic of these layers and are cold once per forward
        chain = [(sp[0], sp[3]) for sp in specs if id(sp[0]) in _prepared and _prepared[id(sp[0])]['path'] == 'aconv']
        for (w0, _), (w1, h1) in zip(chain, chain[1:]):
            _prepared[id(w0)]['next_bank'] = (_frag_weight(w1), _prepared[id(w1)]['b'], h1)
        return len(layers)

    def modconv_release(self):
        _prepared.clear()

    def modconv_pair(self, x, first, second):
        """the two adaptive 3x3 convolutions of one generator block (gp.py:1219-1229: conv1 -> Noise -> leaky_relu -> conv2 -> Noise ->
        leaky_relu, nothing in between) in ONE launch where both are streaming layers of a geometry gg_spair_fwd carries (config 2:
        64 -> 32 -> 32 at 128x128, 32 -> 16 -> 16 at 256x256): the intermediate map stays in LDS. `first` / `second`: dicts with the
        keyword arguments of modconv2d (weights, mod, kernel_mod, demod, eps, noise, noise_weight, act, in_excite). Returns None when
        the pair does not qualify - the caller then runs the two layers one by one. The same arithmetic (128x128: the same bits; 256x256: the
        16-row MFMA form sums in another order)."""
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for d in (first, second)
                                           for t in (x, d['weights'], d['mod'], d.get('kernel_mod'), d.get('noise_weight'),
                                                     d.get('in_excite'))):
            return None
        if not _SPAIR or second.get('in_excite') is not None:
            return None
        x = to_act(x)
        b, _, H, W = x.shape
        w1, w2 = first['weights'], second['weights']
        N1, O1, I1, k1, _ = w1.shape
        N2, O2, I2, k2, _ = w2.shape
        if (k1 != 3 or k2 != 3 or I2 != O1 or x.shape[1] != I1 or not first.get('demod', True) or not second.get('demod', True)
                or first.get('act') not in (None, 'lrelu') or second.get('act') not in (None, 'lrelu')):
            return None
        for w, N, I in ((w1, N1, I1), (w2, N2, I2)):
            if w.dtype != torch.float32 or not w.is_contiguous() or not K.modw_eligible(b, N, I, 9):
                return None
        if (self._modconv_path(b, N1, O1, I1, H, W) != 'sconv' or self._modconv_path(b, N2, O2, I2, H, W) != 'sconv'
                or not K.spair_supported(H, W, I1, O1, O2)):
            return None
        banks = []
        for d, w, N, O, I in ((first, w1, N1, O1, I1), (second, w2, N2, O2, I2)):
            mod, kmod, exc = d['mod'], d.get('kernel_mod'), d.get('in_excite')
            rec = _prepared.pop(id(w), None)
            if rec is not None and (rec['excited'] != (exc is not None) or rec['mod_ptr'] != mod.data_ptr() or rec['path'] != 'sconv'
                                    or rec['b'] != b):
                rec = None
            wm = _wmix_buffer(w, b, I)
            xs_late = None
            if rec is None:     # not announced: this layer's own launch (the excitation folded into its per-sample weights)
                K.modw_fwd(w.detach(), _rows_f32(mod), _rows_f32(kmod) if N > 1 else None, True, d.get('eps', 1e-8), I, O, coef=False,
                           wmix=wm, layout=2, xs=None if exc is None else _rows_f32(exc.reshape(b, I)))
            elif exc is not None:
                xs_late = exc.reshape(b, I).detach().float().contiguous()
            nz = nw = None
            if d.get('noise') is not None:
                nz = d['noise'].reshape(-1).float().contiguous()
                nw = d['noise_weight'].detach().reshape(-1).float().contiguous()
            banks.append((wm, nz, nw, d.get('act'), xs_late))
        (wm1, nz1, nw1, act1, xs1), (wm2, nz2, nw2, act2, _) = banks
        y = K.spair(nhwc(x), wm1, wm2, O1, O2, nz1, nw1, nz2, nw2, act1, act2, LRELU_SLOPE, xs=xs1)
        return nchw(y)

    def modconv2d(self, x, weights, mod, kernel_mod=None, demod=True, eps=1e-8, noise=None, noise_weight=None,
                  act=None, in_excite=None):
        """`in_excite` (b, I[, 1, 1]): a per-sample scale of the input activation (the skip-layer excitation the generator applies
        right before this conv, gp.py:1023-1024); the no-grad path folds it into the per-sample weights / the modulation pass."""
        x = to_act(x)
        b, _, H, W = x.shape
        N, O, I, k, _ = weights.shape
        Ip, Op = _round8(I), _round8(O)
        needs_grad = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (x, weights, mod, kernel_mod, noise_weight, in_excite))
        w_ok = weights.dtype == torch.float32 and weights.is_contiguous()
        if not needs_grad and k == 3 and w_ok and K.modw_eligible(b, N, I, k * k) and Ip == I and Op == O:
            # no-grad forward (the discriminator step's generator pass, generate()): csrc/gg_modfwd.h. The coefficients / per-sample
            # weights come from the forward's ONE batched launch when the generator announced its layers (modconv_prepare),
            # otherwise (and for layers behind a skip-layer excitation, whose scale is not known up front) from a launch here.
            path = self._modconv_path(b, N, O, I, H, W)
            rec = _prepared.pop(id(weights), None)
            if rec is not None and (rec['excited'] != (in_excite is not None) or rec['mod_ptr'] != mod.data_ptr()
                                    or rec['path'] != path or rec['b'] != b):
                rec = None
            xs_late = None          # prepared weights of an excited layer carry no excitation: the convolution applies it
            if rec is not None and in_excite is not None:
                xs_late = in_excite.reshape(b, I).detach().float().contiguous()
            nz = nw = None
            if noise is not None:
                nz = noise.reshape(-1).float().contiguous()
                nw = noise_weight.detach().reshape(-1).float().contiguous()
            wd = weights.detach()
            if rec is None:
                km = _rows_f32(kernel_mod) if N > 1 else None       # column slices of the style network's output: read in place
                md = _rows_f32(mod)
                xs = None if in_excite is None else _rows_f32(in_excite.reshape(b, I))
            if path == 'aconv':
                # 4x4 .. 64x64: the shared bank in MFMA-fragment order streamed straight into registers, the tile's halo parked once
                # for all channels, the N kernels mixed in fp32 after the reduction, demodulation / noise / leaky-relu on the
                # accumulators: one launch, no per-sample weights, no split-K partials (csrc/gg_aconv.h)
                if rec is None:
                    s, a, d = K.modw_fwd(wd, md, km, demod, eps, Ip, Op)
                else:
                    s, a, d = rec['s'], rec['a'], rec['d']
                xs2 = None if in_excite is None else in_excite.reshape(b, I).detach().float().contiguous()
                y = K.aconv(nhwc(x), _frag_weight(weights), s, a if N > 1 else None, d if demod else None, O, nz, nw, act,
                            LRELU_SLOPE, xs=xs2, next_bank=None if rec is None else rec.get('next_bank'))
                return nchw(y)
            if path == 'sconv':
                # narrow high-resolution layers: the reference's per-sample weights (a few KiB each) + the streaming convolution
                wm = _wmix_buffer(weights, b, I)
                if rec is None:
                    K.modw_fwd(wd, md, km, demod, eps, Ip, Op, coef=False, wmix=wm, layout=2, xs=xs)
                return nchw(K.sconv(nhwc(x), wm, O, nz, nw, act, LRELU_SLOPE, xs=xs_late))
            if path == 'pimg':
                # mid resolutions: the per-sample weights are still small next to the activation (<= 32 MiB of bf16), so the
                # reference's formulation (one kernel per sample, algorithmic flops) beats the shared bank's doubled reduction:
                # per-image weight operand, modulation / demodulation folded into the weights
                wm = _wmix_rows(weights, b, O, 9 * I)
                if rec is None:
                    K.modw_fwd(wd, md, km, demod, eps, Ip, Op, coef=False, wmix=wm, layout=1, xs=xs)
                if xs_late is not None and (O > 64 or I > 512):
                    # the excitation folded into the (small) per-sample weights by one element-wise launch: conv(x * e, w) = conv(x, w * e).
                    # (Up to round 5 every excited layer went this way - gg_conv3's SCALED form with one workgroup per CU measured 91 us
                    # against 60 us for the plain one on 128 -> 64 @64x64. Round 6: the 64-column tile with two workgroups per CU takes the
                    # scale on its halo staging at no visible cost - 8.8 + 33.0 us -> 38.5 us in one launch, profiles/r06_conv3_pair_ab.log -
                    # so only shapes that tile does not certainly carry (more than 64 output channels: the planner may pick a wider tile) fold the
                    # scale into the weights.)
                    wm = K.modulate(wm.view(b, O, 9, I), xs_late).view(b, O, 9 * I)
                    xs_late = None
                y = K.conv2d_nhwc(nhwc(x), wm, ksize=3, in_scale=xs_late, noise=nz, noise_w=nw, act=act, act_slope=LRELU_SLOPE,
                                  per_image_weights=True)
                return nchw(y)
            # wide low-resolution layers (weights >> activations): shared bank, the N kernels stacked along the reduction
            if rec is None:
                s, a, d = K.modw_fwd(wd, md, km, demod, eps, Ip, Op, xs=xs)
                insc = None
            else:
                s, a, d, insc = rec['s'], rec['a'], rec['d'], rec['insc']
            wk = None
            if (isinstance(weights, torch.nn.Parameter) and getattr(weights, '_gg_pack_table', None) is not None
                    and not _DEBUG_NO_TABLE):
                wk = _table_pack(weights, 'modk')
            if wk is None:
                wk = wd.permute(1, 3, 4, 0, 2).reshape(O, k * k * N * I).to(ACT_DTYPE).contiguous()
            if H == 16 and W == 16 and N == 2 and I % 32 == 0:
                # a 256-pixel tile is ONE image here: the two kernels of the bank are mixed per image while their tiles are staged
                # (gg_lrconv MIX: a_0 W_0 + a_1 W_1 in the staging registers) and the reduction runs over the I physical channels -
                # the reference's per-sample kernel (gp.py:378-386) without writing it, and half the flops of the stacked form
                y = K.conv2d_nhwc(nhwc(x), wk, ksize=k, cv=N * Ip, in_scale=s.contiguous(), bank_mix=a.contiguous(),
                                  out_scale=d if demod else None, noise=nz, noise_w=nw, act=act, act_slope=LRELU_SLOPE)
            elif ((W >= 8 and H * W >= 64) or (H == 4 and W == 4)) and I % 64 == 0:
                # the per-(sample, stacked channel) scale a_n * s_i rides on the convolution's operand staging (gg_conv3 SCALED: applied
                # once per staged 64-channel halo chunk, shared by the nine taps; 4x4 images: gg_lrconv, plan tile 11): no modulated
                # copy of the activation is written
                if insc is None:
                    insc = (a[:, :, None] * s[:, None, :]).reshape(b, N * Ip).contiguous()
                y = K.conv2d_nhwc(nhwc(x), wk, ksize=k, cv=N * Ip, in_scale=insc, out_scale=d if demod else None, noise=nz,
                                  noise_w=nw, act=act, act_slope=LRELU_SLOPE)
            else:   # other shapes: the N-fold pre-modulated activation through one pointwise pass + the plain gather
                x2 = K.modulate_bank(nhwc(x), s, a)
                y = K.conv2d_nhwc(x2, wk, ksize=k, out_scale=d if demod else None, noise=nz, noise_w=nw, act=act,
                                  act_slope=LRELU_SLOPE)
            return nchw(y)
        if in_excite is not None:        # every other path: the plain multiply (with its fused backward when gradients flow)
            x = self.channel_scale(x, in_excite)
        fused_coef = (demod and not second_order and N <= K.MODCOEF_MAX_N and max(I, O) <= K.MODCOEF_MAX_C
                      and weights.dtype == torch.float32 and weights.is_contiguous())
        s_padded = d_padded = False
        if fused_coef:      # s, a, d in one launch (and two backward), padded to the kernels' channel multiples
            km = kernel_mod.float().contiguous() if N > 1 else None
            if needs_grad:
                s, a, d = ModCoefFn.apply(mod.float().contiguous(), km, weights, eps, Ip, Op)
            else:
                s, a, d = K.modcoef_fwd(weights.detach(), mod.detach().float().contiguous(),
                                        None if km is None else km.detach(), True, eps, Ip, Op)
            s_padded = d_padded = True
        else:
            s = mod.float() + 1.0                                       # (b, I)
            if N > 1:
                a = kernel_mod.float().softmax(dim=-1)                  # (b, N)
            else:
                a = _ones_col(b, x.device)
            d = None
            if demod:
                d = demod_coefficients(weights, s, a, eps)              # (b, O) fp32
        xh = nhwc(x)
        if Ip != I:
            xh = F.pad(xh, (0, Ip - I))
            if not s_padded:
                s = F.pad(s, (0, Ip - I))
        if (not needs_grad and k == 3 and N * Op <= 64 and Ip in (16, 32, 64) and Ip == I
                and (Ip <= 32 or N * Op <= 32) and H % 8 == 0 and W % 32 == 0 and b * H * W >= 65536 and d is not None):
            # the narrow high-resolution layers as direct convolution (style modulation applied on load, the N kernels stacked
            # along the output channels: N*O <= 64) + the mix / demodulate / noise / activation pass, instead of the
            # gather-bound implicit GEMM with the kernels stacked along the reduction. Measured (profiles/r02_modconv_ab.log):
            # 32->32@128x128 127 -> 45 us, 32->16@256x256 417 -> 93 us, 16->16@256x256 242 -> 50 us
            Y = K.conv2d_nhwc(xh, packed_weight(weights, 'fwd'), ksize=3, in_scale=s.contiguous())    # I == Ip here
            d8 = (F.pad(d, (0, Op - O)) if (Op != O and not d_padded) else d).contiguous()
            nz = nw = None
            if noise is not None:
                nz = noise.reshape(b, H * W).float().contiguous()
                nw = noise_weight.reshape(-1).float()
                nw = (F.pad(nw, (0, Op - O)) if Op != O else nw).contiguous()
            y = K.modmix_fwd(Y, a.contiguous(), d8, nz, nw, Op, N, act)
            return nchw(y[..., :O] if Op != O else y)
        if not needs_grad:
            wk = None       # parameters owned by a FlatAdamW: the [co][tap][n][ci] operand lives in the pack table
            if (isinstance(weights, torch.nn.Parameter) and getattr(weights, '_gg_pack_table', None) is not None
                    and not _DEBUG_NO_TABLE):
                wk = _table_pack(weights, 'modk')
            wts = weights
            if wk is None:
                if Ip != I:
                    wts = F.pad(wts, (0, 0, 0, 0, 0, Ip - I))
                if Op != O:
                    wts = F.pad(wts, (0, 0, 0, 0, 0, 0, 0, Op - O))
            y = fused_modconv_forward(xh, wts, s, a, d, noise, noise_weight, act, O, Op, d_padded, wk=wk)
            return nchw(y[..., :O] if Op != O else y)
        # training path: modulate -> ONE conv with the N kernels stacked along output channels -> mix/demod/noise/act
        geom = (k, 1, k // 2, 'oihw')
        if not second_order:
            xs = ModulateFn.apply(xh, s.contiguous())
            Y = ConvFn.apply(xs, weights, None, None, None, geom, 1.0, None)              # (b, H, W, N*Op)
            if N == 1 and d is None and noise is None and act is None:
                y = Y
            else:
                d8 = None if d is None else (F.pad(d, (0, Op - O)) if (Op != O and not d_padded) else d).contiguous()
                nz = nw = None
                if noise is not None:
                    nz = noise.reshape(b, H * W).float().contiguous()
                    nw = noise_weight.reshape(-1).float()
                    nw = (F.pad(nw, (0, Op - O)) if Op != O else nw).contiguous()
                y = ModMixFn.apply(Y, a.contiguous(), d8, nz, nw, Op, N, act)
            return nchw(y[..., :O] if Op != O else y)
        # twice-differentiable variant (a modulated conv inside a gradient-penalty graph: text-conditional predictor)
        Y = ConvFn.apply(xh, weights, None, s.contiguous(), None, geom, 1.0, None)
        y = (Y.view(b, H, W, N, Op).float() * a[:, None, None, :, None]).sum(dim=3)
        if Op != O:
            y = y[..., :O]
        if d is not None:
            y = y * d[:, None, None, :]
        if noise is not None:
            y = y + noise.permute(0, 2, 3, 1).float() * noise_weight.float().view(1, 1, 1, O)
        if act == 'lrelu':
            y = F.leaky_relu(y, LRELU_SLOPE)
        return nchw(y.to(ACT_DTYPE))

    # -- attention (gp.py:538-594 / :617-655): q (B,h,n,d), k/v (B,h,m,d) --------------------------
    def attention(self, q, k, v, *, scale, l2=False, key_mask=None):
        """softmax(sim * scale) v with sim = q.k (dot) or -|q-k|^2 (l2). `key_mask` (B, m) bool keeps keys."""
        B, h, n, dh = q.shape
        m = k.shape[2]
        if dh == 64 and not l2 and B * h <= 65535:
            # heads of 64 features: the fused kernel family (no probability tensor; key-padding mask as a per-key bias; m, n free)
            kb = None
            if key_mask is not None:
                kb = torch.zeros(key_mask.shape, dtype=torch.float32, device=q.device).masked_fill(~key_mask, -1e30).contiguous()
            o = FlashAttnGenFn.apply(_rows_view(q), _rows_view(k), _rows_view(v), None, None, kb, h, float(scale))
            return _fully_masked_rows(o.view(B, n, h, dh).transpose(1, 2), v, key_mask)
        mp = _round8(m)
        q2 = q.reshape(B * h, n, dh).to(ACT_DTYPE).contiguous()
        k2 = k.reshape(B * h, m, dh).to(ACT_DTYPE)
        v2 = v.reshape(B * h, m, dh).to(ACT_DTYPE)
        if mp != m:
            k2 = F.pad(k2, (0, 0, 0, mp - m))
            v2 = F.pad(v2, (0, 0, 0, mp - m))
        k2, v2 = k2.contiguous(), v2.contiguous()
        # logits = alpha * q.k + bias[batch, key]:
        #   dot: alpha = scale;  l2: -|q-k|^2*scale = 2*scale*q.k - scale*|k|^2 - scale*|q|^2, and the per-query
        #   term cancels in the softmax, so alpha = 2*scale and bias = -scale*|k|^2. Key-padding masks are a bias too.
        alpha = 2.0 * scale if l2 else scale
        bias = None
        if l2:
            bias = -scale * (k2.float() ** 2).sum(-1)                                   # (BH, mp)
        if key_mask is not None:
            km = F.pad(key_mask, (0, mp - m), value=False) if mp != m else key_mask
            neg = torch.zeros(km.shape, dtype=torch.float32, device=q.device).masked_fill(~km, -1e30)
            neg = neg[:, None, :].expand(B, h, mp).reshape(B * h, mp)
            bias = neg if bias is None else bias + neg
        attn = AttnProbsFn.apply(q2, k2, None if bias is None else bias.contiguous(), alpha, m)   # bf16 (BH, n, mp)
        out = GemmFn.apply(attn, v2, True, False, (n, dh, mp), None, None, 1.0, False)  # (BH, n, dh)
        return _fully_masked_rows(out.reshape(B, h, n, dh), v, key_mask)

    def self_attention(self, q, k, v, null_kv, *, heads, scale, l2):
        """SelfAttention.forward after the projections (gp.py:562-592); q, k, v logical (b, heads*d, x, y)."""
        b, c, x, y = q.shape
        n = x * y
        if c == heads * 64 and n % 128 == 0:
            tied = k is q
            qh, kh, vh = (nhwc(to_act(t)).view(b, n, c) for t in (q, k, v))
            alpha, beta = (2.0 * scale, -scale) if l2 else (scale, 0.0)
            o = FlashAttnFn.apply(qh, None if tied else kh, vh, null_kv[0], null_kv[1], heads, alpha, beta)
            return nchw(o.view(b, x, y, c))
        from .modules import self_attention_unfused
        return self_attention_unfused(self, q, k, v, null_kv, heads, scale, l2)

    # -- norms / resampling ------------------------------------------------------------------------
    @staticmethod
    def _sinkable(p):
        """the parameter itself when its gradient may be written behind autograd's back under `ops.sinking()` (fp32, dense), else None"""
        return p if (isinstance(p, torch.nn.Parameter) and p.dtype == torch.float32 and p.is_contiguous()) else None

    def channel_rmsnorm(self, x, gamma, act=None, fork=False):
        """F.normalize(x, dim=1) * sqrt(C) * gamma (gp.py:224-232, unet.py:224-234), fp32 statistics, one fused pass
        over NHWC; `act='silu'` is the unet Block's activation (unet.py:268-269). `fork=True` returns (y, x'): x' goes to the
        skip connection around the normalised branch."""
        x = to_act(x)
        c = x.shape[1]
        if fork:
            if c % 8 or act is not None:
                return self.channel_rmsnorm(x, gamma, act), x
            y, xa = RmsNormFn.apply(nhwc(x), gamma.float().reshape(c).contiguous(), True, False, self._sinkable(gamma))
            return nchw(y), nchw(xa)
        if act == 'silu' and c % 8 == 0 and not second_order:
            return nchw(RmsNormFn.apply(nhwc(x), gamma.float().reshape(c).contiguous(), False, True, self._sinkable(gamma)))
        if c % 8:
            _shape_fallback('channel_rmsnorm')
            xf = x.float()
            nrm = xf.norm(dim=1, keepdim=True).clamp(min=1e-12)
            y = (xf / nrm * (c ** 0.5) * gamma.float().view(1, c, 1, 1)).to(ACT_DTYPE)
        else:
            y = nchw(RmsNormFn.apply(nhwc(x), gamma.float().reshape(c).contiguous(), False, False, self._sinkable(gamma)))
        if act == 'silu':
            y = F.silu(y)
        else:
            assert act is None
        return y

    # -- unet pieces (unet_upsampler.py) -----------------------------------------------------------------
    def maxpool_highfreq(self, x):
        """(max_pool2d(x, 2), x - blur(x)): the unet Downsample's pooled map and the high-frequency map that rides the
        skip connection (unet.py:134-160)."""
        x = to_act(x)
        b, C, H, W = x.shape
        if C % 8 or H % 2 or W % 2 or second_order:       # ragged shapes / twice-differentiated graphs: the tensor-algebra form
            if not second_order:
                _shape_fallback('maxpool_highfreq')
            hf = (x.float() - self.blur(x).float()).to(ACT_DTYPE)
            return F.max_pool2d(x, kernel_size=2), hf
        pool, hf = PoolHighFreqFn.apply(nhwc(x))
        return nchw(pool), nchw(hf)

    def linear_attention_qkv(self, qkv, *, heads, scale):
        """the same on the fused to_qkv output (b, 3C, x, y) (channels: q | k | v): returns (b, C, x, y), or None when the
        fused kernels do not apply (heads of 64 features only, first-order graphs)."""
        b, c3, x, y = qkv.shape
        c = c3 // 3
        if second_order or c % heads or c // heads != 64 or c > 512:
            return None
        out = LinearAttnFn.apply(nhwc(to_act(qkv)).view(b, x * y, c3), heads, float(scale))
        return nchw(out.view(b, x, y, c))

    def linear_attention(self, q, k, v, *, heads, scale):
        """LinearAttention core (unet.py:338-348): q softmax over the head features (times scale), k softmax over the
        positions, context = k v^T (d x e per head), out = context^T q. q, k, v logical (b, heads*d, x, y); both
        contractions on the batched MFMA GEMM, softmaxes in fp32."""
        b, c, x, y = q.shape
        n, d = x * y, c // heads
        if not second_order:
            _shape_fallback('linear_attention')
        qf = q.reshape(b, heads, d, n).float().softmax(dim=2) * scale
        kf = k.reshape(b, heads, d, n).float().softmax(dim=3)
        q2, k2, v2 = (t.reshape(b * heads, d, n).transpose(1, 2).to(ACT_DTYPE).contiguous()      # (BH, n, d)
                      for t in (qf, kf, v.reshape(b, heads, d, n)))
        ctx = GemmFn.apply(k2, v2, False, False, (d, d, n), None, None, 1.0, False)               # (BH, d, e)
        out = GemmFn.apply(q2, ctx, True, False, (n, d, d), None, None, 1.0, False)               # (BH, n, e)
        return out.reshape(b, heads, n, d).permute(0, 1, 3, 2).reshape(b, c, x, y)

    def upsample_blur(self, x):
        """nn.Upsample(x2, bilinear, align_corners=False) then the reflect-padded [1,2,1]^2/16 blur
        (gp.py:246-261), as ONE separable-stencil kernel."""
        x = to_act(x)
        H, W = x.shape[-2:]
        return nchw(ResampleFn.apply(nhwc(x), K.ResampleSpec.upsample_blur(H, W)))

    def blur(self, x):
        x = to_act(x)
        H, W = x.shape[-2:]
        return nchw(ResampleFn.apply(nhwc(x), K.ResampleSpec.blur(H, W)))

    def resize_bilinear(self, x, size):
        """F.interpolate(x, size, mode='bilinear') (align_corners=False, no antialias; gp.py:1683-1687)."""
        H, W = x.shape[-2:]
        size = (size, size) if isinstance(size, int) else tuple(size)
        if (H, W) == size:
            return x
        xa = to_act(x)
        return nchw(ResampleFn.apply(nhwc(xa), K.ResampleSpec.bilinear(H, W, *size)))

    def resize_nearest(self, x, size):
        H, W = x.shape[-2:]
        size = (size, size) if isinstance(size, int) else tuple(size)
        if (H, W) == size:
            return x
        xa = to_act(x)
        return nchw(ResampleFn.apply(nhwc(xa), K.ResampleSpec.nearest(H, W, *size)))


_SPAIR = os.environ.get('GG_SPAIR', '1') != '0'      # A/B switch: 0 runs the 128x128 / 256x256 adaptive convs one launch each (gg_sconv)
_ACONV = True      # (round 6: the GG_ACONV A/B switch is gone; the round-3/4 formulations stay reachable for the shapes gg_aconv does not carry)
# widest image the one-launch kernel takes: measured (profiles/r5_aconv_probe*.log, batch 32, hipGraph-timed incl. the modulation
# launch) 28 / 38 / 54+32 / 48+32 us against 63 / 79 / 69+52 / 51+35 us on 4x4 / 8x8 / 16x16 / 32x32, but 68+52 against 53+38 us at
# 64x64 (1024 small workgroups: four rounds of its fixed cost), which therefore stays on per-sample weights + gg_conv3
_ACONV_MAXW = 32

_aconv_plans: dict = {}


def _aconv_ok(b, N, O, I, H, W) -> bool:
    """can gg_aconv_fwd run this layer? (asked of the library itself: gg_aconv_plan; cached per geometry)"""
    key = (b, N, O, I, H, W)
    ok = _aconv_plans.get(key)
    if ok is None:
        ok = _aconv_plans[key] = (N <= 2 and H == W and W <= _ACONV_MAXW and K.aconv_plan(b, H, W, I, O, N) is not None)
    return ok


def _frag_weight(w: torch.Tensor) -> torch.Tensor:
    """the bank (N, O, I, 3, 3) in MFMA-fragment order: from the model's pack table (re-packed with the other operands by the one
    gg_pack_weights launch behind each optimizer step) or, for parameters outside a table, cached on the parameter per version."""
    cacheable = isinstance(w, torch.nn.Parameter)
    if cacheable and getattr(w, '_gg_pack_table', None) is not None and not _DEBUG_NO_TABLE:
        out = _table_pack(w, 'frag')
        if out is not None:
            return out
    if cacheable:
        slot = getattr(w, '_gg_packed', None)
        if slot is not None and slot[0] == _weight_epoch and slot[2] == w._version and 'frag' in slot[1]:
            return slot[1]['frag']
    with torch.no_grad():
        out = K.frag_pack(w)
    if cacheable:
        slot = getattr(w, '_gg_packed', None)
        if slot is None or slot[0] != _weight_epoch or slot[2] != w._version:
            slot = (_weight_epoch, {}, w._version)
            w._gg_packed = slot
        slot[1]['frag'] = out
    return out


def _rows_f32(t: torch.Tensor) -> torch.Tensor:
    """(b, n) fp32 with unit column stride, without a copy when it already is one (a column slice of a wider fp32 matrix)."""
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()


def _fully_masked_rows(o, v, key_mask):
    """reference semantics where a batch item keeps NO key (gp.py:645-647: `sim.masked_fill(~mask, -finfo.max)` makes every score of
    the row the same number, so its softmax is uniform over ALL keys and the output is the plain mean of the values); the kernels'
    per-key bias gives such a row zeros. No host synchronisation: the rows are selected on the device (two small launches, text
    conditioning only)."""
    if key_mask is None:
        return o
    keep = key_mask.any(dim=-1)[:, None, None, None]                     # (B, 1, 1, 1)
    mean_v = v.float().mean(dim=2, keepdim=True).to(o.dtype)            # (B, h, 1, dh)
    return torch.where(keep, o, mean_v.expand_as(o))


def _wmix_buffer(weights, b: int, I: int):
    """the per-sample filter banks of one layer for gg_sconv_fwd, (b, 9, I/16, 32, 16) bf16: persistent per weight tensor (rows
    beyond O stay zero; a captured hipGraph keeps pointing at it), re-filled by gg_modw_fwd on every forward."""
    slot = weights.__dict__.setdefault('_gg_wmix', {}) if isinstance(weights, torch.nn.Parameter) else {}
    buf = slot.get(b)
    if buf is None or buf.device != weights.device:
        buf = slot[b] = torch.zeros((b, 9, I // 16, 32, 16), dtype=ACT_DTYPE, device=weights.device)
    return buf


def _wmix_rows(weights, b: int, O: int, Kw: int):
    """the per-sample weight operands of one layer for the implicit GEMM, (b, O, 9*I) bf16, persistent per weight tensor."""
    slot = weights.__dict__.setdefault('_gg_wmix', {}) if isinstance(weights, torch.nn.Parameter) else {}
    buf = slot.get(('rows', b))
    if buf is None or buf.device != weights.device:
        buf = slot[('rows', b)] = torch.empty((b, O, Kw), dtype=ACT_DTYPE, device=weights.device)
    return buf


def demod_coefficients(weights, s, a, eps):
    """d[b,o] = rsqrt(max(sum_{i,k} (sum_n a[b,n] W[n,o,i,k] s[b,i])^2, eps)) without materialising the
    per-sample weights (gp.py:390-400): a Gram matrix over the kernel bank, contracted with s^2 and a a^T.
    fp32 throughout (these are (b,O)-sized statistics)."""
    N, O, I = weights.shape[:3]
    if not second_order:
        _shape_fallback('demod_coefficients')
    wf = weights.float().flatten(3)                                     # (N, O, I, k*k)
    gram = (wf[:, None] * wf[None, :]).sum(-1)                          # (N, N, O, I) — pointwise, no tiny GEMMs
    t = ((s * s) @ gram.reshape(N * N * O, I).t()).view(-1, N, N, O)    # (b, N, N, O): one (b x I)(I x N^2 O) GEMM
    aa = a[:, :, None] * a[:, None, :]                                  # (b, N, N)
    sumsq = (aa[..., None] * t).sum(dim=(1, 2))
    return sumsq.clamp(min=eps).rsqrt()


def fused_modconv_forward(xh, wts, s, a, d, noise, noise_weight, act, O, Op, d_padded=False, wk=None):
    """no-grad path: the whole adaptive conv (kernel mix, modulation, demodulation, noise, leaky-relu) as
    ONE implicit-GEMM launch with the N kernels stacked along the reduction and batch folded into M. `wk`: the
    pre-packed (Op, k*k*N*Ip) [co][tap][n][ci] operand (pack table), else it is built from `wts` here."""
    b, H, W, Ip = xh.shape
    N, _, _, k, _ = wts.shape
    insc = (a[:, :, None] * s[:, None, :]).reshape(b, N * Ip).contiguous()
    if wk is None:
        wk = wts.permute(1, 3, 4, 0, 2).reshape(Op, k * k * N * Ip).to(ACT_DTYPE).contiguous()
    out_scale = None
    if d is not None:
        out_scale = (F.pad(d, (0, Op - O)) if (Op != O and not d_padded) else d).contiguous()
    nz = nw = None
    if noise is not None:
        nz = noise.reshape(-1).float().contiguous()
        nw = noise_weight.reshape(-1).float()
        nw = (F.pad(nw, (0, Op - O)) if Op != O else nw).contiguous()
    if k == 3 and b * H * W <= 131072:
        # the per-sample scale a_n * s applied to the activation by one pointwise pass per kernel of the bank (channels laid
        # out (n, ci) like the packed reduction), so the convolution is the plain gather of the discriminator's layers - the
        # in-gather scale costs two extra loads and a wait per staged vector. Worth it where the activation is small next to
        # the weights (<= 64x64). Measured (profiles/r02_modconv_ab.log): 8x8 90 -> 71 us, 16x16 138 -> 122 us, 32x32
        # 137 -> 111 us, 64x64 154 -> 120 us / 96 -> 69 us
        x2 = torch.cat([K.modulate(xh, insc[:, n * Ip:(n + 1) * Ip].contiguous()) for n in range(N)], dim=-1) if N > 1 \
            else K.modulate(xh, insc)
        return K.conv2d_nhwc(x2, wk, ksize=k, out_scale=out_scale, noise=nz, noise_w=nw, act=act, act_slope=LRELU_SLOPE)
    return K.conv2d_nhwc(xh, wk, ksize=k, cv=N * Ip, in_scale=insc, out_scale=out_scale, noise=nz, noise_w=nw,
                         act=act, act_slope=LRELU_SLOPE)


impl = HipOps()


@contextmanager
def use_impl(new_impl):
    """Swap the op implementation (tests only: installs the CPU oracle to check model assembly)."""
    global impl
    old = impl
    impl = new_impl
    try:
        yield
    finally:
        impl = old


def get_impl():
    return impl
