"""Flat-buffer fused AdamW — the MI355X counterpart of reference optimizer.py:10-34 (`get_optimizer`).

Semantics kept bit-for-bit in intent: `GigaGAN` passes `weight_decay=` which the
reference's `get_optimizer` swallows in **kwargs, so the effective optimizer is AdamW(wd=1e-2) with decay on
every parameter with ndim >= 2 and none on the rest; parameters that never receive a gradient are skipped
entirely (no moment update, no decay).

Layout: all parameters of a model are re-homed as views into ONE flat fp32 buffer (each parameter starts
on a 256-element boundary), and so are their `.grad`s and both Adam moments. One HIP launch
(`gg_adamw_flat_f32`, 28 B/param of HBM traffic) steps the model, and the same flat gradient buffer is
what RCCL all-reduces for data parallelism (distributed.py) — no per-tensor launches, no bucketing copies.
"""
from __future__ import annotations

import math

import torch

from . import _C
from ._C import ptr

ALIGN = 256


def separate_weight_decayable_params(params):
    wd, no_wd = [], []
    for p in params:
        (no_wd if p.ndim < 2 else wd).append(p)
    return wd, no_wd


class FlatAdamW(torch.optim.Optimizer):
    """AdamW over flat buffers; `inactive` parameters are never stepped (their grad is None in the reference)."""

    def __init__(self, params, lr=1e-4, wd=1e-2, betas=(0.9, 0.99), eps=1e-8, inactive=()):
        params = [p for p in params]
        wd_params, no_wd_params = separate_weight_decayable_params(params)
        groups = [{'params': wd_params, 'weight_decay': wd}, {'params': no_wd_params, 'weight_decay': 0.}]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=wd))
        self.wd = wd
        self._inactive = {id(p) for p in inactive}
        self._all = params
        self.step_count = 0
        self.scaler = None   # reference checks `G_opt.scaler` (accelerate); bf16 training has none
        self._build()

    # -- flat storage ------------------------------------------------------------------------------
    def _build(self):
        device = self._all[0].device
        offs, total = [], 0
        for p in self._all:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = total
        self.offsets = offs
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=device)
        flags = torch.zeros(total // ALIGN, dtype=torch.uint8)
        decay_ids = {id(p) for p in self.param_groups[0]['params']}
        for p, off in zip(self._all, offs):
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)
            p.grad = self.flat_g[off:off + n].view(p.shape)
            p._gg_slot = (p.data_ptr(), (n + ALIGN - 1) // ALIGN * ALIGN)     # (the slot's tail beyond n stays zero: ops._bias8)
            fl = (0 if id(p) in self._inactive else 1) | (2 if id(p) in decay_ids else 0)
            flags[off // ALIGN:(off + n + ALIGN - 1) // ALIGN] = fl
            st = self.state[p]
            st['step'] = torch.tensor(0.)
            st['exp_avg'] = self.flat_m[off:off + n].view(p.shape)
            st['exp_avg_sq'] = self.flat_v[off:off + n].view(p.shape)
        self.flags = flags.to(device)
        self._flags_host = flags
        # INVARIANT: the tail of every parameter's 256-float slot (the floats between its last element and the next slot) is zero.
        # ops._bias8 hands the kernels a view that runs into that tail (a ragged bias read as its zero-padded 8-multiple). It holds by
        # construction - the buffers start as zeros and AdamW with g = m = v = 0 writes 0 - and is re-established after every write
        # that does not go through the optimizer (zero_slot_tails: snapshot restore, state-dict loads)
        tails = [torch.arange(off + p.numel(), off + (p.numel() + ALIGN - 1) // ALIGN * ALIGN) for p, off in zip(self._all, offs)
                 if p.numel() % ALIGN]
        self._tail_idx = (torch.cat(tails) if tails else torch.zeros(0, dtype=torch.long)).to(device)
        self._flag_variants = {}
        # conv weights: persistent bf16 GEMM operands, re-packed by one launch per epoch (ops._table_pack)
        from .kernels import PackTable
        self.pack_table = PackTable(device, capacity=3 * len(self._all) + 16)
        self.pack_table.dirty = True
        from . import ops
        ops._pack_tables.add(self.pack_table)
        for p in self._all:
            if p.ndim >= 4 or p.ndim == 2:      # conv weights / kernel banks; linear weights (ops.LinearFn)
                p._gg_pack_table = self.pack_table

    @torch.no_grad()
    def zero_slot_tails(self):
        """re-establish the slot-tail invariant (see _build) after a write that bypassed the optimizer."""
        if self._tail_idx.numel():
            for buf in (self.flat_p, self.flat_m, self.flat_v):
                buf.index_fill_(0, self._tail_idx, 0.)

    def slot_tails_are_zero(self) -> bool:
        return self._tail_idx.numel() == 0 or not bool(self.flat_p.index_select(0, self._tail_idx).ne(0).any())

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        for p, off in zip(self._all, self.offsets):   # re-attach if user code detached the views
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * 4:
                p.grad = self.flat_g[off:off + p.numel()].view(p.shape)

    def flags_without(self, params):
        """the per-256-block flag table with `params` switched off for one step (their gradient is None in the reference:
        torch optimizers then skip them entirely - no moment update, no weight decay). Cached per parameter set."""
        key = frozenset(id(p) for p in params)
        if not key:
            return self.flags
        fl = self._flag_variants.get(key)
        if fl is None:
            host = self._flags_host.clone()
            for p, off in zip(self._all, self.offsets):
                if id(p) in key:
                    host[off // ALIGN:(off + p.numel() + ALIGN - 1) // ALIGN] &= 0xFE
            fl = self._flag_variants[key] = host.to(self.flags.device)
        return fl

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, skip=()):
        g = self.param_groups[0]
        lr, (b1, b2), eps = g['lr'], g['betas'], g['eps']
        self.step_count += 1
        t = self.step_count
        L = _C.lib()
        if self.flat_p.device.type == 'cpu' and not L.is_emulator:
            raise RuntimeError('FlatAdamW: parameters are on the CPU; the fused optimizer runs on the GPU only')
        L.require(self.flat_p)
        rc = L.lib.gg_adamw_flat_f32(ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v),
                                     ptr(self.flags_without(skip)), self.total, lr, b1, b2, eps, self.wd,
                                     1. - b1 ** t, math.sqrt(1. - b2 ** t), grad_scale, L.stream(self.flat_p))
        L.check(rc, 'gg_adamw_flat_f32')
        from . import ops
        ops.pack_cache_clear()       # per-parameter cached packs are stale now ...
        self.pack_table.refresh()    # ... and the persistent GEMM operands of this model are re-packed right here
        self.pack_table.dirty = False
        self._steps_dirty = True

    def state_dict(self):
        # per-parameter `step` tensors (torch AdamW layout) are only materialised when somebody looks at them
        if getattr(self, '_steps_dirty', False):
            for p in self._all:
                if id(p) not in self._inactive:
                    self.state[p]['step'] = torch.tensor(float(self.step_count))
            self._steps_dirty = False
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """accept a torch AdamW state dict (reference checkpoints) and copy it into the flat buffers."""
        saved = state_dict['state']
        order = [p for grp in self.param_groups for p in grp['params']]
        steps = []
        for idx, p in enumerate(order):
            if idx in saved:
                s = saved[idx]
                self.state[p]['exp_avg'].copy_(s['exp_avg'])
                self.state[p]['exp_avg_sq'].copy_(s['exp_avg_sq'])
                steps.append(int(float(s['step'])))
        if steps:
            self.step_count = max(steps)
        for grp, sg in zip(self.param_groups, state_dict['param_groups']):
            for k in ('lr', 'betas', 'eps'):
                if k in sg:
                    grp[k] = sg[k]
        self.zero_slot_tails()


def get_optimizer(params, lr=1e-4, wd=1e-2, betas=(0.9, 0.99), eps=1e-8, filter_by_requires_grad=True,
                  group_wd_params=True, inactive=(), **kwargs):
    """Reference signature (optimizer.py:10-19); unknown kwargs (e.g. `weight_decay`) are ignored exactly as
    the reference ignores them."""
    params = list(params)
    if filter_by_requires_grad:
        params = [p for p in params if p.requires_grad]
    return FlatAdamW(params, lr=lr, wd=wd, betas=betas, eps=eps, inactive=inactive)
