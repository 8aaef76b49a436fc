"""Exponential moving average of the generator — the semantics the reference gets from `ema_pytorch.EMA`
(gp.py:2173-2185, updated at gp.py:2603): copy the online weights until `update_after_step`, then every
`update_every` calls lerp towards them with weight 1-beta. When the online generator's parameters live in a
FlatAdamW buffer, the EMA copy is flattened with the same layout and the update is ONE HIP launch.
"""
from __future__ import annotations

from copy import deepcopy

import torch
from torch import nn

from . import _C
from ._C import ptr


class EMA(nn.Module):
    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, **_ignored):
        super().__init__()
        self._online = [model]
        self.ema_model = deepcopy(model)
        for p in self.ema_model.parameters():
            p.requires_grad_(False)
        self.beta = beta
        self.update_after_step = update_after_step
        self.update_every = update_every
        self.register_buffer('initted', torch.tensor(False))
        self.register_buffer('step', torch.tensor(0))
        self._step_host = 0
        self._initted_host = False
        self._flat = None

    @property
    def online_model(self):
        return self._online[0]

    def attach_flat(self, flat_p: torch.Tensor, params, offsets):
        """share FlatAdamW's layout so the lerp is a single launch over one buffer."""
        self._flat_online = flat_p
        self._flat = torch.zeros_like(flat_p)
        named_online = {id(p): i for i, p in enumerate(params)}
        ema_params = list(self.ema_model.parameters())
        online_params = list(self.online_model.parameters())
        for pe, po in zip(ema_params, online_params):
            if id(po) not in named_online:
                continue
            off = offsets[named_online[id(po)]]
            n = po.numel()
            self._flat[off:off + n].copy_(pe.detach().reshape(-1))
            pe.data = self._flat[off:off + n].view(pe.shape)

    @torch.no_grad()
    def copy_params_from_model_to_ema(self):
        if self._flat is not None:
            self._flat.copy_(self._flat_online)
        else:
            for pe, po in zip(self.ema_model.parameters(), self.online_model.parameters()):
                pe.copy_(po)
        for be, bo in zip(self.ema_model.buffers(), self.online_model.buffers()):
            be.copy_(bo)

    @torch.no_grad()
    def update(self):
        step = self._step_host
        self._step_host += 1
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not self._initted_host:
            self.copy_params_from_model_to_ema()
            self._initted_host = True
            self.initted.fill_(True)
        if self._flat is not None:
            L = _C.lib()
            rc = L.lib.gg_ema_flat_f32(ptr(self._flat), ptr(self._flat_online), self._flat.numel(), 1. - self.beta,
                                       L.stream(self._flat))
            L.check(rc, 'gg_ema_flat_f32')
        else:
            for pe, po in zip(self.ema_model.parameters(), self.online_model.parameters()):
                pe.lerp_(po, 1. - self.beta)

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
