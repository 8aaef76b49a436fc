"""Exponential moving average of the generator — the semantics the reference gets from `ema_pytorch.EMA`
(gp.py:2173-2185 `EMA(G, update_every=10, update_after_step=100, beta=0.995)`, updated at gp.py:2603): copy the online
weights until `update_after_step`, then every `update_every` calls lerp towards them with weight 1 - decay, where the
decay warms up as ema_pytorch does: decay(step) = clamp(1 - (1 + epoch / inv_gamma) ** -power, min_value, beta) with
epoch = step - update_after_step - 1 evaluated AFTER the step counter was incremented (inv_gamma 1, power 2/3: 0.80 at
the first averaged update, `beta` reached after ~2.8k steps for beta = 0.995). ema_pytorch is not installed in this image
and is not vendored by the reference, so this arithmetic is restated from its published source: **parity unpinned**
(tests/oracle_stubs/ema_pytorch restates the same formula).

State-dict layout follows ema_pytorch with `include_online_model=True` (its default): `ema_model.*`, `online_model.*`
(aliases of the live generator), `initted`, `step` — the reference loads `G_ema` strictly (gp.py:2092).

When the online generator's parameters live in a FlatAdamW buffer, the EMA copy is flattened with the same layout and
the update is ONE HIP launch (`gg_ema_flat_f32`). The counters live in the `step` / `initted` buffers (they travel with
checkpoints); host mirrors avoid a device sync per update and are re-read from the buffers after every state-dict load.
"""
from __future__ import annotations

from collections import OrderedDict
from copy import deepcopy

import torch
from torch import nn

from . import _C
from ._C import ptr


class EMA(nn.Module):
    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1.0, power=2 / 3,
                 min_value=0.0, **_ignored):
        super().__init__()
        self._online = [model]          # not a registered submodule: its keys are emitted by state_dict() below
        self.ema_model = deepcopy(model)
        for p in self.ema_model.parameters():
            p.requires_grad_(False)
        self.beta = beta
        self.update_after_step = update_after_step
        self.update_every = update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.register_buffer('initted', torch.tensor(False))
        self.register_buffer('step', torch.tensor(0))
        self._step_host = 0
        self._initted_host = False
        self._flat = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.sync_host_counters())

    @property
    def online_model(self):
        return self._online[0]

    def sync_host_counters(self):
        """after anything wrote the `step` / `initted` buffers (checkpoint load): one device read, not one per update."""
        self._step_host = int(self.step.item())
        self._initted_host = bool(self.initted.item())

    # -- ema_pytorch's key layout --------------------------------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        out = OrderedDict()
        for k, v in self.online_model.state_dict(prefix=prefix + 'online_model.', keep_vars=keep_vars).items():
            out[k] = v
        for k, v in sd.items():
            out[k] = v
        if hasattr(sd, '_metadata'):
            out._metadata = sd._metadata
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        own = {k: v for k, v in state_dict.items() if not k.startswith('online_model.')}   # the live generator loads its own
        return super().load_state_dict(own, strict=strict, assign=assign)

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

    def get_current_decay(self, step_after_increment: int | None = None) -> float:
        step = self._step_host if step_after_increment is None else step_after_increment
        epoch = max(step - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.
        value = 1. - (1. + epoch / self.inv_gamma) ** (-self.power)
        return min(max(value, self.min_value), self.beta)

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
        decay = self.get_current_decay()
        if self._flat is not None:
            L = _C.lib()
            rc = L.lib.gg_ema_flat_f32(ptr(self._flat), ptr(self._flat_online), self._flat.numel(), 1. - decay,
                                       L.stream(self._flat))
            L.check(rc, 'gg_ema_flat_f32')
        else:
            for pe, po in zip(self.ema_model.parameters(), self.online_model.parameters()):
                pe.lerp_(po, 1. - decay)

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
