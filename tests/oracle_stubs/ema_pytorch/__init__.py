"""Stub of ema_pytorch.EMA with the semantics the reference relies on (gp.py:2173-2185, :2603)."""
from copy import deepcopy
import torch
from torch import nn


class EMA(nn.Module):
    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, **_kw):
        super().__init__()
        self.online_model = [model]  # not registered: mirrors ema_pytorch keeping it out of state_dict
        self.ema_model = deepcopy(model)
        for p in self.ema_model.parameters():
            p.requires_grad_(False)
        self.beta = beta
        self.update_after_step = update_after_step
        self.update_every = update_every
        self.register_buffer('initted', torch.tensor(False))
        self.register_buffer('step', torch.tensor(0))

    @torch.no_grad()
    def copy_params_from_model_to_ema(self):
        for pe, pm in zip(self.ema_model.parameters(), self.online_model[0].parameters()):
            pe.copy_(pm)
        for be, bm in zip(self.ema_model.buffers(), self.online_model[0].buffers()):
            be.copy_(bm)

    @torch.no_grad()
    def update(self):
        step = int(self.step.item())
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not bool(self.initted.item()):
            self.copy_params_from_model_to_ema()
            self.initted.fill_(True)
        for pe, pm in zip(self.ema_model.parameters(), self.online_model[0].parameters()):
            pe.lerp_(pm, 1. - self.beta)
        for be, bm in zip(self.ema_model.buffers(), self.online_model[0].buffers()):
            if be.is_floating_point():
                be.lerp_(bm, 1. - self.beta)
            else:
                be.copy_(bm)

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
