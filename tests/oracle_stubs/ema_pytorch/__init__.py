"""Stub of ema_pytorch.EMA with the semantics the reference relies on (gp.py:2173-2185, :2603), restated from the package's
published source (it is not installed here: parity unpinned): online model registered as a submodule (`include_online_model`
defaults to True, so `online_model.*` keys are in the state dict), warm-up decay with inv_gamma 1 / power 2/3."""
from copy import deepcopy
import torch
from torch import nn


class EMA(nn.Module):
    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1.0, power=2 / 3,
                 min_value=0.0, **_kw):
        super().__init__()
        self.online_model = model
        self.ema_model = deepcopy(model)
        for p in self.ema_model.parameters():
            p.requires_grad_(False)
        self.beta = beta
        self.update_after_step = update_after_step
        self.update_every = update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.register_buffer('initted', torch.tensor(False))
        self.register_buffer('step', torch.tensor(0))

    @torch.no_grad()
    def copy_params_from_model_to_ema(self):
        for pe, pm in zip(self.ema_model.parameters(), self.online_model.parameters()):
            pe.copy_(pm)
        for be, bm in zip(self.ema_model.buffers(), self.online_model.buffers()):
            be.copy_(bm)

    def get_current_decay(self):
        epoch = (self.step - self.update_after_step - 1).clamp(min=0.)
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        if epoch.item() <= 0:
            return 0.
        return value.clamp(min=self.min_value, max=self.beta).item()

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
        decay = self.get_current_decay()
        for pe, pm in zip(self.ema_model.parameters(), self.online_model.parameters()):
            pe.lerp_(pm, 1. - decay)
        for be, bm in zip(self.ema_model.buffers(), self.online_model.buffers()):
            if be.is_floating_point():
                be.lerp_(bm, 1. - decay)
            else:
                be.copy_(bm)

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
