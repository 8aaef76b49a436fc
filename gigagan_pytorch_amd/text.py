"""TextEncoder (gp.py:808-867): learned transformer over frozen CLIP token encodings.

CLIP itself is an external frozen encoder (reference open_clip.py:17-158) and is out of scope for the
MI355X kernels: pass an adapter object exposing `dim_latent` and `embed_texts(texts) ->
(text_embeds, text_encodings)`, or feed pre-computed `text_encodings` (b, 77, dim_latent).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .modules import Linear, Transformer, exists


class TextEncoder(nn.Module):
    def __init__(self, *, dim, depth, clip=None, dim_head=64, heads=8, clip_dim_latent=512):
        super().__init__()
        self.dim = dim
        self.clip = clip
        dim_latent = clip.dim_latent if exists(clip) else clip_dim_latent
        if exists(clip) and isinstance(clip, nn.Module):
            for p in clip.parameters():
                p.requires_grad = False
        self.learned_global_token = nn.Parameter(torch.randn(dim))
        self.project_in = Linear(dim_latent, dim) if dim_latent != dim else nn.Identity()
        self.transformer = Transformer(dim=dim, depth=depth, dim_head=dim_head, heads=heads)

    def forward(self, texts=None, text_encodings=None):
        assert exists(texts) ^ exists(text_encodings)
        if not exists(text_encodings):
            if not exists(self.clip):
                raise RuntimeError('TextEncoder: no CLIP adapter attached; pass pre-computed text_encodings')
            with torch.no_grad():
                _, text_encodings = self.clip.embed_texts(texts)
        mask = (text_encodings != 0.).any(dim=-1)
        x = self.project_in(text_encodings)
        mask_with_global = F.pad(mask, (1, 0), value=True)
        b = x.shape[0]
        g = self.learned_global_token[None, None, :].expand(b, -1, -1).to(x.dtype)
        x = torch.cat((g, x), dim=1)
        x = self.transformer(x, mask=mask_with_global)
        return x[:, 0], x[:, 1:], mask
