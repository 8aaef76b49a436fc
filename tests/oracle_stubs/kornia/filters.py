"""Stub of kornia.filters.filter2d / filter3d: depthwise correlation with border padding.

Restates kornia's documented semantics: `normalized=True` divides the kernel by sum(|k|); the input is
padded by (k-1)//2 per side with `border_type` ('reflect' default for 2d, 'replicate' for 3d); the same
kernel is applied to every channel.
"""
import torch
import torch.nn.functional as F


def _norm(kernel):
    return kernel / kernel.abs().sum(dim=tuple(range(1, kernel.ndim)), keepdim=True)


def filter2d(x, kernel, border_type='reflect', normalized=False, padding='same'):
    if normalized:
        kernel = _norm(kernel)
    b, c, h, w = x.shape
    kh, kw = kernel.shape[-2:]
    k = kernel.to(x)[:, None].expand(-1, c, -1, -1).reshape(-1, 1, kh, kw)
    if k.shape[0] != c:
        k = k[:1].expand(c, -1, -1, -1)
    xp = F.pad(x, (kw // 2, (kw - 1) // 2, kh // 2, (kh - 1) // 2), mode=border_type)
    return F.conv2d(xp, k, groups=c)


def filter3d(x, kernel, border_type='replicate', normalized=False):
    if normalized:
        kernel = _norm(kernel)
    b, c, d, h, w = x.shape
    kd, kh, kw = kernel.shape[-3:]
    k = kernel.to(x)[:1, None].expand(c, 1, -1, -1, -1)
    xp = F.pad(x, (kw // 2, (kw - 1) // 2, kh // 2, (kh - 1) // 2, kd // 2, (kd - 1) // 2), mode=border_type)
    return F.conv3d(xp, k, groups=c)
