"""Datasets (reference data.py:48-113). The benchmarked path uses synthetic batches already resident in HBM;
these are host-side conveniences with the reference's names and item formats."""
from __future__ import annotations

from pathlib import Path

import torch
from torch.utils.data import Dataset, DataLoader


def collate_tensors_or_str(data):
    """reference data.py:28-44: stack tensors, keep strings as lists."""
    is_one = not isinstance(data[0], tuple)
    if is_one:
        return torch.stack(data)
    outs = []
    for column in zip(*data):
        outs.append(torch.stack(column) if torch.is_tensor(column[0]) else list(column))
    return tuple(outs)


class ImageDataset(Dataset):
    """folder of images -> float tensors in [0,1], resized + center-cropped to image_size (data.py:48-85)."""

    def __init__(self, folder, image_size, channels=3, convert_image_to=None, exts=('jpg', 'jpeg', 'png', 'tiff')):
        super().__init__()
        folder = Path(folder)
        assert folder.is_dir(), f'{folder} must be a folder containing images'
        self.paths = [p for ext in exts for p in folder.glob(f'**/*.{ext}')]
        assert len(self.paths) > 0, 'your folder contains no images'
        self.image_size = image_size
        self.mode = convert_image_to or {1: 'L', 3: 'RGB', 4: 'RGBA'}[channels]

    def get_dataloader(self, *args, **kwargs):
        kwargs.setdefault('collate_fn', collate_tensors_or_str)
        return DataLoader(self, *args, **kwargs)

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        from PIL import Image
        import numpy as np
        img = Image.open(self.paths[index]).convert(self.mode)
        w, h = img.size
        s = self.image_size / min(w, h)
        img = img.resize((max(round(w * s), self.image_size), max(round(h * s), self.image_size)), Image.BILINEAR)
        w, h = img.size
        l, t = (w - self.image_size) // 2, (h - self.image_size) // 2
        img = img.crop((l, t, l + self.image_size, t + self.image_size))
        arr = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
        if arr.dim() == 2:
            arr = arr[..., None]
        return arr.permute(2, 0, 1).float() / 255.


class TextImageDataset(Dataset):
    def __init__(self):
        raise NotImplementedError   # same as the reference (data.py:88-89)


class MockTextImageDataset(TextImageDataset):
    """(randn image, 'mock text') pairs (data.py:94-113)."""

    def __init__(self, image_size, length=int(1e5), channels=3):
        self.image_size, self.channels, self.length = image_size, channels, length

    def get_dataloader(self, *args, **kwargs):
        kwargs.setdefault('collate_fn', collate_tensors_or_str)
        return DataLoader(self, *args, **kwargs)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        return torch.randn(self.channels, self.image_size, self.image_size), 'mock text'


class SyntheticImages:
    """an endless in-memory 'dataloader' of uniform [0,1) images already on the device (bench / smoke)."""

    def __init__(self, batch_size, image_size, channels=3, device='cpu', seed=0, n_batches=2):
        g = torch.Generator().manual_seed(seed)
        self.batch_size = batch_size
        self.batches = [torch.rand(batch_size, channels, image_size, image_size, generator=g).to(device)
                        for _ in range(n_batches)]

    def __iter__(self):
        return iter(self.batches)
