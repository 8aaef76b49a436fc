"""Datasets and the input pipeline (reference data.py:48-113; the reference hands its DataLoader to `accelerator.prepare`,
gp.py:2150-2159, which shards it across ranks). The benchmarked path uses synthetic batches already resident in HBM; real
data goes folder -> worker processes -> pinned host batches -> a copy stream (`DevicePrefetcher`), so the 25 MB fp32 batch of
config 2 crosses PCIe while the previous step computes."""
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

    def __init__(self, folder, image_size, exts=('jpg', 'jpeg', 'png', 'tiff'), augment_horizontal_flip=False,
                 convert_image_to=None, channels=3):
        super().__init__()
        folder = Path(folder)
        assert folder.is_dir(), f'{folder} must be a folder containing images'
        self.paths = [p for ext in exts for p in folder.glob(f'**/*.{ext}')]
        assert len(self.paths) > 0, 'your folder contains no images'
        self.image_size = image_size
        self.augment_horizontal_flip = augment_horizontal_flip       # T.RandomHorizontalFlip(), data.py:66
        self.mode = convert_image_to or {1: 'L', 3: 'RGB', 4: 'RGBA'}[channels]

    def get_dataloader(self, *args, **kwargs):
        """reference data.py:76-77: shuffled, ragged last batch dropped (every step then has one shape: one hipGraph)."""
        kwargs.setdefault('collate_fn', collate_tensors_or_str)
        kwargs.setdefault('shuffle', 'sampler' not in kwargs)
        kwargs.setdefault('drop_last', True)
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
        if self.augment_horizontal_flip and torch.rand(()) < 0.5:
            arr = arr.flip(1)
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


def shard_dataloader(dl, rank: int, world: int, seed: int = 0):
    """what `accelerator.prepare(dl)` does for the reference (gp.py:2150-2161): every rank iterates a disjoint shard.
    * a torch DataLoader with a stock sampler over a map-style dataset is rebuilt around a DistributedSampler (same batch size /
      workers / collate), re-seeded every epoch;
    * a DataLoader that carries a custom sampler or batch_sampler keeps ITS sampling scheme: the batch stream it defines is dealt
      out round-robin, rank r taking batches r, r + world, ... (accelerate's BatchSamplerShard with even_batches: an incomplete last
      round is completed from the start of the epoch). The scheme must produce the same stream on every rank (seed its generator
      identically), which is what accelerate requires as well;
    * a DataLoader without automatic batching (`batch_size=None`, no batch_sampler: the dataset yields ready batches) is sharded the
      same way over its sampler's indices;
    * anything that is not a torch DataLoader over a sized dataset (an iterable of ready batches, an IterableDataset) cannot be
      re-dealt from here: under data parallelism the caller must hand over a per-rank stream and say so (`dl.is_rank_sharded =
      True`), otherwise this raises instead of silently training every rank on identical data."""
    if world <= 1:
        return dl
    if not isinstance(dl, DataLoader) or not hasattr(dl.dataset, '__len__') or isinstance(dl.dataset, torch.utils.data.IterableDataset):
        if getattr(dl, 'is_rank_sharded', False) or isinstance(dl, (SyntheticImages, EpochShardedLoader)):
            return dl
        raise ValueError('shard_dataloader: under data parallelism (world size %d) a loader that is not a torch DataLoader over a '
                         'sized map-style dataset cannot be sharded here - every rank would train on identical batches. Build a '
                         'per-rank stream and set `dl.is_rank_sharded = True` on it.' % world)
    from torch.utils.data.distributed import DistributedSampler
    common = dict(num_workers=dl.num_workers, collate_fn=dl.collate_fn, pin_memory=dl.pin_memory,
                  persistent_workers=getattr(dl, 'persistent_workers', False) and dl.num_workers > 0)
    stock = (torch.utils.data.SequentialSampler, torch.utils.data.RandomSampler)
    # the round-robin paths below iterate the caller's sampler in full on every rank: a RANDOM sampler that draws from the process-global
    # generator (`generator=None`: RandomSampler, WeightedRandomSampler, SubsetRandomSampler) would give every rank its own stream and
    # the shards would overlap. accelerate synchronises the sampler RNG across ranks; here every rank installs the same seeded generator
    # (`seed` is the trainer's, identical on all ranks; the generator's state carries over the epochs, identically everywhere)
    for smp in (dl.sampler, getattr(dl.batch_sampler, 'sampler', None)):
        if smp is not None and hasattr(smp, 'generator') and smp.generator is None and not isinstance(smp, torch.utils.data.SequentialSampler):
            smp.generator = torch.Generator().manual_seed(seed)
    if dl.batch_sampler is None:            # automatic batching off: the sampler's indices ARE the batches
        return DataLoader(dl.dataset, batch_size=None, sampler=RoundRobinShard(dl.sampler, rank, world), **common)
    stock_batches = type(dl.batch_sampler) is torch.utils.data.BatchSampler and dl.batch_size is not None
    if not (stock_batches and isinstance(dl.sampler, stock)):
        sharded = DataLoader(dl.dataset, batch_sampler=RoundRobinShard(dl.batch_sampler, rank, world), **common)
        sharded.is_rank_sharded = True
        return sharded
    shuffle = not isinstance(dl.sampler, torch.utils.data.SequentialSampler)
    sampler = DistributedSampler(dl.dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed, drop_last=True)
    inner = DataLoader(dl.dataset, batch_size=dl.batch_size, sampler=sampler, drop_last=True, **common)
    return EpochShardedLoader(inner, sampler)


class RoundRobinShard(torch.utils.data.Sampler):
    """rank r's share of the stream another sampler / batch sampler defines: items r, r + world, r + 2 world, ... (what
    accelerate's BatchSamplerShard does with `even_batches=True`, which the reference relies on through `accelerator.prepare`,
    gp.py:2161): every rank yields the same number of items per epoch; an incomplete last round is filled from the epoch's first
    items. The wrapped sampler is iterated in full on every rank, so it must be deterministic across ranks."""

    def __init__(self, inner, rank: int, world: int):
        assert 0 <= rank < world
        self.inner, self.rank, self.world = inner, rank, world

    def __iter__(self):
        head, group = [], []
        for item in self.inner:
            if len(head) < self.world:
                head.append(item)
            group.append(item)
            if len(group) == self.world:
                yield group[self.rank]
                group = []
        if group:
            i = 0
            while len(group) < self.world:
                group.append(head[i % len(head)])
                i += 1
            yield group[self.rank]

    def __len__(self):
        return (len(self.inner) + self.world - 1) // self.world


class EpochShardedLoader:
    """a per-rank DataLoader whose DistributedSampler is re-seeded at the start of every pass (`sampler.set_epoch`): the trainer
    iterates through `cycle(dl)`, and without this every epoch would replay the first epoch's permutation on every rank
    (accelerate's prepared loader reseeds per epoch, gp.py:2150-2159)."""

    def __init__(self, loader, sampler):
        self.loader, self.sampler, self.epoch = loader, sampler, 0
        self.batch_size, self.dataset = loader.batch_size, loader.dataset

    def __iter__(self):
        self.sampler.set_epoch(self.epoch)
        self.epoch += 1
        return iter(self.loader)

    def __len__(self):
        return len(self.loader)


def _record_stream(item, stream):
    if torch.is_tensor(item):
        if item.is_cuda:
            item.record_stream(stream)
    elif isinstance(item, (tuple, list)):
        for x in item:
            _record_stream(x, stream)


class DevicePrefetcher:
    """wraps an iterable of host batches (tensors, or tuples / lists holding tensors next to captions): the NEXT batch is pinned
    and copied to the device on a side stream while the current one is consumed; `__next__` makes the compute stream wait on
    that copy's event, never on the host. Keeps `batch_size` for the trainer (gp.py:2668)."""

    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, depth)
        self.batch_size = getattr(loader, 'batch_size', None)

    def _to_device(self, item, stream):
        if torch.is_tensor(item):
            if self.device.type != 'cuda':
                return item.to(self.device)
            if not item.is_pinned():
                item = item.pin_memory()
            with torch.cuda.stream(stream):
                return item.to(self.device, non_blocking=True)
        if isinstance(item, (tuple, list)):
            return type(item)(self._to_device(x, stream) for x in item)
        return item

    def __iter__(self):
        from collections import deque
        cuda = self.device.type == 'cuda'
        stream = torch.cuda.Stream(self.device) if cuda else None
        queue = deque()
        it = iter(self.loader)

        def push():
            try:
                host = next(it)
            except StopIteration:
                return False
            dev = self._to_device(host, stream)
            ev = None
            if cuda:
                ev = torch.cuda.Event()
                ev.record(stream)
            queue.append((dev, ev, host))      # the pinned source stays alive until the copy has been consumed
            return True

        while len(queue) < self.depth and push():
            pass
        while queue:
            dev, ev, _host = queue.popleft()
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                _record_stream(dev, cur)     # allocated on the copy stream, consumed on the compute stream
            push()
            yield dev

    def __len__(self):
        return len(self.loader)
