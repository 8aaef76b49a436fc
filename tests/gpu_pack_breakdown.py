"""Where the weight re-pack of the config-2 discriminator / generator spends its time: the trainer's own tables, then
sub-tables grouped by (taps, kind). Run on the GPU box: python tests/gpu_pack_breakdown.py (test infrastructure)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench                                                        # noqa: E402
from gigagan_pytorch_amd import kernels as K                         # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages                 # noqa: E402
from itertools import cycle                                         # noqa: E402


def time_us(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device('cuda', 0)
    gan = bench.build_gan(256, dev, use_hip_graphs=False)
    it = cycle(SyntheticImages(32, 256, device=dev))
    gan.use_hip_graphs = False
    gan.train_step(it, 32)
    for name, opt in (('D', gan.D_opt), ('G', gan.G_opt)):
        tab = opt.pack_table
        us = time_us(tab.refresh)
        print(f'{name} table: {tab.n} entries, {tab.items} items: {us:7.1f} us', flush=True)
        groups = {}
        for src, dst in tab.keep:
            groups.setdefault((tuple(src.shape), tuple(dst.shape)), []).append((src, dst))
        by = {}
        for (ss, ds), lst in groups.items():
            if len(ss) == 4:
                O, I, T = ss[0], ss[1], ss[2] * ss[3]
            elif len(ss) == 3:
                O, I, T = ss
            elif len(ss) == 2:
                O, I, T = ss[0], ss[1], 1
            else:
                continue
            r8 = lambda v: (v + 7) // 8 * 8
            if tuple(ds) == (r8(O), T * r8(I)):
                kind = 'fwd'
            elif tuple(ds) == (r8(I), T * r8(O)):
                kind = 'bwd'
            else:
                continue                    # space-to-depth / bank layouts: not re-registered here
            by.setdefault((T, kind), []).extend(s_.reshape(O, I, T) for s_, _ in lst)
        for (T, kind), srcs in sorted(by.items()):
            t2 = K.PackTable(dev, capacity=len(srcs) + 4)
            tot = 0
            for s in srcs:
                t2.register(s, s.shape[0], s.shape[1], T, kind)
                tot += s.numel()
            u = time_us(t2.refresh)
            print(f'   T={T:2d} {kind}: {len(srcs):3d} weights {tot / 1e6:6.2f} M: {u:7.1f} us  {tot * 6 / u / 1e6:5.2f} TB/s', flush=True)


if __name__ == '__main__':
    main()
