"""ORACLE — test infrastructure only: times the UNMODIFIED reference trainer on host cores (bench.py's `cpu_baseline`, kind
"reference"; SURVEY.md §8d / BASELINE.md §3 protocol).

Runs as its own process with every GPU hidden (so `accelerate` places the reference on the CPU, exactly as in the build
container) and imports the reference as byte code from `oracle/_ref` (oracle/build_ref.py; from /root/reference directly where
that exists and oracle/_ref was not built). Config-2 dims (uncond 256x256, G cap 8 / D cap 16 / dim_max 512), fp32
(`amp=False`), batch 4 (batch 32 needs ~150 GB of host memory), `apply_gradient_penalty_every=4`, synthetic `torch.rand` images
through the reference's own `set_dataloader` / `GigaGAN.forward(steps=...)` loop: ONE warm-up step, then one timed 4-step cycle
(steps 2-5: three plain G+D steps + the gradient-penalty step 4, incl. optimizer updates, EMA, the per-step `.item()` syncs).
Prints one JSON record.

    python oracle/time_reference.py --threads 8 [--warmup-steps 1] [--batch 4]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

for _k in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):
    os.environ[_k] = ''        # before torch is imported: the reference must see no accelerator

import torch                                            # noqa: E402
from torch.utils.data import DataLoader, Dataset        # noqa: E402

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

SIZE = 256
G_CFG = dict(image_size=SIZE, dim_capacity=8, dim_max=512, style_network=dict(dim=64, depth=4), num_skip_layers_excite=4,
             unconditional=True)
D_CFG = dict(image_size=SIZE, dim_capacity=16, dim_max=512, num_skip_layers_excite=4, unconditional=True)


class RandImages(Dataset):
    def __init__(self, n=64):
        self.x = torch.rand(n, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(0))

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--warmup-steps', type=int, default=1)
    args = ap.parse_args()
    assert not torch.cuda.is_available(), 'the reference CPU leg must not see a GPU'
    torch.set_num_threads(args.threads)

    from oracle.build_ref import available, import_reference, STUBS
    if available():
        ref, origin = import_reference(), 'oracle/_ref (byte code of the unmodified reference files)'
    else:
        sys.path.insert(0, str(STUBS))
        sys.path.insert(0, '/root/reference')
        import gigagan_pytorch as ref
        origin = '/root/reference'

    big = dict(save_and_sample_every=10 ** 9, early_save_and_sample_every=10 ** 9, log_steps_every=10 ** 9)
    with tempfile.TemporaryDirectory() as tmp:
        torch.manual_seed(0)
        gan = ref.GigaGAN(generator=dict(G_CFG), discriminator=dict(D_CFG), amp=False, apply_gradient_penalty_every=4,
                          create_ema_generator_at_init=True, model_folder=f'{tmp}/m', results_folder=f'{tmp}/r', **big)
        assert next(gan.G.parameters()).device.type == 'cpu'
        gan.set_dataloader(DataLoader(RandImages(), batch_size=args.batch, shuffle=False, drop_last=True))
        t0 = time.time()
        if args.warmup_steps:
            gan(steps=args.warmup_steps)
        warm = time.time() - t0
        first = int(gan.steps.item())         # the buffer holds the index of the NEXT step (starts at 1, gp.py:2011)
        t0 = time.time()
        gan(steps=4)                     # four consecutive trainer steps: exactly one of them carries the gradient penalty
        cyc = time.time() - t0
    print(json.dumps(dict(kind='reference', origin=origin, threads=args.threads, host_cores=os.cpu_count(), batch=args.batch,
                          image_size=SIZE, dtype='fp32', warmup_steps=args.warmup_steps, warmup_s=warm, timed_steps=[first, first + 3],
                          timed_cycle_s=cyc, images_per_sec=4 * args.batch / cyc)), flush=True)


if __name__ == '__main__':
    main()
