"""The reference's own CPU path, timed by the SURVEY.md §8(d) / BASELINE.md §3 protocol: the UNMODIFIED reference trainer
(/root/reference, inert stubs of tests/oracle_stubs for the absent third-party imports) driven through its public loop
`GigaGAN(...)(steps=4)` — config-2 dims (uncond 256x256, G cap 8 / D cap 16 / dim_max 512), fp32 (`amp=False`), batch 4,
`apply_gradient_penalty_every=4`, synthetic `torch.rand` images from a DataLoader, all host threads. One warm-up 4-step cycle
(also absorbs the step-1 sample/checkpoint the reference writes), then ONE timed 4-step cycle = 3 plain + 1 gradient-penalty
G+D steps incl. optimizer updates, EMA and the per-step `.item()` syncs. Then the same cycle through OUR trainer on the fp32
CPU oracle (`kind: "port"`, what bench.py can time on the GPU box, where /root/reference does not exist): the ratio
port / reference is what bench.py uses to print `cpu_baseline.reference_equivalent`.

Runs only where /root/reference exists (the build container); writes profiles/r04_cpu_baseline_reference_<threads>t.json (one record per
thread count: bench.py reads them all and reports the ratio with its spread).

    python tests/cpu_baseline_reference.py [threads] [first_core]      # pins itself to `threads` cores from `first_core` on
"""
import json
import os
import resource
import sys
import tempfile
import time
from pathlib import Path

import torch
from torch.utils.data import DataLoader, Dataset

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'tests' / 'oracle_stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

BATCH, SIZE = 4, 256
G_CFG = dict(image_size=SIZE, dim_capacity=8, dim_max=512, style_network=dict(dim=64, depth=4), num_skip_layers_excite=4,
             unconditional=True)
D_CFG = dict(image_size=SIZE, dim_capacity=16, dim_max=512, num_skip_layers_excite=4, unconditional=True)


class RandImages(Dataset):
    def __init__(self, n=64):
        g = torch.Generator().manual_seed(0)
        self.x = torch.rand(n, 3, SIZE, SIZE, generator=g)

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i]


def timed_cycles(gan):
    gan.set_dataloader(DataLoader(RandImages(), batch_size=BATCH, shuffle=False, drop_last=True))
    t0 = time.time()
    gan(steps=4)                      # steps 1-4 (warm-up; step 4 carries the gradient penalty)
    warm = time.time() - t0
    t0 = time.time()
    gan(steps=4)                      # steps 5-8: three plain steps + the gradient-penalty step 8
    return warm, time.time() - t0


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    if len(sys.argv) > 2:       # keep the measurement on its own cores (the build container is shared with compiles / tests)
        first = int(sys.argv[2])
        os.sched_setaffinity(0, set(range(first, first + threads)))
    torch.set_num_threads(threads)
    rec = dict(host_cores=os.cpu_count(), threads=threads, pinned=len(sys.argv) > 2, batch=BATCH, image_size=SIZE, dtype='fp32',
               protocol='GigaGAN(...)(steps=4) twice: warm-up cycle, then one timed 4-step cycle (3 plain + 1 GP)')
    big = dict(save_and_sample_every=10 ** 9, early_save_and_sample_every=10 ** 9, log_steps_every=10 ** 9)
    with tempfile.TemporaryDirectory() as tmp:
        import gigagan_pytorch as ref
        torch.manual_seed(0)
        rgan = ref.GigaGAN(generator=dict(G_CFG), discriminator=dict(D_CFG), amp=False, apply_gradient_penalty_every=4,
                           create_ema_generator_at_init=True, model_folder=f'{tmp}/rm', results_folder=f'{tmp}/rr', **big)
        warm, cyc = timed_cycles(rgan)
        rec['reference'] = dict(warmup_cycle_s=warm, timed_cycle_s=cyc, s_per_step=cyc / 4, images_per_sec=4 * BATCH / cyc)
        print('reference', rec['reference'], flush=True)
        del rgan

        from gigagan_pytorch_amd import GigaGAN, ops, _C
        from oracle.torch_ops import OracleOps
        from oracle.cpu_trainer import install_cpu_adamw
        _C.bind(ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so')
        torch.manual_seed(0)
        with ops.use_impl(OracleOps()):
            gan = GigaGAN(generator=dict(G_CFG), discriminator=dict(D_CFG), device='cpu', apply_gradient_penalty_every=4,
                          create_ema_generator_at_init=True, use_hip_graphs=False, model_folder=f'{tmp}/m',
                          results_folder=f'{tmp}/r', **big)
            install_cpu_adamw(gan.G_opt)
            install_cpu_adamw(gan.D_opt)
            warm, cyc = timed_cycles(gan)
        rec['port'] = dict(warmup_cycle_s=warm, timed_cycle_s=cyc, s_per_step=cyc / 4, images_per_sec=4 * BATCH / cyc)
        print('port', rec['port'], flush=True)
    rec['port_vs_reference'] = rec['port']['images_per_sec'] / rec['reference']['images_per_sec']
    rec['peak_rss_gb'] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    rec['note'] = (f'build container, {threads} threads of {os.cpu_count()} cores, fp32, config-2 dims, batch {BATCH}: unmodified reference '
                   f"{rec['reference']['images_per_sec']:.3f} img/s, our trainer on the fp32 CPU oracle {rec['port']['images_per_sec']:.3f} "
                   'img/s over the same 4-step cycle (the port skips the discriminator weight gradients the reference computes and '
                   'discards in the G step and runs D(fake), D(real) as one pass)')
    (ROOT / 'profiles').mkdir(exist_ok=True)
    out = os.environ.get('GG_CAL_OUT', f'r04_cpu_baseline_reference_{threads}t.json')
    (ROOT / 'profiles' / out).write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
