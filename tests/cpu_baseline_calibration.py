"""Calibrates bench.py's `cpu_baseline` (kind "port": our trainer on the fp32 CPU oracle) against the UNMODIFIED reference on
the same host, same weights, same synthetic batches, same thread count: config-2 dims at 256x256, fp32, batch 2, one plain G+D
step and one gradient-penalty G+D step through each trainer's train_discriminator_step / train_generator_step (optimizer
updates included). Runs only where /root/reference exists (the build container); writes
profiles/r02_cpu_baseline_calibration.json, which bench.py copies into its record.

    python tests/cpu_baseline_calibration.py [threads]
"""
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'tests' / 'oracle_stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

import gigagan_pytorch as ref  # noqa: E402
from gigagan_pytorch_amd import GigaGAN, ops, _C  # noqa: E402
from oracle.torch_ops import OracleOps  # noqa: E402
from oracle.cpu_trainer import install_cpu_adamw  # noqa: E402
import c2_common as c2  # noqa: E402


def loader():
    while True:
        yield c2.real_images()


def timed_steps(gan, bs):
    it = loader()
    out = {}
    for name, gp in (('plain', False), ('gp', True)):
        t0 = time.time()
        gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
        gan.train_generator_step(batch_size=bs, dl_iter=it)
        out[name] = time.time() - t0
    out['cycle_mean_step'] = (3 * out['plain'] + out['gp']) / 4
    out['images_per_sec'] = bs / out['cycle_mean_step']
    return out


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    _C.bind(ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so')
    bs = c2.BASE_BATCH
    G, D = c2.build_models()
    with tempfile.TemporaryDirectory() as tmp:
        rG, rD = ref.Generator(**c2.C2_G), ref.Discriminator(**c2.C2_D)
        rG.load_state_dict(G.state_dict())
        rD.load_state_dict(D.state_dict())
        rgan = ref.GigaGAN(generator=rG, discriminator=rD, apply_gradient_penalty_every=4, create_ema_generator_at_init=False,
                           model_folder=f'{tmp}/rm', results_folder=f'{tmp}/rr')
        torch.manual_seed(3)
        r = timed_steps(rgan, bs)
        del rgan, rG, rD
        with ops.use_impl(OracleOps()):
            gan = GigaGAN(generator=G, discriminator=D, device='cpu', apply_gradient_penalty_every=4,
                          create_ema_generator_at_init=False, use_hip_graphs=False, model_folder=f'{tmp}/m',
                          results_folder=f'{tmp}/r')
            install_cpu_adamw(gan.G_opt)
            install_cpu_adamw(gan.D_opt)
            torch.manual_seed(3)
            p = timed_steps(gan, bs)
    rec = dict(host_cores=os.cpu_count(), threads=threads, batch=bs, reference=r, port=p,
               port_vs_reference=p['images_per_sec'] / r['images_per_sec'],
               note=f'build container, {threads} threads, fp32, config-2 dims, batch {bs}: port {p["images_per_sec"]:.3f} img/s vs '
                    f'unmodified reference {r["images_per_sec"]:.3f} img/s (cycle mean of one plain + one gradient-penalty step); the '
                    'port skips the discriminator weight gradients the reference computes and discards in the G step and runs '
                    'D(fake) and D(real) as one pass')
    (ROOT / 'profiles').mkdir(exist_ok=True)
    (ROOT / 'profiles' / 'r02_cpu_baseline_calibration.json').write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
