"""Generates tests/golden/c2_step1.pt: step-1 quantities of BASELINE config 2 (256x256, dim_max 512) at batch 2 on the CPU from

  (a) the UNMODIFIED reference (/root/reference with the inert stubs of tests/oracle_stubs), fp32 — key `ref`;
  (b) our trainer on the fp32 oracle ops — key `oracle_f32` (must agree with (a) to fp32 rounding: this pins the oracle and the
      host-side assembly at config-2 dims, not only at the toy sizes of the other fixtures);
  (c) our trainer on the oracle with bf16-rounded contraction operands — key `oracle_bf16`, what the MFMA kernels compute.

The GPU test (tests/test_config2_parity.py) rebuilds the same weights from the same seed (checksum stored here), runs the
MI355X path at batch 32 (16 copies of the 2 samples) and compares. Flat gradients are stored subsampled (every 389th element)
plus per-parameter norms. Run in the build container (takes a few minutes on 8 cores):

    python tests/golden/make_golden_c2.py
"""
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / 'tests' / 'oracle_stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

import gigagan_pytorch as ref  # noqa: E402
from gigagan_pytorch_amd import ops, _C  # noqa: E402
from oracle.torch_ops import OracleOps  # noqa: E402
import c2_common as c2  # noqa: E402

OUT = Path(__file__).resolve().parent


def compress(out, gan):
    o = dict(out)
    for k, opt in (('d_plain_grad', gan.D_opt), ('d_gp_grad', gan.D_opt), ('g_grad', gan.G_opt)):
        flat = o.pop(k)
        o[k + '_sub'] = flat[::c2.GRAD_STRIDE].clone()
        o[k + '_pnorm'] = c2.param_norms(opt, flat)
        o[k + '_norm'] = float(flat.norm())
    o.pop('img_all', None)
    return o


def reference_step_one(G, D, tmp):
    """the same quantities from the unmodified reference modules / trainer (two-pass discriminator step, gp.py:2227-2430)."""
    rG, rD = ref.Generator(**c2.C2_G), ref.Discriminator(**c2.C2_D)
    rG.load_state_dict(G.state_dict())
    rD.load_state_dict(D.state_dict())
    gan = ref.GigaGAN(generator=rG, discriminator=rD, apply_gradient_penalty_every=4, discr_aux_recon_loss_weight=0.,
                      create_ema_generator_at_init=False, model_folder=f'{tmp}/rm', results_folder=f'{tmp}/rr')
    out = {}
    G_, D_ = gan.unwrapped_G, gan.unwrapped_D
    G_.train(); D_.train()
    with c2.randn_replay(), torch.no_grad():
        img, rgbs = G_(noise=c2.latents(), return_all_rgbs=True)
    out['img'], out['rgbs'] = img.clone(), [r.clone() for r in rgbs]
    real = c2.real_images()
    with torch.no_grad():
        logits, ms, _ = D_(real, D_.real_images_to_rgbs(real), calc_aux_loss=False)
    out['logits'] = logits.clone()
    out['ms'] = [m.reshape(-1, c2.BASE_BATCH, *m.shape[1:]).clone() for m in ms]

    def loader():
        while True:
            yield c2.real_images()

    def flat(params):
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in params])

    for name, gp in (('d_plain', False), ('d_gp', True)):
        # the reference steps its optimizer inside train_discriminator_step; keep the weights where they are
        sd = {k: v.clone() for k, v in D_.state_dict().items()}
        step = gan.D_opt.step
        gan.D_opt.step = lambda *a, **k: None
        try:
            with c2.randn_replay():
                losses = gan.train_discriminator_step(dl_iter=loader(), apply_gradient_penalty=gp)
        finally:
            gan.D_opt.step = step
        D_.load_state_dict(sd)
        out[name] = dict(divergence=float(losses.divergence), multiscale=float(losses.multiscale_divergence),
                         gradient_penalty=float(losses.gradient_penalty))
        out[name + '_gradnorm'] = float(flat(D_.parameters()).norm())
    step = gan.G_opt.step
    gan.G_opt.step = lambda *a, **k: None
    try:
        with c2.randn_replay():
            losses = gan.train_generator_step(batch_size=c2.BASE_BATCH, dl_iter=loader())
    finally:
        gan.G_opt.step = step
    out['g'] = dict(divergence=float(losses.divergence), multiscale=float(losses.multiscale_divergence))
    out['g_gradnorm'] = float(flat(G_.parameters()).norm())
    return out


def main():
    _C.bind(ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so')     # FlatAdamW's pack table wants a bound library on the CPU
    torch.set_num_threads(8)
    G, D = c2.build_models()
    fx = dict(checksum=c2.weights_checksum(G, D), grad_stride=c2.GRAD_STRIDE)
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        fx['ref'] = reference_step_one(G, D, tmp)
        print(f'reference: {time.time() - t0:.0f} s', fx['ref']['d_plain'], fx['ref']['d_gp'], fx['ref']['g'], flush=True)
        for key, impl in (('oracle_f32', OracleOps()), ('oracle_bf16', OracleOps(bf16_operands=True))):
            t0 = time.time()
            gan = c2.make_trainer(G, D, 'cpu', tmp)
            with ops.use_impl(impl):
                fx[key] = compress(c2.run_step_one(gan, c2.BASE_BATCH), gan)
            fx[key]['rgbs'] = fx[key]['rgbs'][:-1]          # the last one is the image itself
            print(f'{key}: {time.time() - t0:.0f} s', fx[key]['d_plain'], fx[key]['d_gp'], fx[key]['g'], flush=True)
            del gan
    # the fp32 oracle only has to prove that it IS the reference at these dims: keep the numbers, not the tensors
    of, rf = fx.pop('oracle_f32'), fx['ref']
    rel = lambda a, b: float((a - b).norm() / b.norm())
    fx['oracle_f32_vs_reference'] = dict(
        img=rel(of['img'], rf['img']), logits=rel(of['logits'], rf['logits']),
        ms=[rel(a, b) for a, b in zip(of['ms'], rf['ms'])],
        losses={k: {n: (of[k][n], rf[k][n]) for n in of[k]} for k in ('d_plain', 'd_gp', 'g')},
        grad_norms={k: (of[k + '_grad_norm'], rf[k + '_gradnorm']) for k in ('d_plain', 'd_gp', 'g')})
    fx['oracle_f32_grad_norms'] = {k: of[k + '_grad_norm'] for k in ('d_plain', 'd_gp', 'g')}
    # ... and the subsampled fp32 gradients: the yardstick for the bf16 paths (the test bounds the HIP path's distance to these
    # by the bf16-operand oracle's own distance to them)
    fx['oracle_f32_grad_sub'] = {k: of[k + '_grad_sub'] for k in ('d_plain', 'd_gp', 'g')}
    rf.pop('rgbs')
    print(fx['oracle_f32_vs_reference'])
    torch.save(fx, OUT / 'c2_step1.pt')
    print('wrote', OUT / 'c2_step1.pt')


if __name__ == '__main__':
    main()
