"""Generates tests/golden/c4_step1.pt and tests/golden/c5_step1.pt: step-1 quantities of BASELINE configs 4 and 5 at their
BASELINE dimensions (tests/c45_common.py), batch 2, CPU, from

  (a) the UNMODIFIED reference trainer (/root/reference with the inert stubs of tests/oracle_stubs), fp32 — key `ref`;
  (b) our trainer on the fp32 oracle ops — reduced to `oracle_f32_vs_reference` (pins oracle + host assembly at these dims);
  (c) our trainer on the oracle with bf16-rounded contraction operands — key `oracle_bf16` (what the MFMA kernels compute).

Two shims around the reference, both outside its arithmetic: (1) CLIP is replaced by a parameter-less adapter that returns
the fixture's pre-computed (b, 77, 512) token encodings for the loader's caption ids (open_clip weights do not exist
offline; the reference supports exactly this through `TextEncoder(clip=...)`, gp.py:815-826); (2) `UnetUpsampler.forward`
is given the keyword name the reference's own trainer calls it with (`lowres_image=`, gp.py:2212 vs unet_upsampler.py:659 —
SURVEY.md Appendix B.2: without it `train_upsampler=True` raises TypeError before computing anything).

    python tests/golden/make_golden_c45.py [c4|c5 ...]
"""
import sys
import tempfile
import time
from pathlib import Path

import torch
from torch import nn

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / 'tests' / 'oracle_stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

import gigagan_pytorch as ref  # noqa: E402
from gigagan_pytorch_amd import ops, _C  # noqa: E402
from oracle.torch_ops import OracleOps  # noqa: E402
import c2_common as c2  # noqa: E402
import c45_common as cc  # noqa: E402

OUT = Path(__file__).resolve().parent


class FixtureClip(nn.Module):
    dim_latent = cc.CLIP_DIM

    def embed_texts(self, texts):
        enc = cc.text_encodings()
        return None, torch.stack([enc[int(t)] for t in texts])


def _flat(params):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in params])


def reference_step_one(cfg, G, D, tmp):
    if cfg == 'c4':
        rG = ref.Generator(text_encoder=ref.TextEncoder(clip=FixtureClip(), dim=64, depth=4), **cc.C4_G)
        rD = ref.Discriminator(text_encoder=ref.TextEncoder(clip=FixtureClip(), dim=64, depth=4), **cc.C4_D)
    else:
        rG = ref.UnetUpsampler(**cc.C5_G)
        rD = ref.Discriminator(**cc.C5_D)
        fwd = ref.UnetUpsampler.forward
        if 'lowres_image' not in fwd.__code__.co_varnames:
            def forward(self, lowres_image_or_video=None, *a, lowres_image=None, **k):
                return fwd(self, lowres_image if lowres_image_or_video is None else lowres_image_or_video, *a, **k)
            ref.UnetUpsampler.forward = forward
    missing, unexpected = rG.load_state_dict(G.state_dict(), strict=False)
    assert not [k for k in missing if 'clip' not in k] and not unexpected, (missing[:5], unexpected[:5])
    missing, unexpected = rD.load_state_dict(D.state_dict(), strict=False)
    assert not [k for k in missing if 'clip' not in k] and not unexpected, (missing[:5], unexpected[:5])
    gan = ref.GigaGAN(generator=rG, discriminator=rD, train_upsampler=cfg == 'c5', apply_gradient_penalty_every=4,
                      discr_aux_recon_loss_weight=0., generator_contrastive_loss_weight=0.,
                      create_ema_generator_at_init=False, model_folder=f'{tmp}/rm', results_folder=f'{tmp}/rr')
    G_, D_ = gan.unwrapped_G, gan.unwrapped_D
    G_.train(); D_.train()
    out = {}
    real = cc.real_images()
    text = dict(texts=['0', '1']) if cfg == 'c4' else {}
    with c2.randn_replay(), torch.no_grad():
        if cfg == 'c5':
            lowres = torch.nn.functional.interpolate(real, (64, 64))
            img, rgbs = G_(lowres, noise=cc.latents(), return_all_rgbs=True)
        else:
            img, rgbs = G_(noise=cc.latents(), return_all_rgbs=True, **text)
    out['img'], out['rgbs'] = img.clone(), [r.clone() for r in rgbs]
    with torch.no_grad():
        logits, ms, _ = D_(real, D_.real_images_to_rgbs(real), calc_aux_loss=False, **text)
    out['logits'] = logits.clone()
    out['ms'] = [m.reshape(-1, cc.BASE_BATCH, *m.shape[1:]).clone() for m in ms]

    def loader():
        while True:
            yield (cc.real_images(), ['0', '1']) if cfg == 'c4' else cc.real_images()

    it = loader()
    for name, gp in (('d_plain', False), ('d_gp', True)):
        step = gan.D_opt.step
        gan.D_opt.step = lambda *a, **k: None
        try:
            with c2.randn_replay():
                L = gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
        finally:
            gan.D_opt.step = step
        out[name] = dict(divergence=float(L.divergence), multiscale=float(L.multiscale_divergence),
                         gradient_penalty=float(L.gradient_penalty), matching_aware=float(L.total_matching_aware_loss))
        out[name + '_gradnorm'] = float(_flat([p for p in D_.parameters() if p.requires_grad]).norm())
    step = gan.G_opt.step
    gan.G_opt.step = lambda *a, **k: None
    try:
        with c2.randn_replay():
            L = gan.train_generator_step(batch_size=cc.BASE_BATCH, dl_iter=it)
    finally:
        gan.G_opt.step = step
    out['g'] = dict(divergence=float(L.divergence), multiscale=float(L.multiscale_divergence))
    out['g_gradnorm'] = float(_flat([p for p in G_.parameters() if p.requires_grad]).norm())
    return out


def main():
    _C.bind(ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so')
    torch.set_num_threads(8)
    for cfg in (sys.argv[1:] or cc.CONFIGS):
        G, D = cc.build_models(cfg)
        fx = dict(config=cfg, checksum=c2.weights_checksum(G, D), grad_stride=cc.GRAD_STRIDE)
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.time()
            fx['ref'] = reference_step_one(cfg, G, D, tmp)
            print(f'{cfg} reference: {time.time() - t0:.0f} s', fx['ref']['d_plain'], fx['ref']['d_gp'], fx['ref']['g'], flush=True)
            for key, impl in (('oracle_f32', OracleOps()), ('oracle_bf16', OracleOps(bf16_operands=True))):
                t0 = time.time()
                gan = cc.make_trainer(cfg, G, D, 'cpu', tmp)
                gan.merge_discriminator_passes = key == 'oracle_bf16'      # fp32: the reference's two-pass formulation
                with ops.use_impl(impl):
                    fx[key] = cc.compress(cc.run_step_one(gan, cfg, cc.BASE_BATCH), gan)
                print(f'{cfg} {key}: {time.time() - t0:.0f} s', fx[key]['d_plain'], fx[key]['d_gp'], fx[key]['g'], flush=True)
                del gan
        of, rf = fx.pop('oracle_f32'), fx['ref']
        rel = lambda a, b: float((a - b).norm() / b.norm())
        fx['oracle_f32_vs_reference'] = dict(
            img=rel(of['img'], rf['img']), rgbs=[rel(a, b) for a, b in zip(of['rgbs'], rf['rgbs'])],
            logits=rel(of['logits'], rf['logits']), ms=[rel(a, b) for a, b in zip(of['ms'], rf['ms'])],
            losses={k: {n: (of[k][n], rf[k][n]) for n in of[k]} for k in ('d_plain', 'd_gp', 'g')},
            grad_norms={k: (of[k + '_grad_norm'], rf[k + '_gradnorm']) for k in ('d_plain', 'd_gp', 'g')})
        fx['oracle_f32_grad_norms'] = {k: of[k + '_grad_norm'] for k in ('d_plain', 'd_gp', 'g')}
        fx['oracle_f32_grad_sub'] = {k: of[k + '_grad_sub'] for k in ('d_plain', 'd_gp', 'g')}
        # keep the fixture small: reference image at bf16-exact fp16? no - fp32 image only, rgbs dropped (the oracle's are kept)
        rf.pop('rgbs')
        fx['oracle_bf16']['rgbs'] = [r for r in fx['oracle_bf16']['rgbs'][:-1]]
        print(cfg, fx['oracle_f32_vs_reference'], flush=True)
        torch.save(fx, OUT / f'{cfg}_step1.pt')
        print('wrote', OUT / f'{cfg}_step1.pt', (OUT / f'{cfg}_step1.pt').stat().st_size, flush=True)


if __name__ == '__main__':
    main()
