"""Generates tests/golden/c2_traj.pt: a FOUR-STEP training trajectory of BASELINE config 2 (256x256, dim_max 512) at batch 2 on the CPU
from the UNMODIFIED reference trainer (/root/reference: GigaGAN.train_discriminator_step / train_generator_step, gp.py:2227-2610,
the loop body of gp.py:2681-2748 with its torch.optim.AdamW), fp32: three plain steps and the gradient-penalty step (step 4), every
torch.randn draw replayed (tests/c2_common.randn_replay, re-armed at the start of every half-step), aux reconstruction loss off (its
dropout mask / patch choice cannot be replayed across devices), no EMA.

Stored per step: the losses, and every 389th element of the flat (FlatAdamW-ordered) parameter vectors of G and D AFTER the step,
plus the same subsample of the initial parameters. tests/test_config2_parity.py::test_config2_four_step_trajectory_vs_reference_trainer
runs our trainer on the MI355X from the same weights (batch 32 = 16 copies) and compares parameter UPDATES step by step: AdamW + the
weight re-pack + the flat buffers are in the loop there, which the step-1 fixture (c2_step1.pt) does not cover.

    python tests/golden/make_golden_c2_traj.py          (build container, ~3 minutes on 8 cores)
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
from gigagan_pytorch_amd import _C  # noqa: E402
import c2_common as c2  # noqa: E402

OUT = Path(__file__).resolve().parent
STEPS = 4
LR = 1e-5        # the reference's default 2e-4 makes this 'training' (uniform-noise images) explode by 10-100x per step (G loss 68 -> 8.5e3 ->
                 # 9.2e4 -> 4.0e5): a chaotic trajectory amplifies bf16 rounding into O(1) differences by step 3 and says nothing about the
                 # optimizer path. At 1e-5 the four steps stay in one regime and the parameter UPDATES remain comparable step by step


def main():
    _C.bind(ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so')
    torch.set_num_threads(8)
    G, D = c2.build_models()
    fx = dict(checksum=c2.weights_checksum(G, D), stride=c2.GRAD_STRIDE, lr=LR, steps=[])
    with tempfile.TemporaryDirectory() as tmp:
        # FlatAdamW's parameter order (what the GPU test compares in): the order of module.parameters() - the reference modules carry
        # the same parameters under the same names, so its state dict is read back in OUR order
        ours = c2.make_trainer(G, D, 'cpu', tmp)
        order_G = [n for n, _ in ours.G.named_parameters()]
        order_D = [n for n, _ in ours.D.named_parameters()]
        assert [id(p) for p in ours.G_opt._all] == [id(p) for _, p in ours.G.named_parameters()]
        assert [id(p) for p in ours.D_opt._all] == [id(p) for _, p in ours.D.named_parameters()]
        del ours
        rG, rD = ref.Generator(**c2.C2_G), ref.Discriminator(**c2.C2_D)
        rG.load_state_dict(G.state_dict())
        rD.load_state_dict(D.state_dict())
        gan = ref.GigaGAN(generator=rG, discriminator=rD, learning_rate=LR, apply_gradient_penalty_every=4, discr_aux_recon_loss_weight=0.,
                          create_ema_generator_at_init=False, model_folder=f'{tmp}/rm', results_folder=f'{tmp}/rr')
        G_, D_ = gan.unwrapped_G, gan.unwrapped_D

        def flat(mod, order):
            sd = dict(mod.named_parameters())
            return torch.cat([sd[n].detach().flatten() for n in order])[::c2.GRAD_STRIDE].clone()

        def loader():
            while True:
                yield c2.real_images()
        fx['p0'] = dict(G=flat(G_, order_G), D=flat(D_, order_D))
        it = loader()
        for step in range(1, STEPS + 1):
            t0 = time.time()
            gp = step % 4 == 0
            with c2.randn_replay():
                dl = gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=gp)
            with c2.randn_replay():
                gl = gan.train_generator_step(batch_size=c2.BASE_BATCH, dl_iter=it)
            rec = dict(gp=gp, d=dict(divergence=float(dl.divergence), multiscale=float(dl.multiscale_divergence),
                                     gradient_penalty=float(dl.gradient_penalty)),
                       g=dict(divergence=float(gl.divergence), multiscale=float(gl.multiscale_divergence)),
                       G=flat(G_, order_G), D=flat(D_, order_D))
            fx['steps'].append(rec)
            print(f'step {step}: {time.time() - t0:.0f} s', rec['d'], rec['g'], flush=True)
        # the yardstick: the same four steps through OUR trainer on the bf16-operand CPU oracle (what the MFMA kernels compute, with
        # torch's AdamW arithmetic): how far bf16 contraction operands ALONE move this trajectory away from the fp32 reference. AdamW's
        # first steps are sign-like, so the rounding noise of near-zero gradient elements flips whole lr-sized updates
        from gigagan_pytorch_amd import ops
        from oracle.torch_ops import OracleOps
        from oracle.cpu_trainer import install_cpu_adamw
        import itertools
        ours = c2.make_trainer(G, D, 'cpu', tmp, learning_rate=LR)
        install_cpu_adamw(ours.G_opt)
        install_cpu_adamw(ours.D_opt)
        sub = lambda opt: torch.cat([p.detach().flatten() for p in opt._all])[::c2.GRAD_STRIDE].clone()
        it2 = itertools.repeat(c2.real_images())
        fx['oracle_bf16'] = []
        with ops.use_impl(OracleOps(bf16_operands=True)):
            for step, want in enumerate(fx['steps'], start=1):
                t0 = time.time()
                with c2.randn_replay():
                    dl = ours.train_discriminator_step(dl_iter=it2, apply_gradient_penalty=want['gp'])
                with c2.randn_replay():
                    gl = ours.train_generator_step(batch_size=c2.BASE_BATCH, dl_iter=it2)
                rec = dict(d=dict(divergence=float(dl.divergence), multiscale=float(dl.multiscale_divergence),
                                  gradient_penalty=float(dl.gradient_penalty)),
                           g=dict(divergence=float(gl.divergence), multiscale=float(gl.multiscale_divergence)))
                for m, opt in (('G', ours.G_opt), ('D', ours.D_opt)):
                    got = sub(opt)
                    du, dw = got - fx['p0'][m], want[m] - fx['p0'][m]
                    rec[m] = dict(update_cosine=float(torch.dot(du, dw) / (du.norm() * dw.norm())),
                                  max_abs_diff_in_lr=float((got - want[m]).abs().max() / LR),
                                  rel_l2=float((got - want[m]).norm() / want[m].norm()))
                fx['oracle_bf16'].append(rec)
                print(f'oracle_bf16 step {step}: {time.time() - t0:.0f} s', rec, flush=True)
    torch.save(fx, OUT / 'c2_traj.pt')
    print('wrote', OUT / 'c2_traj.pt')


if __name__ == '__main__':
    main()
