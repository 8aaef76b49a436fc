"""End-to-end parity of the optimisation step with the unmodified reference trainer (gp.py:2227-2610 + optimizer.py:10-34):
same initial weights, same loader batches, same RNG stream -> the same losses for the discriminator step (plain and with the
gradient penalty's double backward), the generator step, and the same parameters after both AdamW updates. Ours runs on the
fp32 oracle ops with the reference's two-pass discriminator formulation (`merge_discriminator_passes = False`; the merged
pass is proven equivalent in test_trainer_cpu.py) and steps through the fused flat AdamW kernel. Live only."""
import torch

from gigagan_pytorch_amd import GigaGAN, ops
from oracle.torch_ops import OracleOps
from helpers import rel_err, TINY_G, TINY_D


def _flat(params):
    return torch.cat([p.detach().flatten() for p in params])


def test_training_steps_match_the_reference_trainer(reference, tmp_path):
    torch.manual_seed(0)
    ref_gan = reference.GigaGAN(generator=dict(TINY_G), discriminator=dict(TINY_D), apply_gradient_penalty_every=2,
                                model_folder=str(tmp_path / 'rm'), results_folder=str(tmp_path / 'rr'))
    gan = GigaGAN(generator=dict(TINY_G), discriminator=dict(TINY_D), apply_gradient_penalty_every=2, device='cpu',
                  create_ema_generator_at_init=False, model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    gan.merge_discriminator_passes = False
    gan.G.load_state_dict(ref_gan.unwrapped_G.state_dict())
    gan.D.load_state_dict(ref_gan.unwrapped_D.state_dict())
    ops.bump_weight_epoch()
    batches = [torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(7 + i)) for i in range(4)]

    def loader():
        while True:
            for t in batches:
                yield t.clone()

    def num(x):
        return float(x) if x is not None else None

    for with_penalty in (False, True):
        it_ref, it_ours = loader(), loader()
        torch.manual_seed(11)
        d_ref = ref_gan.train_discriminator_step(dl_iter=it_ref, apply_gradient_penalty=with_penalty)
        with ops.use_impl(OracleOps()):
            torch.manual_seed(11)
            d_ours = gan.train_discriminator_step(dl_iter=it_ours, apply_gradient_penalty=with_penalty)
        for a, b in zip(d_ours, d_ref):
            assert (a is None) == (b is None)
            if a is not None:
                assert abs(num(a) - num(b)) <= 1e-5 * max(1., abs(num(b))), (with_penalty, d_ours, d_ref)
        assert (num(d_ref.gradient_penalty) > 0) == with_penalty
        torch.manual_seed(12)
        g_ref = ref_gan.train_generator_step(batch_size=2, dl_iter=it_ref)
        with ops.use_impl(OracleOps()):
            torch.manual_seed(12)
            g_ours = gan.train_generator_step(batch_size=2, dl_iter=it_ours)
        for a, b in zip(g_ours, g_ref):
            assert abs(num(a) - num(b)) <= 1e-5 * max(1., abs(num(b))), (g_ours, g_ref)
        # both optimizers have stepped: the reference's torch AdamW vs the fused flat-buffer kernel
        assert rel_err(_flat(gan.D.parameters()), _flat(ref_gan.unwrapped_D.parameters())) < 1e-6
        assert rel_err(_flat(gan.G.parameters()), _flat(ref_gan.unwrapped_G.parameters())) < 1e-6


def test_text_conditional_training_steps_match_the_reference_trainer(reference, tmp_path):
    """config 4's step semantics: raw captions from the loader through a (deterministic stand-in for the frozen) CLIP adapter,
    TextEncoder, cross attention, text-modulated predictors, the matching-aware loss on rolled (mismatched) captions."""
    from torch import nn
    from helpers import TEXT_ENC, TEXT_CLIP_DIM, TEXT_G, TEXT_D, text_encodings

    class StandInClip(nn.Module):
        dim_latent = TEXT_CLIP_DIM

        def embed_texts(self, texts):
            enc = torch.cat([text_encodings(batch=1, seed=int(t)) for t in texts])
            for i, t in enumerate(texts):
                enc[i, 3 + int(t) % 4:] = 0                     # ragged caption lengths
            return None, enc

    torch.manual_seed(0)
    ref_gan = reference.GigaGAN(
        generator=reference.Generator(text_encoder=reference.TextEncoder(clip=StandInClip(), **TEXT_ENC), **TEXT_G),
        discriminator=reference.Discriminator(text_encoder=reference.TextEncoder(clip=StandInClip(), **TEXT_ENC), **TEXT_D),
        generator_contrastive_loss_weight=0., model_folder=str(tmp_path / 'rm'), results_folder=str(tmp_path / 'rr'))
    te = dict(clip=StandInClip(), **TEXT_ENC)
    gan = GigaGAN(generator=dict(text_encoder=dict(te), **TEXT_G), discriminator=dict(text_encoder=dict(te), **TEXT_D),
                  generator_contrastive_loss_weight=0., device='cpu', create_ema_generator_at_init=False,
                  model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    gan.merge_discriminator_passes = False
    gan.G.load_state_dict(ref_gan.unwrapped_G.state_dict())
    gan.D.load_state_dict(ref_gan.unwrapped_D.state_dict())
    ops.bump_weight_epoch()

    def loader():
        i = 0
        while True:
            yield torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(7 + i)), [str(2 * i), str(2 * i + 1)]
            i += 1

    def num(x):
        return float(x) if x is not None else None

    for with_penalty in (False, True):
        it_ref, it_ours = loader(), loader()
        torch.manual_seed(11)
        d_ref = ref_gan.train_discriminator_step(dl_iter=it_ref, apply_gradient_penalty=with_penalty)
        with ops.use_impl(OracleOps()):
            torch.manual_seed(11)
            d_ours = gan.train_discriminator_step(dl_iter=it_ours, apply_gradient_penalty=with_penalty)
        for a, b in zip(d_ours, d_ref):
            assert abs(num(a) - num(b)) <= 1e-5 * max(1., abs(num(b))), (with_penalty, d_ours, d_ref)
        assert num(d_ref.total_matching_aware_loss) > 0
        torch.manual_seed(12)
        g_ref = ref_gan.train_generator_step(batch_size=2, dl_iter=it_ref)
        with ops.use_impl(OracleOps()):
            torch.manual_seed(12)
            g_ours = gan.train_generator_step(batch_size=2, dl_iter=it_ours)
        for a, b in zip(g_ours, g_ref):
            assert abs(num(a) - num(b)) <= 1e-5 * max(1., abs(num(b))), (g_ours, g_ref)
        assert rel_err(_flat(gan.D.parameters()), _flat(ref_gan.unwrapped_D.parameters())) < 1e-6
        assert rel_err(_flat(gan.G.parameters()), _flat(ref_gan.unwrapped_G.parameters())) < 1e-6
