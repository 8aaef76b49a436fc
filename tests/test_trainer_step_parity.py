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


def test_generator_contrastive_loss_matches_the_reference_trainer(reference, tmp_path):
    """config 4's generator step with the CLIP contrastive loss on (gp.py:174-188, :2530-2592, weight 0.1): images of both
    micro-batches and their captions go through the (stand-in, deterministic) frozen adapter's `contrastive_loss`; the reported
    loss and the generator after the AdamW update equal the reference trainer's. CLIP's own arithmetic is third-party."""
    import torch.nn.functional as F
    from torch import nn
    from helpers import TEXT_ENC, TEXT_CLIP_DIM, TEXT_G, TEXT_D, text_encodings

    class StandInClip(nn.Module):
        dim_latent = TEXT_CLIP_DIM

        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(99)
            self.register_buffer('img_proj', torch.randn(3 * 4 * 4, 12, generator=g))
            self.register_buffer('txt_table', F.normalize(torch.randn(64, 12, generator=g), dim=-1))

        def embed_texts(self, texts):
            enc = torch.cat([text_encodings(batch=1, seed=int(t)) for t in texts])
            for i, t in enumerate(texts):
                enc[i, 3 + int(t) % 4:] = 0
            return self.txt_table[[int(t) % 64 for t in texts]], enc

        def embed_images(self, images):
            feats = F.adaptive_avg_pool2d(images, 4).flatten(1) @ self.img_proj
            return F.normalize(feats, dim=-1), None

        def contrastive_loss(self, images, texts=None, text_embeds=None):        # the arithmetic of open_clip.py:139-158
            if text_embeds is None:
                text_embeds, _ = self.embed_texts(texts)
            image_embeds, _ = self.embed_images(images)
            sim = torch.einsum('i d, j d -> i j', text_embeds, image_embeds) * 14.3
            labels = torch.arange(text_embeds.shape[0])
            return (F.cross_entropy(sim, labels) + F.cross_entropy(sim.t(), labels)) / 2

    torch.manual_seed(0)
    ref_gan = reference.GigaGAN(
        generator=reference.Generator(text_encoder=reference.TextEncoder(clip=StandInClip(), **TEXT_ENC), **TEXT_G),
        discriminator=reference.Discriminator(text_encoder=reference.TextEncoder(clip=StandInClip(), **TEXT_ENC), **TEXT_D),
        generator_contrastive_loss_weight=0.1, model_folder=str(tmp_path / 'rm'), results_folder=str(tmp_path / 'rr'))
    te = dict(clip=StandInClip(), **TEXT_ENC)
    gan = GigaGAN(generator=dict(text_encoder=dict(te), **TEXT_G), discriminator=dict(text_encoder=dict(te), **TEXT_D),
                  generator_contrastive_loss_weight=0.1, device='cpu', create_ema_generator_at_init=False,
                  model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    assert gan.need_contrastive_loss
    gan.G.load_state_dict(ref_gan.unwrapped_G.state_dict())
    gan.D.load_state_dict(ref_gan.unwrapped_D.state_dict())
    ops.bump_weight_epoch()

    def loader():
        i = 0
        while True:
            yield torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(7 + i)), [str(2 * i), str(2 * i + 1)]
            i += 1

    it_ref, it_ours = loader(), loader()
    torch.manual_seed(12)
    g_ref = ref_gan.train_generator_step(batch_size=2, dl_iter=it_ref, grad_accum_every=2)
    with ops.use_impl(OracleOps()):
        torch.manual_seed(12)
        g_ours = gan.train_generator_step(batch_size=2, dl_iter=it_ours, grad_accum_every=2)
    assert float(g_ref.contrastive_loss) > 0
    for a, b in zip(g_ours, g_ref):
        assert abs(float(a) - float(b)) <= 1e-5 * max(1., abs(float(b))), (g_ours, g_ref)
    assert rel_err(_flat(gan.G.parameters()), _flat(ref_gan.unwrapped_G.parameters())) < 1e-6


def test_training_loop_with_gradient_accumulation_matches_the_reference(reference, tmp_path, capsys):
    """`GigaGAN.__call__(steps=, grad_accum_every=)` (gp.py:2665-2750) over a real DataLoader: three steps with two
    micro-batches each (the second step with the gradient penalty) leave the same weights, step counter, log lines and
    EMA generator output as the reference's loop."""
    from torch.utils.data import DataLoader, Dataset

    class Images(Dataset):
        def __len__(self):
            return 16

        def __getitem__(self, i):
            return torch.rand(3, 16, 16, generator=torch.Generator().manual_seed(i))

    torch.manual_seed(0)
    kw = dict(apply_gradient_penalty_every=2, log_steps_every=1)
    ref_gan = reference.GigaGAN(generator=dict(TINY_G), discriminator=dict(TINY_D), model_folder=str(tmp_path / 'rm'),
                                results_folder=str(tmp_path / 'rr'), **kw)
    gan = GigaGAN(generator=dict(TINY_G), discriminator=dict(TINY_D), device='cpu', model_folder=str(tmp_path / 'm'),
                  results_folder=str(tmp_path / 'r'), **kw)
    gan.merge_discriminator_passes = False
    gan.G.load_state_dict(ref_gan.unwrapped_G.state_dict())
    gan.D.load_state_dict(ref_gan.unwrapped_D.state_dict())
    gan.G_ema.load_state_dict(ref_gan.G_ema.state_dict())
    ops.bump_weight_epoch()
    ref_gan.set_dataloader(DataLoader(Images(), batch_size=2, shuffle=False))
    gan.set_dataloader(DataLoader(Images(), batch_size=2, shuffle=False))
    capsys.readouterr()
    torch.manual_seed(5)
    ref_gan(steps=3, grad_accum_every=2)
    log_ref = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('G:')]
    with ops.use_impl(OracleOps()):
        torch.manual_seed(5)
        gan(steps=3, grad_accum_every=2)
    log_ours = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('G:')]
    assert len(log_ref) == 3 and log_ours == log_ref
    assert int(ref_gan.steps.item()) == gan._steps_host == 4
    assert rel_err(_flat(gan.D.parameters()), _flat(ref_gan.unwrapped_D.parameters())) < 1e-6
    assert rel_err(_flat(gan.G.parameters()), _flat(ref_gan.unwrapped_G.parameters())) < 1e-6
    z = torch.randn(2, 32)
    with ops.use_impl(OracleOps()):
        torch.manual_seed(1)
        ours = gan.generate(noise=z)
    torch.manual_seed(1)
    theirs = ref_gan.generate(noise=z)
    assert rel_err(ours, theirs) < 1e-5
