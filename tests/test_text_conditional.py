"""Text-conditional GigaGAN (BASELINE config 4): TextEncoder / cross attention / text-modulated
predictor against the reference's own outputs (tests/golden/text_small.pt from tests/golden/make_golden_text.py; live
reference where /root/reference exists), the HIP op set against the oracle on the kernel emulator (CPU) and on the MI355X
(-m gpu). CLIP itself is an external frozen encoder: every case feeds pre-computed token encodings (gp.py:843-852)."""
from pathlib import Path

import pytest
import torch

from gigagan_pytorch_amd import ops, Generator, Discriminator
from gigagan_pytorch_amd.gigagan import aux_matching_loss, gradient_penalty
from oracle.torch_ops import OracleOps
from helpers import rel_err, TEXT_ENC, TEXT_CLIP_DIM, TEXT_G, TEXT_D, text_encodings

GOLD = Path(__file__).resolve().parent / 'golden'
TOL = 1e-5          # fp32 restatement vs fp32 reference
TOL_BF16 = 3e-2     # bf16 operands and activations end to end vs the fp32 reference


@pytest.fixture(scope='module')
def fx():
    return torch.load(GOLD / 'text_small.pt', weights_only=False)


def _models(fx, dev='cpu'):
    te = dict(clip_dim_latent=TEXT_CLIP_DIM, **TEXT_ENC)
    G = Generator(text_encoder=dict(te), **TEXT_G)
    D = Discriminator(text_encoder=dict(te), **TEXT_D)
    assert list(G.state_dict().keys()) == list(fx['G'].keys())
    assert list(D.state_dict().keys()) == list(fx['D'].keys())
    G.load_state_dict(fx['G']); D.load_state_dict(fx['D'])
    return G.to(dev), D.to(dev).eval()


# ---- oracle + host logic vs the reference (fp32) ---------------------------------------------------------------

def test_text_models_on_oracle_match_reference_fixture(fx):
    G, D = _models(fx)
    with ops.use_impl(OracleOps()):
        g_tok, fine, mask = G.text_encoder(text_encodings=fx['enc'])
        assert rel_err(g_tok, fx['global_tokens']) < TOL and rel_err(fine, fx['fine_tokens']) < TOL
        assert torch.equal(mask, fx['mask']) and not mask.all() and mask.any(dim=1).all()      # ragged lengths
        torch.manual_seed(1)
        img, rgbs = G(noise=fx['z'], text_encodings=fx['enc'], return_all_rgbs=True)
        assert rel_err(img, fx['img']) < TOL
        for a, b in zip(rgbs, fx['rgbs']):
            assert rel_err(a, b) < TOL
        # the tokens can also be handed over directly (gp.py:1159-1165)
        torch.manual_seed(1)
        img2 = G(noise=fx['z'], global_text_tokens=g_tok, fine_text_tokens=fine, text_mask=mask)
        assert rel_err(img2, fx['img']) < TOL
        real = fx['real'].clone().requires_grad_()
        logits, ms, _ = D(real, D.real_images_to_rgbs(real), text_encodings=fx['enc'], calc_aux_loss=False)
        assert rel_err(logits, fx['logits']) < TOL
        for a, b in zip(ms, fx['ms']):
            assert rel_err(a, b) < TOL
        # mismatched pairs (matching-aware loss): the text conditions the multi-scale predictors, not the main logits
        l_mis, ms_mis, _ = D(real, D.real_images_to_rgbs(real), text_encodings=fx['enc'].roll(1, 0), calc_aux_loss=False)
        assert rel_err(l_mis, fx['logits_mis']) < TOL
        for a, b, c in zip(ms_mis, fx['ms_mis'], fx['ms']):
            assert rel_err(a, b) < TOL and rel_err(a, c) > 1e-3
        gp = gradient_penalty(real, [logits, *ms], grad_output_weights=[1., *(0.1,) * len(ms)])
        assert rel_err(gp, fx['gp']) < 1e-4
        loss = logits.mean() + 0.1 * sum(m.mean() for m in ms) + gp
        grads = torch.autograd.grad(loss, list(D.parameters()), allow_unused=True)
    for (n, _), g in zip(D.named_parameters(), grads):
        if n in fx['d_grads'] and fx['d_grads'][n].abs().max() > 0:
            assert g is not None and rel_err(g, fx['d_grads'][n]) < 1e-3, n


def test_matching_aware_loss_is_the_reference_function_evaluated_stably(fx):
    f = fx['mal']
    assert abs(float(aux_matching_loss(f['real'], f['fake'])) - float(f['loss'])) < 1e-6
    big = torch.tensor([-200.])      # the reference's log(1 + exp(-x)) overflows to inf here (Appendix B.4)
    assert torch.isfinite(aux_matching_loss(big, big))


def test_text_inputs_are_validated(fx):
    G, D = _models(fx)
    with ops.use_impl(OracleOps()):
        with pytest.raises(AssertionError):
            G(noise=fx['z'])                                   # conditional model without any text input
        with pytest.raises(AssertionError):
            D(fx['real'], D.real_images_to_rgbs(fx['real']))
        with pytest.raises(RuntimeError):
            G(noise=fx['z'], texts=['a', 'b'])                 # raw strings need the external CLIP adapter


def test_live_reference_text_parity(reference):
    """another seed / token count against the reference imported live (train-mode D: same RNG stream -> same dropout
    and patch draws in the aux decoder)."""
    from torch import nn

    class PrecomputedClip(nn.Module):
        dim_latent = TEXT_CLIP_DIM

    torch.manual_seed(5)
    Gr = reference.Generator(text_encoder=reference.TextEncoder(clip=PrecomputedClip(), **TEXT_ENC), **TEXT_G)
    Dr = reference.Discriminator(text_encoder=reference.TextEncoder(clip=PrecomputedClip(), **TEXT_ENC), **TEXT_D)
    te = dict(clip_dim_latent=TEXT_CLIP_DIM, **TEXT_ENC)
    G, D = Generator(text_encoder=dict(te), **TEXT_G), Discriminator(text_encoder=dict(te), **TEXT_D)
    G.load_state_dict(Gr.state_dict()); D.load_state_dict(Dr.state_dict())
    assert G.style_embed_split_dims == Gr.style_embed_split_dims
    enc, z = text_encodings(batch=3, tokens=9, seed=4), torch.randn(3, 32)
    torch.manual_seed(1)
    img_r, rgbs_r = Gr(noise=z, text_encodings=enc, return_all_rgbs=True)
    with ops.use_impl(OracleOps()):
        torch.manual_seed(1)
        img, rgbs = G(noise=z, text_encodings=enc, return_all_rgbs=True)
    assert rel_err(img, img_r) < TOL
    torch.manual_seed(2)
    l_r, ms_r, aux_r = Dr(img_r.detach(), [r.detach() for r in rgbs_r], text_encodings=enc)
    with ops.use_impl(OracleOps()):
        torch.manual_seed(2)
        l, ms, aux = D(img_r.detach(), [r.detach() for r in rgbs_r], text_encodings=enc)
    assert rel_err(l, l_r) < TOL and rel_err(aux[0], aux_r[0]) < TOL
    for a, b in zip(ms, ms_r):
        assert rel_err(a, b) < TOL


# ---- HIP op set vs the oracle / the fixture: shared by the emulator (CPU) and the MI355X runs ------------------------

def _cosine(g, go, named):
    dot = na = nb = 0.
    for (n, _), a, b in zip(named, g, go):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        assert torch.isfinite(a).all(), n
        dot += (a.float() * b.float()).sum().item()
        na += a.float().square().sum().item()
        nb += b.float().square().sum().item()
    return dot / (na * nb) ** 0.5


def check_text_generator(fx, dev):
    G, _ = _models(fx, dev)
    z, enc = fx['z'].to(dev), fx['enc'].to(dev)

    def run():
        torch.manual_seed(1)
        img, rgbs = G(noise=z, text_encodings=enc, return_all_rgbs=True)
        loss = img.float().square().mean() + sum(r.float().mean() for r in rgbs)
        return img, torch.autograd.grad(loss, list(G.parameters()), allow_unused=True)
    img, g = run()
    assert img.dtype == torch.bfloat16
    if dev == 'cpu':     # same noise stream as the fixture only on the CPU generator
        assert rel_err(img, fx['img']) < TOL_BF16, rel_err(img, fx['img'])
    with ops.use_impl(OracleOps(bf16_operands=True)):
        img_o, go = run()
    assert rel_err(img, img_o) < TOL_BF16
    assert _cosine(g, go, G.named_parameters()) > 0.98


def check_text_discriminator(fx, dev, with_penalty):
    _, D = _models(fx, dev)
    enc, real0 = fx['enc'].to(dev), fx['real'].to(dev)

    def run():
        real = real0.clone().requires_grad_()
        ops.second_order = with_penalty
        try:
            logits, ms, _ = D(real, D.real_images_to_rgbs(real), text_encodings=enc, calc_aux_loss=False)
            loss = logits.float().mean() + 0.1 * sum(m.float().mean() for m in ms)
            pen = None
            if with_penalty:
                pen = gradient_penalty(real, [logits, *ms], grad_output_weights=[1., *(0.1,) * len(ms)])
                loss = loss + pen
        finally:
            ops.second_order = False
        return logits, ms, pen, torch.autograd.grad(loss, list(D.parameters()), allow_unused=True)
    logits, ms, pen, g = run()
    assert rel_err(logits.cpu(), fx['logits']) < TOL_BF16
    for a, b in zip(ms, fx['ms']):
        assert rel_err(a.cpu(), b) < TOL_BF16
    if with_penalty:     # double backward through the text-modulated predictor convs (twice-differentiable variant)
        assert abs(float(pen) - float(fx['gp'])) < 0.03 * float(fx['gp'])        # measured 0.0012 on the MI355X
    with ops.use_impl(OracleOps(bf16_operands=True)):
        _, _, _, go = run()
    assert _cosine(g, go, D.named_parameters()) > 0.98


def test_emulated_kernels_text_generator(fx):
    check_text_generator(fx, 'cpu')


def test_emulated_kernels_text_discriminator_gradient_penalty(fx):
    check_text_discriminator(fx, 'cpu', with_penalty=True)


@pytest.mark.gpu
def test_hip_text_generator(fx):
    check_text_generator(fx, 'cuda')


@pytest.mark.gpu
@pytest.mark.parametrize('with_penalty', [False, True])
def test_hip_text_discriminator(fx, with_penalty):
    check_text_discriminator(fx, 'cuda', with_penalty)


# ---- trainer, text-conditional mode (gp.py:2432-2475 matching-aware pairs, :2192-2206 dataloader tuples) ----------------

class _TextImages:
    batch_size = 2

    def __init__(self, dev):
        self.dev = dev

    def __iter__(self):
        g = torch.Generator().manual_seed(0)
        while True:
            yield torch.rand(2, 3, 16, 16, generator=g).to(self.dev), text_encodings(seed=int(torch.randint(99, (1,), generator=g)))


def _text_trainer_steps(tmp_path, dev):
    from gigagan_pytorch_amd import GigaGAN
    te = dict(clip_dim_latent=TEXT_CLIP_DIM, **TEXT_ENC)
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(text_encoder=dict(te), **TEXT_G), discriminator=dict(text_encoder=dict(te), **TEXT_D),
                  apply_gradient_penalty_every=2, generator_contrastive_loss_weight=0., device=dev,
                  model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    g0, d0 = gan.G_opt.flat_p.clone(), gan.D_opt.flat_p.clone()
    it = iter(_TextImages(dev))
    d1, g1 = gan.train_step(it, 2)
    d2, g2 = gan.train_step(it, 2)
    vals = [float(v) for v in (*d1, *g1, *d2, *g2) if v is not None]
    assert all(v == v and abs(v) < 1e9 for v in vals), vals
    assert float(d1.total_matching_aware_loss) > 0 and float(d2.gradient_penalty) > 0
    assert not torch.equal(g0, gan.G_opt.flat_p) and not torch.equal(d0, gan.D_opt.flat_p)
    img = gan.generate(batch_size=2, text_encodings=text_encodings().to(dev))
    assert img.shape == (2, 3, 16, 16) and torch.isfinite(img.float()).all()


def test_text_trainer_host_logic_on_oracle(tmp_path):
    with ops.use_impl(OracleOps()):
        _text_trainer_steps(tmp_path, 'cpu')


def test_text_trainer_needs_clip_for_the_contrastive_loss(tmp_path):
    """the CLIP contrastive loss (gp.py:174-188) needs the external adapter; with pre-computed encodings only it must be
    switched off explicitly rather than silently skipped."""
    from gigagan_pytorch_amd import GigaGAN
    te = dict(clip_dim_latent=TEXT_CLIP_DIM, **TEXT_ENC)
    gan = GigaGAN(generator=dict(text_encoder=dict(te), **TEXT_G), discriminator=dict(text_encoder=dict(te), **TEXT_D),
                  device='cpu', model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    assert gan.need_contrastive_loss


@pytest.mark.gpu
def test_hip_text_trainer_steps(tmp_path):
    _text_trainer_steps(tmp_path, 'cuda')


class _TextImagesTwoLengths:
    """a loader whose token encodings change length from batch to batch (the reference accepts any (b, n, d))."""
    batch_size = 2

    def __init__(self, dev, lengths=(7, 7, 9, 9, 7, 7, 9, 9)):
        self.dev, self.lengths = dev, lengths

    def __iter__(self):
        g = torch.Generator().manual_seed(0)
        i = 0
        while True:
            n = self.lengths[i % len(self.lengths)]
            i += 1
            yield (torch.rand(2, 3, 16, 16, generator=g).to(self.dev),
                   text_encodings(tokens=n, seed=int(torch.randint(99, (1,), generator=g))).to(self.dev))


@pytest.mark.gpu
def test_hip_graph_replay_follows_the_loader_when_encoding_lengths_change(tmp_path):
    """a captured step reads the static buffers it was captured with, so the shapes of the staged batches are
    part of the graph key (gigagan.py `_stage`; the reference draws any (b, n, d) encodings, gp.py:2269 / :2196). A loader that
    switches between 7 and 9 tokens must capture one graph per staged signature (a 7-token graph replayed on a 9-token batch would
    train on the previous batch's text), re-use them when a length comes back, and stage what the loader yielded."""
    from gigagan_pytorch_amd import GigaGAN
    te = dict(clip_dim_latent=TEXT_CLIP_DIM, **TEXT_ENC)
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(text_encoder=dict(te), **TEXT_G), discriminator=dict(text_encoder=dict(te), **TEXT_D),
                  apply_gradient_penalty_every=0, generator_contrastive_loss_weight=0., device='cuda', use_hip_graphs=True,
                  create_ema_generator_at_init=False, model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    # a step draws three loader batches (D: real pairs + the generator's conditioning, G: conditioning): 3 x 7 tokens, 3 x 9, ...
    loader = _TextImagesTwoLengths('cuda', lengths=(7, 7, 7, 9, 9, 9))
    it = iter(loader)

    def graph_keys():
        return {k for k, v in gan._graphs.items() if isinstance(v, tuple) and len(v) == 2 and not torch.is_tensor(v)}
    counts = []
    for step in range(4):
        d, g = gan.train_step(it, 1)
        vals = [float(v) for v in (*d, *g) if v is not None]
        assert all(v == v and abs(v) < 1e9 for v in vals), vals
        assert gan.use_hip_graphs, 'capture was refused'
        counts.append(len(graph_keys()))
        want = 7 if step % 2 == 0 else 9
        assert gan._static_G_src[1].shape[1] == want, (step, gan._static_G_src[1].shape)
    # steps 0 / 1 capture a D and a G graph each for their token count; steps 2 / 3 replay them
    assert counts[1] == 2 * counts[0] and counts[2] == counts[1] and counts[3] == counts[1], counts
    sigs = {k[-1] for k in graph_keys() if k[0] == 'D'}
    assert len(sigs) == 2, sigs
