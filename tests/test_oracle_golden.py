"""CPU suite: the oracle restatement and our host-side model logic against the reference's own outputs
(committed golden fixtures from tests/golden/make_golden.py; live reference where /root/reference exists)."""
from pathlib import Path

import pytest
import torch

from gigagan_pytorch_amd import ops
from gigagan_pytorch_amd.generator import Generator
from gigagan_pytorch_amd.discriminator import Discriminator
from gigagan_pytorch_amd.gigagan import gradient_penalty
from gigagan_pytorch_amd.modules import SelfAttention
from oracle.torch_ops import OracleOps
from helpers import rel_err, SMALL_G, SMALL_D, C1_G, C1_D

GOLD = Path(__file__).resolve().parent / 'golden'
TOL = 1e-5   # fp32 restatement vs fp32 reference


@pytest.fixture(scope='module')
def fx_ops():
    return torch.load(GOLD / 'ops_small.pt', weights_only=False)


@pytest.fixture(scope='module')
def fx_model():
    return torch.load(GOLD / 'model_small.pt', weights_only=False)


def test_oracle_modconv_matches_reference(fx_ops):
    O = OracleOps()
    f = fx_ops['modconv']
    assert rel_err(O.modconv2d(f['x'], f['weights'], f['mod'], f['kernel_mod']), f['y']) < TOL
    f = fx_ops['torgb']
    assert rel_err(O.modconv2d(f['x'], f['weights'], f['mod'], None, demod=False), f['y']) < TOL


def test_demod_gram_identity_matches_reference(fx_ops):
    """the Gram-matrix demodulation used by the HIP path equals the reference's per-sample-weight norm."""
    f = fx_ops['modconv']
    w = f['weights']
    s = f['mod'] + 1
    a = f['kernel_mod'].softmax(-1)
    d = ops.demod_coefficients(w, s, a, 1e-8)
    wb = (w[None] * a[:, :, None, None, None, None]).sum(1) * s[:, None, :, None, None]
    ref = wb.pow(2).sum(dim=(2, 3, 4)).clamp(min=1e-8).rsqrt()
    assert rel_err(d, ref) < 1e-5


@pytest.mark.parametrize('dot', [0, 1])
def test_oracle_self_attention_matches_reference(fx_ops, dot):
    f = fx_ops[f'selfattn_dot{dot}']
    attn = SelfAttention(16, dim_head=8, heads=2, dot_product=bool(dot))
    attn.load_state_dict(f['state'])
    with ops.use_impl(OracleOps()):
        y = attn(f['x'])
    assert rel_err(y, f['y']) < TOL


def test_oracle_stencils_and_norm_match_reference(fx_ops):
    O = OracleOps()
    assert rel_err(O.upsample_blur(fx_ops['upsample']['x']), fx_ops['upsample']['y']) < TOL
    f = fx_ops['rmsnorm']
    assert rel_err(O.channel_rmsnorm(f['x'], f['gamma']), f['y']) < TOL
    f = fx_ops['resize']
    assert rel_err(O.resize_bilinear(f['x'], 8), f['y8']) < TOL
    assert rel_err(O.resize_bilinear(f['x'], 16), f['y16']) < TOL


def test_models_on_oracle_match_reference_fixture(fx_model):
    fx = fx_model
    G, D = Generator(**SMALL_G), Discriminator(**SMALL_D)
    assert list(G.state_dict().keys()) == list(fx['G'].keys())
    assert list(D.state_dict().keys()) == list(fx['D'].keys())
    G.load_state_dict(fx['G']); D.load_state_dict(fx['D'])
    with ops.use_impl(OracleOps()):
        torch.manual_seed(1)
        img, rgbs = G(noise=fx['z'], return_all_rgbs=True)
        assert rel_err(img, fx['img']) < TOL
        for a, b in zip(rgbs, fx['rgbs']):
            assert rel_err(a, b) < TOL
        D.eval()
        real = fx['real'].clone().requires_grad_()
        logits, ms, _ = D(real, D.real_images_to_rgbs(real), calc_aux_loss=False)
        assert rel_err(logits, fx['logits']) < TOL
        for a, b in zip(ms, fx['ms']):
            assert rel_err(a, b) < TOL
        gp = gradient_penalty(real, [logits, *ms], grad_output_weights=[1., *(0.1,) * len(ms)])
        assert rel_err(gp, fx['gp']) < 1e-4
        loss = logits.mean() + 0.1 * sum(m.mean() for m in ms) + gp
        grads = torch.autograd.grad(loss, list(D.parameters()), allow_unused=True)
    for (n, _), g in zip(D.named_parameters(), grads):
        if n in fx['d_grads']:
            ref = fx['d_grads'][n]
            if ref.abs().max() > 0:
                assert g is not None and rel_err(g, ref) < 1e-3, n


def test_live_reference_parity_config1(reference):
    """BASELINE config 1 (64x64, capacity 8, batch 2) against the reference imported live: state-dict keys,
    generator images, discriminator logits (train mode, same RNG stream -> same dropout / patch draws)."""
    torch.manual_seed(0)
    Gr, Dr = reference.Generator(**C1_G), reference.Discriminator(**C1_D)
    G, D = Generator(**C1_G), Discriminator(**C1_D)
    assert list(G.state_dict().keys()) == list(Gr.state_dict().keys())
    assert list(D.state_dict().keys()) == list(Dr.state_dict().keys())
    assert [tuple(v.shape) for v in G.state_dict().values()] == [tuple(v.shape) for v in Gr.state_dict().values()]
    G.load_state_dict(Gr.state_dict()); D.load_state_dict(Dr.state_dict())
    assert G.style_embed_split_dims == Gr.style_embed_split_dims
    z = torch.randn(2, 64)
    torch.manual_seed(1)
    img_r, rgbs_r = Gr(noise=z, return_all_rgbs=True)
    with ops.use_impl(OracleOps()):
        torch.manual_seed(1)
        img, rgbs = G(noise=z, return_all_rgbs=True)
    assert rel_err(img, img_r) < TOL
    Dr.train(); D.train()
    torch.manual_seed(2)
    l_r, ms_r, aux_r = Dr(img_r.detach(), [r.detach() for r in rgbs_r])
    with ops.use_impl(OracleOps()):
        torch.manual_seed(2)
        l, ms, aux = D(img_r.detach(), [r.detach() for r in rgbs_r])
    assert rel_err(l, l_r) < TOL and rel_err(aux[0], aux_r[0]) < TOL
    for a, b in zip(ms, ms_r):
        assert rel_err(a, b) < TOL


def test_optimizer_is_adamw_with_reference_grouping(reference):
    """Appendix B.1: the reference's effective optimizer is AdamW(wd=1e-2) with no decay on ndim<2 params."""
    from gigagan_pytorch_amd.optimizer import separate_weight_decayable_params
    from gigagan_pytorch.optimizer import get_optimizer as ref_get
    G = Generator(**SMALL_G)
    ro = ref_get(G.parameters(), lr=2e-4, betas=(0.5, 0.9), weight_decay=0.)
    assert type(ro).__name__ == 'AdamW' and ro.param_groups[0]['weight_decay'] == 1e-2
    wd, no_wd = separate_weight_decayable_params(list(G.parameters()))
    assert len(ro.param_groups[0]['params']) == len(wd) and len(ro.param_groups[1]['params']) == len(no_wd)
