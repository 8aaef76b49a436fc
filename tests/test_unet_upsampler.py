"""UnetUpsampler (BASELINE config 5): the oracle's unet ops and our host-side model assembly against
the reference's own outputs (tests/golden/unet_small.pt from tests/golden/make_golden_unet.py, live reference where
/root/reference exists), the HIP op set against the oracle on the kernel emulator (CPU) and on the MI355X (-m gpu)."""
from pathlib import Path

import pytest
import torch

from gigagan_pytorch_amd import ops
from gigagan_pytorch_amd import unet_upsampler as uu
from gigagan_pytorch_amd.unet_upsampler import UnetUpsampler
from oracle.torch_ops import OracleOps
from helpers import rel_err, UNET_SMALL

GOLD = Path(__file__).resolve().parent / 'golden'
TOL = 1e-5          # fp32 restatement vs fp32 reference
TOL_BF16 = 3e-2     # bf16 operands and activations end to end (~40 layers) vs the fp32 reference, as for G and D

PARTS = {
    'linattn': lambda: uu.LinearAttention(16, heads=2, dim_head=8),
    'attn': lambda: uu.Attention(16, heads=2, dim_head=8),
    'lintr': lambda: uu.LinearTransformer(16, heads=2, dim_head=8),
    'tr': lambda: uu.Transformer(16, heads=2, dim_head=8),
}


@pytest.fixture(scope='module')
def fx():
    return torch.load(GOLD / 'unet_small.pt', weights_only=False)


def _resnet(fx):
    blk = uu.ResnetBlock(24, 16, num_conv_kernels=2)
    blk.load_state_dict(fx['resnet']['state'])
    return blk


# ---- oracle + host logic vs the reference (fp32) ---------------------------------------------------------------

@pytest.mark.parametrize('name', list(PARTS))
def test_oracle_unet_blocks_match_reference(fx, name):
    f = fx[name]
    mod = PARTS[name]()
    assert list(mod.state_dict().keys()) == list(f['state'].keys())
    mod.load_state_dict(f['state'])
    with ops.use_impl(OracleOps()):
        assert rel_err(mod(f['x']), f['y']) < TOL


def test_oracle_downsample_and_resnet_block_match_reference(fx):
    f = fx['down']
    down = uu.Downsample(16, 24)
    down.load_state_dict(f['state'])
    with ops.use_impl(OracleOps()):
        y, hf = down(f['x'])
        assert rel_err(y, f['y']) < TOL and rel_err(hf, f['hf']) < TOL
        f = fx['resnet']
        assert rel_err(_resnet(fx)(f['x'], conv_mods_iter=iter(f['mods'])), f['y']) < TOL


def test_unet_on_oracle_matches_reference_fixture(fx):
    f = fx['unet']
    U = UnetUpsampler(**UNET_SMALL)
    assert list(U.state_dict().keys()) == list(f['state'].keys())
    assert [tuple(v.shape) for v in U.state_dict().values()] == [tuple(v.shape) for v in f['state'].values()]
    assert U.style_embed_split_dims == f['split_dims'] and U.allowable_rgb_resolutions == [8, 16]
    U.load_state_dict(f['state'])
    with ops.use_impl(OracleOps()):
        img, rgbs = U(lowres_image=f['lowres'], noise=f['z'], return_all_rgbs=True)     # the trainer's keyword (gp.py:2212)
        img2 = U(f['lowres'], noise=f['z'])
    assert rel_err(img, f['img']) < TOL and torch.equal(img, img2)
    assert len(rgbs) == len(f['rgbs']) and torch.equal(rgbs[0], f['lowres'])
    for a, b in zip(rgbs, f['rgbs']):
        assert a.shape == b.shape and rel_err(a, b) < TOL


def test_live_reference_unet_parity(reference):
    """a second configuration against the reference imported live (three no-downsample stages, one linear stage)."""
    cfg = dict(dim=8, image_size=64, input_image_size=8, style_network=dict(dim=16, depth=2), dim_mults=(1, 2, 2, 4),
               full_attn=(False, False, False, True), self_attn_dim_head=8, self_attn_heads=2, cross_attn_dim_head=8,
               unconditional=True)
    torch.manual_seed(3)
    Ur = reference.UnetUpsampler(**cfg)
    U = UnetUpsampler(**cfg)
    assert list(U.state_dict().keys()) == list(Ur.state_dict().keys())
    U.load_state_dict(Ur.state_dict())
    assert U.total_params == Ur.total_params
    x, z = torch.rand(1, 3, 8, 8), torch.randn(1, 16)
    with torch.no_grad():
        img_r, rgbs_r = Ur(x, noise=z, return_all_rgbs=True)
        with ops.use_impl(OracleOps()):
            img, rgbs = U(x, noise=z, return_all_rgbs=True)
    assert rel_err(img, img_r) < TOL
    for a, b in zip(rgbs, rgbs_r):
        assert rel_err(a, b) < TOL


def test_video_layers_are_refused():
    with pytest.raises(NotImplementedError):
        UnetUpsampler(**{**UNET_SMALL, 'has_temporal_layers': True})


# ---- HIP op set vs the oracle: shared by the emulator (CPU) and the MI355X runs --------------------------------

def _check_block_gradients(name, mod, x, call, dev, tol=2.5e-2):
    mod = mod.to(dev)
    x = x.to(dev).requires_grad_()
    ps = [x] + list(mod.parameters())
    with ops.use_impl(OracleOps(bf16_operands=True)):
        yo = call(mod, x)
        yo = yo if isinstance(yo, tuple) else (yo,)
        cs = [torch.randn_like(t) for t in yo]
        go = torch.autograd.grad(sum((t * c).sum() for t, c in zip(yo, cs)), ps)
    y = call(mod, x)
    y = y if isinstance(y, tuple) else (y,)
    g = torch.autograd.grad(sum((t.float() * c).sum() for t, c in zip(y, cs)), ps)
    for a, b in zip(y, yo):
        assert a.dtype == torch.bfloat16 and rel_err(a, b) < 1e-2, (name, rel_err(a, b))
    for a, b in zip(g, go):
        assert rel_err(a, b) < tol, (name, tuple(b.shape), rel_err(a, b))


def check_unet_blocks_first_order(fx, dev):
    torch.manual_seed(0)
    for name in PARTS:
        mod = PARTS[name]()
        mod.load_state_dict(fx[name]['state'])
        _check_block_gradients(name, mod, fx[name]['x'], lambda m, x: m(x), dev)
    down = uu.Downsample(16, 24)
    down.load_state_dict(fx['down']['state'])
    # max-pool winners can flip between a bf16 and an fp32 conv output: a few routed gradients differ outright
    _check_block_gradients('down', down, fx['down']['x'], lambda m, x: m(x), dev, tol=8e-2)
    mods = [t.to(dev) for t in fx['resnet']['mods']]
    _check_block_gradients('resnet', _resnet(fx), fx['resnet']['x'], lambda m, x: m(x, conv_mods_iter=iter(mods)), dev)


def check_unet_forward(fx, dev):
    f = fx['unet']
    U = UnetUpsampler(**UNET_SMALL)
    U.load_state_dict(f['state'])
    U = U.to(dev)
    with torch.no_grad():
        img, rgbs = U(f['lowres'].to(dev), noise=f['z'].to(dev), return_all_rgbs=True)
    assert img.dtype == torch.bfloat16 and img.shape == f['img'].shape
    assert rel_err(img.cpu(), f['img']) < TOL_BF16, rel_err(img.cpu(), f['img'])
    for a, b in zip(rgbs[1:], f['rgbs'][1:]):
        assert rel_err(a.cpu(), b) < TOL_BF16
    return U


def test_emulated_kernels_unet_blocks_first_order(fx):
    check_unet_blocks_first_order(fx, 'cpu')


def test_emulated_kernels_unet_forward_vs_reference_fixture(fx):
    check_unet_forward(fx, 'cpu')


@pytest.mark.gpu
def test_hip_unet_blocks_first_order(fx):
    check_unet_blocks_first_order(fx, 'cuda')


@pytest.mark.gpu
def test_hip_unet_forward_and_training_gradients(fx):
    """images vs the reference's fp32 fixture; a training-style backward produces finite gradients for every parameter
    and, summed over the model, points the same way as the oracle's."""
    U = check_unet_forward(fx, 'cuda')
    f = fx['unet']
    lowres, z = f['lowres'].cuda(), f['z'].cuda()
    params = list(U.parameters())

    def grads():
        img, rgbs = U(lowres, noise=z, return_all_rgbs=True)
        loss = img.float().square().mean() + sum(r.float().mean() for r in rgbs[1:])
        return torch.autograd.grad(loss, params, allow_unused=True)
    g = grads()
    with ops.use_impl(OracleOps(bf16_operands=True)):
        go = grads()
    dot = na = nb = 0.
    for (n, _), a, b in zip(U.named_parameters(), g, go):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        assert torch.isfinite(a).all(), n
        dot += (a.float() * b.float()).sum().item()
        na += a.float().square().sum().item()
        nb += b.float().square().sum().item()
    assert dot / (na * nb) ** 0.5 > 0.99          # measured 0.9989 on the MI355X


# ---- trainer with train_upsampler=True (gp.py:1912-1960, :2208-2212) ------------------------------------------------

UP_G = dict(dim=8, image_size=16, input_image_size=8, style_network=dict(dim=16, depth=2), dim_mults=(1, 2),
            full_attn=(False, True), self_attn_dim_head=8, self_attn_heads=2, cross_attn_dim_head=8, unconditional=True)
UP_D = dict(image_size=16, dim_capacity=8, dim_max=32, unconditional=True, attn_resolutions=(8,), attn_heads=2,
            attn_dim_head=16, multiscale_input_resolutions=(8,))


def _upsampler_trainer_steps(tmp_path, dev):
    from gigagan_pytorch_amd import GigaGAN
    from gigagan_pytorch_amd.data import SyntheticImages
    from gigagan_pytorch_amd.gigagan import cycle
    torch.manual_seed(0)
    gan = GigaGAN(train_upsampler=True, generator=dict(UP_G), discriminator=dict(UP_D), apply_gradient_penalty_every=2,
                  device=dev, model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    g0, d0 = gan.G_opt.flat_p.clone(), gan.D_opt.flat_p.clone()
    it = cycle(SyntheticImages(2, 16, device=dev))
    d1, g1 = gan.train_step(it, 2)          # plain step
    d2, g2 = gan.train_step(it, 2)          # gradient-penalty step (double backward through the discriminator)
    vals = [float(v) for v in (*d1, *g1, *d2, *g2) if v is not None]
    assert all(v == v and abs(v) < 1e9 for v in vals), vals
    assert float(d2.gradient_penalty) > 0 and float(d1.gradient_penalty) == 0
    assert not torch.equal(g0, gan.G_opt.flat_p) and not torch.equal(d0, gan.D_opt.flat_p)
    if dev != 'cpu':                        # on the GPU the step kinds are hipGraphs by now: replay each of them again
        assert gan.use_hip_graphs and len(gan._graphs) > 0
        for _ in range(3):
            dn, gn = gan.train_step(it, 2)
            vals = [float(v) for v in (*dn, *gn) if v is not None]
            assert all(v == v and abs(v) < 1e9 for v in vals), vals
        assert torch.isfinite(gan.G_opt.flat_p).all() and torch.isfinite(gan.D_opt.flat_p).all()
        assert gan.use_hip_graphs, 'a capture was refused: the upsampler step fell back to eager launches'
    lowres = torch.rand(2, 3, 8, 8, device=dev)
    img = gan.generate(lowres_image=lowres)
    assert img.shape == (2, 3, 16, 16) and torch.isfinite(img.float()).all()
    return vals


def test_upsampler_trainer_host_logic_on_oracle(tmp_path):
    """the trainer's upsampler mode end to end (low-res conditioning from the real batch, rgbs restricted to the allowed
    resolutions, D multi-scale inputs) on the fp32 oracle ops; the same steps run on the HIP kernels in the -m gpu test."""
    with ops.use_impl(OracleOps()):
        _upsampler_trainer_steps(tmp_path, 'cpu')


def test_upsampler_trainer_rejects_disallowed_multiscale_resolutions(tmp_path):
    from gigagan_pytorch_amd import GigaGAN
    with pytest.raises(AssertionError):
        GigaGAN(train_upsampler=True, generator=dict(UP_G), discriminator={**UP_D, 'multiscale_input_resolutions': (4,)},
                device='cpu', model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))


@pytest.mark.gpu
def test_hip_upsampler_trainer_steps(tmp_path):
    _upsampler_trainer_steps(tmp_path, 'cuda')
