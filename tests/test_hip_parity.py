"""GPU suite (-m gpu): the gfx950 kernels through the C ABI against the oracle, the reference golden fixtures,
and size-independent properties at BASELINE config-2 shapes. Tolerances: the kernels compute bf16 x bf16 products
with fp32 accumulation; against an oracle that rounds the same operands to bf16 the only differences are the
accumulation order (fp32 outputs: 1e-4 relative L2) and the final bf16 store (4e-3)."""
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

from gigagan_pytorch_amd import kernels as K, ops, GigaGAN
from gigagan_pytorch_amd.generator import Generator
from gigagan_pytorch_amd.discriminator import Discriminator
from gigagan_pytorch_amd.gigagan import gradient_penalty, cycle
from gigagan_pytorch_amd.data import SyntheticImages
from oracle.torch_ops import OracleOps
from helpers import rel_err, bf, SMALL_G, SMALL_D, C1_G, C1_D

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / 'golden'
F32_TOL, BF16_TOL = 1e-4, 4e-3


def dev():
    return torch.device('cuda', 0)


def test_native_library_is_the_gfx950_build():
    from gigagan_pytorch_amd import _C
    assert _C.lib().is_emulator is False
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        K.gemm(bf(torch.randn(8, 8)), bf(torch.randn(8, 8)))


@pytest.mark.parametrize('shape', [(130, 70, 104, 2), (257, 129, 40, 3), (1024, 1032, 64, 4), (512, 512, 4608, 1)])
def test_gemm_vs_cpu_oracle(shape):
    M, N, Kd, batch = shape
    torch.manual_seed(0)
    a = bf(torch.randn(batch, M, Kd)); b = bf(torch.randn(batch, N, Kd))
    ref = torch.einsum('bmk,bnk->bmn', a.float(), b.float())          # CPU fp32 on bf16-rounded operands
    ad, bd = a.to(dev()), b.to(dev())
    for tile in (0, 1, 2, 3):
        assert rel_err(K.gemm(ad, bd, out_dtype=torch.float32, force_tile=tile).cpu(), ref) < F32_TOL
    assert rel_err(K.gemm(ad, bd, out_dtype=torch.float32, force_splitk=2).cpu(), ref) < F32_TOL
    assert rel_err(K.gemm(ad, bd).cpu(), ref) < BF16_TOL
    if M % 8 == 0 and N % 8 == 0:
        at, bt = ad.transpose(1, 2).contiguous(), bd.transpose(1, 2).contiguous()
        for ta, tb, x, y in ((True, True, at, bd), (False, False, ad, bt), (True, False, at, bt)):
            assert rel_err(K.gemm(x, y, trans_a=ta, trans_b=tb, out_dtype=torch.float32).cpu(), ref) < F32_TOL


def test_gemm_k_tail_and_epilogue():
    torch.manual_seed(0)
    a = bf(torch.randn(4, 256, 1032)); b = bf(torch.randn(4, 64, 1032)); bias = torch.randn(64)
    ref = torch.einsum('bmk,bnk->bmn', a.float()[..., :1025], b.float()[..., :1025])
    out = K.gemm(a.to(dev()), b.to(dev()), k_valid=1025, out_dtype=torch.float32)
    assert rel_err(out.cpu(), ref) < F32_TOL
    out = K.gemm(a.to(dev()), b.to(dev()), k_valid=1025, alpha=0.5, bias=bias.to(dev()), act='lrelu', out_dtype=torch.float32)
    assert rel_err(out.cpu(), F.leaky_relu(ref * 0.5 + bias, 0.2)) < F32_TOL


@pytest.mark.parametrize('cfg', [(2, 8, 8, 16, 24, 3), (3, 5, 7, 8, 40, 7), (2, 16, 16, 32, 136, 1), (4, 32, 32, 64, 64, 3),
                                 (2, 64, 64, 128, 128, 3), (1, 4, 4, 512, 512, 3)])
def test_conv_forward_dgrad_wgrad_vs_cpu_oracle(cfg):
    n, H, W, Ci, Co, ks = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, Ci, H, W)); w = bf(torch.randn(Co, Ci, ks, ks) * 0.1); dy = bf(torch.randn(n, Co, H, W))
    xf, wf = x.float().requires_grad_(), w.float().requires_grad_()
    ref = F.conv2d(xf, wf, padding=ks // 2)
    ref.backward(dy.float())
    xh, dyh = x.permute(0, 2, 3, 1).contiguous().to(dev()), dy.permute(0, 2, 3, 1).contiguous().to(dev())
    wh = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().to(dev())
    assert rel_err(K.conv2d_nhwc(xh, wh, ksize=ks, out_dtype=torch.float32).cpu().permute(0, 3, 1, 2), ref) < F32_TOL
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks).cpu(), wf.grad.permute(2, 3, 1, 0).reshape(-1, Co)) < F32_TOL
    wT = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, -1).contiguous().to(dev())
    assert rel_err(K.conv2d_nhwc(dyh, wT, ksize=ks, out_dtype=torch.float32).cpu().permute(0, 3, 1, 2), xf.grad) < F32_TOL


@pytest.mark.parametrize('cfg', [(4, 32, 32, 64, 128, 3, 5), (2, 64, 64, 128, 256, 3, 4), (8, 16, 16, 256, 512, 3, 4),
                                 (16, 8, 8, 512, 512, 3, 4), (4, 32, 32, 256, 1024, 1, 4)])
def test_large_tile_kernel_conv_vs_cpu_oracle(cfg):
    """the 8-wave 256-row tile kernel (gg_gemm2.h): forward, data gradient, weight gradient (with and without
    split-K) against fp32 CPU convolution of the same bf16 operands."""
    n, H, W, Ci, Co, ks, tile = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, Ci, H, W)); w = bf(torch.randn(Co, Ci, ks, ks) * 0.05); dy = bf(torch.randn(n, Co, H, W))
    xf, wf = x.float().requires_grad_(), w.float().requires_grad_()
    ref = F.conv2d(xf, wf, padding=ks // 2)
    ref.backward(dy.float())
    xh, dyh = x.permute(0, 2, 3, 1).contiguous().to(dev()), dy.permute(0, 2, 3, 1).contiguous().to(dev())
    wh = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().to(dev())
    out = K.conv2d_nhwc(xh, wh, ksize=ks, out_dtype=torch.float32, force_tile=tile)
    assert rel_err(out.cpu().permute(0, 3, 1, 2), ref) < F32_TOL
    dw_ref = wf.grad.permute(2, 3, 1, 0).reshape(-1, Co)
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks, force_tile=tile).cpu(), dw_ref) < F32_TOL
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks, force_tile=tile, force_splitk=1).cpu(), dw_ref) < F32_TOL
    wT = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, -1).contiguous().to(dev())
    dx = K.conv2d_nhwc(dyh, wT, ksize=ks, out_dtype=torch.float32, force_tile=tile)
    assert rel_err(dx.cpu().permute(0, 3, 1, 2), xf.grad) < F32_TOL


def test_strided_convs_and_depth_to_space_vs_cpu_oracle():
    torch.manual_seed(0)
    n, H, W, C, O = 4, 16, 16, 64, 128
    x = bf(torch.randn(n, C, H, W)); xh = x.permute(0, 2, 3, 1).contiguous().to(dev())
    w2 = bf(torch.randn(O, 4 * C, 1, 1) * 0.1)
    xf, wf = x.float().requires_grad_(), w2.float().requires_grad_()
    s2d = xf.reshape(n, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(n, 4 * C, H // 2, W // 2)
    ref = F.conv2d(s2d, wf)
    dy = bf(torch.randn_like(ref)); ref.backward(dy.float())
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev())
    w2h = w2.reshape(O, C, 2, 2).permute(0, 2, 3, 1).reshape(O, 4 * C).contiguous().to(dev())
    for tile in (0, 1, 5):
        out = K.conv2d_nhwc(xh, w2h, ksize=2, stride=2, pad=0, out_dtype=torch.float32, force_tile=tile)
        assert rel_err(out.cpu().permute(0, 3, 1, 2), ref) < F32_TOL
        dw = K.conv2d_wgrad_nhwc(xh, dyh, ksize=2, stride=2, pad=0, force_tile=tile)
        assert rel_err(dw.cpu(), wf.grad.reshape(O, C, 2, 2).permute(2, 3, 1, 0).reshape(4 * C, O)) < F32_TOL
        dx = K.conv2d_dgrad_d2s(dyh, w2h, cell=2, taps=2, force_tile=tile)
        assert rel_err(dx.cpu().permute(0, 3, 1, 2), xf.grad) < BF16_TOL
    w1 = bf(torch.randn(O, C, 1, 1) * 0.1)
    xf, wf = x.float().requires_grad_(), w1.float().requires_grad_()
    ref = F.conv2d(xf, wf, stride=2); ref.backward(dy.float())
    w1h = w1.reshape(O, C).contiguous().to(dev())
    out = K.conv2d_nhwc(xh, w1h, ksize=1, stride=2, pad=0, out_dtype=torch.float32)
    assert rel_err(out.cpu().permute(0, 3, 1, 2), ref) < F32_TOL
    dx = K.conv2d_dgrad_d2s(dyh, w1h, cell=2, taps=1)
    assert rel_err(dx.cpu().permute(0, 3, 1, 2), xf.grad) < BF16_TOL


@pytest.mark.parametrize('cfg', [(2, 1024, 8, True), (4, 256, 8, False), (1, 1024, 2, False)])
def test_fused_attention_vs_cpu_reference(cfg):
    """flash-style attention kernels at the model's shapes against fp32 tensor algebra on the same bf16 inputs."""
    B, n, h, l2 = cfg
    torch.manual_seed(0)
    scale = 64 ** -0.5
    q = bf(torch.randn(B, n, h * 64) * 0.7); v = bf(torch.randn(B, n, h * 64))
    k = q.clone() if l2 else bf(torch.randn(B, n, h * 64) * 0.7)
    k0 = bf(torch.randn(h, 64) * 0.7); v0 = bf(torch.randn(h, 64))
    alpha, beta = (2 * scale, -scale) if l2 else (scale, 0.)
    d_o = bf(torch.randn(B, n, h * 64))

    def ref(q, k, v, k0, v0):
        qh, kh, vh = (t.view(B, n, h, 64).permute(0, 2, 1, 3) for t in (q, k, v))
        kk = torch.cat((k0[None, :, None, :].expand(B, -1, -1, -1), kh), 2)
        vv = torch.cat((v0[None, :, None, :].expand(B, -1, -1, -1), vh), 2)
        x = alpha * qh @ kk.transpose(-1, -2) + beta * (kk * kk).sum(-1)[:, :, None, :]
        return (x.softmax(-1) @ vv).permute(0, 2, 1, 3).reshape(B, n, h * 64), x.logsumexp(-1).reshape(B * h, n)

    ins = [t.float().requires_grad_() for t in (q, k, v, k0, v0)]
    o_ref, lse_ref = ref(*ins)
    g = torch.autograd.grad(o_ref, ins, d_o.float())
    d = dev()
    qd, kd, vd, k0d, v0d, dod = (t.to(d) for t in (q, k, v, k0, v0, d_o))
    o, lse = K.attn_fwd(qd, kd, vd, k0d, v0d, h, alpha, beta)
    assert rel_err(o.cpu(), o_ref) < 6e-3 and rel_err(lse.cpu(), lse_ref) < 1e-4
    dq, dk, dv, dk0q, dv0, dbias0 = K.attn_bwd(qd, kd, vd, k0d, v0d, o, lse, dod, h, alpha, beta)
    assert rel_err(dq.cpu(), g[0]) < 2e-2 and rel_err(dk.cpu(), g[1]) < 2e-2 and rel_err(dv.cpu(), g[2]) < 2e-2
    dk0 = dk0q.cpu() + 2 * beta * dbias0.cpu()[:, None] * k0.float()
    assert rel_err(dk0, g[3]) < 2e-2 and rel_err(dv0.cpu(), g[4]) < 2e-2


@pytest.mark.parametrize('cfg', [(2, 256, 4, True), (1, 1024, 2, False)])
def test_fused_attention_second_order_vs_cpu_autograd(cfg):
    """gg_attn_bwd2 on the GPU vs double backward of fp32 tensor algebra on the same bf16 inputs."""
    B, n, h, l2 = cfg
    torch.manual_seed(0)
    scale = 64 ** -0.5
    mk = lambda *s, m=1.0: bf(torch.randn(*s) * m)
    q, v, k0, v0, d_o = mk(B, n, h * 64, m=0.7), mk(B, n, h * 64), mk(h, 64, m=0.7), mk(h, 64), mk(B, n, h * 64)
    k = q.clone() if l2 else mk(B, n, h * 64, m=0.7)
    aq, ak, av, ak0, av0 = mk(B, n, h * 64), mk(B, n, h * 64), mk(B, n, h * 64), mk(h, 64), mk(h, 64)
    alpha, beta = (2 * scale, -scale) if l2 else (scale, 0.)

    def fwd(q, k, v, k0, v0):
        qh, kh, vh = (t.view(B, n, h, 64).permute(0, 2, 1, 3) for t in (q, k, v))
        kk = torch.cat((k0[None, :, None, :].expand(B, -1, -1, -1), kh), 2)
        vv = torch.cat((v0[None, :, None, :].expand(B, -1, -1, -1), vh), 2)
        x = alpha * qh @ kk.transpose(-1, -2) + beta * (kk * kk).sum(-1)[:, :, None, :]
        return (x.softmax(-1) @ vv).permute(0, 2, 1, 3).reshape(B, n, h * 64)

    ins = [t.float().requires_grad_() for t in (q, k, v, k0, v0)]
    dof = d_o.float().requires_grad_()
    first = torch.autograd.grad(fwd(*ins), ins, dof, create_graph=True)
    F_ = sum((a.float() * g).sum() for a, g in zip((aq, ak, av, ak0, av0), first))
    ref = torch.autograd.grad(F_, [*ins, dof])
    d = dev()
    qd, kd, vd, k0d, v0d, dod, aqd, akd, avd, ak0d, av0d = (t.to(d) for t in (q, k, v, k0, v0, d_o, aq, ak, av, ak0, av0))
    o, lse = K.attn_fwd(qd, kd, vd, k0d, v0d, h, alpha, beta)
    *_, dvec = K.attn_bwd(qd, kd, vd, k0d, v0d, o, lse, dod, h, alpha, beta, return_dvec=True)
    outs = K.attn_bwd2(qd, kd, vd, k0d, v0d, dod, lse, dvec, aqd, akd, avd, ak0d, av0d, h, alpha, beta)
    gq, gk, gv, gdo, gk0, gv0 = (t.cpu() for t in outs)
    for name, a, b_ in zip(('gq', 'gk', 'gv', 'gk0', 'gv0', 'gdo'), (gq, gk, gv, gk0, gv0, gdo), ref):
        assert rel_err(a, b_) < 2e-2, (name, rel_err(a, b_))


def test_ops_match_reference_golden_fixture():
    fx = torch.load(GOLD / 'ops_small.pt', weights_only=False)
    H_ = ops.HipOps()
    d = dev()
    f = fx['modconv']
    with torch.no_grad():
        y = H_.modconv2d(f['x'].to(d), f['weights'].to(d), f['mod'].to(d), f['kernel_mod'].to(d))
    assert rel_err(y.cpu(), f['y']) < 1e-2            # bf16 kernel vs fp32 reference output
    y = H_.modconv2d(f['x'].to(d).requires_grad_(), f['weights'].to(d), f['mod'].to(d), f['kernel_mod'].to(d))
    assert rel_err(y.cpu(), f['y']) < 1e-2            # training (stacked-output) path
    f = fx['upsample']
    assert rel_err(H_.upsample_blur(f['x'].to(d)).cpu(), f['y']) < 1e-2
    f = fx['resize']
    assert rel_err(H_.resize_bilinear(f['x'].to(d), 8).cpu(), f['y8']) < 1e-2
    f = fx['rmsnorm']
    assert rel_err(H_.channel_rmsnorm(f['x'].to(d), f['gamma'].to(d)).cpu(), f['y']) < 1e-2


def test_ops_gradients_vs_oracle():
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps(bf16_operands=True)
    d = dev()

    def check(fn, inputs, tol=2e-2):
        ih = [t.clone().to(d).requires_grad_() for t in inputs]; io = [t.clone().requires_grad_() for t in inputs]
        yh, yo = fn(H_, *ih), fn(O_, *io)
        g = torch.randn_like(yo)
        gh = torch.autograd.grad(yh.float(), ih, g.to(d)); go = torch.autograd.grad(yo, io, bf(g).float())
        assert rel_err(yh.cpu(), yo) < tol
        for a, b in zip(gh, go):
            assert rel_err(a.cpu(), b) < 2 * tol

    x = torch.randn(2, 16, 8, 8)
    check(lambda I, x, w, b: I.conv2d(x, w, b, act='lrelu'), [x, torch.randn(24, 16, 3, 3) * 0.1, torch.randn(24)])
    check(lambda I, x, w: I.conv2d(x, w, None), [torch.randn(2, 3, 8, 8), torch.randn(16, 3, 7, 7) * 0.1])
    check(lambda I, x, w, b: I.linear(x, w, b), [torch.randn(6, 20), torch.randn(5, 20), torch.randn(5)])
    q, k, v = torch.randn(2, 2, 16, 16), torch.randn(2, 2, 17, 16), torch.randn(2, 2, 17, 16)
    check(lambda I, q, k, v: I.attention(q, k, v, scale=0.25), [q, k, v])
    check(lambda I, q, k, v: I.attention(q, k, v, scale=0.25, l2=True), [q, k, v])
    check(lambda I, x, w, m, k: I.modconv2d(x, w, m, k), [x, torch.randn(2, 24, 16, 3, 3) * 0.1, torch.randn(2, 16) * 0.5, torch.randn(2, 2)])
    check(lambda I, x: I.upsample_blur(x), [x])
    check(lambda I, x: I.resize_bilinear(x, 4), [torch.rand(2, 3, 16, 16)])


def test_models_vs_reference_golden_fixture():
    """generator images / discriminator logits / gradient penalty on the GPU vs the reference's fp32 CPU outputs."""
    fx = torch.load(GOLD / 'model_small.pt', weights_only=False)
    d = dev()
    G, D = Generator(**SMALL_G), Discriminator(**SMALL_D)
    G.load_state_dict(fx['G']); D.load_state_dict(fx['D'])
    G, D = G.to(d), D.to(d).eval()
    # per-layer noise comes from the device RNG, so reproduce the reference's CPU draws by replaying them
    draws = []
    torch.manual_seed(1)
    for r in (4, 4, 8, 8, 16, 16, 32, 32):
        draws.append(torch.randn(2, 1, r, r))
    it = iter(draws)
    orig = torch.randn
    torch.randn = lambda *a, **k: next(it).to(k.get('device', 'cpu'))
    try:
        with torch.no_grad():
            img, rgbs = G(noise=fx['z'].to(d), return_all_rgbs=True)
    finally:
        torch.randn = orig
    assert rel_err(img.cpu(), fx['img']) < 3e-2
    real = fx['real'].to(d).requires_grad_()
    logits, ms, _ = D(real, D.real_images_to_rgbs(real), calc_aux_loss=False)
    assert rel_err(logits.cpu(), fx['logits']) < 3e-2
    for a, b in zip(ms, fx['ms']):
        assert rel_err(a.cpu(), b) < 3e-2
    gp = gradient_penalty(real, [logits, *ms], grad_output_weights=[1., *(0.1,) * len(ms)])
    assert rel_err(gp.cpu(), fx['gp']) < 0.04         # second-order quantity through bf16 leaky-relu masks; measured 0.020


def test_train_step_runs_and_is_finite(tmp_path):
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=dev(),
                  model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    it = cycle(SyntheticImages(2, 64, device=dev()))
    for _ in range(2):
        d_l, g_l = gan.train_step(it, 2)
    vals = [float(v) for v in (*d_l, *g_l) if v is not None]
    assert all(v == v and abs(v) < 1e9 for v in vals), vals
    assert float(d_l.gradient_penalty) > 0


def test_config2_shapes_linearity_and_adjointness():
    """size-independent properties at BASELINE config-2 sizes (batch 32): conv is linear in x, and
    <conv(x, w), y> == <x, conv(y, flipT(w))> == <w, wgrad(x, y)> (fp32-accumulated inner products)."""
    torch.manual_seed(0)
    d = dev()
    for (n, R, Ci, Co) in [(32, 64, 128, 64), (32, 256, 16, 16), (128, 32, 256, 256)]:
        x = bf(torch.randn(n, R, R, Ci, device=d)); x2 = bf(torch.randn(n, R, R, Ci, device=d))
        y = bf(torch.randn(n, R, R, Co, device=d))
        w = bf(torch.randn(Co, Ci, 3, 3, device=d) * 0.05)
        wh = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()
        wT = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, -1).contiguous()
        c1 = K.conv2d_nhwc(x, wh, ksize=3, out_dtype=torch.float32)
        c2 = K.conv2d_nhwc(x2, wh, ksize=3, out_dtype=torch.float32)
        xs = bf(x.float() + x2.float())
        exact = (xs.float() == x.float() + x2.float())         # keep only positions where the bf16 sum is exact
        c12 = K.conv2d_nhwc(torch.where(exact, xs, torch.zeros_like(xs)), wh, ksize=3, out_dtype=torch.float32)
        cz1 = K.conv2d_nhwc(torch.where(exact, x, torch.zeros_like(x)), wh, ksize=3, out_dtype=torch.float32)
        cz2 = K.conv2d_nhwc(torch.where(exact, x2, torch.zeros_like(x2)), wh, ksize=3, out_dtype=torch.float32)
        assert rel_err(c12, cz1 + cz2) < 1e-4
        lhs = (c1.double() * y.double()).sum()
        dg = K.conv2d_nhwc(y, wT, ksize=3, out_dtype=torch.float32)
        rhs = (dg.double() * x.double()).sum()
        wg = K.conv2d_wgrad_nhwc(x, y, ksize=3)
        rhs2 = (wg.double() * wh.t().double()).sum()
        assert abs(lhs - rhs) / abs(lhs) < 1e-3 and abs(lhs - rhs2) / abs(lhs) < 1e-3
        del c1, c2, c12, cz1, cz2, dg


@pytest.mark.parametrize('shape', [(16, 24, 9), (3, 40, 49), (35, 3, 1), (512, 512, 9), (64, 16, 4)])
def test_weight_pack_table_and_wgrad_finish(shape):
    from helpers import check_pack_table_and_wgrad_finish
    check_pack_table_and_wgrad_finish(shape, dev())


def test_gelu_first_and_second_order():
    from helpers import check_gelu_first_and_second_order
    check_gelu_first_and_second_order(dev())


def test_fused_modconv_uses_bank_operand_from_pack_table():
    from helpers import check_fused_modconv_uses_bank_operand_from_pack_table
    check_fused_modconv_uses_bank_operand_from_pack_table(dev())


def test_style_network_runs_on_linear_fn_and_matches_oracle():
    from helpers import check_style_network_on_linear_fn
    check_style_network_on_linear_fn(dev())


def test_flat_optimizer_packs_and_grad_sink_match_autograd():
    from helpers import check_flat_optimizer_packs_and_grad_sink
    check_flat_optimizer_packs_and_grad_sink(dev())


def test_trained_generator_matches_oracle_after_optimizer_steps(tmp_path):
    """two trainer steps (plain + gradient penalty, eager) on the GPU, then the updated generator against the CPU
    oracle on the same weights: the packed operands follow the optimizer, nothing is stale or non-finite."""
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=dev(),
                  use_hip_graphs=False, model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    it = cycle(SyntheticImages(2, C1_G['image_size'], device=dev()))
    for _ in range(2):
        d, g = gan.train_step(it, 2)
        vals = [float(v) for v in (*d, *g) if v is not None]
        assert all(v == v and abs(v) != float('inf') for v in vals), vals
    assert torch.isfinite(gan.G_opt.flat_p).all() and torch.isfinite(gan.D_opt.flat_p).all()
    z = torch.randn(2, 64, device=dev())
    gan.G.eval()
    with torch.no_grad():
        torch.manual_seed(1)
        img = gan.G(noise=z).float().cpu()
        Gc = Generator(**C1_G)
        Gc.load_state_dict({k: v.detach().cpu() for k, v in gan.G.state_dict().items()})
        with ops.use_impl(OracleOps()):
            torch.manual_seed(1)
            ref = Gc(noise=z.cpu()).float()
    assert rel_err(img, ref) < 5e-2


def test_hipgraph_replays_keep_gradients_and_weights_sane(tmp_path):
    """the trainer with hipGraph capture + replays (plain and gradient-penalty steps interleaved): every replay must
    leave finite, plausibly sized flat gradients (garbage from a stale buffer shows up as 1e30+ or NaN), and the
    trained generator still matches the CPU oracle on its own weights."""
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=dev(),
                  use_hip_graphs=True, model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    it = cycle(SyntheticImages(2, C1_G['image_size'], device=dev()))
    for step in range(5):
        d, g = gan.train_step(it, 2)
        for name, opt in (('D', gan.D_opt), ('G', gan.G_opt)):
            gmax = float(opt.flat_g.abs().max())
            assert gmax == gmax and gmax < 1e6, (step, name, gmax)
            assert torch.isfinite(opt.flat_p).all(), (step, name)
    assert gan.use_hip_graphs and len(gan._graphs) >= 3          # capture was not refused
    z = torch.randn(2, 64, device=dev())
    gan.G.eval()
    with torch.no_grad():
        torch.manual_seed(1)
        img = gan.G(noise=z).float().cpu()
        Gc = Generator(**C1_G)
        Gc.load_state_dict({k: v.detach().cpu() for k, v in gan.G.state_dict().items()})
        with ops.use_impl(OracleOps()):
            torch.manual_seed(1)
            ref = Gc(noise=z.cpu()).float()
    assert rel_err(img, ref) < 5e-2


@pytest.mark.parametrize('cfg', [(4, 128, 128, 32, 32), (2, 256, 256, 64, 64), (3, 64, 96, 16, 24), (2, 128, 128, 64, 16)])
def test_direct_conv_vs_implicit_gemm_and_cpu(cfg):
    """gg_dconv (the narrow high-resolution layers) against the implicit-GEMM kernel on the same operands (bf16 store
    rounding only) and against fp32 CPU math, plain and full epilogue."""
    n, H, W, ci, co = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, 9 * ci) * 0.1)
    bias = torch.randn(co); res = bf(torch.randn(n, H, W, co))
    xd, wd, bd, rd = x.to(dev()), w.to(dev()), bias.to(dev()), res.to(dev())
    got = K.conv2d_nhwc(xd, wd, ksize=3, force_tile=9)
    assert rel_err(got, K.conv2d_nhwc(xd, wd, ksize=3, force_tile=1)) < BF16_TOL
    exact = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert rel_err(got.cpu(), exact) < BF16_TOL
    kw = dict(ksize=3, bias=bd, act='lrelu', alpha=0.5, bias_scale=0.5, residual=rd)
    assert rel_err(K.conv2d_nhwc(xd, wd, force_tile=9, **kw), K.conv2d_nhwc(xd, wd, force_tile=1, **kw)) < BF16_TOL


@pytest.mark.parametrize('cfg', [(8, 16, 16, 512, 512, 7), (13, 8, 8, 256, 520, 7), (2, 64, 64, 128, 128, 8), (4, 32, 16, 64, 200, 8),
                                 (4, 32, 32, 256, 256, 7)])
def test_halo_staged_conv3_vs_implicit_gemm_and_cpu(cfg):
    """gg_conv3 (halo staged once per channel chunk) against the implicit-GEMM kernel on the same operands and against fp32 CPU
    convolution, forward and data-gradient form, plain and full epilogue."""
    n, H, W, ci, co, tile = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, 9 * ci) * 0.05)
    bias = torch.randn(co); res = bf(torch.randn(n, H, W, co))
    xd, wd, bd, rd = x.to(dev()), w.to(dev()), bias.to(dev()), res.to(dev())
    K.plan_log = []
    got = K.conv2d_nhwc(xd, wd, ksize=3, out_dtype=torch.float32, force_tile=tile, force_splitk=1)
    assert K.plan_log == [(tile, 1)]
    K.plan_log = None
    exact = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert rel_err(got.cpu(), exact) < F32_TOL
    assert rel_err(got, K.conv2d_nhwc(xd, wd, ksize=3, out_dtype=torch.float32, force_tile=1)) < F32_TOL
    kw = dict(ksize=3, bias=bd, act='lrelu', alpha=0.5, bias_scale=0.5, residual=rd)
    assert rel_err(K.conv2d_nhwc(xd, wd, force_tile=tile, force_splitk=1, **kw), K.conv2d_nhwc(xd, wd, force_tile=1, **kw)) < BF16_TOL
    # split over the channel chunks (automatic and forced): fp32 partials + finish == unsplit
    for sk in (0, 2, 4):
        K.plan_log = []
        got = K.conv2d_nhwc(xd, wd, ksize=3, out_dtype=torch.float32, force_tile=tile, force_splitk=sk)
        assert K.plan_log[-1][0] == tile and (sk == 0 or K.plan_log[-1][1] == min(sk, ci // 64)), K.plan_log
        K.plan_log = None
        assert rel_err(got.cpu(), exact) < F32_TOL
    assert rel_err(K.conv2d_nhwc(xd, wd, force_tile=tile, force_splitk=2, **kw), K.conv2d_nhwc(xd, wd, force_tile=1, **kw)) < BF16_TOL


@pytest.mark.parametrize('cfg', [
    # (n, H, W, C = K, N, epilogue)
    (32, 32, 32, 256, 1024, 'aux'), (32, 32, 32, 1024, 256, 'res'), (16, 64, 64, 128, 64, 'bias'), (32, 16, 16, 512, 2048, 'aux'),
    (3, 50, 50, 192, 136, 'res'), (64, 32, 32, 64, 128, 'plain'), (2, 64, 64, 576, 128, 'plain'),
])
def test_persistent_short_k_contraction_vs_tiled_kernel_and_cpu(cfg):
    """gg_pgemm (plan tile 15: persistent workgroups, LDS-DMA loader waves, runs of 128 x 128 tiles) on the step's 1x1 shapes and on
    ragged ones (7500 rows, 136 columns): bit-identical to gg_gemm2<128,128> on the same operands - same k order, same rounding
    points, same epilogue arithmetic - incl. the FeedForward GELU aux modes (gp.py:726-740), and within bf16 rounding of fp32 math."""
    n, H, W, ci, co, epi = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, ci) / ci ** 0.5)
    bias = torch.randn(co); res = bf(torch.randn(n, H, W, co))
    xd, wd, bd, rd = x.to(dev()), w.to(dev()), bias.to(dev()), res.to(dev())
    kw = dict(ksize=1, pad=0, alpha=0.75)
    if epi in ('bias', 'res', 'aux'):
        kw.update(bias=bd, bias_scale=0.5)
    if epi == 'res':
        kw.update(residual=rd, res_scale=0.25, act='lrelu')
    out = {}
    for tile in (15, 6):
        K.plan_log = []
        if epi == 'aux':
            aux = torch.zeros(n, H, W, co, device=dev(), dtype=torch.bfloat16)
            y = K.conv2d_nhwc(xd, wd, gelu_aux=aux, gelu_mode=1, force_tile=tile, **kw)
            g = K.conv2d_nhwc(xd, wd, gelu_aux=aux, gelu_mode=2, force_tile=tile, ksize=1, pad=0)
            out[tile] = (y, aux, g)
        else:
            out[tile] = (K.conv2d_nhwc(xd, wd, force_tile=tile, **kw),)
        assert all(p == (tile, 1) for p in K.plan_log), K.plan_log
        K.plan_log = None
    for a, b in zip(out[15], out[6]):
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
    exact = 0.75 * torch.einsum('nhwc,oc->nhwo', x.float(), w.float())
    if epi != 'plain':
        exact = exact + 0.5 * bias
    if epi == 'res':
        exact = F.leaky_relu(exact, 0.2).bfloat16().float() + 0.25 * res.float()
    if epi == 'aux':
        y, aux, g = out[15]
        assert rel_err(aux.cpu(), exact) < BF16_TOL and rel_err(y.cpu(), F.gelu(aux.cpu().float())) < BF16_TOL
        h = aux.cpu().float().requires_grad_()
        F.gelu(h).sum().backward()
        lin = torch.einsum('nhwc,oc->nhwo', x.float(), w.float())
        assert rel_err(g.cpu(), lin.bfloat16().float() * h.grad) < BF16_TOL
    else:
        assert rel_err(out[15][0].cpu(), exact) < BF16_TOL


@pytest.mark.parametrize('cfg', [(32, 8, 8, 512, 2, 512, 8, 8), (32, 16, 16, 256, 2, 256, 8, 4), (5, 8, 8, 64, 3, 264, 7, 0),
                                 (4, 32, 32, 128, 2, 128, 8, 1)])
def test_halo_staged_conv3_with_bank_modulation_on_the_operand_staging(cfg):
    """the generator's shared-bank adaptive conv as ONE contraction at config-2 layer shapes (batch 32): N kernels stacked along
    the reduction, a[b,n] * s[b,i] applied when the halo chunk is parked in LDS (gg_conv3 SCALED), split over the channel chunks,
    against the convolution of the explicitly modulated N-fold activation on the implicit GEMM."""
    n, H, W, ci, N, co, tile, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)).to(dev()); w = bf(torch.randn(co, 9 * N * ci) * 0.05).to(dev())
    insc = (torch.rand(n, N * ci) + 0.5).to(dev())
    x2 = bf(torch.cat([x.float() * insc[:, None, None, j * ci:(j + 1) * ci] for j in range(N)], dim=-1))
    want = K.conv2d_nhwc(x2, w, ksize=3, out_dtype=torch.float32, force_tile=1)
    K.plan_log = []
    got = K.conv2d_nhwc(x, w, ksize=3, cv=N * ci, in_scale=insc, out_dtype=torch.float32, force_tile=tile, force_splitk=sk)
    assert K.plan_log[-1][0] == tile, K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < F32_TOL
    d = (torch.rand(n, co) + 0.5).to(dev()); nz = torch.randn(n * H * W).to(dev()); nw = torch.randn(co).to(dev())
    epi = dict(out_scale=d, noise=nz, noise_w=nw, act='lrelu')
    assert rel_err(K.conv2d_nhwc(x, w, ksize=3, cv=N * ci, in_scale=insc, **epi), K.conv2d_nhwc(x2, w, ksize=3, force_tile=1, **epi)) < BF16_TOL


@pytest.mark.parametrize('cfg', [(32, 32, 32, 256, 128, 8, 1), (32, 32, 32, 128, 128, 8, 2), (8, 64, 64, 128, 64, 8, 0), (3, 16, 32, 64, 264, 7, 0)])
def test_halo_staged_conv3_with_per_image_weights_on_gpu(cfg):
    """per-sample weights (gp.py:390-409) through gg_conv3 at config-2's 32x32 / 64x64 layer shapes vs one F.conv2d per image."""
    n, H, W, ci, co, tile, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(n, co, 9 * ci) * 0.05)
    want = torch.stack([F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2), w[i].float().view(co, 3, 3, ci).permute(0, 3, 1, 2),
                                 padding=1)[0].permute(1, 2, 0) for i in range(n)])
    K.plan_log = []
    got = K.conv2d_nhwc(x.to(dev()), w.to(dev()), ksize=3, per_image_weights=True, out_dtype=torch.float32, force_tile=tile,
                        force_splitk=sk)
    assert K.plan_log[-1][0] == tile, K.plan_log
    K.plan_log = None
    assert rel_err(got.cpu(), want) < F32_TOL


def test_multi_layer_modulation_launch_on_gpu():
    """gg_modw_multi_fwd at config-2's generator shapes (batch 32) against one gg_modw_fwd per layer: coefficients to fp32 rounding
    (the Gram-less path is the same code), per-sample weights to bf16 rounding (the batched launch builds them eight channels per
    thread: the compiler contracts the fp32 products differently, a last-bit flip on rare elements)."""
    torch.manual_seed(0)
    b = 32
    shapes = [(512, 512, 'coef'), (512, 256, 'coef'), (256, 128, 'rows'), (64, 64, 'rows'), (32, 32, 'bank'), (16, 16, 'bank')]
    layers, want = [], []
    for I, O, kind in shapes:
        w = (torch.randn(2, O, I, 3, 3) * 0.1).to(dev())
        mod, kmod = (torch.randn(b, I) * 0.5).to(dev()), torch.randn(b, 2).to(dev())
        ly = dict(w=w, mod=mod, kmod=kmod, demod=True, eps=1e-8, Ip=I, Op=O)
        if kind == 'coef':
            want.append(K.modw_fwd(w, mod, kmod, True, 1e-8, I, O))
        else:
            shape = (b, O, 9 * I) if kind == 'rows' else (b, 9, I // 16, 32, 16)
            wm = torch.zeros(shape, dtype=torch.bfloat16, device=dev())
            K.modw_fwd(w, mod, kmod, True, 1e-8, I, O, coef=False, wmix=wm, layout=1 if kind == 'rows' else 2)
            want.append(wm)
            ly.update(coef=False, wmix=torch.zeros_like(wm), layout=1 if kind == 'rows' else 2)
        layers.append(ly)
    outs = K.modw_multi(layers)
    torch.cuda.synchronize()
    for ly, o, w_ in zip(layers, outs, want):
        if torch.is_tensor(w_):
            assert rel_err(ly['wmix'], w_) < 1e-3 and float((ly['wmix'] != w_).float().mean()) < 0.05
        else:
            s, a, d = w_
            assert torch.equal(o['s'], s) and torch.allclose(o['a'], a, rtol=1e-6, atol=1e-7) and torch.allclose(o['d'], d, rtol=2e-6, atol=1e-7)
            assert torch.allclose(o['insc'], (a[:, :, None] * s[:, None, :]).reshape(b, -1), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('shape', [(2, 64, 64, 64), (3, 8, 16, 24), (16, 32, 32, 128)])
def test_maxpool_highfreq_kernels_vs_torch(shape):
    """gg_poolhf_fwd / _bwd (unet Downsample tail, unet_upsampler.py:134-160) against max_pool2d + the reflect-padded blur in fp32
    under autograd: values and the gradient of both outputs."""
    from helpers import check_maxpool_highfreq
    check_maxpool_highfreq(shape, dev())


@pytest.mark.parametrize('cfg', [(64, 16, 16, 512, 512, 0), (40, 8, 8, 256, 520, 3), (8, 32, 32, 128, 256, 0), (2, 64, 64, 64, 128, 16)])
def test_nine_tap_weight_gradient_vs_implicit_gemm_and_cpu(cfg):
    """gg_wgrad9 against the 4-wave implicit-GEMM weight gradient on the same operands and against fp32 CPU autograd."""
    n, H, W, ci, co, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); dy = bf(torch.randn(n, H, W, co))
    xd, dyd = x.to(dev()), dy.to(dev())
    K.plan_log = []
    got = K.conv2d_wgrad_nhwc(xd, dyd, ksize=3, force_tile=10, force_splitk=sk)
    assert K.plan_log[-1][0] == 10
    K.plan_log = None
    assert rel_err(got, K.conv2d_wgrad_nhwc(xd, dyd, ksize=3, force_tile=1)) < F32_TOL
    w = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert rel_err(got.cpu(), w.grad.permute(2, 3, 1, 0).reshape(-1, co)) < F32_TOL


@pytest.mark.parametrize('cfg', [(5, 2, 10, 12, 3), (32, 2, 512, 512, 3), (32, 2, 32, 64, 3), (32, 1, 3, 32, 1)])
def test_fused_adaptive_conv_coefficients_match_tensor_algebra(cfg):
    from helpers import check_modcoef
    check_modcoef(cfg, dev())


@pytest.mark.parametrize('cfg', [(16, 8, 256, 256, False, False, True), (16, 8, 64, 64, False, False, True), (16, 8, 1024, 77, False, True, False),
                                 (16, 8, 78, 78, True, True, False), (3, 2, 130, 50, True, False, False),
                                 (6, 8, 1024, 77, False, 2, False), (6, 2, 96, 77, True, 2, False)])
def test_general_fused_attention_on_gpu(cfg):
    """gg_attn_gen_* at the shapes configs 4 / 5 run: the unet's Attend at 16x16 and 8x8 (8 heads of 64 on slices of to_qkv), the
    generator's cross attention at 32x32 over 77 masked text tokens, the text transformer's attention (78 tokens, null key, mask),
    a ragged case, and key masks that switch off the LEADING keys (with and without a null token) - forward and backward against
    fp32 autograd of the reference's masked_fill formulation (gp.py:645-647)."""
    from helpers import check_general_attention
    check_general_attention(cfg, dev())
