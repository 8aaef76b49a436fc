"""CPU suite: the real kernel sources compiled for the host-side emulator, called through the same C ABI."""
import ctypes
import re
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

from gigagan_pytorch_amd import kernels as K, ops, _C
from oracle.torch_ops import OracleOps
from helpers import rel_err, bf

ROOT = Path(__file__).resolve().parents[1]


def test_c_abi_exports_every_declared_symbol():
    header = (ROOT / 'include' / 'gigagan_amd.h').read_text()
    names = set(re.findall(r'\b(gg_[a-z0-9_]+)\s*\(', header))
    assert {'gg_gemm_bf16', 'gg_resample_nhwc_bf16', 'gg_adamw_flat_f32', 'gg_ema_flat_f32', 'gg_last_error'} <= names
    for lib in (ROOT / 'gigagan_pytorch_amd' / 'libgigagan_amd.so', ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so'):
        h = ctypes.CDLL(str(lib))
        for n in names:
            assert hasattr(h, n), f'{lib.name} does not export {n}'
    assert _C.Library(ROOT / 'gigagan_pytorch_amd' / 'libgigagan_amd.so').is_emulator is False


def test_product_path_fails_loudly_without_gpu_tensors():
    lib = _C.Library(ROOT / 'gigagan_pytorch_amd' / 'libgigagan_amd.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        lib.require(torch.zeros(4))
    with pytest.raises(RuntimeError, match='not found'):
        _C.Library(ROOT / 'gigagan_pytorch_amd' / 'does_not_exist.so')


def test_argument_errors_are_reported():
    a = bf(torch.randn(16, 12)); b = bf(torch.randn(8, 12))   # pitch 12 is not a multiple of 8
    with pytest.raises(RuntimeError, match='multiple of 8'):
        K.gemm(a, b)


@pytest.mark.parametrize('shape', [(130, 70, 100, 2), (64, 33, 1032, 1), (257, 129, 40, 3)])
def test_gemm_all_layouts_tiles_and_splitk(shape):
    M, N, Kd, batch = shape
    torch.manual_seed(0)
    Kp = (Kd + 7) // 8 * 8
    a = bf(torch.randn(batch, M, Kp)); b = bf(torch.randn(batch, N, Kp))
    ref = torch.einsum('bmk,bnk->bmn', a.float()[..., :Kd], b.float()[..., :Kd])
    for tile in (1, 2, 3):
        assert rel_err(K.gemm(a, b, k_valid=Kd, out_dtype=torch.float32, force_tile=tile), ref) < 1e-5
    assert rel_err(K.gemm(a, b, k_valid=Kd, out_dtype=torch.float32, force_splitk=3), ref) < 1e-5
    assert rel_err(K.gemm(a, b, k_valid=Kd), ref) < 4e-3     # bf16 output rounding


def test_gemm_transposed_operands_and_epilogue():
    torch.manual_seed(0)
    M, N, Kd, batch = 72, 48, 50, 2
    A = torch.randn(batch, M, Kd); B = torch.randn(batch, N, Kd)
    pad = lambda t: F.pad(t, (0, (-t.shape[-1]) % 8))
    ref = torch.einsum('bmk,bnk->bmn', bf(A).float(), bf(B).float())
    for ta in (False, True):
        for tb in (True, False):
            a = bf(pad(A.transpose(1, 2).contiguous() if ta else A))
            b = bf(pad(B if tb else B.transpose(1, 2).contiguous()))
            out = K.gemm(a, b, trans_a=ta, trans_b=tb, k_valid=Kd, m_valid=M, n_valid=N, out_dtype=torch.float32)
            assert rel_err(out, ref) < 1e-5, (ta, tb)
    bias = torch.randn(N)
    out = K.gemm(bf(pad(A)), bf(pad(B)), k_valid=Kd, bias=bias, act='lrelu', alpha=0.5, out_dtype=torch.float32)
    assert rel_err(out, F.leaky_relu(ref * 0.5 + bias, 0.2)) < 1e-5


@pytest.mark.parametrize('cfg', [(2, 8, 8, 16, 24, 3), (3, 5, 7, 8, 40, 7), (2, 16, 16, 32, 136, 1), (1, 4, 4, 64, 64, 3)])
def test_conv_forward_dgrad_wgrad(cfg):
    n, H, W, Ci, Co, ks = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, Ci, H, W)); w = bf(torch.randn(Co, Ci, ks, ks) * 0.1); dy = bf(torch.randn(n, Co, H, W))
    xf, wf = x.float().requires_grad_(), w.float().requires_grad_()
    ref = F.conv2d(xf, wf, padding=ks // 2)
    ref.backward(dy.float())
    xh, dyh = x.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
    wh = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()
    assert rel_err(K.conv2d_nhwc(xh, wh, ksize=ks, out_dtype=torch.float32).permute(0, 3, 1, 2), ref) < 1e-5
    dw_ref = wf.grad.permute(2, 3, 1, 0).reshape(-1, Co)
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks), dw_ref) < 1e-5
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks, force_splitk=3), dw_ref) < 1e-5
    wT = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, -1).contiguous()
    assert rel_err(K.conv2d_nhwc(dyh, wT, ksize=ks, out_dtype=torch.float32).permute(0, 3, 1, 2), xf.grad) < 1e-5


def test_fused_adaptive_conv_single_launch():
    """kernel mix + modulation + demod scale + noise + leaky-relu in ONE implicit-GEMM launch vs the oracle."""
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps(bf16_operands=True)
    x = torch.randn(2, 16, 8, 8); wm = torch.randn(2, 24, 16, 3, 3) * 0.1
    mod = torch.randn(2, 16) * 0.5; km = torch.randn(2, 2); nz = torch.randn(2, 1, 8, 8); nw = torch.randn(24, 1, 1)
    with torch.no_grad():
        y = H_.modconv2d(x, wm, mod, km, noise=nz, noise_weight=nw, act='lrelu')
    ref = O_.modconv2d(x, wm, mod, km, noise=nz, noise_weight=nw, act='lrelu')
    assert rel_err(y, ref) < 1e-2
    with torch.no_grad():   # toRGB: 1x1, single kernel, no demod, 3 output channels
        wr = torch.randn(1, 3, 16, 1, 1) * 0.1
        assert rel_err(H_.modconv2d(x, wr, mod, None, demod=False), O_.modconv2d(x, wr, mod, None, demod=False)) < 1e-2


def test_ops_gradients_match_oracle():
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps(bf16_operands=True)

    def check(fn, inputs, tol=2e-2):
        ih = [t.clone().requires_grad_() for t in inputs]; io = [t.clone().requires_grad_() for t in inputs]
        yh, yo = fn(H_, *ih), fn(O_, *io)
        g = torch.randn_like(yo)
        gh = torch.autograd.grad(yh.float(), ih, g); go = torch.autograd.grad(yo, io, bf(g).float())
        assert rel_err(yh, yo) < tol
        for a, b in zip(gh, go):
            assert rel_err(a, b) < 2 * tol      # bf16 operand rounding accumulates through the backward pass

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


def test_conv_double_backward_matches_oracle():
    """gradient-penalty pattern: d/dw of |d out / d x|^2 through ConvFn <-> WgradFn <-> ConvFn."""
    torch.manual_seed(0)

    def run(I):
        x = torch.randn(2, 8, 6, 6).requires_grad_(); w = (torch.randn(8, 8, 3, 3) * 0.2).requires_grad_()
        w2 = (torch.randn(8, 8, 3, 3) * 0.2).requires_grad_()
        y = I.conv2d(I.conv2d(x, w, None), w2, None).float()
        gx, = torch.autograd.grad(y.pow(2).sum(), x, create_graph=True)
        return torch.autograd.grad(gx.float().pow(2).sum(), [w, w2])

    torch.manual_seed(0); gh = run(ops.HipOps())
    torch.manual_seed(0); go = run(OracleOps(bf16_operands=True))
    for a, b in zip(gh, go):
        assert rel_err(a, b) < 3e-2


def test_resample_matches_reference_semantics():
    torch.manual_seed(0)
    O_ = OracleOps()
    for shape in [(2, 8, 4, 4), (1, 16, 8, 6), (2, 3, 5, 5)]:
        x = bf(torch.randn(*shape))
        assert rel_err(ops.HipOps().upsample_blur(x), O_.upsample_blur(x.float())) < 4e-3
    x = torch.rand(2, 3, 32, 32)
    for size in (8, 16):
        assert rel_err(ops.HipOps().resize_bilinear(x, size), F.interpolate(bf(x).float(), size, mode='bilinear')) < 4e-3
    assert rel_err(ops.HipOps().resize_nearest(x, 16), F.interpolate(bf(x).float(), (16, 16))) == 0.


@pytest.mark.parametrize('kind', ['upblur', 'blur', 'bilinear_up', 'bilinear_down', 'upblur_T', 'custom_far'])
def test_resample_two_by_two_blocks_match_the_dense_operator(kind, monkeypatch):
    """gg_resample_taps2x2_kernel (a 2 x 2 block of output pixels per thread from one shared window; taken for 2 / 3-tap up-sampling and
    same-size filters with C % 8 == 0 and even output extents) and the per-pixel kernels (GG_RESAMPLE_2X2=0) against the dense separable
    operator out = My x Mx^T, incl. non-square maps, a table whose neighbouring windows are further apart than one pixel (the kernel's
    per-block fallback) and operators the blocked kernel does not take (down-sampling, 6-tap adjoints)."""
    torch.manual_seed(0)
    H, W, C, n = 6, 10, 16, 2
    if kind == 'upblur':
        spec = K.ResampleSpec.upsample_blur(H, W)
    elif kind == 'blur':
        spec = K.ResampleSpec.blur(H, W)
    elif kind == 'bilinear_up':
        spec = K.ResampleSpec.bilinear(H, W, 2 * H, 2 * W)
    elif kind == 'bilinear_down':
        spec = K.ResampleSpec.bilinear(H, W, H // 2, W // 2)
    elif kind == 'upblur_T':
        spec = K.ResampleSpec.upsample_blur(H // 2, W // 2).transposed()
    else:       # 3-tap rows whose windows jump by two input pixels between neighbouring outputs, on a map that does not shrink
        my = torch.zeros(H, H); mx = torch.zeros(W, W)
        for o in range(H):
            for a in range(3):
                my[o, (2 * o) % (H - 2) + a] = 0.2 + 0.1 * a
        for o in range(W):
            for a in range(3):
                mx[o, (3 * o) % (W - 2) + a] = 0.3 - 0.05 * a
        spec = K.ResampleSpec(my, mx, ('custom_far', H, W))
        assert spec.ty == 3 and spec.tx == 3
    for Cc in (C, 3):            # (3: the rgb maps take gg_resample_taps_small_kernel - every tap loaded unconditionally)
        x = bf(torch.randn(n, spec.ih, spec.iw, Cc))
        want = torch.einsum('oh,nhwc->nowc', spec.my.float(), x.float())
        want = torch.einsum('pw,nowc->nopc', spec.mx.float(), want)
        outs = {}
        for env in ('1', '0'):
            monkeypatch.setenv('GG_RESAMPLE_2X2', env)
            outs[env] = K.resample_nhwc(x, spec)
            assert rel_err(outs[env].float(), want) < 4e-3, (kind, env, Cc)
        assert rel_err(outs['1'].float(), outs['0'].float()) < 3e-3


def test_fused_adamw_and_ema_match_torch():
    from gigagan_pytorch_amd.optimizer import FlatAdamW
    torch.manual_seed(0)
    p0, p1 = torch.nn.Parameter(torch.randn(50, 33)), torch.nn.Parameter(torch.randn(77))
    p2 = torch.nn.Parameter(torch.randn(9, 9))     # inactive: never stepped, never decayed
    r0, r1 = torch.nn.Parameter(p0.detach().clone()), torch.nn.Parameter(p1.detach().clone())
    p2_before = p2.detach().clone()
    fo = FlatAdamW([p0, p1, p2], lr=2e-4, betas=(0.5, 0.9), inactive=[p2])
    to = torch.optim.AdamW([{'params': [r0]}, {'params': [r1], 'weight_decay': 0.}], lr=2e-4, betas=(0.5, 0.9), weight_decay=1e-2)
    for _ in range(3):
        g0, g1 = torch.randn_like(p0), torch.randn_like(p1)
        fo.zero_grad()
        p0.grad.add_(g0); p1.grad.add_(g1)
        r0.grad, r1.grad = g0.clone(), g1.clone()
        fo.step(); to.step()
    assert rel_err(p0, r0) < 1e-6 and rel_err(p1, r1) < 1e-6
    assert torch.equal(p2.detach(), p2_before)
    assert rel_err(fo.state[p0]['exp_avg'], to.state[r0]['exp_avg']) < 1e-6


# ---- 8-wave large-tile kernel (gg_gemm2.h) and the generalised conv geometry ---------------------------------

@pytest.mark.parametrize('tile', [4, 5, 6])
def test_gemm2_dense_all_layouts(tile):
    torch.manual_seed(0)
    M, N, Kd, batch = 264, 136, 136, 2
    A = torch.randn(batch, M, Kd); B = torch.randn(batch, N, Kd)
    ref = torch.einsum('bmk,bnk->bmn', bf(A).float(), bf(B).float())
    for ta in (False, True):
        for tb in (True, False):
            a = bf(A.transpose(1, 2).contiguous() if ta else A)
            b = bf(B if tb else B.transpose(1, 2).contiguous())
            out = K.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, force_tile=tile)
            assert rel_err(out, ref) < 1e-5, (ta, tb)
    out = K.gemm(bf(A), bf(B), out_dtype=torch.float32, force_tile=tile, force_splitk=2)
    assert rel_err(out, ref) < 1e-5
    bias = torch.randn(N)
    out = K.gemm(bf(A), bf(B), bias=bias, act='lrelu', alpha=0.5, out_dtype=torch.float32, force_tile=tile)
    assert rel_err(out, F.leaky_relu(ref * 0.5 + bias, 0.2)) < 1e-5


@pytest.mark.parametrize('cfg', [(2, 12, 12, 64, 136, 3, 4), (1, 9, 7, 128, 128, 3, 5), (3, 8, 8, 64, 104, 1, 5),
                                 (2, 8, 8, 64, 128, 3, 6)])
def test_gemm2_conv_forward_dgrad_wgrad(cfg):
    n, H, W, Ci, Co, ks, tile = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, Ci, H, W)); w = bf(torch.randn(Co, Ci, ks, ks) * 0.1); dy = bf(torch.randn(n, Co, H, W))
    xf, wf = x.float().requires_grad_(), w.float().requires_grad_()
    ref = F.conv2d(xf, wf, padding=ks // 2)
    ref.backward(dy.float())
    xh, dyh = x.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
    wh = w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()
    out = K.conv2d_nhwc(xh, wh, ksize=ks, out_dtype=torch.float32, force_tile=tile)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 1e-5
    dw_ref = wf.grad.permute(2, 3, 1, 0).reshape(-1, Co)
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks, force_tile=tile), dw_ref) < 1e-5
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=ks, force_tile=tile, force_splitk=2), dw_ref) < 1e-5
    if Co % 64 == 0:
        wT = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, -1).contiguous()
        dx = K.conv2d_nhwc(dyh, wT, ksize=ks, out_dtype=torch.float32, force_tile=tile)
        assert rel_err(dx.permute(0, 3, 1, 2), xf.grad) < 1e-5


def test_gemm2_conv_in_scale_virtual_channels_and_epilogue():
    """the no-grad adaptive-conv launch shape: CV = 2*C virtual channels, per-sample in_scale, out_scale, noise."""
    torch.manual_seed(0)
    n, H, W, C, O = 2, 6, 6, 32, 128
    x = bf(torch.randn(n, H, W, C)); w = bf(torch.randn(O, 9 * 2 * C) * 0.1)
    insc = torch.rand(n, 2 * C) + 0.5; osc = torch.rand(n, O) + 0.5
    nz = torch.randn(n * H * W); nw = torch.randn(O)
    res = bf(torch.randn(n, H, W, O))
    outs = [K.conv2d_nhwc(x, w, ksize=3, cv=2 * C, in_scale=insc, out_scale=osc, noise=nz, noise_w=nw, act='lrelu',
                          residual=res, res_scale=0.5, out_dtype=torch.float32, force_tile=t) for t in (1, 4)]
    assert rel_err(outs[1], outs[0]) < 1e-5
    # independent restatement
    xs = bf(torch.cat([x.float(), x.float()], -1) * insc[:, None, None, :]).float()        # (n,H,W,2C)
    wk = w.float().view(O, 3, 3, 2 * C).permute(0, 3, 1, 2)
    y = F.conv2d(xs.permute(0, 3, 1, 2), wk, padding=1).permute(0, 2, 3, 1)
    y = y * osc[:, None, None, :] + nz.view(n, H, W, 1) * nw
    y = F.leaky_relu(y, 0.2) + 0.5 * res.float()
    assert rel_err(outs[1], y) < 1e-5
    # bf16 output: the LDS-staged row-contiguous store path of the 8-wave kernel (residual added on the way out)
    for t in (4, 5, 6):
        ob = K.conv2d_nhwc(x, w, ksize=3, cv=2 * C, in_scale=insc, out_scale=osc, noise=nz, noise_w=nw, act='lrelu',
                           residual=res, res_scale=0.5, force_tile=t)
        assert ob.dtype == torch.bfloat16 and rel_err(ob, y) < 4e-3
        ob = K.conv2d_nhwc(x, w, ksize=3, cv=2 * C, in_scale=insc, force_tile=t)          # plain epilogue, staged
        assert rel_err(ob, F.conv2d(xs.permute(0, 3, 1, 2), wk, padding=1).permute(0, 2, 3, 1)) < 4e-3


@pytest.mark.parametrize('tile', [0, 4])
def test_strided_convs_and_depth_to_space_dgrad(tile):
    """stride-2 1x1 (gp.py:1612) and space-to-depth + 1x1 (gp.py:289-293) as gather variants, with the
    depth-to-space scatter store as their data gradient."""
    torch.manual_seed(0)
    n, H, W, C, O = 2, 8, 8, 64, 128
    x = bf(torch.randn(n, C, H, W)); xh = x.permute(0, 2, 3, 1).contiguous()
    # (a) 1x1 stride 2
    w1 = bf(torch.randn(O, C, 1, 1) * 0.1)
    xf, wf = x.float().requires_grad_(), w1.float().requires_grad_()
    ref = F.conv2d(xf, wf, stride=2)
    dy = bf(torch.randn_like(ref)); ref.backward(dy.float())
    dyh = dy.permute(0, 2, 3, 1).contiguous()
    w1h = w1.reshape(O, C).contiguous()
    out = K.conv2d_nhwc(xh, w1h, ksize=1, stride=2, pad=0, out_dtype=torch.float32, force_tile=tile)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 1e-5
    assert rel_err(K.conv2d_wgrad_nhwc(xh, dyh, ksize=1, stride=2, pad=0, force_tile=tile), wf.grad.reshape(O, C).t()) < 1e-5
    dx = K.conv2d_dgrad_d2s(dyh, w1h, cell=2, taps=1, force_tile=tile)
    assert rel_err(dx.permute(0, 3, 1, 2), xf.grad) < 4e-3
    # (b) space-to-depth + 1x1: reference channel order (c, s1, s2); kernel weight order [co][s1][s2][c]
    w2 = bf(torch.randn(O, 4 * C, 1, 1) * 0.1)
    xf, wf = x.float().requires_grad_(), w2.float().requires_grad_()
    s2d = xf.reshape(n, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(n, 4 * C, H // 2, W // 2)
    ref = F.conv2d(s2d, wf)
    ref.backward(dy.float())
    w2h = w2.reshape(O, C, 2, 2).permute(0, 2, 3, 1).reshape(O, 4 * C).contiguous()
    bias = torch.randn(O); res = bf(torch.randn(n, H // 2, W // 2, O))
    out = K.conv2d_nhwc(xh, w2h, ksize=2, stride=2, pad=0, out_dtype=torch.float32, force_tile=tile,
                        alpha=0.7, bias=bias, bias_scale=0.7, residual=res)
    assert rel_err(out, 0.7 * (ref.detach().permute(0, 2, 3, 1) + bias) + res.float()) < 1e-5
    dw = K.conv2d_wgrad_nhwc(xh, dyh, ksize=2, stride=2, pad=0, force_tile=tile)          # ((s1,s2,c), O)
    dw_ref = wf.grad.reshape(O, C, 2, 2).permute(2, 3, 1, 0).reshape(4 * C, O)
    assert rel_err(dw, dw_ref) < 1e-5
    dx = K.conv2d_dgrad_d2s(dyh, w2h, cell=2, taps=2, force_tile=tile)
    assert rel_err(dx.permute(0, 3, 1, 2), xf.grad) < 4e-3


def test_softmax_second_order_pass_matches_tensor_algebra():
    torch.manual_seed(0)
    nb, n, m, ld = 2, 16, 21, 24
    S = torch.zeros(nb, n, ld); S[..., :m] = torch.randn(nb, n, m).softmax(-1)
    S, dS, g_dx = bf(S), bf(torch.randn(nb, n, ld)), bf(torch.randn(nb, n, ld))
    g_db = torch.randn(nb, ld)
    alpha = 0.7
    Sf, dSf = S.float()[..., :m], dS.float()[..., :m]
    gt = alpha * g_dx.float()[..., :m] + g_db[:, None, :m]
    r = (Sf * dSf).sum(-1, keepdim=True); gs = (gt * Sf).sum(-1, keepdim=True)
    ref_gdS = Sf * (gt - gs); ref_gS = gt * (dSf - r) - dSf * gs
    g_S, g_dS = K.softmax_bwd2(S, dS, g_dx, g_db, alpha, m)
    assert rel_err(g_S[..., :m], ref_gS) < 4e-3 and rel_err(g_dS[..., :m], ref_gdS) < 4e-3
    assert float(g_S[..., m:].abs().max()) == 0 and float(g_dS[..., m:].abs().max()) == 0
    g_S, g_dS = K.softmax_bwd2(S, dS, g_dx, None, alpha, m)
    gt = alpha * g_dx.float()[..., :m]; gs = (gt * Sf).sum(-1, keepdim=True)
    assert rel_err(g_dS[..., :m], Sf * (gt - gs)) < 4e-3


def test_attention_double_backward_matches_oracle():
    """gradient-penalty pattern through the attention Functions (fused softmax fwd / bwd / second-order passes)."""
    def run(I, l2):
        torch.manual_seed(0)
        q = (torch.randn(1, 2, 16, 16) * 0.5).requires_grad_(); k = (torch.randn(1, 2, 17, 16) * 0.5).requires_grad_()
        v = torch.randn(1, 2, 17, 16).requires_grad_()
        out = I.attention(q, k, v, scale=0.25, l2=l2).float()
        gq, = torch.autograd.grad(out.pow(2).sum(), q, create_graph=True)
        return torch.autograd.grad(gq.float().pow(2).sum(), [k, v])

    for l2 in (False, True):
        gh = run(ops.HipOps(), l2); go = run(OracleOps(bf16_operands=True), l2)
        for a, b in zip(gh, go):
            assert rel_err(a, b) < 6e-2, l2


def test_adaptive_conv_training_path_all_gradients():
    """modulate -> stacked conv -> mix/demod/noise/lrelu Functions: every input gradient against the oracle."""
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps(bf16_operands=True)
    x = torch.randn(2, 16, 6, 6); w = torch.randn(2, 24, 16, 3, 3) * 0.1; mod = torch.randn(2, 16) * 0.5
    km = torch.randn(2, 2); nz = torch.randn(2, 1, 6, 6); nw = torch.randn(24, 1, 1) * 0.5

    def run(I):
        ins = [t.clone().requires_grad_() for t in (x, w, mod, km, nw)]
        y = I.modconv2d(ins[0], ins[1], ins[2], ins[3], noise=nz, noise_weight=ins[4], act='lrelu')
        g = torch.autograd.grad(y.float(), ins, torch.ones_like(y.float()) * torch.linspace(-1, 1, y.numel()).view_as(y))
        return y, g

    yh, gh = run(H_); yo, go = run(O_)
    assert rel_err(yh, yo) < 2e-2
    for a, b, name in zip(gh, go, ('x', 'weights', 'mod', 'kernel_mod', 'noise_weight')):
        assert rel_err(a, b) < 5e-2, name
    # toRGB: single kernel, 1x1, no demodulation, 3 output channels
    wr = torch.randn(1, 3, 16, 1, 1) * 0.1

    def run_rgb(I):
        ins = [t.clone().requires_grad_() for t in (x, wr, mod)]
        y = I.modconv2d(ins[0], ins[1], ins[2], None, demod=False)
        return y, torch.autograd.grad(y.float().pow(2).sum(), ins)

    yh, gh = run_rgb(H_); yo, go = run_rgb(O_)
    assert rel_err(yh, yo) < 2e-2
    for a, b in zip(gh, go):
        assert rel_err(a, b) < 5e-2


@pytest.mark.parametrize('l2', [False, True])
def test_fused_attention_forward_backward_vs_reference(l2):
    """flash-style attention kernels (null key/value folded into the online-softmax state) vs plain tensor algebra."""
    torch.manual_seed(0)
    B, n, h = 1, 128, 2
    scale = 64 ** -0.5
    q = bf(torch.randn(B, n, h * 64) * 0.7); v = bf(torch.randn(B, n, h * 64))
    k = q.clone() if l2 else bf(torch.randn(B, n, h * 64) * 0.7)
    k0 = bf(torch.randn(h, 64) * 0.7); v0 = bf(torch.randn(h, 64))
    alpha, beta = (2 * scale, -scale) if l2 else (scale, 0.)
    d_o = bf(torch.randn(B, n, h * 64))

    def ref(q, k, v, k0, v0):
        qh, kh, vh = (t.view(B, n, h, 64).permute(0, 2, 1, 3) for t in (q, k, v))          # (B,h,n,64)
        kk = torch.cat((k0[None, :, None, :].expand(B, -1, -1, -1), kh), 2)
        vv = torch.cat((v0[None, :, None, :].expand(B, -1, -1, -1), vh), 2)
        x = alpha * qh @ kk.transpose(-1, -2) + beta * (kk * kk).sum(-1)[:, :, None, :]
        o = x.softmax(-1) @ vv
        return o.permute(0, 2, 1, 3).reshape(B, n, h * 64), x.logsumexp(-1).reshape(B * h, n)

    ins = [t.float().requires_grad_() for t in (q, k, v, k0, v0)]
    o_ref, lse_ref = ref(*ins)
    g = torch.autograd.grad(o_ref, ins, d_o.float())
    o, lse = K.attn_fwd(q, k, v, k0, v0, h, alpha, beta)
    assert rel_err(o, o_ref) < 6e-3 and rel_err(lse, lse_ref) < 1e-4
    dq, dk, dv, dk0q, dv0, dbias0 = K.attn_bwd(q, k, v, k0, v0, o, lse, d_o, h, alpha, beta)
    assert rel_err(dq, g[0]) < 2e-2 and rel_err(dk, g[1]) < 2e-2 and rel_err(dv, g[2]) < 2e-2
    dk0 = dk0q + 2 * beta * dbias0[:, None] * k0.float()
    assert rel_err(dk0, g[3]) < 2e-2 and rel_err(dv0, g[4]) < 2e-2


def test_channel_rmsnorm_first_and_second_order_vs_oracle():
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps()
    x0 = torch.randn(2, 24, 5, 5); gamma0 = torch.rand(24, 1, 1) + 0.5

    def run(I):
        x = bf(x0).float().requires_grad_(); gamma = gamma0.clone().requires_grad_()
        y = I.channel_rmsnorm(x, gamma).float()
        w = torch.linspace(-1, 1, y.numel()).view_as(y)
        gx, = torch.autograd.grad((y * w).sum() + y.pow(2).sum(), x, create_graph=True)
        gg = torch.autograd.grad(gx.float().pow(2).sum(), [x, gamma])
        g1 = torch.autograd.grad((I.channel_rmsnorm(x, gamma).float() * w).sum(), [x, gamma])
        return y, gx, gg, g1

    yh, gxh, ggh, g1h = run(H_); yo, gxo, ggo, g1o = run(O_)
    assert rel_err(yh, yo) < 6e-3 and rel_err(gxh, gxo) < 2e-2
    for a, b in zip(g1h, g1o):
        assert rel_err(a, b) < 2e-2
    for a, b in zip(ggh, ggo):
        assert rel_err(a, b) < 5e-2


@pytest.mark.parametrize('dot', [True, False])
def test_self_attention_module_on_fused_kernels_vs_oracle(dot):
    """SelfAttention (norm -> 1x1 projections -> fused attention with null kv -> 1x1 + skip) on the kernel path vs the
    same module on the oracle, outputs and all parameter / input gradients."""
    from gigagan_pytorch_amd.modules import SelfAttention
    torch.manual_seed(0)
    attn = SelfAttention(16, dim_head=64, heads=1, dot_product=dot)
    x0 = torch.randn(1, 16, 16, 16)

    def run(I):
        with ops.use_impl(I):
            x = x0.clone().requires_grad_()
            y = attn(x, skip=True).float()
            g = torch.autograd.grad((y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum(), [x, *attn.parameters()])
        return y, g

    yh, gh = run(ops.HipOps()); yo, go = run(OracleOps(bf16_operands=True))
    assert rel_err(yh, yo) < 2e-2
    for a, b in zip(gh, go):
        assert rel_err(a, b) < 6e-2


@pytest.mark.parametrize('l2', [False, True])
def test_fused_attention_second_order_vs_autograd(l2):
    """gg_attn_bwd2 (gradient of <A, first-backward outputs>) vs double backward of plain fp32 tensor algebra."""
    torch.manual_seed(0)
    B, n, h = 1, 128, 2
    scale = 64 ** -0.5
    mk = lambda *s, m=1.0: bf(torch.randn(*s) * m)
    q, v, k0, v0, d_o = mk(B, n, h * 64, m=0.7), mk(B, n, h * 64), mk(h, 64, m=0.7), mk(h, 64), mk(B, n, h * 64)
    k = mk(B, n, h * 64, m=0.7)
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

    o, lse = K.attn_fwd(q, k, v, k0, v0, h, alpha, beta)
    *_, dvec = K.attn_bwd(q, k, v, k0, v0, o, lse, d_o, h, alpha, beta, return_dvec=True)
    gq, gk, gv, gdo, gk0, gv0 = K.attn_bwd2(q, k, v, k0, v0, d_o, lse, dvec, aq, ak, av, ak0, av0, h, alpha, beta)
    for name, a, b_ in zip(('gq', 'gk', 'gv', 'gk0', 'gv0', 'gdo'), (gq, gk, gv, gk0, gv0, gdo), ref):
        assert rel_err(a, b_) < 4e-2, (name, rel_err(a, b_))


@pytest.mark.parametrize('l2', [False, True])
def test_self_attention_op_double_backward_on_fused_kernels(l2):
    """gradient-penalty pattern through HipOps.self_attention: FlashAttnFn -> FlashAttnBwdFn -> fused second-order pass,
    against the oracle's plain autograd (null key/value parameters included)."""
    def run(I):
        torch.manual_seed(0)
        q = (torch.randn(1, 64, 16, 8) * 0.5).requires_grad_(); v = torch.randn(1, 64, 16, 8).requires_grad_()
        k = q if l2 else (torch.randn(1, 64, 16, 8) * 0.5).requires_grad_()
        null_kv = (torch.randn(2, 1, 64) * 0.5).requires_grad_()
        out = I.self_attention(q, k, v, null_kv, heads=1, scale=0.125, l2=l2).float()
        gq, = torch.autograd.grad((out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum(), q, create_graph=True)
        leaves = [q, v, null_kv] if l2 else [q, k, v, null_kv]
        return torch.autograd.grad(gq.float().pow(2).sum(), leaves)

    gh = run(ops.HipOps()); go = run(OracleOps(bf16_operands=True))
    for a, b in zip(gh, go):
        assert rel_err(a, b) < 6e-2


@pytest.mark.parametrize('shape', [(16, 24, 9), (3, 40, 49), (35, 3, 1), (64, 16, 4)])
def test_weight_pack_table_and_wgrad_finish(shape):
    from helpers import check_pack_table_and_wgrad_finish
    check_pack_table_and_wgrad_finish(shape, 'cpu')


def test_gelu_first_and_second_order():
    from helpers import check_gelu_first_and_second_order
    check_gelu_first_and_second_order('cpu')


def test_fused_modconv_uses_bank_operand_from_pack_table():
    from helpers import check_fused_modconv_uses_bank_operand_from_pack_table
    check_fused_modconv_uses_bank_operand_from_pack_table('cpu')


def test_style_network_runs_on_linear_fn_and_matches_oracle():
    from helpers import check_style_network_on_linear_fn
    check_style_network_on_linear_fn('cpu')


def test_flat_optimizer_packs_and_grad_sink_match_autograd():
    from helpers import check_flat_optimizer_packs_and_grad_sink
    check_flat_optimizer_packs_and_grad_sink('cpu')


@pytest.mark.parametrize('C,act', [(32, None), (24, None), (64, 'silu')])
def test_rmsnorm_gain_gradient_through_the_finish_queue_matches_autograd(C, act):
    """under `ops.sinking()` ChannelRMSNorm's gain gradient is a queued column-sum finish into the parameter's .grad (no partial-sum fold,
    no AccumulateGrad launch); without it autograd delivers it - same numbers, also when the gain is used twice, and .grad keeps
    what was in it (accumulation)."""
    torch.manual_seed(0)
    H_ = ops.HipOps()
    x = bf(torch.randn(2, C, 5, 6)).float()
    w = torch.randn(2, C, 5, 6)
    gamma0 = torch.rand(C, 1, 1) + 0.5
    res = {}
    for tag in ('autograd', 'sink'):
        gamma = torch.nn.Parameter(gamma0.clone())
        gamma.grad = torch.full_like(gamma, 0.25)
        xi = x.clone().requires_grad_()
        with ops.use_impl(H_):
            y = H_.channel_rmsnorm(xi, gamma, act=act).float()
            y2 = H_.channel_rmsnorm(xi * 0.5, gamma, act=act).float()
            loss = (y * w).sum() + (y2 * w).sum() * 0.3
            if tag == 'sink':
                with ops.sinking():
                    loss.backward()
            else:
                loss.backward()
        res[tag] = (gamma.grad.clone(), xi.grad.clone())
    assert rel_err(res['sink'][0], res['autograd'][0]) < 1e-5 and rel_err(res['sink'][1], res['autograd'][1]) < 1e-6


@pytest.mark.parametrize('cfg', [(5, 2, 24, 40, 9), (3, 1, 16, 8, 9), (17, 3, 72, 33, 4), (32, 4, 40, 24, 1), (2, 2, 130, 70, 9)])
def test_modcoef_through_gram_rows_matches_the_direct_kernels(cfg):
    """gg_modgram + gg_modcoef_gram_fwd / _bwd (the coefficients through the bank's Gram rows: the tap sum leaves the per-sample work)
    against gg_modcoef_fwd / _bwd (walks (b, o, i, t)) and against autograd of the reference formulas (gp.py:378-400): s, a, d; gmod,
    gkernel_mod and the weights' gradient accumulated onto what gw held; with and without the external gs / ga; a clamped row."""
    b, N, O, I, T = cfg
    k = int(T ** 0.5)
    torch.manual_seed(b * 7 + N)
    w = torch.randn(N, O, I, k, k) * 0.2
    mod = torch.randn(b, I) * 0.4
    mod[0] = -1.0                                    # s = 0 for sample 0: sumsq = 0 -> the eps clamp is active there
    km = torch.randn(b, N) if N > 1 else None
    Ip, Op = (I + 7) // 8 * 8, (O + 7) // 8 * 8
    eps = 1e-8
    s0, a0, d0 = K.modcoef_fwd(w, mod, km, True, eps, Ip, Op)
    gram = K.modgram(w)
    s1, a1, d1, tsum = K.modcoef_gram_fwd(gram, N, mod, km, eps, Ip, Op)
    H = torch.einsum('noit,moit->nmoi', w.view(N, O, I, T), w.view(N, O, I, T))
    q = 0
    for n in range(N):
        for m in range(n, N):
            assert rel_err(gram[q], H[n, m] * (1. if n == m else 2.)) < 1e-5
            q += 1
    assert torch.equal(s1, s0) and rel_err(a1, a0) < 1e-6
    assert rel_err(d1[1:], d0[1:]) < 1e-5 and bool((d1[0, :O] == d0[0, :O]).all()) and float(d1[:, O:].abs().max() if Op > O else 0.) == 0.
    for with_ext in (True, False):
        gs = torch.randn(b, Ip) if with_ext else None
        ga = torch.randn(b, N) if (with_ext and N > 1) else None
        gd = torch.randn(b, Op)
        gw0 = torch.randn_like(w) * 0.1
        gw1 = gw0.clone()
        gm0, gk0 = K.modcoef_bwd(w, km, s0, d0, gs, ga, gd, gw0, eps)
        gm1, gk1 = K.modcoef_gram_bwd(w, gram, km, s1, d1, tsum, gs, ga, gd, gw1, eps)
        assert rel_err(gm1, gm0) < 2e-5 and rel_err(gw1, gw0) < 2e-5
        if N > 1:
            assert rel_err(gk1, gk0) < 2e-5
        gm2, _ = K.modcoef_gram_bwd(w, gram, km, s1, d1, tsum, gs, ga, gd, None, eps)       # no weight gradient wanted
        assert rel_err(gm2, gm0) < 2e-5
    # autograd of the reference formulas
    wr, modr = w.clone().requires_grad_(), mod.clone().requires_grad_()
    kmr = km.clone().requires_grad_() if N > 1 else None
    a = kmr.softmax(-1) if N > 1 else torch.ones(b, 1)
    sr = modr + 1.
    mix = torch.einsum('bn,noikl->boikl', a, wr) * sr[:, None, :, None, None]
    dr = mix.pow(2).sum((2, 3, 4)).clamp(min=eps).rsqrt()
    gd = torch.randn(b, O)
    grads = torch.autograd.grad((dr * gd).sum(), [modr, wr] + ([kmr] if N > 1 else []))
    gdp = torch.zeros(b, Op); gdp[:, :O] = gd
    gwz = torch.zeros_like(w)
    gm, gk = K.modcoef_gram_bwd(w, gram, km, s1, d1, tsum, None, None, gdp, gwz, eps)
    assert rel_err(gm[1:], grads[0][1:]) < 1e-4 and rel_err(gwz, grads[1]) < 1e-4
    if N > 1:
        assert rel_err(gk, grads[2]) < 1e-4


def test_generator_layer_major_modulations_equal_the_column_slices():
    """Generator._layer_major (the differentiable pass: one gather hands every layer a dense (b, I) block of the style -> modulation
    projection) against the plain `.split()` column slices: same blocks, and the same gradient back at the projection's output."""
    from gigagan_pytorch_amd.generator import Generator
    torch.manual_seed(0)
    G = Generator(image_size=32, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2),
                  unconditional=True, num_skip_layers_excite=2, self_attn_resolutions=())
    dims = tuple(G.style_embed_split_dims)
    mods = torch.randn(3, sum(dims), requires_grad=True)
    blocks = G._layer_major(mods, 3)
    want = mods.split(dims, dim=-1)
    assert len(blocks) == len(want) and all(b.is_contiguous() and torch.equal(b, w) for b, w in zip(blocks, want))
    probe = [torch.randn_like(w) for w in want]
    ga, = torch.autograd.grad(sum((b * p).sum() for b, p in zip(blocks, probe)), mods)
    gb, = torch.autograd.grad(sum((w * p).sum() for w, p in zip(want, probe)), mods)
    assert torch.equal(ga, gb)
    assert G._layer_major(mods, 3)[0].data_ptr() != blocks[0].data_ptr() and len(G._layer_major_idx) == 1      # (index table cached per batch)


@pytest.mark.parametrize('cfg', [(3, 40, 40, 32, 2, True, True), (2, 12, 9, 16, 1, False, True), (1, 64, 64, 64, 3, True, False)])
def test_modmix_backward_partials_folded_by_one_launch(cfg):
    """K.modmix_bwd (gg_modmix_bwd's chunk-major partial stacks folded by ONE gg_reduce_multi launch) against autograd of
    y = act(d * sum_n a_n Y_n + noise_w * noise) (gp.py:344-409 in the batched form), several chunks per image included."""
    b, H, W, O, N, demod, with_noise = cfg
    torch.manual_seed(0)
    Y = bf(torch.randn(b, H, W, N * O))
    a = torch.softmax(torch.randn(b, N), -1)
    d = torch.rand(b, O) + 0.5 if demod else None
    noise = torch.randn(b, H * W) if with_noise else None
    nw = torch.randn(O) * 0.3 if with_noise else None
    dy = bf(torch.randn(b, H, W, O))
    assert K._chunks(H * W, O) > 1 or H * W * O < 9000
    Yr, ar = Y.float().requires_grad_(), a.clone().requires_grad_()
    dr = d.clone().requires_grad_() if demod else None
    nwr = nw.clone().requires_grad_() if with_noise else None
    z = torch.einsum('bn,bhwno->bhwo', ar, Yr.view(b, H, W, N, O))
    if demod:
        z = z * dr[:, None, None, :]
    if with_noise:
        z = z + noise.view(b, H, W, 1) * nwr
    yr = torch.nn.functional.leaky_relu(z, 0.2)
    grads = torch.autograd.grad((yr * dy.float()).sum(), [Yr, ar] + ([dr] if demod else []) + ([nwr] if with_noise else []))
    y = K.modmix_fwd(Y, a, d, noise, nw, O, N, 'lrelu')
    dY, da, dd, dnw = K.modmix_bwd(dy, y, Y, a, d, noise, O, N, 'lrelu')
    mask = (yr.detach().abs() > 1e-2).all()         # (the sign of y decides the leaky-relu branch: bf16 y and fp32 y agree away from 0)
    assert rel_err(dY.float(), grads[0]) < 2e-2
    if N > 1:
        assert rel_err(da, grads[1]) < 2e-2
    k = 2
    if demod:
        assert rel_err(dd, grads[k]) < 2e-2
        k += 1
    if with_noise:
        assert rel_err(dnw, grads[k]) < 2e-2


def test_many_way_splitk_reduce_and_xcd_slice_mapping():
    """split counts above 8 take the wave-per-64-outputs reduce, and few-tile split-K launches of the 4-wave kernel use
    the slice-major (XCD-aware) 1-D grid: same numbers as the unsplit launch."""
    torch.manual_seed(0)
    a = bf(torch.randn(1, 72, 32 * 40)); b = bf(torch.randn(1, 24, 32 * 40))
    ref = torch.einsum('bmk,bnk->bmn', a.float(), b.float())
    for sk in (3, 12, 40):
        out = K.gemm(a, b, out_dtype=torch.float32, force_splitk=sk, force_tile=3)
        assert rel_err(out, ref) < 1e-5
    x = bf(torch.randn(2, 12, 12, 8)); dy = bf(torch.randn(2, 12, 12, 16))
    want = K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_splitk=1, force_tile=3)
    got = K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_splitk=9, force_tile=3)
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize('cfg', [(2, 16, 32, 16, 24), (1, 8, 64, 32, 32), (3, 16, 32, 64, 56), (1, 24, 32, 32, 64), (2, 8, 32, 64, 16)])
def test_direct_conv_matches_implicit_gemm(cfg):
    """gg_dconv (persistent workgroups, LDS halo tile, taps by pixel shift) == the implicit-GEMM kernel bit for bit
    (same bf16 products, fp32 accumulation in the same k order per MFMA chain is not guaranteed -> compare at fp32
    rounding), with the plain and the full epilogue (bias, leaky-relu, residual, alpha)."""
    n, H, W, ci, co = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, 9 * ci) * 0.1)
    bias = torch.randn(co); res = bf(torch.randn(n, H, W, co))
    ref = K.conv2d_nhwc(x, w, ksize=3, force_tile=1)
    got = K.conv2d_nhwc(x, w, ksize=3, force_tile=9)
    assert rel_err(got, ref) < 2e-3 and got.shape == ref.shape
    ref = K.conv2d_nhwc(x, w, ksize=3, bias=bias, act='lrelu', alpha=0.5, bias_scale=0.5, residual=res, force_tile=1)
    got = K.conv2d_nhwc(x, w, ksize=3, bias=bias, act='lrelu', alpha=0.5, bias_scale=0.5, residual=res, force_tile=9)
    assert rel_err(got, ref) < 2e-3
    exact = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert rel_err(K.conv2d_nhwc(x, w, ksize=3, force_tile=9), exact) < 4e-3
    s_in = torch.rand(n, ci) + 0.5                 # per-sample style modulation applied while staging the tile
    assert rel_err(K.conv2d_nhwc(x, w, ksize=3, in_scale=s_in, force_tile=9), K.conv2d_nhwc(x, w, ksize=3, in_scale=s_in, force_tile=1)) < 2e-3


@pytest.mark.parametrize('cfg', [(1, 16, 16, 64, 256, 7), (5, 8, 8, 128, 264, 7), (1, 8, 32, 64, 128, 8), (3, 16, 8, 64, 136, 8),
                                 (1, 64, 64, 64, 128, 7), (2, 16, 16, 128, 64, 8), (5, 8, 8, 64, 40, 8)])
def test_halo_staged_conv3_matches_fp32_convolution(cfg):
    """gg_conv3 (activation halo staged once per 64-channel chunk, nine taps read it at a tap-uniform offset, weight tiles by
    LDS-DMA into swizzled rows) against fp32 convolution of the same bf16 operands: whole-image tiles (H*W < 256, ragged
    image count), row tiles (H*W >= 256, non-square), ragged N, several channel chunks; plain, bias + activation and residual
    epilogues; and the planner hands ineligible geometries to the implicit GEMM."""
    n, H, W, ci, co, tile = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, 9 * ci) * 0.1)
    bias = torch.randn(co); res = bf(torch.randn(n, H, W, co))
    exact = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    K.plan_log = []
    got = K.conv2d_nhwc(x, w, ksize=3, out_dtype=torch.float32, force_tile=tile, force_splitk=1)
    assert K.plan_log == [(tile, 1)]
    K.plan_log = None
    assert rel_err(got, exact) < 1e-5
    got = K.conv2d_nhwc(x, w, ksize=3, bias=bias, act='lrelu', alpha=0.5, bias_scale=0.5, force_tile=tile, force_splitk=1)
    assert rel_err(got, F.leaky_relu(0.5 * exact + 0.5 * bias, 0.2)) < 4e-3
    got = K.conv2d_nhwc(x, w, ksize=3, residual=res, res_scale=0.5, force_tile=tile, force_splitk=1)
    assert rel_err(got, exact + 0.5 * res.float()) < 4e-3


@pytest.mark.parametrize('cfg', [(3, 8, 8, 128, 264, 7, 2), (1, 16, 16, 192, 128, 8, 3), (2, 32, 32, 64, 128, 8, 0)])
def test_halo_staged_conv3_split_over_channel_chunks(cfg):
    """gg_conv3 with the reduction split over its 64-channel chunks (the generator's 8x8 / 16x16 layers at batch 32 have too few
    row tiles for 256 CUs): fp32 partials + the split-K finish give the unsplit result, with the full epilogue (per-sample
    output scale, noise, leaky-relu) applied by the finish; forced and automatic split counts."""
    n, H, W, ci, co, tile, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, 9 * ci) * 0.1)
    exact = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    K.plan_log = []
    got = K.conv2d_nhwc(x, w, ksize=3, out_dtype=torch.float32, force_tile=tile, force_splitk=sk)
    assert K.plan_log[-1][0] == tile and (sk == 0 or K.plan_log[-1][1] == sk), K.plan_log
    K.plan_log = None
    assert rel_err(got, exact) < 1e-5
    d = torch.rand(n, co) + 0.5; nz = torch.randn(n * H * W); nw = torch.randn(co)
    got = K.conv2d_nhwc(x, w, ksize=3, out_scale=d, noise=nz, noise_w=nw, act='lrelu', force_tile=tile, force_splitk=max(sk, 2))
    want = F.leaky_relu(exact * d[:, None, None, :] + nz.view(n, H, W, 1) * nw, 0.2)
    assert rel_err(got, want) < 4e-3


@pytest.mark.parametrize('cfg', [(5, 8, 8, 64, 2, 264, 7, 0), (2, 16, 16, 128, 2, 128, 8, 2), (1, 32, 32, 64, 3, 128, 8, 1)])
def test_halo_staged_conv3_applies_the_bank_modulation_on_its_operand_staging(cfg):
    """the shared-bank adaptive conv (gp.py:378-409) as ONE contraction: N kernels stacked along the reduction (weights
    [co][tap][n][ci], CV = N * C virtual channels over C physical ones) with the per-(sample, stacked channel) scale a[b,n] * s[b,i]
    applied when the halo chunk is parked in LDS (gg_conv3 SCALED) == the convolution of the explicitly modulated N-fold
    activation; whole-image tiles with several images per tile (8x8), split and unsplit."""
    n, H, W, ci, N, co, tile, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, 9 * N * ci) * 0.1)
    insc = (torch.rand(n, N * ci) + 0.5)
    x2 = bf(torch.cat([x.float() * insc[:, None, None, j * ci:(j + 1) * ci] for j in range(N)], dim=-1))     # (n, H, W, N*ci)
    want = K.conv2d_nhwc(x2, w, ksize=3, out_dtype=torch.float32, force_tile=1)
    K.plan_log = []
    got = K.conv2d_nhwc(x, w, ksize=3, cv=N * ci, in_scale=insc, out_dtype=torch.float32, force_tile=tile, force_splitk=sk)
    assert K.plan_log[-1][0] == tile, K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < 1e-5
    # unforced: the planner may pick it by modelled cost; whatever runs must agree
    got = K.conv2d_nhwc(x, w, ksize=3, cv=N * ci, in_scale=insc, out_dtype=torch.float32)
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize('cfg', [(19, 4, 4, 64, 1, 72, 2), (5, 8, 8, 32, 1, 64, 1), (3, 16, 16, 96, 1, 136, 3), (18, 4, 4, 32, 2, 64, 2),
                                 (6, 8, 8, 64, 2, 40, 4), (2, 16, 16, 32, 3, 64, 0)])
def test_low_resolution_conv_matches_fp32_convolution_and_the_modulated_bank(cfg):
    """gg_lrconv (plan tile 11: 4x4 / 8x8 / 16x16 images, whole images per 256-pixel tile, 32-channel chunks carrying all nine taps,
    partial image tiles and output-channel tiles, split over the channel chunks or not) == the fp32 convolution; with a stacked
    bank (CV = N * C, per-(sample, stacked channel) scale on the halo store) == the convolution of the explicitly modulated
    N-fold activation; the finish applies the adaptive conv's epilogue (demodulation, noise, leaky-relu)."""
    n, H, W, ci, N, co, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(co, 9 * N * ci) * 0.1)
    if N == 1:
        want = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        kw = {}
    else:
        insc = (torch.rand(n, N * ci) + 0.5)
        x2 = bf(torch.cat([x.float() * insc[:, None, None, j * ci:(j + 1) * ci] for j in range(N)], dim=-1))
        want = K.conv2d_nhwc(x2, w, ksize=3, out_dtype=torch.float32, force_tile=1)
        kw = dict(cv=N * ci, in_scale=insc)
    K.plan_log = []
    got = K.conv2d_nhwc(x, w, ksize=3, out_dtype=torch.float32, force_tile=11, force_splitk=sk, **kw)
    assert K.plan_log[-1][0] == 11 and (sk == 0 or K.plan_log[-1][1] == min(sk, N * ci // 32)), K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < 1e-5
    d = torch.rand(n, co) + 0.5; nz = torch.randn(n * H * W); nw = torch.randn(co)
    got = K.conv2d_nhwc(x, w, ksize=3, out_scale=d, noise=nz, noise_w=nw, act='lrelu', force_tile=11, force_splitk=sk, **kw)
    ref = F.leaky_relu(want * d[:, None, None, :] + nz.view(n, H, W, 1) * nw, 0.2)
    assert rel_err(got, ref) < 4e-3


@pytest.mark.parametrize('cfg', [(3, 64, 72, 0), (2, 96, 64, 3), (1, 32, 40, 1)])
def test_low_resolution_conv_mixes_the_bank_per_image(cfg):
    """gg_lrconv MIX (16x16 images, two stacked banks): w_img = a[img,0] W_0 + a[img,1] W_1 formed while the weight blocks are staged,
    the activation scaled by s[img, i], the reduction over the C physical channels == the per-image convolution with the
    reference's mixed kernel (gp.py:378-386), bf16-rounded where the kernel rounds; split and unsplit, full epilogue in the finish."""
    n, ci, co, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, 16, 16, ci)); w = bf(torch.randn(co, 9, 2, ci) * 0.1)
    s = torch.rand(n, ci) + 0.5; a = torch.softmax(torch.randn(n, 2), dim=-1)
    xs = bf(x.float() * s[:, None, None, :]).float()
    wm = bf(w.float()[None, :, :, 0, :] * a[:, 0, None, None, None] + w.float()[None, :, :, 1, :] * a[:, 1, None, None, None]).float()
    want = torch.stack([F.conv2d(xs[i:i + 1].permute(0, 3, 1, 2), wm[i].view(co, 3, 3, ci).permute(0, 3, 1, 2), padding=1)[0]
                        for i in range(n)]).permute(0, 2, 3, 1)
    K.plan_log = []
    got = K.conv2d_nhwc(x, w.view(co, -1), ksize=3, cv=2 * ci, in_scale=s, bank_mix=a, out_dtype=torch.float32, force_splitk=sk)
    assert K.plan_log[-1][0] == 11, K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < 1e-5
    d = torch.rand(n, co) + 0.5; nz = torch.randn(n * 256); nw = torch.randn(co)
    got = K.conv2d_nhwc(x, w.view(co, -1), ksize=3, cv=2 * ci, in_scale=s, bank_mix=a, out_scale=d, noise=nz, noise_w=nw, act='lrelu',
                        force_splitk=sk)
    assert rel_err(got, F.leaky_relu(want * d[:, None, None, :] + nz.view(n, 16, 16, 1) * nw, 0.2)) < 4e-3
    with pytest.raises(RuntimeError):       # other shapes are refused, not silently run unmixed
        K.conv2d_nhwc(bf(torch.randn(2, 8, 8, 32)), bf(torch.randn(64, 9 * 64)), ksize=3, cv=64, in_scale=torch.ones(2, 32),
                      bank_mix=torch.ones(2, 2) * 0.5)


@pytest.mark.parametrize('cfg', [(3, 16, 16, 64, 128, 8, 1), (2, 32, 32, 128, 256, 7, 2), (2, 16, 32, 64, 72, 8, 0), (2, 32, 32, 64, 64, 8, 1),
                                 (2, 16, 16, 128, 48, 8, 2)])
def test_halo_staged_conv3_with_per_image_weights(cfg):
    """per-sample weights (the reference's own formulation of the adaptive conv, gp.py:390-409): image i is convolved with w[i];
    row tiles lie inside one image (H * W >= 256)."""
    n, H, W, ci, co, tile, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); w = bf(torch.randn(n, co, 9 * ci) * 0.1)
    want = torch.stack([F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2), w[i].float().view(co, 3, 3, ci).permute(0, 3, 1, 2),
                                 padding=1)[0].permute(1, 2, 0) for i in range(n)])
    K.plan_log = []
    got = K.conv2d_nhwc(x, w, ksize=3, per_image_weights=True, out_dtype=torch.float32, force_tile=tile, force_splitk=sk)
    assert K.plan_log[-1][0] == tile, K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < 1e-5
    nz = torch.randn(n * H * W); nw = torch.randn(co)
    got = K.conv2d_nhwc(x, w, ksize=3, per_image_weights=True, noise=nz, noise_w=nw, act='lrelu', force_tile=tile, force_splitk=sk)
    assert rel_err(got, F.leaky_relu(want + nz.view(n, H, W, 1) * nw, 0.2)) < 4e-3


@pytest.mark.parametrize('cfg', [(1, 16, 16, 32, 256, 0), (3, 8, 8, 64, 264, 2), (1, 32, 32, 32, 128, 4), (1, 8, 64, 32, 40, 0),
                                 (2, 16, 8, 96, 256, 1)])
def test_nine_tap_weight_gradient_matches_autograd(cfg):
    """gg_wgrad9 (all nine taps of a 32-channel slice per workgroup: dy tile shared, x operand = the k-tile's halo read through
    transpose reads at a tap-uniform offset) against autograd's conv weight gradient on the same bf16 operands: whole-image and
    row k-tiles, several channel slices, ragged output-channel tiles, automatic and forced split-K."""
    n, H, W, ci, co, sk = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); dy = bf(torch.randn(n, H, W, co))
    w = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    want = w.grad.permute(2, 3, 1, 0).reshape(-1, co)
    K.plan_log = []
    got = K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=10, force_splitk=sk)
    assert K.plan_log[-1][0] == 10 and (sk == 0 or K.plan_log[-1][1] == sk), K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < 1e-5
    # ineligible geometries (4x4 images, 24 channels, stride 2) are planned on the implicit GEMM
    K.plan_log = []
    K.conv2d_wgrad_nhwc(bf(torch.randn(2, 4, 4, 32)), bf(torch.randn(2, 4, 4, 64)), ksize=3, force_tile=10)
    K.conv2d_wgrad_nhwc(bf(torch.randn(1, 8, 8, 24)), bf(torch.randn(1, 8, 8, 64)), ksize=3, force_tile=10)
    assert all(t != 10 for t, _ in K.plan_log), K.plan_log
    K.plan_log = None


@pytest.mark.parametrize('sk', [2, 5, 8, 12])
def test_splitk_finish_paths_bf16_batched_and_epilogue(sk):
    """the vectorised finish (2..8 slices, four columns per thread) and the many-slice finish give the unsplit result: bf16 and
    fp32 outputs, a batched launch, bias + leaky-relu applied after the slices are summed."""
    torch.manual_seed(0)
    A = bf(torch.randn(2, 40, 32 * sk)); B = bf(torch.randn(2, 24, 32 * sk)); bias = torch.randn(24)
    for kw in (dict(out_dtype=torch.float32), dict(), dict(bias=bias, act='lrelu', alpha=0.25)):
        want = K.gemm(A, B, force_tile=1, force_splitk=1, **kw)
        K.plan_log = []
        got = K.gemm(A, B, force_tile=1, force_splitk=sk, **kw)
        assert K.plan_log[-1] == (1, sk), K.plan_log
        K.plan_log = None
        assert rel_err(got, want) < (1e-5 if kw.get('out_dtype') is torch.float32 else 4e-3), (sk, kw.keys())


def test_committed_plan_table_loads_and_is_honoured_by_the_planner():
    """plans/gfx950.json (tests/gpu_plan_sweep.py output) is accepted entry by entry by gg_gemm_plan_table, every tile it names is
    one the planner knows, and an installed entry (here: the nine-tap weight gradient with a measured split) overrides the cost
    model for exactly its geometry; ineligible entries are ignored."""
    import json
    from gigagan_pytorch_amd import _C
    L = _C.lib()
    entries = json.loads(_C.PLAN_TABLE.read_text())['entries']
    assert len(entries) > 50 and all(1 <= e['tile'] <= 15 and e['splitk'] >= 1 for e in entries)
    assert sum(e['tile'] == 15 for e in entries) >= 50          # (round 5: the persistent short-K contraction's geometries)
    try:
        L.load_plan_table(entries)
        assert L.plan_entries == len(entries)
        x = bf(torch.randn(1, 16, 16, 32)); dy = bf(torch.randn(1, 16, 16, 64))
        key = dict(M=288, N=64, K=256, batch=1, a_layout=1, b_layout=1, a_conv=1, H=16, W=16, C=32, CV=32, R=3, conv_stride=1,
                   conv_pad=1, c_is_f32=1, d2s=0, epi=0, scaled=0)
        L.load_plan_table([dict(key, tile=10, splitk=2), dict(key, N=72, tile=7, splitk=1)])
        K.plan_log = []
        got = K.conv2d_wgrad_nhwc(x, dy, ksize=3)
        K.conv2d_wgrad_nhwc(x, bf(torch.randn(1, 16, 16, 72)), ksize=3)     # tile 7 is not a weight-gradient kernel: entry ignored
        assert K.plan_log[0] == (10, 2) and K.plan_log[1][0] not in (7, 8), K.plan_log
        assert rel_err(got, K.conv2d_wgrad_nhwc(x, dy, ksize=3, force_tile=1)) < 1e-5
        # a tile-15 entry (gg_pgemm) on a 1x1 convolution far below the planner's own row threshold; ineligible (fp32 output): ignored
        k1 = dict(M=256, N=64, K=64, batch=1, a_layout=0, b_layout=0, a_conv=1, H=16, W=16, C=64, CV=64, R=1, conv_stride=1,
                  conv_pad=0, c_is_f32=0, d2s=0, epi=0, scaled=0)
        L.load_plan_table([dict(k1, tile=15, splitk=1), dict(k1, c_is_f32=1, tile=15, splitk=1)])
        x1 = bf(torch.randn(1, 16, 16, 64)); w1 = bf(torch.randn(64, 64) * 0.1)
        assert K.conv2d_nhwc(x1, w1, ksize=1, pad=0, plan_only=True) == (15, 1)
        assert K.conv2d_nhwc(x1, w1, ksize=1, pad=0, out_dtype=torch.float32, plan_only=True)[0] != 15
        assert torch.equal(K.conv2d_nhwc(x1, w1, ksize=1, pad=0), K.conv2d_nhwc(x1, w1, ksize=1, pad=0, force_tile=6))
    finally:
        K.plan_log = None
        L.load_plan_table([])


def test_halo_staged_conv3_is_not_planned_for_ineligible_geometries():
    torch.manual_seed(0)
    K.plan_log = []
    for (H, W, ci, ks, kw) in [(4, 4, 64, 3, {}), (16, 16, 32, 3, {}), (12, 16, 64, 3, {}), (16, 16, 64, 1, {})]:
        x = bf(torch.randn(2, H, W, ci)); w = bf(torch.randn(128, ks * ks * ci) * 0.1)
        got = K.conv2d_nhwc(x, w, ksize=ks, force_tile=7, **kw)
        assert rel_err(got, K.conv2d_nhwc(x, w, ksize=ks, force_tile=1, **kw)) < 2e-3
    assert all(t not in (7, 8) for t, _ in K.plan_log[::2]), K.plan_log
    K.plan_log = None


@pytest.mark.parametrize('cfg', [(5, 2, 10, 12, 3), (18, 1, 8, 16, 1), (3, 4, 6, 20, 3)])
def test_fused_adaptive_conv_coefficients_match_tensor_algebra(cfg):
    from helpers import check_modcoef
    check_modcoef(cfg, 'cpu')


def test_narrow_modconv_layers_take_the_streaming_convolution():
    """no-grad adaptive conv on a narrow high-resolution layer: per-sample weights (gg_modw_fwd) + the streaming direct
    convolution (gg_sconv_fwd), no implicit-GEMM launch at all, against the oracle."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    torch.manual_seed(0)
    I, O, b, H, W = 16, 16, 2, 128, 256
    conv = AdaptiveConv2DMod(I, O, 3, num_conv_kernels=2)
    x, mod, km = torch.randn(b, I, H, W), torch.randn(b, I) * 0.3, torch.randn(b, 2)
    nz, nw = torch.randn(b, 1, H, W), torch.randn(O, 1, 1) * 0.1
    try:
        with torch.no_grad():
            K.plan_log = []
            y1 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu')
            assert K.plan_log == []                           # no contraction launch: gg_modw_fwd + gg_sconv_fwd
    finally:
        K.plan_log = None
    with torch.no_grad(), ops.use_impl(OracleOps(bf16_operands=True)):
        y0 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu')
    assert rel_err(y1, y0) < 1e-2


def test_shared_bank_modconv_paths_match_oracle():
    """wide low-resolution no-grad adaptive conv (shared bank, the N kernels stacked along the reduction): from 8x8 up the
    per-(sample, stacked channel) scale rides on the convolution's operand staging (in_scale, no modulated copy of the activation:
    gg_conv3 SCALED from 8x8 up, the low-resolution kernel - plan tile 11 - on 4x4 images); odd sizes keep one pointwise pass for
    both kernels of the bank + the plain gather. Demodulation / noise / activation in the epilogue; all against the oracle."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    torch.manual_seed(0)
    conv = AdaptiveConv2DMod(64, 72, 3, num_conv_kernels=2)
    for res, want_calls in ((8, 0), (4, 0), (2, 1)):
        x, mod, km = torch.randn(2, 64, res, res), torch.randn(2, 64) * 0.3, torch.randn(2, 2)
        nz, nw = torch.randn(2, 1, res, res), torch.randn(72, 1, 1) * 0.1
        calls, orig = [], K.modulate_bank
        K.desc_log = []
        try:
            K.modulate_bank = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            K.plan_log = []
            with torch.no_grad():
                y1 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu')
            assert len(calls) == want_calls, (res, calls)
            assert res != 4 or K.plan_log[-1][0] == 11, K.plan_log
            from gigagan_pytorch_amd._C import GemmDesc
            d = GemmDesc.from_buffer_copy(K.desc_log[-1])
            assert bool(d.in_scale) == (want_calls == 0) and d.CV == (128 if want_calls == 0 else d.C)
        finally:
            K.modulate_bank, K.desc_log, K.plan_log = orig, None, None
        with torch.no_grad(), ops.use_impl(OracleOps(bf16_operands=True)):
            y0 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu')
        assert rel_err(y1, y0) < 1e-2, res


# ---- no-grad forward of the adaptive convolution (gg_modfwd.h) -----------------------------------------------------------

@pytest.mark.parametrize('cfg', [(5, 2, 24, 32, 3), (3, 1, 16, 16, 3), (4, 3, 8, 48, 3), (2, 2, 40, 64, 1)])
def test_modw_coefficients_and_per_sample_weights_match_the_reference_formulation(cfg):
    """gg_modw_fwd: s, a, d and the per-sample weights (both layouts) against the reference's tensor algebra (gp.py:378-400:
    softmax over the kernels, (mod + 1), demodulation over (i, k))."""
    b, N, O, I, k = cfg
    torch.manual_seed(0)
    w = torch.randn(N, O, I, k, k) * 0.2
    mod, kmod = torch.randn(b, I) * 0.5, (torch.randn(b, N) if N > 1 else None)
    r8 = lambda n: (n + 7) // 8 * 8
    attn = kmod.softmax(-1) if N > 1 else torch.ones(b, 1)
    wts = (w[None] * attn[:, :, None, None, None, None]).sum(1) * (mod[:, None, :, None, None] + 1)
    inv = wts.pow(2).sum(dim=(2, 3, 4), keepdim=True).clamp(min=1e-8).rsqrt()
    ref_w = wts * inv                                                       # (b, O, I, k, k): the reference's per-sample weights
    s, a, d = K.modw_fwd(w, mod, kmod, True, 1e-8, r8(I), r8(O))
    assert torch.allclose(s[:, :I], mod + 1, atol=1e-6) and torch.allclose(a, attn, atol=1e-6)
    assert rel_err(d[:, :O], inv.reshape(b, O)) < 1e-5 and float(d[:, O:].abs().max() if r8(O) > O else 0.) == 0.
    wm1 = torch.zeros(b, O, k * k * I, dtype=torch.bfloat16)
    K.modw_fwd(w, mod, kmod, True, 1e-8, r8(I), r8(O), coef=False, wmix=wm1, layout=1)
    assert rel_err(wm1.float().view(b, O, k * k, I), ref_w.reshape(b, O, I, k * k).transpose(2, 3)) < 4e-3
    if I % 16 == 0 and O <= 32 and k == 3:
        wm2 = torch.zeros(b, 9, I // 16, 32, 16, dtype=torch.bfloat16)
        K.modw_fwd(w, mod, kmod, True, 1e-8, r8(I), r8(O), coef=False, wmix=wm2, layout=2)
        got = wm2.float().permute(0, 3, 2, 4, 1).reshape(b, 32, I, 9)     # (b, o, i, tap)
        assert rel_err(got[:, :O], ref_w.reshape(b, O, I, 9)) < 4e-3 and float(got[:, O:].abs().max() if O < 32 else 0.) == 0.
    _, _, d0 = K.modw_fwd(w, mod, kmod, False, 1e-8, r8(I), r8(O))
    assert torch.equal(d0[:, :O], torch.ones(b, O))


def test_multi_layer_modulation_launch_equals_the_per_layer_launches():
    """gg_modw_multi_fwd (every layer's coefficients / per-sample weights in one launch) == one gg_modw_fwd per layer, bit for
    bit: a coefficient-only layer (+ the stacked input scale a[b,n] * s[b,i]), a per-image-weight layer (layout 1), a streaming
    layer (layout 2), a single-kernel layer (N = 1), with modulation rows that are column slices of one wide matrix."""
    torch.manual_seed(0)
    b = 5
    cfgs = [(2, 24, 32, 'coef'), (2, 16, 24, 'rows'), (2, 16, 32, 'bank'), (1, 8, 16, 'coef'), (3, 8, 8, 'coef')]
    wide = torch.randn(b, sum(I + N for N, O, I, _ in cfgs)) * 0.5            # the style projection all slices come from
    layers, want, col = [], [], 0
    for N, O, I, kind in cfgs:
        w = torch.randn(N, O, I, 3, 3) * 0.2
        mod, kmod = wide[:, col:col + I], (wide[:, col + I:col + I + N] if N > 1 else None)
        col += I + N
        ly = dict(w=w, mod=mod, kmod=kmod, demod=True, eps=1e-8, Ip=I, Op=O)
        if kind == 'coef':
            s, a, d = K.modw_fwd(w, mod, kmod, True, 1e-8, I, O)
            want.append(dict(s=s, a=a, d=d, insc=(a[:, :, None] * s[:, None, :]).reshape(b, N * I)))
        elif kind == 'rows':
            wm = torch.zeros(b, O, 9 * I, dtype=torch.bfloat16)
            K.modw_fwd(w, mod, kmod, True, 1e-8, I, O, coef=False, wmix=wm, layout=1)
            want.append(dict(wmix=wm))
            ly.update(coef=False, wmix=torch.zeros_like(wm), layout=1)
        else:
            wm = torch.zeros(b, 9, I // 16, 32, 16, dtype=torch.bfloat16)
            K.modw_fwd(w, mod, kmod, True, 1e-8, I, O, coef=False, wmix=wm, layout=2)
            want.append(dict(wmix=wm))
            ly.update(coef=False, wmix=torch.zeros_like(wm), layout=2)
        layers.append(ly)
    outs = K.modw_multi(layers)
    for ly, o, w_ in zip(layers, outs, want):
        if 'wmix' in w_:
            assert torch.equal(ly['wmix'], w_['wmix']) and o['d'] is None
        else:
            assert torch.equal(o['s'], w_['s']) and torch.equal(o['a'], w_['a']) and torch.equal(o['d'], w_['d'])
            assert torch.allclose(o['insc'], w_['insc'], rtol=1e-6, atol=1e-7)
    # cached Gram rows (pack table kind 2): coefficient-only layers then run a wave per channel without touching the bank
    for ly in layers:
        wf = ly['w'].flatten(3)
        N = wf.shape[0]
        ly['gram'] = torch.stack([(1. if n == m else 2.) * (wf[n] * wf[m]).sum(-1) for n in range(N) for m in range(n, N)]).contiguous()
    outs_g = K.modw_multi(layers)
    for ly, o, w_ in zip(layers, outs_g, want):
        if 'wmix' in w_:
            assert rel_err(ly['wmix'], w_['wmix']) < 1e-6
        else:
            assert torch.equal(o['s'], w_['s']) and torch.allclose(o['a'], w_['a'], rtol=1e-6, atol=1e-7)
            assert torch.allclose(o['d'], w_['d'], rtol=2e-6, atol=1e-7) and torch.allclose(o['insc'], w_['insc'], rtol=1e-6, atol=1e-7)
    # more layers than one launch holds (16): split transparently
    many = [dict(w=layers[0]['w'], mod=layers[0]['mod'], kmod=layers[0]['kmod'], demod=True, eps=1e-8, Ip=32, Op=24) for _ in range(19)]
    outs = K.modw_multi(many)
    assert len(outs) == 19 and all(torch.equal(o['d'], want[0]['d']) for o in outs)


def test_generator_announces_its_adaptive_convs_and_every_layer_uses_the_batched_modulation():
    """no-grad Generator.forward on the kernels: ONE gg_modw_multi_fwd launch at the top for the 3x3 demodulated layers (layers
    behind a skip-layer excitation included where their convolution can apply the excitation to the staged weights), per-layer
    gg_modw_fwd only for the rest, Gram rows from the pack table when the parameters are optimizer-owned, the registry is empty
    afterwards; images equal the per-layer-launch path (prepare disabled; bf16 rounding of the excited layers' weights differs)
    and the oracle within bf16 tolerance."""
    from gigagan_pytorch_amd.generator import Generator
    from helpers import SMALL_G
    torch.manual_seed(0)
    G = Generator(**SMALL_G).eval()
    z = torch.randn(2, 32)
    calls = []
    real_multi, real_single = K.modw_multi, K.modw_fwd
    K.modw_multi = lambda layers: (calls.append(('multi', len(layers))), real_multi(layers))[1]
    K.modw_fwd = lambda *a, **k: (calls.append(('single',)), real_single(*a, **k))[1]
    try:
        with torch.no_grad():
            torch.manual_seed(1)
            img = G(noise=z)
            n_multi = [c for c in calls if c[0] == 'multi']
            n_single = [c for c in calls if c[0] == 'single']
            assert len(n_multi) == 1 and not ops._prepared, (calls, list(ops._prepared))
            excited = max(G.num_layers - G.num_skip_layers_excite, 0) if G.num_skip_layers_excite else 0
            assert n_multi[0][1] + len(n_single) == 1 + 2 * G.num_layers and len(n_single) <= excited, (calls, excited)
            calls.clear()
            prep, ops.HipOps.modconv_prepare = ops.HipOps.modconv_prepare, lambda self, specs: 0
            try:
                torch.manual_seed(1)
                img0 = G(noise=z)
            finally:
                ops.HipOps.modconv_prepare = prep
            assert not [c for c in calls if c[0] == 'multi'] and rel_err(img, img0) < 5e-3
            with ops.use_impl(OracleOps(bf16_operands=True)):
                torch.manual_seed(1)
                ref = G(noise=z)
            assert rel_err(img, ref) < 2e-2
            # optimizer-owned parameters: the Gram rows come from the pack table and follow an optimizer step
            from gigagan_pytorch_amd.optimizer import FlatAdamW
            opt = FlatAdamW(list(G.parameters()), lr=1e-2)
            torch.manual_seed(1)
            img1 = G(noise=z)
            assert '_gg_tpacks' in G.init_conv.weights.__dict__ and 'gram' in G.init_conv.weights._gg_tpacks
            assert rel_err(img1, img) < 2e-2        # (the style network and the packed operands take their table paths too now)
            with torch.no_grad():
                opt.flat_g.normal_()
            opt.step()
            wts = G.init_conv.weights.detach().flatten(3)
            want = torch.stack([(wts[0] * wts[0]).sum(-1), 2 * (wts[0] * wts[1]).sum(-1), (wts[1] * wts[1]).sum(-1)])
            assert torch.allclose(G.init_conv.weights._gg_tpacks['gram'][0], want, rtol=1e-5, atol=1e-6)
    finally:
        K.modw_multi, K.modw_fwd = real_multi, real_single


@pytest.mark.parametrize('cfg', [(2, 16, 64, 16, 16), (1, 8, 32, 32, 32), (3, 8, 64, 64, 24), (2, 5, 32, 32, 8)])
def test_streaming_conv_matches_per_sample_convolution(cfg):
    """gg_sconv_fwd against F.conv2d with groups = batch on the same bf16 operands, with noise + leaky-relu, image borders and
    a shared bank."""
    b, H, W, C, O = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(b, H, W, C))
    wps = bf(torch.randn(b, O, C, 3, 3) * 0.1)
    wm = torch.zeros(b, 9, C // 16, 32, 16, dtype=torch.bfloat16)
    wm[:, :, :, :O] = wps.reshape(b, O, C // 16, 16, 9).permute(0, 4, 2, 1, 3)
    nz, nw = torch.randn(b * H * W), torch.randn(O) * 0.3
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).reshape(1, b * C, H, W), wps.float().reshape(b * O, C, 3, 3), padding=1, groups=b)
    ref = ref.reshape(b, O, H, W) + nz.view(b, 1, H, W) * nw.view(1, O, 1, 1)
    y = K.sconv(x, wm, O, nz, nw, 'lrelu')
    assert rel_err(y.permute(0, 3, 1, 2), F.leaky_relu(ref, 0.2)) < 4e-3
    y1 = K.sconv(x, wm[:1].contiguous(), O)                               # one shared bank, plain epilogue
    ref1 = F.conv2d(x.float().permute(0, 3, 1, 2), wps[0].float(), padding=1)
    assert rel_err(y1.permute(0, 3, 1, 2), ref1) < 4e-3


def _layout2(wps, C):
    b, O = wps.shape[:2]
    wm = torch.zeros(b, 9, C // 16, 32, 16, dtype=torch.bfloat16)
    wm[:, :, :, :O] = wps.reshape(b, O, C // 16, 16, 9).permute(0, 4, 2, 1, 3)
    return wm


@pytest.mark.parametrize('model', ['early', 'late', 'early-reverse'])
@pytest.mark.parametrize('cfg', [
    # b, H, W, C0, C1, C2, excitation, noise, shared banks
    (1, 11, 256, 32, 16, 16, True, True, False),       # the 256x256 pair of config 2 (two strips: 8 + 3 rows), every operand
    (2, 5, 128, 64, 32, 32, True, True, False),        # the 128x128 pair (conv2's bank in LDS), one strip per image
    (1, 4, 256, 32, 16, 8, False, False, True),        # no noise maps, no excitation, shared banks, ragged output channel count
    (1, 19, 128, 64, 32, 24, False, True, True),       # three strips (8 + 8 + 3 rows)
    (1, 6, 256, 32, 16, 24, True, True, False),        # 24 output channels at 256 wide: the 32x32x16 form of that geometry
])
def test_fused_streaming_pair_is_bit_identical_to_two_streaming_convolutions(cfg, model, monkeypatch):
    """gg_spair_fwd (loader wave + LDS-DMA x ring with the two noise maps riding along, conv1 -> bf16 mid ring in LDS -> conv2, one
    barrier per row) against gg_sconv_fwd(gg_sconv_fwd(x)): the same accumulation order and epilogue expressions, so torch.equal -
    under both DMA landing models of the emulator and both fiber orders (a mis-counted wait or a ring slot refilled too early shows)."""
    b, H, W, C0, C1, C2, excite, noise, shared = cfg
    if model.startswith('late'):
        monkeypatch.setenv('GG_EMU_DMA', 'late')
    if model.endswith('reverse'):
        monkeypatch.setenv('GG_EMU_REVERSE', '1')
    assert K.spair_supported(H, W, C0, C1, C2) and not K.spair_supported(H, W, C0, C1 * 2, C2)
    torch.manual_seed(0)
    x = bf(torch.randn(b, H, W, C0))
    nb = 1 if shared else b
    w1 = _layout2(bf(torch.randn(nb, C1, C0, 3, 3) * 0.1), C0)
    w2 = _layout2(bf(torch.randn(nb, C2, C1, 3, 3) * 0.2), C1)
    xs = torch.rand(b, C0) + 0.5 if excite else None
    n1 = n2 = nw1 = nw2 = None
    if noise:
        n1, n2 = torch.randn(b * H * W), torch.randn(b * H * W)
        nw1, nw2 = torch.randn(C1) * 0.3, torch.randn(C2) * 0.3
    mid = K.sconv(x, w1, C1, n1, nw1, 'lrelu', xs=xs)
    want = K.sconv(mid, w2, C2, n2, nw2, 'lrelu')
    got = K.spair(x, w1, w2, C1, C2, n1, nw1, n2, nw2, 'lrelu', 'lrelu', xs=xs)
    # the 32 -> 16 -> <= 16 block runs on 16x16x32 MFMAs (K = 32 per instruction, conv2's taps paired): the same products in another
    # summation order - equal to bf16 rounding, and exact on integer operands (tests/test_exact_integer.py); every other geometry: the same bits
    same = (lambda a, b_: torch.equal(a, b_)) if not (C0 == 32 and C2 <= 16) else (lambda a, b_: rel_err(a, b_) < 4e-3)
    assert got.shape == want.shape and same(got, want)
    if not noise:       # the plain epilogues (no activation on either stage)
        want = K.sconv(K.sconv(x, w1, C1, xs=xs), w2, C2)
        assert same(K.spair(x, w1, w2, C1, C2, xs=xs), want)


def test_modulate_bank_lays_the_kernels_side_by_side():
    torch.manual_seed(0)
    x, s, a = bf(torch.randn(2, 4, 4, 16)), torch.rand(2, 16) + 0.5, torch.rand(2, 3)
    out = K.modulate_bank(x, s, a)
    want = torch.cat([x.float() * (s * a[:, n:n + 1])[:, None, None, :] for n in range(3)], dim=-1)
    assert out.shape == (2, 4, 4, 48) and rel_err(out, want) < 4e-3


@pytest.mark.parametrize('cfg', [(16, 16, 32, 64, True), (64, 32, 16, 32, True), (64, 72, 8, 8, True), (32, 24, 8, 8, False),
                                 (64, 40, 32, 32, True), (128, 128, 32, 32, True)])
def test_no_grad_adaptive_conv_paths_match_oracle(cfg):
    """the whole no-grad forward through ops.modconv2d: the streaming path (narrow layers), per-sample weights through the implicit
    GEMM (mid resolutions) and the stacked path (wide layers),
    with demodulation on and off, noise and activation, against the oracle."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    I, O, H, W, demod = cfg
    torch.manual_seed(0)
    conv = AdaptiveConv2DMod(I, O, 3, num_conv_kernels=2, demod=demod)
    b = 4 if H * W >= 2048 else 2
    x, mod, km = torch.randn(b, I, H, W), torch.randn(b, I) * 0.3, torch.randn(b, 2)
    nz, nw = torch.randn(b, 1, H, W), torch.randn(O, 1, 1) * 0.1
    with torch.no_grad():
        y1 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu')
        with ops.use_impl(OracleOps(bf16_operands=True)):
            y0 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu')
    assert rel_err(y1, y0) < 1e-2


@pytest.mark.parametrize('dh', [64, 32])
def test_fully_masked_rows_attend_uniformly_like_the_reference(dh):
    """a batch item whose key mask keeps nothing: the reference fills every score with -finfo.max (gp.py:645-647), so the softmax is
    uniform over all keys and the output the plain mean of the values - ops.attention returns that (fused family at 64 features per
    head, probability-tensor path otherwise), output and gradients, next to ordinary and leading-masked items."""
    torch.manual_seed(0)
    B, h, n, m = 3, 2, 16, 12
    q, k, v = (bf(torch.randn(B, h, t, dh)).float().requires_grad_() for t in (n, m, m))
    mask = torch.ones(B, m, dtype=torch.bool)
    mask[0, 7:] = False
    mask[1] = False                      # the caption without a single token
    mask[2, :2] = False
    probe = torch.randn(B, h, n, dh)
    sim = torch.einsum('bhid,bhjd->bhij', q, k) * dh ** -0.5
    sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(sim.dtype).max)
    want = torch.einsum('bhij,bhjd->bhid', sim.softmax(-1), v)
    gw = torch.autograd.grad((want * probe).sum(), (q, k, v))
    got = ops.HipOps().attention(q, k, v, scale=dh ** -0.5, key_mask=mask)
    gg = torch.autograd.grad((got.float() * probe).sum(), (q, k, v))
    assert rel_err(got[1], v[1].mean(dim=1, keepdim=True).expand(-1, n, -1)) < 8e-3
    assert rel_err(got, want) < 8e-3
    for a, b_ in zip(gg, gw):
        assert rel_err(a, b_) < 2e-2
    assert float(gg[0][1].abs().max()) == 0 and float(gg[1][1].abs().max()) == 0      # a uniform row does not depend on q or k


def test_no_grad_block_pair_runs_as_one_launch_and_matches_the_two_layer_path():
    """ops.modconv_pair (what Generator._synthesise calls for a block's conv1 -> noise -> leaky-relu -> conv2 -> noise -> leaky-relu in a
    no-grad pass): at a geometry gg_spair_fwd carries both layers run as ONE launch, bit-identical to the two modconv2d calls - with
    the first layer behind a skip-layer excitation, announced (modconv_prepare) or not; other geometries return None."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    torch.manual_seed(0)
    b, H, W = 2, 128, 128
    c1, c2 = AdaptiveConv2DMod(64, 32, 3, num_conv_kernels=2), AdaptiveConv2DMod(32, 32, 3, num_conv_kernels=2)
    x = torch.randn(b, 64, H, W)
    m1, k1, m2, k2 = torch.randn(b, 64) * 0.3, torch.randn(b, 2), torch.randn(b, 32) * 0.3, torch.randn(b, 2)
    n1, n2 = torch.randn(b, 1, H, W), torch.randn(b, 1, H, W)
    nw1, nw2 = torch.randn(32, 1, 1) * 0.1, torch.randn(32, 1, 1) * 0.1
    exc = torch.rand(b, 64, 1, 1) + 0.5
    impl = ops.HipOps()
    first = dict(weights=c1.weights, mod=m1, kernel_mod=k1, demod=True, eps=1e-8, noise=n1, noise_weight=nw1, act='lrelu', in_excite=exc)
    second = dict(weights=c2.weights, mod=m2, kernel_mod=k2, demod=True, eps=1e-8, noise=n2, noise_weight=nw2, act='lrelu', in_excite=None)
    calls = []
    real = K.spair
    K.spair = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad(), ops.use_impl(impl):
            mid = c1(x, m1, k1, noise=n1, noise_weight=nw1, act='lrelu', in_excite=exc)
            want = c2(mid, m2, k2, noise=n2, noise_weight=nw2, act='lrelu')
            got = impl.modconv_pair(x, first, second)
            assert calls == [1] and torch.equal(got, want)
            # announced: the per-sample weights come from the forward's batched launch, the excitation rides on the pair's weight staging
            specs = [(c1.weights, m1, k1, H, W, True, True, 1e-8), (c2.weights, m2, k2, H, W, False, True, 1e-8)]
            assert impl.modconv_prepare(specs) == 2
            want2 = c2(c1(x, m1, k1, noise=n1, noise_weight=nw1, act='lrelu', in_excite=exc), m2, k2, noise=n2, noise_weight=nw2, act='lrelu')
            assert impl.modconv_prepare(specs) == 2
            got2 = impl.modconv_pair(x, first, second)
            impl.modconv_release()
            assert calls == [1, 1] and torch.equal(got2, want2)
            assert rel_err(got2, want) < 8e-3          # ((w s d) rounded, then * e rounded: one rounding more than (w s e d))
            # a pair the kernel does not carry (64 -> 64): the caller runs the layers one by one
            c3 = AdaptiveConv2DMod(32, 64, 3, num_conv_kernels=2)
            assert impl.modconv_pair(mid, dict(second, weights=c3.weights, noise_weight=torch.randn(64, 1, 1)), second) is None
            ops._SPAIR, keep = False, ops._SPAIR                            # GG_SPAIR=0: the A/B switch hands every pair back
            try:
                assert impl.modconv_pair(x, first, second) is None
            finally:
                ops._SPAIR = keep
        with ops.use_impl(impl):                                         # gradients flow: never fused
            assert impl.modconv_pair(x.requires_grad_(), first, second) is None
    finally:
        K.spair = real


def test_forked_conv_and_norm_match_the_plain_fork_first_and_second_order():
    """`fork=True` (the second consumer's gradient joins inside the op's backward pass: dgrad GEMM epilogue / rmsnorm_bwd carry)
    gives the same first- and second-order gradients as letting autograd accumulate the two branches."""
    torch.manual_seed(0)
    H_ = ops.HipOps()
    x0 = bf(torch.randn(2, 16, 8, 8)).float()
    w1 = (torch.randn(24, 16, 3, 3) * 0.2).requires_grad_()
    w2 = (torch.randn(16, 16, 1, 1) * 0.3).requires_grad_()
    gamma = (torch.rand(16, 1, 1) + 0.5).requires_grad_()

    def run(fork):
        x = x0.clone().requires_grad_()
        with ops.use_impl(H_):
            if fork:
                n, xs = H_.channel_rmsnorm(x, gamma, fork=True)
                y, n2 = H_.conv2d(n, w1, None, act='lrelu', fork=True)
            else:
                n, xs = H_.channel_rmsnorm(x, gamma), x
                y, n2 = H_.conv2d(n, w1, None, act='lrelu'), n
            z = H_.conv2d(n2, w2, None, residual=xs)          # second consumer of n; the skip joins in its epilogue
            loss = (y.float() * torch.linspace(-1, 1, y.numel()).view_as(y)).sum() + z.float().pow(2).sum()
            g1 = torch.autograd.grad(loss, [w1, w2, gamma], retain_graph=True)
            gx, = torch.autograd.grad(loss, x, create_graph=True)
            g2 = torch.autograd.grad(gx.float().pow(2).sum(), [x, w1, w2, gamma])
        return gx.detach(), g1, g2

    gxa, g1a, g2a = run(True); gxb, g1b, g2b = run(False)
    assert rel_err(gxa, gxb) < 1e-2
    for a, b in zip(g1a, g1b):
        assert rel_err(a, b) < 1e-2
    for a, b in zip(g2a, g2b):
        assert rel_err(a, b) < 3e-2


def test_pool_mean_kernels_and_row_fork_match_tensor_algebra():
    """gg_pool_mean_fwd / _bwd (SqueezeExcite's pool and the one-pass merge of its gradient with the trunk's) and RowsForkFn
    (a predictor's rows gradient added into the trunk gradient in place) vs plain autograd."""
    torch.manual_seed(0)
    H_ = ops.HipOps()
    x0 = bf(torch.randn(3, 24, 6, 10)).float()
    wt = torch.randn(3, 24)

    def run(fork):
        x = x0.clone().requires_grad_()
        with ops.use_impl(H_):
            xa = H_.prepare(x)
            if fork:
                m, xt = H_.global_mean(xa, fork=True)
                rows, xt = H_.take_rows(xt, 2, fork=True)
            else:
                m, xt = xa.float().mean(dim=(2, 3)), xa
                rows = xt[:2]
            loss = (m * wt).sum() + (xt.float() * 0.5).pow(2).sum() + rows.float().sin().sum()
            g, = torch.autograd.grad(loss, x)
        return m.detach(), g

    ma, ga = run(True); mb, gb = run(False)
    assert rel_err(ma, mb) < 1e-5
    assert rel_err(ga, gb) < 1e-2
    # the plain broadcast (no trunk gradient)
    gs = torch.randn(3, 24)
    y = K.pool_mean_bwd(gs.contiguous(), (3, 6, 10, 24))
    assert rel_err(y.float(), gs[:, None, None, :].expand(3, 6, 10, 24)) < 5e-3


def test_add_cat_merge_matches_tensor_algebra():
    """gg_addcat_fwd / _bwd (the discriminator's multi-scale input merge, gp.py:1797-1803) vs cat((x + tile(f), tile(f)))."""
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps()
    x0, f0 = bf(torch.randn(6, 16, 4, 6)).float(), bf(torch.randn(2, 16, 4, 6)).float()
    w = torch.randn(12, 16, 4, 6)

    def run(I):
        x, f = x0.clone().requires_grad_(), f0.clone().requires_grad_()
        with ops.use_impl(I):
            y = I.add_cat(I.prepare(x), I.prepare(f))
        gx, gf = torch.autograd.grad((y.float() * w).sum(), [x, f])
        return y.float(), gx, gf

    ya, gxa, gfa = run(H_); yb, gxb, gfb = run(O_)
    assert rel_err(ya, yb) < 5e-3 and rel_err(gxa, gxb) < 5e-3 and rel_err(gfa, gfb) < 1e-2


@pytest.mark.parametrize('gp', [False, True])
def test_discriminator_parameter_gradients_on_kernels_vs_oracle(gp):
    """a 4-stage discriminator with skip-layer excitation, two multi-scale inputs and predictors: every parameter gradient of
    (logits + multi-scale logits [+ gradient penalty]) on the kernel path (forked convs, in-place row merge, pool / add-cat
    kernels, predictor merge) against the bf16-operand oracle. This is the check that catches a wrong backward formula in the
    fused glue (forward values alone do not)."""
    from gigagan_pytorch_amd.discriminator import Discriminator
    from gigagan_pytorch_amd.gigagan import gradient_penalty
    torch.manual_seed(0)
    D = Discriminator(image_size=32, dim_capacity=8, dim_max=32, unconditional=True, num_skip_layers_excite=2,
                      attn_resolutions=(), multiscale_input_resolutions=(16, 8))
    imgs = torch.rand(2, 3, 32, 32)

    def run(I):
        D.zero_grad()
        with ops.use_impl(I):
            x = imgs.clone().requires_grad_(gp)
            rgbs = D.real_images_to_rgbs(x)
            ops.second_order = gp
            try:
                logits, ms, _ = D(x, rgbs, return_multiscale_outputs=True, calc_aux_loss=False)
            finally:
                ops.second_order = False
            loss = logits.float().sum() + sum(m.float().sum() for m in ms)
            if gp:
                loss = loss + gradient_penalty(x, outputs=[logits, *ms], grad_output_weights=[1.] * (1 + len(ms)))
            loss.backward()
        return float(loss), torch.cat([p.grad.flatten() for p in D.parameters() if p.grad is not None]).clone()

    la, ga = run(ops.HipOps()); lb, gb = run(OracleOps(bf16_operands=True))
    assert abs(la - lb) < 1e-2 * abs(lb)
    assert rel_err(ga, gb) < (4e-2 if gp else 2e-2), rel_err(ga, gb)


def test_generator_parameter_gradients_through_the_discriminator_vs_oracle():
    """generator step in miniature: loss = -(D(G(z)) logits + multi-scale logits) summed; every generator parameter gradient
    on the kernel path (adaptive convs with grad, skip-layer excitation fork, upsampling, forks in D) vs the bf16-operand oracle."""
    from gigagan_pytorch_amd.discriminator import Discriminator
    from gigagan_pytorch_amd.generator import Generator
    torch.manual_seed(0)
    G = Generator(image_size=32, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2),
                  unconditional=True, num_skip_layers_excite=2, self_attn_resolutions=())
    D = Discriminator(image_size=32, dim_capacity=8, dim_max=32, unconditional=True, num_skip_layers_excite=2,
                      attn_resolutions=(), multiscale_input_resolutions=(16, 8))
    for n in (m for m in G.modules() if hasattr(m, 'weight') and m.__class__.__name__ == 'Noise'):
        torch.nn.init.normal_(n.weight, std=0.1)
    z = torch.randn(2, 32)

    def run(I):
        G.zero_grad()
        torch.manual_seed(1)
        with ops.use_impl(I):
            img, rgbs = G(noise=z, return_all_rgbs=True)
            logits, ms, _ = D(img, rgbs, return_multiscale_outputs=True, calc_aux_loss=False)
            loss = -(logits.float().sum() + sum(m.float().sum() for m in ms))
            loss.backward()
        return float(loss), torch.cat([p.grad.flatten() for p in G.parameters() if p.grad is not None]).clone()

    la, ga = run(ops.HipOps()); lb, gb = run(OracleOps(bf16_operands=True))
    assert abs(la - lb) < 2e-2 * abs(lb)
    assert rel_err(ga, gb) < 5e-2, rel_err(ga, gb)


@pytest.mark.parametrize('shape', [(2, 2, 8, 16), (3, 2, 8, 8)])
def test_fused_linear_attention_on_qkv_slices_vs_oracle(shape):
    """LinearAttnFn (gg_linattn_q / _k softmax passes + strided head-view GEMMs on the fused to_qkv tensor) vs the oracle's
    einsum formulation (unet.py:338-348): output and the gradient w.r.t. the fused qkv tensor; batched over the heads of one image
    (b <= heads) and over the images of one head (b > heads: the upsampler's batch 16 x 8 heads)."""
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps(bf16_operands=True)
    b, heads, x, y = shape
    d = 64
    c = heads * d
    qkv0 = bf(torch.randn(b, 3 * c, x, y) * 1.5).float()
    w = torch.randn(b, c, x, y)

    def run(fused):
        qkv = qkv0.clone().requires_grad_()
        if fused:
            with ops.use_impl(H_):
                out = H_.linear_attention_qkv(qkv, heads=heads, scale=d ** -0.5)
            assert out is not None
        else:
            q, k, v = qkv.chunk(3, dim=1)
            out = O_.linear_attention(q, k, v, heads=heads, scale=d ** -0.5)
        g, = torch.autograd.grad((out.float() * w).sum(), qkv)
        return out.float().detach(), g

    oa, ga = run(True); ob, gb = run(False)
    assert rel_err(oa, ob) < 1.5e-2, rel_err(oa, ob)
    assert rel_err(ga, gb) < 3e-2, rel_err(ga, gb)
    c3 = 3 * c
    for name, sl in (('dq', slice(0, c)), ('dk', slice(c, 2 * c)), ('dv', slice(2 * c, c3))):
        assert rel_err(ga[:, sl], gb[:, sl]) < 4e-2, (name, rel_err(ga[:, sl], gb[:, sl]))


def test_rmsnorm_with_fused_silu_matches_oracle():
    """gg_rmsnorm with act = silu (forward and first-order backward incl. dgamma) vs norm followed by silu on the oracle."""
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps()
    x0 = bf(torch.randn(2, 24, 6, 5)).float(); gamma0 = torch.rand(24) + 0.5
    w = torch.randn(2, 24, 6, 5)

    def run(I):
        x = x0.clone().requires_grad_(); gamma = gamma0.clone().requires_grad_()
        with ops.use_impl(I):
            y = I.channel_rmsnorm(x, gamma.view(24, 1, 1), act='silu').float()
        return y, torch.autograd.grad((y * w).sum(), [x, gamma])

    ya, ga = run(H_); yb, gb = run(O_)
    assert rel_err(ya, yb) < 6e-3
    assert rel_err(ga[0], gb[0]) < 2e-2 and rel_err(ga[1], gb[1]) < 2e-2


@pytest.mark.parametrize('C', [32, 64, 128, 256, 512])
@pytest.mark.parametrize('silu', [False, True])
def test_rmsnorm_rows_kernel_matches_the_wave_per_row_kernel(C, silu, monkeypatch):
    """gg_rmsnorm_rows_kernel (C = 8 * LPR <= 512: LPR lanes per pixel row, 64 / LPR rows per wavefront, two passes in flight) against
    gg_rmsnorm_kernel (a wavefront per row; GG_RMS_ROWS=0) on ragged row counts, with and without the carry: same summation order
    (the group butterflies are the leading steps of the 64-lane ones), so y and dx agree except where the compiler contracts an fma
    in one kernel and not the other (<= 1 bf16 ulp, or 1e-4 absolute where the terms cancel, on <= 0.1 % of the elements); the gain gradient to fp32 summation order; and both
    against the fp32 formulas (gp.py:224-232)."""
    torch.manual_seed(C + silu)
    for rows in (1, 37, 64 * 4 * 2 + 5, 1201):
        x = bf(torch.randn(rows, C) * 1.7)
        x[rows // 2] = x[rows // 2] * 1e-9            # a row below eps (the clamped branch)
        g = bf(torch.randn(rows, C))
        carry = bf(torch.randn(rows, C))
        gamma = torch.rand(C) + 0.5
        res = {}
        for tag, env in (('rows', '1'), ('wave', '0')):
            monkeypatch.setenv('GG_RMS_ROWS', env)
            y = K.rmsnorm_fwd(x, gamma, silu)
            dx, dgam = K.rmsnorm_bwd(x, g, gamma, True, None, silu=silu)
            dxc, _ = K.rmsnorm_bwd(x, g, gamma, False, carry, silu=silu)
            res[tag] = (y, dx, dxc, dgam)
            if not silu:          # the second-order pass (gradient penalty): v = carry as the gradient w.r.t. dx
                gx2, gg2, dg2 = K.rmsnorm_bwd2(x, g, carry, gamma, True)
                res[tag + '2'] = (gx2, gg2, dg2)
        for a, b in zip(res['rows'][:3], res['wave'][:3]):
            d = (a.float() - b.float()).abs()
            assert float((d > 0).float().mean()) <= 1e-3 and bool((d <= 2. ** -7 * b.float().abs() + 1e-4).all()), (C, silu, rows)
        assert rel_err(res['rows'][3], res['wave'][3]) < 1e-5
        if not silu:
            for a, b in zip(res['rows2'][:2], res['wave2'][:2]):
                assert rel_err(a.float(), b.float()) < 2e-3 and float(((a.float() - b.float()).abs() > 0).float().mean()) <= 2e-2, (C, rows)
            assert rel_err(res['rows2'][2], res['wave2'][2]) < 1e-5
        xf = x.float().requires_grad_()
        gm = gamma.clone().requires_grad_()
        n = xf.norm(dim=-1, keepdim=True).clamp(min=K.RMS_EPS)
        yr = xf / n * C ** 0.5 * gm
        if silu:
            yr = torch.nn.functional.silu(yr)
        gx, gg = torch.autograd.grad((yr * g.float()).sum(), [xf, gm])
        assert rel_err(res['rows'][0].float(), yr.detach()) < 6e-3
        assert rel_err(res['rows'][1].float(), gx) < 1e-2 and rel_err(res['rows'][3], gg) < 1e-2
        assert rel_err(res['rows'][2].float(), gx + carry.float()) < 1e-2


@pytest.mark.parametrize('shape', [(2, 8, 8, 16), (1, 6, 4, 8), (2, 2, 2, 24)])
def test_maxpool_highfreq_kernels_vs_torch(shape):
    from helpers import check_maxpool_highfreq
    check_maxpool_highfreq(shape, 'cpu')


@pytest.mark.parametrize('cfg', [(2, 2, 64, 64, False, False, True), (1, 2, 200, 77, False, True, False), (2, 1, 78, 78, True, True, False),
                                 (1, 1, 130, 130, False, False, True), (3, 1, 96, 77, False, 2, False), (3, 1, 70, 40, True, 2, False)])
def test_general_fused_attention_forward_backward_vs_autograd(cfg):
    from helpers import check_general_attention
    check_general_attention(cfg, 'cpu')


def test_shape_fallbacks_are_counted_and_only_for_shape_reasons():
    """ops.shape_fallbacks (what the BASELINE-dimension GPU parity tests assert to stay empty): a ragged channel count sends RMSNorm and
    GELU-on-a-strided-view to their tensor-algebra forms and is counted; the same ops at kernel-friendly shapes are not; forms taken
    because a graph is differentiated twice (ops.second_order) are by design and not counted."""
    torch.manual_seed(0)
    h = ops.HipOps()
    ops.shape_fallbacks.clear()
    with ops.use_impl(h), torch.no_grad():
        h.channel_rmsnorm(torch.randn(2, 16, 4, 4), torch.ones(16))
        h.gelu(bf(torch.randn(2, 16, 4, 4)))
        assert not ops.shape_fallbacks, ops.shape_fallbacks
        y = h.channel_rmsnorm(torch.randn(2, 12, 4, 4), torch.ones(12))                 # C % 8 != 0
        assert torch.isfinite(y.float()).all() and ops.shape_fallbacks.get('channel_rmsnorm') == 1
        h.gelu(bf(torch.randn(2, 16, 4, 6))[..., ::2])                                 # not a dense view
        assert ops.shape_fallbacks.get('gelu') == 1, ops.shape_fallbacks
        ops.shape_fallbacks.clear()
        ops.second_order = True
        try:
            h.maxpool_highfreq(torch.randn(2, 8, 4, 4))
        finally:
            ops.second_order = False
        assert not ops.shape_fallbacks, ops.shape_fallbacks
    ops.shape_fallbacks.clear()


@pytest.mark.parametrize('cfg', [
    # n, H, W, ci, co, ksize, splitk      (W x channels decide pixels per step and prefetch depth; splitk 0 = the planner's choice)
    (1, 64, 64, 32, 32, 3, 0),       # one block, four k-step shares, four image rows per step
    (2, 64, 64, 8, 32, 3, 3),        # the discriminator stem (3 -> 8 padded channels): 16-byte slots, garbage rows never stored
    (1, 128, 128, 16, 8, 3, 5),      # to-rgb-like narrow output, two rows per step
    (1, 64, 64, 64, 64, 3, 2),       # four blocks (one per wave), 128-pixel steps
    (1, 64, 64, 32, 64, 3, 7),       # two output blocks, two k-step shares
    (1, 256, 256, 32, 32, 3, 16),    # full-width rows of a 256 x 256 image: one image row per step, the benchmark's layer
    (2, 64, 64, 16, 32, 1, 0),       # 1x1 (no halo ring)
    (1, 64, 128, 64, 8, 1, 3),
    (1, 128, 128, 32, 32, 2, 4),     # 2x2 / stride 2 (space-to-depth + 1x1): super-pixel slots of 64 channels, two ring rows per output row
    (2, 128, 128, 8, 16, 2, 0),
    (1, 128, 256, 16, 64, -1, 3),    # 1x1 / stride 2 (ksize -1 here): even input rows, the first C of the 2C slot channels stored
    (2, 128, 128, 8, 32, -1, 0),
])
def test_streaming_weight_gradient_matches_autograd(cfg):
    """gg_wgrads (plan tile 13: rows streamed once through LDS rings by LDS-DMA, counted vmcnt waits, raw barriers, transpose-read
    fragments) against autograd's conv weight gradient on the same bf16 operands. Run under BOTH DMA landing models of the emulator
    in CI (GG_EMU_DMA=late retires a transfer only at the counted wait that covers it: a mis-counted wait reads stale rows)."""
    n, H, W, ci, co, ks, sk = cfg
    stride, pad = (2, 0) if ks in (2, -1) else (1, ks // 2)
    ks = abs(ks)
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci)); dy = bf(torch.randn(n, H // stride, W // stride, co))
    w = torch.zeros(co, ci, ks, ks, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, stride=stride, padding=pad).backward(dy.float().permute(0, 3, 1, 2))
    want = w.grad.permute(2, 3, 1, 0).reshape(-1, co)
    K.plan_log = []
    got = K.conv2d_wgrad_nhwc(x, dy, ksize=ks, stride=stride, pad=pad, force_tile=13, force_splitk=sk)
    assert K.plan_log[-1][0] == 13 and (sk == 0 or K.plan_log[-1][1] == sk), K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < 1e-5


def test_streaming_weight_gradient_takes_over_the_thin_layers_only():
    """the planner substitutes tile 13 for the 4-wave kernel on >= 64K-pixel launches of eligible geometry; everything else (wide
    layers, small maps, strided windows, forced tiles) stays where it was."""
    K.plan_log = []
    K.conv2d_wgrad_nhwc(bf(torch.randn(16, 64, 64, 32)), bf(torch.randn(16, 64, 64, 32)), ksize=3)         # 64K pixels: streamed
    K.conv2d_wgrad_nhwc(bf(torch.randn(2, 64, 64, 32)), bf(torch.randn(2, 64, 64, 32)), ksize=3)           # 8K pixels: not worth it
    K.conv2d_wgrad_nhwc(bf(torch.randn(1, 32, 32, 32)), bf(torch.randn(1, 32, 32, 32)), ksize=3, force_tile=13)   # 32-wide: ineligible
    K.conv2d_wgrad_nhwc(bf(torch.randn(1, 64, 64, 24)), bf(torch.randn(1, 64, 64, 32)), ksize=3, force_tile=13)   # 24 channels
    K.conv2d_wgrad_nhwc(bf(torch.randn(16, 64, 64, 32)), bf(torch.randn(16, 64, 64, 32)), ksize=3, force_tile=3)
    tiles = [t for t, _ in K.plan_log]
    K.plan_log = None
    assert tiles[0] == 13 and all(t != 13 for t in tiles[1:]), tiles


@pytest.mark.parametrize('cfg', [
    # n, H, W, ci, co, per-image weights, epilogue
    (1, 64, 64, 32, 32, False, 'plain'),        # four image rows per step, every wave 64 pixels x 32 channels
    (2, 64, 64, 8, 32, False, 'bias_act'),      # the discriminator stem: 16-byte slots, zero weight fragments beyond 8 channels
    (1, 128, 128, 16, 16, False, 'noise_act'),  # narrow output: 32 staged bytes per pixel
    (1, 64, 128, 64, 64, False, 'residual'),    # 64 channels in (two planes, 128-pixel steps) and out (channel halves on wave pairs)
    (1, 64, 64, 32, 64, False, 'bias_act'),
    (1, 128, 128, 64, 32, False, 'plain'),
    (1, 256, 256, 32, 16, False, 'noise_act'),  # full-width rows of a 256 x 256 image
    (3, 64, 64, 32, 32, True, 'noise_act'),     # per-image weights (the adaptive convolution's no-grad form) reloaded at image changes
    (2, 128, 128, 16, 8, True, 'scale'),        # + per-image input scale folded into the weight fragments
    (2, 64, 64, 32, 24, False, 'scale'),        # ragged output channel count (24), shared weights with a per-image input scale
])
def test_streaming_forward_convolution_matches_conv2d(cfg):
    """gg_sfwd (plan tile 14: loader wave + LDS-DMA row ring, weights as register-resident MFMA fragments, per-wave output staging)
    against F.conv2d on the same bf16 operands, every epilogue it carries."""
    n, H, W, ci, co, per_img, epi = cfg
    torch.manual_seed(0)
    x = bf(torch.randn(n, H, W, ci))
    w = bf(torch.randn(n if per_img else 1, co, 9 * ci) * 0.1)
    kw, scale = {}, None
    if epi == 'scale':
        scale = torch.rand(n, ci) + 0.5
        kw['in_scale'] = scale
    xs = x.float() if scale is None else bf(x.float() * scale[:, None, None, :]).float()    # (the kernel scales the weights instead)
    want = torch.stack([F.conv2d(xs[i:i + 1].permute(0, 3, 1, 2), w[i if per_img else 0].float().view(co, 3, 3, ci).permute(0, 3, 1, 2),
                                 padding=1)[0].permute(1, 2, 0) for i in range(n)])
    if epi == 'bias_act':
        b = torch.randn(co); kw.update(bias=b, act='lrelu', alpha=0.5, bias_scale=0.5)
        want = F.leaky_relu(0.5 * (want + b), 0.2)
    elif epi == 'noise_act':
        nz = torch.randn(n * H * W); nw = torch.randn(co); kw.update(noise=nz, noise_w=nw, act='lrelu')
        want = F.leaky_relu(want + nz.view(n, H, W, 1) * nw, 0.2)
    elif epi == 'residual':
        r = bf(torch.randn(n, H, W, co)); kw.update(residual=r, res_scale=0.5)
        want = want + 0.5 * r.float()
    K.plan_log = []
    got = K.conv2d_nhwc(x, w if per_img else w[0], ksize=3, per_image_weights=per_img, force_tile=14, **kw)
    assert K.plan_log[-1][0] == 14, K.plan_log
    K.plan_log = None
    assert rel_err(got, want) < (8e-3 if epi == 'scale' else 4e-3)


def test_streaming_forward_takes_over_where_it_was_measured_faster():
    """planner policy for tile 14 (profiles/r04_sfwd_ab.log): every eligible >= 64K-pixel launch the 4-wave kernel would run (the stem,
    64 -> 64), and the direct convolution's launches only when they carry a bias / activation / noise / residual epilogue."""
    K.plan_log = []
    x8, x32 = bf(torch.randn(16, 64, 64, 8)), bf(torch.randn(16, 64, 64, 32))
    K.conv2d_nhwc(x8, bf(torch.randn(32, 72)), ksize=3)                                        # stem: 4-wave -> streamed
    K.conv2d_nhwc(x32, bf(torch.randn(32, 288)), ksize=3, bias=torch.randn(32), act='lrelu')   # direct conv with a full epilogue -> streamed
    K.conv2d_nhwc(x32, bf(torch.randn(32, 288)), ksize=3)                                      # plain: the direct convolution stays
    K.conv2d_nhwc(x32[:2], bf(torch.randn(32, 288)), ksize=3, bias=torch.randn(32))            # 8K pixels: not worth a persistent launch
    K.conv2d_nhwc(x32, bf(torch.randn(32, 288)), ksize=3, bias=torch.randn(32), act='gelu')    # no GELU in the branch-free epilogue
    tiles = [t for t, _ in K.plan_log]
    K.plan_log = None
    assert tiles[0] == 14 and tiles[1] == 14 and tiles[2] == 9 and tiles[3] != 14 and tiles[4] != 14, tiles


def test_queued_finishes_match_the_single_launches():
    """kernels.FinishQueue -> gg_reduce_multi + gg_finish_multi: weight-gradient finishes (plain, with up to 16 split-K slices summed on
    the way in, and deep slice stacks folded by the batched reduce that a flush runs first), bias
    column sums and dense accumulations, more items than one batch carries (40), a repeated destination (flushes before it is queued
    again) - against the single-launch kernels / plain tensor algebra. Notifications run after the launch that wrote their item."""
    torch.manual_seed(0)
    q = K.FinishQueue()
    want, dsts, fired = [], [], []
    for j in range(45):
        O, I, T = 8 * (1 + j % 5), 8 * (1 + j % 3), (9, 1, 4)[j % 3]
        nsplit = (1, 3, 5)[j % 3] if j % 7 else (17, 40, 131)[j % 3]         # (> 16 slices: folded by the flush's batched reduce first)
        g = torch.randn(nsplit, T * I, O) if nsplit > 1 else torch.randn(T * I, O)
        dst = torch.randn(O, I, T)
        ref = dst + 0.5 * K.wgrad_finish(g.sum(0) if nsplit > 1 else g, O, I, T, 1.0).view(O, I, T)
        q.add_wgrad(g, O, I, T, 0.5, dst.view(-1), nsplit=nsplit, notify=lambda j=j: fired.append(j))
        want.append(ref); dsts.append(dst)
    part, bias = torch.randn(37, 64), torch.randn(50)
    want.append(bias + 2.0 * part[:, :50].sum(0)); dsts.append(bias)
    q.add_colsum(part, 50, 2.0, bias)
    src, acc = torch.randn(1003), torch.randn(1003)
    want.append(acc - 0.25 * src); dsts.append(acc)
    q.add_axpy(src, -0.25, acc)
    assert len(fired) == 40, 'the first full batch was not flushed when the 40th item arrived'
    again = torch.randn(9 * 8, 8)
    ref2 = want[0] + K.wgrad_finish(again, 8, 8, 9, 1.0).view(8, 8, 9)
    q.add_wgrad(again, 8, 8, 9, 1.0, dsts[0].view(-1))          # same destination as item 0 (already written: no flush needed)
    q.add_wgrad(again, 8, 8, 9, 1.0, dsts[0].view(-1))          # ... and again while it is queued: flushes first
    q.flush()
    assert len(fired) == 45 and not q.items
    for got, ref in zip(dsts[1:], want[1:]):
        assert rel_err(got, ref) < 1e-6
    assert rel_err(dsts[0], ref2 + K.wgrad_finish(again, 8, 8, 9, 1.0).view(8, 8, 9)) < 1e-6


def test_feedforward_with_gelu_on_the_gemm_epilogues_matches_the_separate_passes():
    """ops.FFTailFn (gp.py:726-740 behind the norm): the up-projection's staged epilogue stores h and gelu(h) (gelu_mode 1), the
    down-projection's data gradient leaves the GEMM already multiplied by gelu'(h) (gelu_mode 2). Output, input gradient and every
    parameter gradient against the separate conv / gg_gelu / conv Functions (bit-identical forward: the same bf16 rounding points and
    the same Phi / phi arithmetic) and against fp32 math; no-grad form (GELU as the epilogue activation) included."""
    from gigagan_pytorch_amd.modules import FeedForward, _ff_residual
    torch.manual_seed(0)
    ff = FeedForward(dim=128, mult=4, channel_first=True)
    for p_ in ff.parameters():
        if p_.dim() == 1:
            torch.nn.init.normal_(p_, std=0.3) if p_.shape[0] != 128 or p_ is ff[3].bias else None
    x = torch.randn(2, 128, 16, 16)
    probe = torch.randn(2, 128, 16, 16)

    def run(fused, grad=True):
        xx = x.clone().requires_grad_(grad)
        for p_ in ff.parameters():
            p_.grad = None
        saved, ops.second_order = ops.second_order, not fused           # (second-order graphs keep the separate Functions)
        K.plan_log = []
        try:
            with ops.use_impl(ops.HipOps()), torch.set_grad_enabled(grad):
                y = _ff_residual(ff, xx)
                if grad:
                    (y.float() * probe).sum().backward()
        finally:
            ops.second_order = saved
            plans, K.plan_log = K.plan_log, None
        return y.detach().float(), (xx.grad.float() if grad else None), [p_.grad.clone() for p_ in ff.parameters()] if grad else None, plans

    y1, dx1, g1, plans1 = run(True)
    y0, dx0, g0, _ = run(False)
    assert sum(1 for t, sk in plans1 if 4 <= t <= 6 and sk == 1) >= 2 and len(plans1) == 6, plans1   # both GELU-carrying launches on the staged epilogue; 2 forward + 4 backward contractions, no GELU pass
    assert torch.equal(y1, y0), float((y1 - y0).abs().max())
    assert rel_err(dx1, dx0) < 4e-3
    for a, b in zip(g1, g0):
        assert rel_err(a, b) < 4e-3, (a.shape, rel_err(a, b))
    y2, _, _, _ = run(True, grad=False)
    assert rel_err(y2, y0) < 4e-3
    with ops.use_impl(OracleOps()):
        xx = x.clone().requires_grad_()
        for p_ in ff.parameters():
            p_.grad = None
        yr = _ff_residual(ff, xx)
        (yr * probe).sum().backward()
    assert rel_err(y1, yr.detach()) < 1e-2 and rel_err(dx1, xx.grad) < 3e-2
    for a, p_ in zip(g1, ff.parameters()):
        assert rel_err(a, p_.grad) < 3e-2, (a.shape, rel_err(a, p_.grad))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_fused_hinge_losses_match_the_reference_formulation(dtype):
    """ops.HingeFn (gg_hinge: one launch per logit tensor, one more for its backward) against gp.py:157-163 on the halves of a merged
    discriminator batch (fake rows first) and on the generator's mean, values and gradients."""
    from gigagan_pytorch_amd.gigagan import discriminator_hinge_loss, generator_hinge_loss
    torch.manual_seed(0)
    b = 6
    for shape in ((1, 2 * b), (3, 2 * b, 4, 4), (2, 2 * b, 5), (2, 2 * b, 37, 41), (1, 2 * b, 128, 128)):   # (the last two: 18 / 64 workgroups)
        x = (torch.randn(shape) * 1.5).to(dtype)
        xr = x.clone().float().requires_grad_()
        want = discriminator_hinge_loss(xr[:, b:], xr[:, :b]) * 0.7
        want.backward()
        xh = x.clone().requires_grad_()
        got = ops.HipOps().hinge(xh, b) * 0.7
        got.backward()
        assert abs(float(got) - float(want)) <= 1e-5 * max(1., abs(float(want))), (shape, float(got), float(want))
        assert rel_err(xh.grad.float(), xr.grad) < (4e-3 if dtype == torch.bfloat16 else 1e-6)
        xr = x.clone().float().requires_grad_()
        want = generator_hinge_loss(xr)
        want.backward()
        xh = x.clone().requires_grad_()
        got = ops.HipOps().hinge(xh)
        got.backward()
        assert abs(float(got) - float(want)) <= 1e-5 * max(1., abs(float(want)))
        assert rel_err(xh.grad.float(), xr.grad) < (4e-3 if dtype == torch.bfloat16 else 1e-6)
    xt = torch.randn(12, 4, requires_grad=True)                            # a non-dense view is copied once (still 2 launches for ~16)
    assert abs(float(ops.HipOps().hinge(xt.t(), 6)) - float(discriminator_hinge_loss(xt.t()[:, 6:], xt.t()[:, :6]))) < 1e-5


# ---- gg_aconv: the one-launch adaptive convolution on a fragment-ordered shared bank (round 5) ------------------------------------
def _aconv_reference(x, W, s, xs, a, d, nz, nw, act):
    """fp32 math on the operands as the kernel rounds them: (x * s * xs) -> bf16, banks -> bf16, fp32 accumulation, the banks mixed
    in fp32 AFTER the reduction (gp.py:378-409 with the sum over kernels pulled out of the convolution), then d, noise, leaky-relu."""
    b, H, Wd, Cc = x.shape
    N, O = W.shape[:2]
    sc = s if xs is None else s * xs
    xm = bf(x.float() * sc[:, None, None, :]).float().permute(0, 3, 1, 2)
    y = 0.
    for n in range(N):
        yn = F.conv2d(xm, bf(W[n]).float(), padding=1)
        y = y + (a[:, n].view(b, 1, 1, 1) if a is not None else 1.) * yn
    if d is not None:
        y = y * d.view(b, O, 1, 1)
    if nz is not None:
        y = y + nz.view(b, 1, H, Wd) * nw.view(1, O, 1, 1)
    if act:
        y = F.leaky_relu(y, 0.2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize('cfg', [(3, 4, 32, 32, 2, 1, 1), (2, 8, 16, 64, 2, 2, 2), (2, 16, 32, 128, 2, 2, 4), (1, 16, 16, 64, 1, 4, 2),
                                 (1, 32, 16, 128, 2, 4, 4), (1, 64, 16, 64, 2, 4, 2), (2, 8, 64, 32, 2, 1, 1), (2, 8, 32, 64, 2, 0, 0),
                                 (2, 4, 16, 64, 2, 1, 2), (1, 8, 16, 32, 1, 2, 1)])
def test_aconv_matches_fp32_math_on_every_tile_shape(cfg):
    """whole-image tiles (4x4: two images per 32 pixels, a batch that leaves a tile half empty), row tiles of 8..64-wide images,
    every (pixels x channels x K-slices) wavefront arrangement, one and two kernels in the bank, with and without the excitation
    scale / demodulation / noise / activation."""
    b, R, Cc, O, NB, tm, nwn = cfg
    torch.manual_seed(sum(cfg))
    x = bf(torch.randn(b, R, R, Cc))
    W = torch.randn(NB, O, Cc, 3, 3) * 0.2
    s, xs = torch.rand(b, Cc) + 0.5, torch.rand(b, Cc) + 0.5
    a = torch.softmax(torch.randn(b, NB), -1) if NB > 1 else None
    d = torch.rand(b, O) + 0.5
    nz, nw = torch.randn(b * R * R), torch.randn(O) * 0.3
    wf = K.frag_pack(W)
    plan = K.aconv_plan(b, R, R, Cc, O, NB, tm, nwn)
    assert plan is not None and (tm == 0 or plan[0] == tm) and (nwn == 0 or plan[1] == nwn) and plan[1] * plan[2] == 8, plan
    for use_xs, use_d, use_nz, act in ((True, True, True, 'lrelu'), (False, False, False, None), (False, True, False, 'lrelu')):
        y = K.aconv(x, wf, s, a, d if use_d else None, O, nz if use_nz else None, nw if use_nz else None, act,
                    xs=xs if use_xs else None, force_tm=tm, force_nwn=nwn)
        ref = _aconv_reference(x, W, s, xs if use_xs else None, a, d if use_d else None, nz if use_nz else None, nw, act)
        assert y.shape == (b, R, R, O) and rel_err(y, ref) < 4e-3, (cfg, use_xs, use_d, use_nz, rel_err(y, ref))


def test_aconv_refuses_what_it_cannot_run():
    assert K.aconv_plan(2, 8, 8, 24, 64, 2) is None            # C not a power of two
    assert K.aconv_plan(2, 8, 8, 32, 40, 2) is None            # O % 32
    assert K.aconv_plan(2, 128, 128, 32, 32, 2) is None        # beyond 64-wide images: the streaming kernels' territory
    assert K.aconv_plan(2, 8, 8, 32, 64, 3) is None            # banks of more than two kernels
    with pytest.raises(RuntimeError, match='gg_aconv'):
        K.aconv(bf(torch.randn(1, 8, 8, 32)), K.frag_pack(torch.randn(2, 64, 32, 3, 3)), torch.ones(1, 32), None, None, 64)


def test_pack_table_keeps_a_bank_in_fragment_order():
    """gg_pack_weights kind 3 == kernels.frag_pack (the tensor-algebra statement of the layout), through the device-resident table."""
    torch.manual_seed(0)
    tab = K.PackTable('cpu', capacity=16)
    for N, O, I in ((2, 64, 32), (1, 32, 16), (2, 32, 512)):
        w = torch.randn(N, O, I, 3, 3)
        dst = tab.register_frag(w.view(N, O, I, 9), N, O, I, 9)
        tab.refresh()
        assert torch.equal(dst, K.frag_pack(w)), (N, O, I)


def test_no_grad_adaptive_conv_takes_the_one_launch_kernel_where_it_can():
    """ops.modconv2d routes the 4x4 .. 64x64 layers to gg_aconv (and keeps the narrow high-resolution layers on gg_sconv); the result
    matches the oracle with and without the skip-layer excitation, from a generator-style batched modulation and stand-alone."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    torch.manual_seed(0)
    for I, O, R, b in ((32, 32, 4, 2), (16, 64, 8, 2), (32, 64, 16, 1)):
        assert ops.HipOps._modconv_path(b, 2, O, I, R, R) == 'aconv'
        conv = AdaptiveConv2DMod(I, O, 3, num_conv_kernels=2)
        x, mod, km = torch.randn(b, I, R, R), torch.randn(b, I) * 0.3, torch.randn(b, 2)
        nz, nw, ex = torch.randn(b, 1, R, R), torch.randn(O, 1, 1) * 0.1, torch.rand(b, I, 1, 1) + 0.5
        for excite in (None, ex):
            with torch.no_grad():
                with ops.use_impl(ops.HipOps()):
                    y1 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu', in_excite=excite)
                with ops.use_impl(OracleOps(bf16_operands=True)):
                    y0 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu', in_excite=excite)
            assert rel_err(y1, y0) < 1e-2, (I, O, R, excite is not None, rel_err(y1, y0))
    assert ops.HipOps._modconv_path(32, 2, 32, 64, 128, 128) == 'sconv'


def test_aconv_requests_exactly_the_next_banks_bytes():
    """the next-bank hint: every XCD's workgroups together touch exactly the byte range that XCD's workgroups of the NEXT launch will
    stream (the emulator traps on a touch outside the bank and counts the bytes), and the result of the launch itself is unchanged."""
    torch.manual_seed(0)
    lib = _C.lib().lib
    lib.gg_emu_touched.restype = ctypes.c_ulonglong
    b, R, Cc, O = 4, 8, 32, 64
    x = bf(torch.randn(b, R, R, Cc))
    W = torch.randn(2, O, Cc, 3, 3) * 0.2
    s, a, d = torch.rand(b, Cc) + 0.5, torch.softmax(torch.randn(b, 2), -1), torch.rand(b, O) + 0.5
    wf = K.frag_pack(W)
    y0 = K.aconv(x, wf, s, a, d, O)
    lib.gg_emu_touched()
    for nb, nR, nC, nO in ((4, 8, 64, 256), (4, 16, 32, 128), (8, 4, 32, 512)):
        nwf = K.frag_pack(torch.randn(2, nO, nC, 3, 3))
        y1 = K.aconv(x, wf, s, a, d, O, next_bank=(nwf, nb, nR))
        assert torch.equal(y0, y1)
        tm, nwn, nwk, lds, grid = K.aconv_plan(nb, nR, nR, nC, nO, 2)
        mt = grid // (nO // (32 * nwn))
        tn_bytes = nwn * 2 * 9 * (nC // 16) * 1024
        want = 0
        for xcd in range(8):
            q, r = grid >> 3, grid & 7
            lo = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
            cnt = q + (1 if xcd < r else 0)
            if cnt:
                want += min(((lo + cnt - 1) // mt + 1) * tn_bytes, nwf.numel() * 2) - (lo // mt) * tn_bytes
        got = lib.gg_emu_touched()
        assert got == want and got >= nwf.numel() * 2, (got, want, nwf.numel() * 2)


# ---- gg_pgemm.h: the persistent short-K contraction (plan tile 15) --------------------------------------------------------------

@pytest.fixture
def pgemm_wgs(monkeypatch):
    def set_wgs(n):
        monkeypatch.setenv('GG_PGEMM_WGS', str(n))
    return set_wgs


@pytest.mark.parametrize('cfg', [
    # (n, H, W, C = K, N, workgroups, bias, act, residual)
    (2, 12, 12, 128, 192, 2, True, 'lrelu', True),      # M = 288: a ragged last row tile, a ragged column tile, runs of 3 tiles
    (1, 16, 16, 64, 128, 1, False, None, False),        # K = 64: one stage per tile, the ring runs three tiles ahead
    (1, 16, 24, 256, 136, 3, True, None, True),         # four stages per tile, N = 136 (a column tile of 8)
    (1, 16, 16, 320, 128, 1, True, 'gelu', False),      # five stages per tile: the ring wraps inside a tile
])
@pytest.mark.parametrize('other_order', [False, True])     # (the host picks round-robin tiles or contiguous runs by shape; GG_PGEMM_ORDER=flip takes the other)
def test_persistent_short_k_contraction_is_bit_identical_to_the_tiled_kernel(cfg, pgemm_wgs, other_order, monkeypatch):
    """gg_pgemm_kernel (plan tile 15) against gg_gemm2_kernel<128,128> (tile 6) on 1x1 convolutions: same k order, same rounding
    points, same epilogue arithmetic -> the same bits; and against fp32 math. Runs of several tiles per workgroup (GG_PGEMM_WGS)
    exercise the cross-tile prefetch, the bias slots and the ring wrap, in both tile orders."""
    n, H, W, Cc, N, wgs, with_bias, act, with_res = cfg
    pgemm_wgs(wgs)
    if other_order:
        monkeypatch.setenv('GG_PGEMM_ORDER', 'flip')
    torch.manual_seed(0)
    x = torch.randn(n, H, W, Cc).bfloat16()
    w = (torch.randn(N, Cc) / Cc ** 0.5).bfloat16()
    bias = torch.randn(N) if with_bias else None
    res = torch.randn(n, H, W, N).bfloat16() if with_res else None
    kw = dict(ksize=1, pad=0, alpha=0.7, bias=bias, bias_scale=0.5, act=act, residual=res, res_scale=0.25)
    assert K.conv2d_nhwc(x, w, force_tile=15, plan_only=True, **kw) == (15, 1)
    y15 = K.conv2d_nhwc(x, w, force_tile=15, **kw)
    y6 = K.conv2d_nhwc(x, w, force_tile=6, **kw)
    assert torch.equal(y15, y6), float((y15.float() - y6.float()).abs().max())
    ref = 0.7 * torch.einsum('nhwc,oc->nhwo', x.float(), w.float())
    if with_bias:
        ref = ref + 0.5 * bias
    if act == 'lrelu':
        ref = F.leaky_relu(ref, 0.2)
    elif act == 'gelu':
        ref = F.gelu(ref)
    if with_res:
        ref = ref.bfloat16().float() + 0.25 * res.float()
    assert rel_err(y15, ref) < 6e-3


@pytest.mark.parametrize('Kd', [128, 192])          # (two stages per tile: the epilogue follows the k-loop; three: it rides in the next tile's)
def test_persistent_short_k_contraction_gelu_aux_modes_and_dense_operands(pgemm_wgs, Kd):
    """the FeedForward epilogues (gelu_mode 1: h to aux, gelu(h) out; gelu_mode 2: out = staged * gelu'(aux)) and a dense row-major
    A with a row pitch larger than K, both bit-identical to tile 6."""
    pgemm_wgs(2)
    torch.manual_seed(1)
    x = torch.randn(1, 16, 20, Kd).bfloat16()
    w = (torch.randn(256, Kd) / Kd ** 0.5).bfloat16()
    bias = torch.randn(256)
    out = {}
    for tile in (15, 6):
        aux = torch.zeros(1, 16, 20, 256).bfloat16()
        y = K.conv2d_nhwc(x, w, ksize=1, pad=0, bias=bias, gelu_aux=aux, gelu_mode=1, force_tile=tile)
        g = torch.randn(1, 16, 20, Kd, generator=torch.Generator().manual_seed(2)).bfloat16()
        wt = (torch.randn(256, Kd, generator=torch.Generator().manual_seed(3)) / Kd ** 0.5).bfloat16()
        if tile == 15:
            assert K.conv2d_nhwc(g, wt, ksize=1, pad=0, gelu_aux=aux, gelu_mode=2, force_tile=15, plan_only=True) == (15, 1)
        dy = K.conv2d_nhwc(g, wt, ksize=1, pad=0, gelu_aux=aux, gelu_mode=2, force_tile=tile)
        out[tile] = (y, aux, dy)
    for a, b in zip(out[15], out[6]):
        assert torch.equal(a, b)
    y, aux, _ = out[15]
    h = torch.einsum('nhwc,oc->nhwo', x.float(), w.float()) + bias
    assert rel_err(aux, h) < 6e-3 and rel_err(y, F.gelu(aux.float())) < 6e-3

    a = torch.randn(320, 264).bfloat16()[:, :Kd + 64]        # (row pitch 264 elements)
    b = (torch.randn(128, Kd + 64) / 192 ** 0.5).bfloat16()
    c15 = K.gemm(a, b, force_tile=15, alpha=1.5)
    c6 = K.gemm(a, b, force_tile=6, alpha=1.5)
    assert torch.equal(c15, c6)
    assert rel_err(c15[0], 1.5 * a.float() @ b.float().t()) < 6e-3


def test_persistent_short_k_contraction_is_the_planned_kernel_for_large_short_k_launches():
    """planner: an eligible row-major launch of >= 8K rows and K <= 1024 that the 8-wave tiles would take runs on tile 15; launches
    it cannot express (an input scale, a 3x3 window, fp32 output) keep their kernels."""
    x = torch.zeros(32, 32, 32, 256).bfloat16()
    w = torch.zeros(512, 256).bfloat16()
    assert K.conv2d_nhwc(x, w, ksize=1, pad=0, plan_only=True) == (15, 1)
    assert K.conv2d_nhwc(x, w, ksize=1, pad=0, in_scale=torch.ones(32, 256), plan_only=True)[0] != 15
    assert K.conv2d_nhwc(x, w, ksize=1, pad=0, out_dtype=torch.float32, plan_only=True)[0] != 15
    w3 = torch.zeros(512, 9 * 256).bfloat16()
    assert K.conv2d_nhwc(x, w3, ksize=3, plan_only=True)[0] != 15
    xs = torch.zeros(2, 32, 32, 256).bfloat16()
    assert K.conv2d_nhwc(xs, w, ksize=1, pad=0, plan_only=True)[0] != 15          # 2048 rows: the tiled kernels
    w5 = torch.zeros(512, 512).bfloat16()
    x5 = torch.zeros(32, 32, 32, 512).bfloat16()
    assert K.conv2d_nhwc(x5, w5, ksize=1, pad=0, plan_only=True)[0] != 15         # alpha-only epilogue at K = 512: the 256 x 256 tile
    assert K.conv2d_nhwc(x5, w5, ksize=1, pad=0, bias=torch.zeros(512), plan_only=True) == (15, 1)


@pytest.mark.parametrize('cfg', [(5, 40, 32, 24, True), (3, 8, 4, 8, False), (2, 300, 70, 130, True), (1, 1100, 20, 33, True), (2, 30, 10, 13, True)])
def test_squeeze_excite_mlp_kernels_match_the_module_stack(cfg):
    """gg_se_mlp_fwd / _bwd on the emulator: ragged widths (rows shorter / longer than a 16-lane DPP row, column counts below and
    above the workgroup), with and without biases, against fp32 autograd through Linear -> SiLU -> Linear -> Sigmoid."""
    from helpers import check_se_mlp
    b, C, H, O, bias = cfg
    check_se_mlp(torch.device('cpu'), b, C, H, O, bias)


def test_squeeze_excite_container_runs_the_fused_node_and_matches_the_modules():
    """SqueezeExciteNet (the reference's six-slot Sequential, same state-dict keys) on ops.HipOps: the fused node gives the modules'
    excitation and gradients (input and parameters; grad-sink route into a pre-filled .grad included), falls back to the modules
    for gradient-penalty graphs, and the oracle op set never takes it."""
    from gigagan_pytorch_amd import modules as M
    from oracle.torch_ops import OracleOps
    torch.manual_seed(0)
    se = M.SqueezeExcite(16, 24)
    assert list(se.state_dict()) == ['1.weight', '1.bias', '3.weight', '3.bias']
    x0 = bf(torch.randn(3, 16, 8, 8)).float()
    g = torch.randn(3, 24, 1, 1)

    def run(impl, second_order=False, sink=False):
        for p_ in se.parameters():
            p_.grad = torch.full_like(p_, 0.5) if sink else None
        x = x0.clone().requires_grad_()
        with ops.use_impl(impl):
            ops.second_order = second_order
            try:
                e = se(x)
                assert e.shape == (3, 24, 1, 1)
                seen, todo = set(), [e.grad_fn]
                while todo:
                    fn = todo.pop()
                    if fn is not None and fn not in seen:
                        seen.add(fn)
                        todo += [n for n, _ in fn.next_functions]
                kind = 'fused' if any('SeMlpFn' in type(fn).__name__ for fn in seen) else 'modules'
                if sink:
                    with ops.sinking():
                        e.float().backward(g, inputs=[x, *se.parameters()])
                else:
                    e.float().backward(g, inputs=[x, *se.parameters()])
            finally:
                ops.second_order = False
        return e.detach().float(), x.grad.float(), [p_.grad.clone() for p_ in se.parameters()], kind

    e_o, gx_o, gp_o, kind_o = run(OracleOps())
    assert kind_o == 'modules'
    e_h, gx_h, gp_h, node = run(ops.HipOps())
    assert node == 'fused'
    assert rel_err(e_h, e_o) < 1e-5 and rel_err(gx_h, gx_o) < 1e-2
    for a, b_ in zip(gp_h, gp_o):
        assert rel_err(a, b_) < 1e-4
    e_s, gx_s, gp_s, _ = run(ops.HipOps(), sink=True)               # accumulated into a pre-filled flat .grad by the queued finishes
    assert torch.equal(e_s, e_h) and torch.equal(gx_s, gx_h)
    for a, b_ in zip(gp_s, gp_h):
        assert rel_err(a - 0.5, b_) < 1e-5
    e_2, gx_2, gp_2, kind_2 = run(ops.HipOps(), second_order=True)       # twice-differentiable graphs keep the modules
    assert kind_2 == 'modules'
    assert rel_err(e_2, e_o) < 1e-2 and rel_err(gx_2, gx_o) < 2e-2


@pytest.mark.parametrize('P,C,n', [(1, 8, 8), (7, 64, 50), (31, 72, 72), (32, 64, 64), (100, 136, 130), (1024, 64, 64)])
def test_bias_gradient_finish_folds_every_partial_row_in_one_workgroup_per_column_block(P, C, n):
    """gg_colsum_finish (and gg_finish_multi's column-sum items) with the default one workgroup per 64-channel block: the batched-load
    loop (eight rows in flight per wavefront) and its remainder, ragged column counts, accumulate mode - against the fp64 column sums."""
    torch.manual_seed(P + C)
    part = torch.randn(P, C)
    want = 0.5 * part[:, :n].double().sum(0)
    got = K.colsum_finish(part, n, 0.5)
    assert rel_err(got.double(), want) < 1e-6
    acc = torch.full((n,), 2.0)
    K.colsum_finish(part, n, 0.5, out=acc, accumulate=True)
    assert rel_err(acc.double(), want + 2.0) < 1e-6
    sink = torch.full((n,), -1.0)
    K.finish_queue.add_colsum(part, n, 0.5, sink)
    K.finish_queue.flush()
    assert rel_err(sink.double(), want - 1.0) < 1e-6


@pytest.mark.parametrize('rows,C', [(1, 8), (7, 16), (37, 32), (1030, 64), (4099, 24), (301, 2056)])
def test_bias_activation_backward_on_ragged_row_counts(rows, C):
    """gg_bias_act_bwd (autograd of bias + leaky_relu, gp.py:109, :1608-1621) walks its rows four per trip with the loads batched
    (round 6): row counts that are not a multiple of anything, a single row, more than 256 column groups. dz bit for bit against
    the tensor expression; the column sums against fp64 sums of that dz."""
    torch.manual_seed(rows + C)
    dy, y = bf(torch.randn(rows, C)), bf(torch.randn(rows, C))
    dz, db = K.bias_act_bwd(dy, y, True)
    want = bf(dy.float() * torch.where(y.float() > 0, 1.0, 0.2))
    assert torch.equal(dz, want)
    ref = want.double().sum(0)
    assert (db.double() - ref).abs().max() <= 1e-5 * (1 + want.double().abs().sum(0).max())
    same, db2 = K.bias_act_bwd(dy, None, True)
    assert same is dy and (db2.double() - dy.double().sum(0)).abs().max() <= 1e-5 * (1 + dy.double().abs().sum(0).max())
