"""Exact-arithmetic parity of the contraction kernels: on SMALL-INTEGER operands every product and every partial sum is an integer
that bf16 (operands, outputs up to 256) and fp32 (accumulators, up to 2^24) represent exactly, so a correct kernel returns the fp32
reference BIT FOR BIT whatever its tiling, summation order or rounding points - and an indexing / accumulation / epilogue bug of any
size fails `torch.equal`. This is the check the bf16 tolerance of the other parity tests cannot give ("a 1 %
kernel bug can hide inside the bf16 floor"); it is NOT an fp32 mode of the kernels - there is none, activations are bf16 in HBM.

Covers every plan tile of gg_gemm_bf16 (4-wave GEMM 1-3, 8-wave GEMM 4-6, halo-staged conv 7 / 8 / 12, direct conv 9, nine-tap weight
gradient 10, streaming weight gradient 13, streaming forward 14, persistent short-K GEMM 15), forward / data gradient / weight
gradient, plus the streaming adaptive convolutions gg_sconv_fwd / gg_spair_fwd and gg_aconv_fwd with unit coefficients.
reference: nn.Conv2d / F.conv2d of gp.py:1608-1621, :1454-1470, :378-409."""
import pytest
import torch
import torch.nn.functional as F

from gigagan_pytorch_amd import kernels as K


def _ints(shape, p_nonzero, gen):
    """values in {-1, 0, 1}, non-zero with probability p_nonzero"""
    v = torch.randint(0, 2, shape, generator=gen) * 2 - 1
    return (v * (torch.rand(shape, generator=gen) < p_nonzero)).float()


def _dev(gpu):
    return torch.device('cuda', 0) if gpu else torch.device('cpu')


def check_conv(cfg, gpu):
    n, H, W, C, N, ks, tile, kind = cfg
    g = torch.Generator().manual_seed(hash(cfg) % (1 << 31))
    dev = _dev(gpu)
    pad = ks // 2
    # sparsity so that |sums| stay below 256 (the last integer bf16 holds exactly): expected non-zero products 9 C p^2
    p = min(0.5, (40.0 / (ks * ks * C)) ** 0.5)
    x = _ints((n, H, W, C), p, g)
    w = _ints((N, C, ks, ks), p, g)
    xb = x.to(torch.bfloat16).to(dev)
    if kind == 'fwd':
        bias = torch.randint(-3, 4, (N,), generator=g).float()
        ref = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=pad)
        assert float(ref.abs().max()) <= 256
        wk = w.permute(0, 2, 3, 1).reshape(N, ks * ks * C).to(torch.bfloat16).to(dev)
        kw = dict(ksize=ks, bias=bias.to(dev), force_tile=tile)
        assert K.conv2d_nhwc(xb, wk, plan_only=True, **kw)[0] == tile, (cfg, K.conv2d_nhwc(xb, wk, plan_only=True, **kw))
        y = K.conv2d_nhwc(xb, wk, **kw)
        assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref), cfg
        # leaky-relu with slope 0.5 keeps integers' halves exact in bf16
        y2 = K.conv2d_nhwc(xb, wk, act='lrelu', act_slope=0.5, **kw)
        assert torch.equal(y2.float().cpu().permute(0, 3, 1, 2), F.leaky_relu(ref, 0.5)), cfg
    elif kind == 'dgrad':
        # the data gradient is a convolution of dy with the flipped, transposed kernel: the same entry point, [ci][kh'][kw'][co] weights
        dy = _ints((n, H, W, N), p, g)
        ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2), w, padding=pad)              # (n, C, H, W)
        assert float(ref.abs().max()) <= 256
        wt = w.flip(2, 3).permute(1, 2, 3, 0).reshape(C, ks * ks * N).to(torch.bfloat16).to(dev)
        kw = dict(ksize=ks, force_tile=tile)
        dyb = dy.to(torch.bfloat16).to(dev)
        assert K.conv2d_nhwc(dyb, wt, plan_only=True, **kw)[0] == tile, cfg
        dx = K.conv2d_nhwc(dyb, wt, **kw)
        assert torch.equal(dx.float().cpu().permute(0, 3, 1, 2), ref), cfg
    else:
        dy = _ints((n, H, W, N), p, g)
        wz = torch.zeros(N, C, ks, ks, requires_grad=True)
        F.conv2d(x.permute(0, 3, 1, 2), wz, padding=pad).backward(dy.permute(0, 3, 1, 2))
        ref = wz.grad.permute(2, 3, 1, 0).reshape(ks * ks * C, N)                      # fp32 integers < 2^24
        K.plan_log = []
        try:
            got = K.conv2d_wgrad_nhwc(xb, dy.to(torch.bfloat16).to(dev), ksize=ks, force_tile=tile)
            assert K.plan_log[-1][0] == tile, (cfg, K.plan_log)
        finally:
            K.plan_log = None
        assert torch.equal(got.float().cpu(), ref), cfg


# (n, H, W, C, N, ksize, plan tile, pass)
CPU_CASES = [
    (1, 8, 8, 16, 24, 3, 1, 'fwd'), (1, 8, 8, 16, 40, 3, 2, 'fwd'), (2, 8, 8, 8, 16, 3, 3, 'fwd'),
    (1, 16, 16, 64, 136, 3, 6, 'fwd'), (1, 16, 16, 64, 256, 3, 7, 'fwd'), (1, 16, 16, 64, 128, 3, 8, 'fwd'), (1, 16, 16, 64, 64, 3, 12, 'fwd'),
    (1, 64, 64, 32, 32, 3, 14, 'fwd'), (1, 16, 16, 64, 128, 1, 15, 'fwd'),
    (1, 16, 16, 64, 128, 3, 8, 'dgrad'), (1, 8, 8, 16, 24, 3, 1, 'wgrad'), (1, 16, 16, 32, 256, 3, 10, 'wgrad'), (1, 64, 64, 32, 32, 3, 13, 'wgrad'),
]
GPU_CASES = CPU_CASES + [
    (4, 32, 32, 128, 256, 3, 4, 'fwd'), (4, 32, 32, 128, 128, 3, 5, 'fwd'), (8, 32, 32, 256, 256, 3, 7, 'fwd'), (8, 16, 16, 512, 512, 3, 7, 'fwd'),
    (8, 64, 64, 128, 128, 3, 8, 'fwd'), (4, 64, 64, 128, 64, 3, 12, 'fwd'), (2, 256, 256, 32, 32, 3, 9, 'fwd'), (2, 256, 256, 32, 32, 3, 14, 'fwd'),
    (2, 128, 128, 64, 64, 3, 14, 'fwd'), (16, 32, 32, 512, 1024, 1, 15, 'fwd'), (16, 32, 32, 256, 1024, 1, 15, 'fwd'),
    (8, 32, 32, 256, 256, 3, 7, 'dgrad'), (8, 16, 16, 512, 512, 3, 7, 'dgrad'), (2, 256, 256, 32, 32, 3, 14, 'dgrad'),
    (8, 16, 16, 512, 512, 3, 10, 'wgrad'), (8, 32, 32, 256, 256, 3, 10, 'wgrad'), (2, 256, 256, 32, 32, 3, 13, 'wgrad'),
    (4, 32, 32, 128, 256, 3, 4, 'wgrad'), (4, 32, 32, 128, 128, 3, 6, 'wgrad'),
]


@pytest.mark.parametrize('cfg', CPU_CASES)
def test_contraction_kernels_are_exact_on_integer_operands(cfg):
    check_conv(cfg, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', GPU_CASES)
def test_contraction_kernels_are_exact_on_integer_operands_gpu(cfg):
    check_conv(cfg, gpu=True)


def _layout2(wps, C):
    b, O = wps.shape[:2]
    wm = torch.zeros(b, 9, C // 16, 32, 16, dtype=torch.bfloat16)
    wm[:, :, :, :O] = wps.reshape(b, O, C // 16, 16, 9).permute(0, 4, 2, 1, 3).to(torch.bfloat16)
    return wm


def check_streaming_adaptive(cfg, gpu):
    """gg_sconv_fwd and gg_spair_fwd on per-image integer banks, with integer noise and a leaky-relu of slope 0.5: the fp32 grouped
    convolution, bit for bit (the fused pair's intermediate map holds integers' halves: exact in bf16 as well)."""
    b, H, W, C0, C1, C2 = cfg
    g = torch.Generator().manual_seed(7)
    dev = _dev(gpu)
    x = _ints((b, H, W, C0), 0.3, g)
    w1 = _ints((b, C1, C0, 3, 3), 4.0 / (9 * C0 * 0.3), g)             # ~4 non-zero products per output: a small-valued intermediate map
    w2 = _ints((b, C2, C1, 3, 3), 3.0 / (9 * C1), g) * 2          # (even weights: the halves conv1's leaky-relu produces stay integers)
    n1, n2 = _ints((b * H * W,), 0.5, g), _ints((b * H * W,), 0.5, g)
    nw1, nw2 = torch.randint(-2, 3, (C1,), generator=g).float(), torch.randint(-2, 3, (C2,), generator=g).float()

    def conv(t, w, nz, nw):      # t (b, C, H, W)
        r = F.conv2d(t.reshape(1, -1, H, W), w.reshape(-1, w.shape[2], 3, 3), padding=1, groups=b).reshape(b, -1, H, W)
        return F.leaky_relu(r + nz.view(b, 1, H, W) * nw.view(1, -1, 1, 1), 0.5)
    mid = conv(x.permute(0, 3, 1, 2), w1, n1, nw1)
    ref = conv(mid, w2, n2, nw2)
    assert float(mid.abs().max()) <= 128 and float(ref.abs().max()) <= 128          # (halves of integers below 128: 8 significant bits)
    xb = x.to(torch.bfloat16).to(dev)
    wm1, wm2 = _layout2(w1, C0).to(dev), _layout2(w2, C1).to(dev)
    d = lambda t: t.to(dev)
    m = K.sconv(xb, wm1, C1, d(n1), d(nw1), 'lrelu', slope=0.5)
    assert torch.equal(m.float().cpu().permute(0, 3, 1, 2), mid)
    y = K.sconv(m, wm2, C2, d(n2), d(nw2), 'lrelu', slope=0.5)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref)
    if K.spair_supported(H, W, C0, C1, C2):
        y2 = K.spair(xb, wm1, wm2, C1, C2, d(n1), d(nw1), d(n2), d(nw2), 'lrelu', 'lrelu', slope=0.5)
        assert torch.equal(y2.float().cpu().permute(0, 3, 1, 2), ref)


def test_streaming_adaptive_convolutions_are_exact_on_integer_operands():
    check_streaming_adaptive((1, 6, 256, 32, 16, 16), gpu=False)
    check_streaming_adaptive((2, 5, 128, 64, 32, 32), gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [(4, 256, 256, 32, 16, 16), (4, 128, 128, 64, 32, 32), (2, 128, 128, 32, 32, 16)])
def test_streaming_adaptive_convolutions_are_exact_on_integer_operands_gpu(cfg):
    check_streaming_adaptive(cfg, gpu=True)


def check_shared_bank(cfg, gpu):
    """gg_aconv_fwd (fragment-ordered shared bank, K-slices summed through LDS, the banks mixed in fp32) with integer scales: s in
    {1, 2}, a = (1, 1), d = 1 -> y = conv(x * s, W_0 + W_1) + noise * nw, integers throughout."""
    b, R, C, O = cfg
    g = torch.Generator().manual_seed(11)
    dev = _dev(gpu)
    p = (20.0 / (18 * C)) ** 0.5
    x = _ints((b, R, R, C), min(0.5, p), g)
    w = _ints((2, O, C, 3, 3), min(0.5, p), g)
    s = torch.randint(1, 3, (b, C), generator=g).float()
    a = torch.ones(b, 2)
    dm = torch.ones(b, O)
    nz, nw = _ints((b * R * R,), 0.5, g), torch.randint(-2, 3, (O,), generator=g).float()
    ref = F.conv2d((x * s[:, None, None, :]).permute(0, 3, 1, 2), w[0] + w[1], padding=1) + nz.view(b, 1, R, R) * nw.view(1, O, 1, 1)
    assert float(ref.abs().max()) <= 256
    d = lambda t: t.to(dev)
    y = K.aconv(d(x.to(torch.bfloat16)), K.frag_pack(d(w)), d(s), d(a), d(dm), O, d(nz), d(nw), None)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref), cfg


@pytest.mark.parametrize('cfg', [(2, 4, 64, 64), (1, 8, 64, 32)])
def test_shared_bank_adaptive_convolution_is_exact_on_integer_operands(cfg):
    check_shared_bank(cfg, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [(32, 4, 512, 512), (32, 8, 512, 512), (32, 16, 512, 256), (32, 16, 256, 256), (32, 32, 256, 128), (32, 32, 128, 128)])
def test_shared_bank_adaptive_convolution_is_exact_on_integer_operands_gpu(cfg):
    check_shared_bank(cfg, gpu=True)
