import torch


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-12)).item()


def bf(t):
    return t.to(torch.bfloat16)


SMALL_G = dict(image_size=32, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2),
               unconditional=True, num_skip_layers_excite=2, self_attn_heads=2, self_attn_dim_head=16)
SMALL_D = dict(image_size=32, dim_capacity=8, dim_max=32, unconditional=True, num_skip_layers_excite=2, attn_heads=2,
               attn_dim_head=16)
C1_G = dict(image_size=64, dim_capacity=8, style_network=dict(dim=64, depth=4), unconditional=True, num_skip_layers_excite=4)
C1_D = dict(image_size=64, dim_capacity=8, unconditional=True, num_skip_layers_excite=4)
# trainer / multi-rank tests: the smallest model that still has every block type (attention at 8x8, one multi-scale
# input, predictor, aux decoder, squeeze-excite) — the host-side emulator pays ~1 s per shuffle-heavy launch
TINY_G = dict(image_size=16, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2),
              unconditional=True, num_skip_layers_excite=1, self_attn_resolutions=(8,), self_attn_heads=2,
              self_attn_dim_head=16)
TINY_D = dict(image_size=16, dim_capacity=8, dim_max=32, unconditional=True, num_skip_layers_excite=1,
              attn_resolutions=(8,), attn_heads=2, attn_dim_head=16, multiscale_input_resolutions=(8,))


def check_pack_table_and_wgrad_finish(shape, device):
    """gg_pack_weights (one launch, table on the device) equals the permute/flip/pad/cast chain, follows entries
    registered later, and gg_wgrad_finish equals the transpose (+ in-place accumulate) of the GEMM's [tap][ci][co]."""
    import torch.nn.functional as F
    from gigagan_pytorch_amd import kernels as K
    O, I, T = shape
    k = int(round(T ** 0.5))
    torch.manual_seed(0)
    tab = K.PackTable(device, capacity=8)
    w1 = torch.randn(O, I, k, k).to(device)
    w2 = torch.randn(I + 5, O + 1, k, k).to(device)
    f1 = tab.register(w1.view(O, I, T), O, I, T, 'fwd')
    b1 = tab.register(w1.view(O, I, T), O, I, T, 'bwd')
    tab.refresh()
    f2 = tab.register(w2.view(I + 5, O + 1, T), I + 5, O + 1, T, 'fwd')     # appended after a first launch
    w1.mul_(2.0)                                                            # the table reads the live parameter
    tab.refresh()
    r8 = lambda n: (n + 7) // 8 * 8

    def ref(w, kind):
        w = w.cpu()
        wp = F.pad(w, (0, 0, 0, 0, 0, r8(w.shape[1]) - w.shape[1], 0, r8(w.shape[0]) - w.shape[0]))
        if kind == 'fwd':
            return wp.permute(0, 2, 3, 1).reshape(wp.shape[0], -1).to(torch.bfloat16)
        return wp.flip(2, 3).permute(1, 2, 3, 0).reshape(wp.shape[1], -1).to(torch.bfloat16)
    assert torch.equal(f1.cpu(), ref(w1, 'fwd')) and torch.equal(b1.cpu(), ref(w1, 'bwd')) and torch.equal(f2.cpu(), ref(w2, 'fwd'))
    c8, o8 = r8(I), r8(O)
    g = torch.randn(T * c8, o8)
    want = (g.view(T, c8, o8)[:, :I, :O].permute(2, 1, 0) * 0.5).contiguous()
    got = K.wgrad_finish(g.to(device), O, I, T, 0.5)
    assert torch.equal(got.cpu(), want)
    acc = torch.ones(O, I, T, device=device)
    K.wgrad_finish(g.to(device), O, I, T, 0.5, out=acc, accumulate=True)
    assert torch.allclose(acc.cpu(), want + 1.0)


def check_flat_optimizer_packs_and_grad_sink(device):
    """parameters owned by FlatAdamW: conv weights come from the persistent pack table (fresh after every step), and
    with ops.grad_sink the weight gradients land in the flat gradient buffer exactly as autograd's accumulation."""
    from gigagan_pytorch_amd import ops
    from gigagan_pytorch_amd.modules import Conv2d, Downsample
    from gigagan_pytorch_amd.optimizer import FlatAdamW
    torch.manual_seed(0)
    net = torch.nn.ModuleList([Conv2d(8, 16, 3, padding=1), Conv2d(16, 16, 1), Downsample(16)]).to(device)
    opt = FlatAdamW(list(net.parameters()), lr=1e-2)
    x = torch.randn(2, 8, 8, 8).to(device)

    def loss():
        h = net[1](net[0](x))
        return (net[2](h) ** 2).mean() + h.mean()
    with ops.use_impl(ops.HipOps()):
        for it in range(2):
            opt.zero_grad()
            loss().backward()
            ref = opt.flat_g.clone()
            opt.zero_grad()
            with ops.sinking():
                loss().backward()
            assert opt.flat_g.abs().sum() > 0 and rel_err(opt.flat_g, ref) < 1e-6
            w = net[0].weight
            packed = w._gg_tpacks['fwd'][0]
            opt.step()
            fresh = ops.packed_weight(w, 'fwd')
            assert fresh.data_ptr() == packed.data_ptr()
            assert torch.equal(fresh, w.detach().permute(0, 2, 3, 1).reshape(16, -1).to(torch.bfloat16))


def check_style_network_on_linear_fn(device):
    """StyleNetwork owned by a FlatAdamW: every EqualLinear + LeakyReLU pair runs as ops.LinearFn (bf16 operand from the pack
    table, lr multiplier and activation in the GEMM epilogue; gradients through one mask / column-sum pass and two GEMMs, into
    the flat gradient buffer with the grad sink). Output and every parameter / input gradient against the oracle formulation of
    gp.py:871-921; operands stay current after an optimizer step."""
    from gigagan_pytorch_amd import ops
    from gigagan_pytorch_amd.modules import StyleNetwork
    from gigagan_pytorch_amd.optimizer import FlatAdamW
    from oracle.torch_ops import OracleOps
    torch.manual_seed(0)
    net = StyleNetwork(64, 3, lr_mul=0.1).to(device)
    for m in net.net[0::2]:
        torch.nn.init.normal_(m.bias, std=0.5)
    opt = FlatAdamW(list(net.parameters()), lr=1e-2)
    z = torch.randn(6, 64).to(device).requires_grad_()
    probe = torch.randn(6, 64).to(device)

    def run(sink):
        opt.zero_grad()
        z.grad = None
        y = net(z)
        with ops.sinking(sink):
            (y.float() * probe).sum().backward()
        return y.detach().float().cpu(), opt.flat_g.clone().cpu(), z.grad.clone().cpu()
    with ops.use_impl(OracleOps(bf16_operands=True)):
        y_ref, g_ref, dz_ref = run(False)
    with ops.use_impl(ops.HipOps()):
        y, g, dz = run(False)
        assert '_gg_tpacks' in net.net[0].weight.__dict__, 'the pack-table path was not taken'
        y2, g2, dz2 = run(True)
        assert rel_err(y, y_ref) < 1e-2 and rel_err(g, g_ref) < 2e-2 and rel_err(dz, dz_ref) < 2e-2, (rel_err(y, y_ref), rel_err(g, g_ref), rel_err(dz, dz_ref))
        assert rel_err(g2, g) < 1e-6 and torch.equal(y2, y) and torch.equal(dz2, dz)
        opt.step()
        w = net.net[0].weight
        assert torch.equal(ops.packed_weight(w, 'fwd'), w.detach().to(torch.bfloat16))
        with torch.no_grad():
            y3 = net(z).float().cpu()
    with ops.use_impl(OracleOps(bf16_operands=True)), torch.no_grad():
        assert rel_err(y3, net(z).float().cpu()) < 1e-2


def check_modcoef(cfg, device):
    """gg_modcoef (s, a, d in one launch; gradients w.r.t. mod, kernel_mod and the kernel bank in two) against the
    tensor-algebra formulation of gp.py:378-400, values and gradients."""
    from gigagan_pytorch_amd import kernels as K, ops
    b, N, O, I, k = cfg
    torch.manual_seed(0)
    w = (torch.randn(N, O, I, k, k) * 0.2).to(device).requires_grad_()
    mod = (torch.randn(b, I) * 0.5).to(device).requires_grad_()
    kmod = torch.randn(b, N).to(device).requires_grad_() if N > 1 else None
    r8 = lambda n: (n + 7) // 8 * 8
    cs, ca, cd = torch.randn(b, r8(I)).to(device), torch.randn(b, N).to(device), torch.randn(b, r8(O)).to(device)

    def reference():
        s = mod + 1.0
        a = kmod.softmax(-1) if N > 1 else torch.ones(b, 1, device=device)
        d = ops.demod_coefficients(w, s, a, 1e-8)
        return s, a, d
    s0, a0, d0 = reference()
    loss0 = (s0 * cs[:, :I]).sum() + (a0 * ca).sum() + (d0 * cd[:, :O]).sum()
    g0 = torch.autograd.grad(loss0, [t for t in (mod, kmod, w) if t is not None])
    s1, a1, d1 = ops.ModCoefFn.apply(mod, kmod, w, 1e-8, r8(I), r8(O))
    assert torch.allclose(s1[:, :I], s0, atol=1e-6) and (s1[:, I:] == 0).all()
    assert torch.allclose(a1, a0, atol=1e-6) and rel_err(d1[:, :O], d0) < 1e-5 and (d1[:, O:] == 0).all()
    loss1 = (s1 * cs).sum() + (a1 * ca).sum() + (d1 * cd).sum()
    g1 = torch.autograd.grad(loss1, [t for t in (mod, kmod, w) if t is not None])
    for x, y in zip(g1, g0):
        assert rel_err(x, y) < 2e-5, (x.shape, rel_err(x, y))
    # forward-only entry (no-grad generator pass): the direct (b, o, i, t) kernel; ModCoefFn goes through the bank's Gram rows
    s2, a2, d2 = K.modcoef_fwd(w.detach(), mod.detach(), None if kmod is None else kmod.detach(), True, 1e-8, r8(I), r8(O))
    assert torch.equal(s2, s1) and rel_err(d2, d1) < 1e-5


# UnetUpsampler (BASELINE config 5 at toy size): two no-downsample stages (8 -> 32), linear attention in the first stage,
# full attention in the last two, high-frequency skip maps from the downsampling stages
UNET_SMALL = dict(dim=8, image_size=32, input_image_size=8, style_network=dict(dim=16, depth=2), dim_mults=(1, 2, 4),
                  full_attn=(False, True, True), self_attn_dim_head=8, self_attn_heads=2, cross_attn_dim_head=8,
                  unconditional=True)


# text-conditional GigaGAN (BASELINE config 4 at toy size): CLIP is an external frozen encoder, the models are fed
# pre-computed token encodings (b, tokens, clip_dim_latent) with zero rows as padding
TEXT_ENC = dict(dim=16, depth=1, heads=2, dim_head=8)
TEXT_CLIP_DIM = 24
TEXT_G = dict(image_size=16, dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2, dim_text_latent=16),
              unconditional=False, self_attn_resolutions=(8,), cross_attn_resolutions=(8,), self_attn_heads=2,
              self_attn_dim_head=16, cross_attn_heads=2, cross_attn_dim_head=16, num_skip_layers_excite=1)
TEXT_D = dict(image_size=16, dim_capacity=8, dim_max=32, unconditional=False, attn_resolutions=(8,), attn_heads=2,
              attn_dim_head=16, num_skip_layers_excite=1, multiscale_input_resolutions=(8,))


def text_encodings(batch=2, tokens=7, seed=0):
    import torch
    g = torch.Generator().manual_seed(seed)
    enc = torch.randn(batch, tokens, TEXT_CLIP_DIM, generator=g)
    for i in range(batch):
        enc[i, tokens - 2 - 2 * i:] = 0          # ragged lengths: zero rows are padding (gp.py:853)
    return enc


def check_fused_modconv_uses_bank_operand_from_pack_table(device):
    """no-grad adaptive conv with optimizer-owned parameters: the [co][tap][n][ci] operand comes from the pack table (one
    strided entry per kernel of the bank), equals the permute/pad/cast chain, follows optimizer steps, and the fused launch
    gives the same result as with the per-call pack."""
    from gigagan_pytorch_amd import ops
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    from gigagan_pytorch_amd.optimizer import FlatAdamW
    torch.manual_seed(0)
    conv = AdaptiveConv2DMod(12, 24, 3, num_conv_kernels=2).to(device)      # ragged I: zero padded to 16
    opt = FlatAdamW(list(conv.parameters()), lr=1e-2)
    x = torch.randn(2, 12, 8, 8).to(device)
    mod, kmod = (torch.randn(2, 12) * 0.3).to(device), torch.randn(2, 2).to(device)

    def want_operand():
        w = conv.weights.detach().cpu()
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 4))
        return w.permute(1, 3, 4, 0, 2).reshape(24, -1).to(torch.bfloat16)
    with ops.use_impl(ops.HipOps()), torch.no_grad():
        y1 = conv(x, mod, kmod)
        packed = conv.weights._gg_tpacks['modk'][0]
        assert torch.equal(packed.cpu(), want_operand())
        conv.weights._gg_pack_table, tab = None, conv.weights._gg_pack_table
        y0 = conv(x, mod, kmod)                      # per-call pack
        conv.weights._gg_pack_table = tab
        assert torch.equal(y0, y1)
    with ops.use_impl(ops.HipOps()):
        opt.zero_grad()
        conv(x, mod, kmod).float().square().mean().backward()
        opt.step()
        with torch.no_grad():
            conv(x, mod, kmod)
        assert conv.weights._gg_tpacks['modk'][0].data_ptr() == packed.data_ptr()
        assert torch.equal(packed.cpu(), want_operand())


def check_gelu_first_and_second_order(device):
    """gg_gelu (forward, backward, backward of the backward) against torch's exact GELU under autograd, on a channels_last
    activation and on a token tensor; values, first-order gradient, and the gradient-penalty style double backward."""
    import torch.nn.functional as F
    from gigagan_pytorch_amd import ops
    torch.manual_seed(0)
    H = ops.HipOps()
    for shape, cl in (((2, 24, 6, 6), True), ((3, 7, 16), False)):
        x0 = (torch.randn(shape) * 2).to(torch.bfloat16).to(device)
        if cl:
            x0 = x0.contiguous(memory_format=torch.channels_last)
        c1, c2 = torch.randn(shape).to(device), torch.randn(shape).to(device)

        def run(fn):
            x = x0.clone().requires_grad_()
            y = fn(x)
            g, = torch.autograd.grad((y.float() * c1).sum(), x, create_graph=True)
            gg, = torch.autograd.grad((g.float() * c2).sum(), x)
            return y.detach(), g.detach(), gg.detach()
        got = run(H.gelu)
        want = run(lambda x: F.gelu(x.float()))
        assert got[0].dtype == torch.bfloat16 and got[0].shape == x0.shape
        for a, b, tol in zip(got, want, (4e-3, 8e-3, 2e-2)):
            assert rel_err(a, b) < tol, (shape, rel_err(a, b))


def check_maxpool_highfreq(shape, device):
    """ops.HipOps.maxpool_highfreq (one HIP pass forward, one backward) against F.max_pool2d and x - blur(x) with the reflect-padded
    normalised [1,2,1]^2 filter (the oracle's restatement of kornia.filter2d) in fp32 under autograd."""
    import torch.nn.functional as F
    from gigagan_pytorch_amd import ops
    b, H, W, C = shape
    torch.manual_seed(0)
    x0 = torch.randn(b, C, H, W).to(torch.bfloat16)
    c1, c2 = torch.randn(b, C, H // 2, W // 2), torch.randn(b, C, H, W)
    f = torch.tensor([1., 2., 1.])
    k = (f[:, None] * f[None, :] / 16.)[None, None].repeat(C, 1, 1, 1)

    def ref(x):
        xf = x.float()
        blur = F.conv2d(F.pad(xf, (1, 1, 1, 1), mode='reflect'), k, groups=C)
        return F.max_pool2d(xf, 2), xf - blur
    xr = x0.clone().requires_grad_()
    p0, h0 = ref(xr)
    g0, = torch.autograd.grad((p0 * c1).sum() + (h0 * c2).sum(), xr)
    xd = x0.to(device).contiguous(memory_format=torch.channels_last).requires_grad_()
    p1, h1 = ops.HipOps().maxpool_highfreq(xd)
    g1, = torch.autograd.grad((p1.float() * c1.to(device)).sum() + (h1.float() * c2.to(device)).sum(), xd)
    assert p1.dtype == torch.bfloat16 and torch.equal(p1.float().cpu(), p0.detach())          # a max of bf16 values is exact
    assert rel_err(h1.cpu(), h0.detach()) < 6e-3 and rel_err(g1.cpu(), g0) < 8e-3, (rel_err(h1.cpu(), h0.detach()), rel_err(g1.cpu(), g0))
    # gradients of one output only
    ga, = torch.autograd.grad((ops.HipOps().maxpool_highfreq(xd)[0].float() * c1.to(device)).sum(), xd)
    gb, = torch.autograd.grad((ref(xr)[0] * c1).sum(), xr)
    assert rel_err(ga.cpu(), gb) < 8e-3


def check_general_attention(cfg, device):
    """ops.FlashAttnGenFn (gg_attn_gen_fwd / _bwd: n queries x m keys, optional null key / value, optional key-padding mask, ragged
    n and m, q / k / v as strided channel slices of one fused projection) against softmax(q k^T * scale + mask) v in fp32 autograd
    on the same bf16 operands: output and the gradients of q, k, v (and the null token)."""
    from gigagan_pytorch_amd import ops
    B, h, n, m, null, mask, fused_qkv = cfg
    torch.manual_seed(0)
    scale = 64 ** -0.5
    if fused_qkv:       # self attention on channel slices of one (B, n, 3*h*64) projection (the unet's to_qkv)
        qkv = torch.randn(B, n, 3 * h * 64).to(torch.bfloat16).to(device).requires_grad_()
        q, k, v = (qkv[..., i * h * 64:(i + 1) * h * 64].view(B, n, h, 64).transpose(1, 2) for i in range(3))
        leaves = [qkv]
    else:
        q0 = torch.randn(B, n, h * 64).to(torch.bfloat16).to(device).requires_grad_()
        kv = torch.randn(B, m, 2 * h * 64).to(torch.bfloat16).to(device).requires_grad_()
        q = q0.view(B, n, h, 64).transpose(1, 2)
        k, v = (kv[..., i * h * 64:(i + 1) * h * 64].view(B, m, h, 64).transpose(1, 2) for i in range(2))
        leaves = [q0, kv]
    nkv = (torch.randn(2, h, 64) * 0.5).to(device).requires_grad_() if null else None
    km = None
    if mask:
        km = torch.ones(B, m, dtype=torch.bool, device=device)
        for i in range(B):
            km[i, max(2, m - 5 - 11 * i):] = False        # (at least two keys stay: the KERNEL gives a fully masked row zeros; ops.attention
                                                          # puts the reference's uniform average there: test_fully_masked_rows_..)
        if mask == 2:       # LEADING masked keys as well (without a null token the online softmax is seeded from key 0's score: the
            for i in range(B):                            # seed must not leak into the result when that key is masked; gp.py:645-647)
                km[i, :1 + i % 3] = False
                km[i, 3] = True
    probe = torch.randn(B, h, n, 64).to(device)

    def reference():
        qf, kf, vf = q.float(), k.float(), v.float()
        bias = torch.zeros(B, 1, 1, kf.shape[2], device=device)
        if km is not None:
            bias = bias.masked_fill(~km[:, None, None, :], -1e30)
        if nkv is not None:
            nk = nkv[0].to(torch.bfloat16).float()[None, :, None, :].expand(B, -1, -1, -1)
            nv = nkv[1].to(torch.bfloat16).float()[None, :, None, :].expand(B, -1, -1, -1)
            kf, vf = torch.cat((nk, kf), 2), torch.cat((nv, vf), 2)
            bias = torch.nn.functional.pad(bias, (1, 0))
        att = (qf @ kf.transpose(-1, -2) * scale + bias).softmax(-1)
        return att @ vf
    want = reference()
    gw = torch.autograd.grad((want * probe).sum(), leaves + ([nkv] if null else []))
    kb = None if km is None else torch.zeros(B, m, device=device).masked_fill(~km, -1e30)
    got = ops.FlashAttnGenFn.apply(ops._rows_view(q), ops._rows_view(k), ops._rows_view(v), None if nkv is None else nkv[0],
                                   None if nkv is None else nkv[1], kb, h, scale)
    got = got.view(B, n, h, 64).transpose(1, 2)
    gg = torch.autograd.grad((got.float() * probe).sum(), leaves + ([nkv] if null else []))
    assert rel_err(got, want) < 8e-3, rel_err(got, want)
    for a, b_ in zip(gg, gw):
        assert rel_err(a, b_) < 2e-2, (a.shape, rel_err(a, b_))
    if fused_qkv:       # the channel slices were read in place
        assert ops._rows_view(q).data_ptr() == qkv.data_ptr()


def check_se_mlp(device, b=5, C=40, H=32, O=24, bias=True, tol=2e-5):
    """gg_se_mlp_fwd / _bwd (SqueezeExcite's excitation MLP, gp.py:297-307) against the reference's module stack in fp32 autograd:
    excitation, gradient w.r.t. the pooled rows and all four parameter gradients; then the same through ops.HipOps / the
    SqueezeExciteNet container (fused node) against the module-by-module path on the same parameters."""
    import torch.nn.functional as F
    from gigagan_pytorch_amd import kernels as K
    torch.manual_seed(b * 1000 + C + H + O)
    m = torch.randn(b, C)
    w1, w2 = torch.randn(H, C) / C ** 0.5, torch.randn(O, H) / H ** 0.5
    b1, b2 = (torch.randn(H) * 0.3, torch.randn(O) * 0.3) if bias else (None, None)
    de = torch.randn(b, O)
    ref_in = [t.clone().requires_grad_() for t in (m, w1, w2)] + ([t.clone().requires_grad_() for t in (b1, b2)] if bias else [])
    rm, rw1, rw2 = ref_in[:3]
    rb1, rb2 = (ref_in[3], ref_in[4]) if bias else (None, None)
    h_ref = F.linear(rm, rw1, rb1)
    e_ref = torch.sigmoid(F.linear(F.silu(h_ref), rw2, rb2))
    grads = torch.autograd.grad(e_ref, ref_in, de)
    dv = lambda t: None if t is None else t.to(device)
    h, hs, e = K.se_mlp_fwd(dv(m), dv(w1), dv(b1), dv(w2), dv(b2))
    assert rel_err(h.cpu(), h_ref.detach()) < tol and rel_err(hs.cpu(), F.silu(h_ref).detach()) < tol
    assert rel_err(e.cpu(), e_ref.detach()) < tol
    dm, gw = K.se_mlp_bwd(dv(de), e, h, hs, dv(m), dv(w1), dv(w2))
    assert rel_err(dm.cpu(), grads[0]) < 10 * tol
    assert rel_err(gw[0].cpu(), grads[1]) < 10 * tol and rel_err(gw[2].cpu(), grads[2]) < 10 * tol
    if bias:
        assert rel_err(gw[1].cpu(), grads[3]) < 10 * tol and rel_err(gw[3].cpu(), grads[4]) < 10 * tol
    dm2, gw2 = K.se_mlp_bwd(dv(de), e, h, hs, dv(m), dv(w1), dv(w2), want_dm=False, want_gw=False)
    assert dm2 is None and gw2 is None
    return e_ref.detach(), grads
