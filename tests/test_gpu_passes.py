"""-m gpu: the HBM-bound passes that so far were only compared on the CPU emulator — fused AdamW / EMA (reference optimizer.py:10-34,
gp.py:2603), softmax / RMSNorm second-order passes (what the gradient penalty's double backward runs, gp.py:120-155), the
resampling stencil and its adjoint (gp.py:246-261, :1683-1687), the squeeze-excite pool (gp.py:297-307) — each on the MI355X
against fp32 torch math of the same op at model-sized shapes. Expectations are stated per test."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from gigagan_pytorch_amd import kernels as K, ops
from oracle.torch_ops import OracleOps
from helpers import rel_err, bf

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda', 0)


def test_fused_adamw_matches_torch_adamw_on_the_gpu():
    """gg_adamw_flat_f32 vs torch.optim.AdamW (fp32, same hyper-parameters as the trainer: lr 2e-4, betas (0.5, 0.9), weight
    decay 1e-2 on ndim >= 2 only) over 5 steps on ~3 M parameters. Same formula, different operation order (the kernel folds
    the bias corrections into two scalars): parameters agree to 2e-7 relative L2 and 4e-6 max-abs (a few fp32 ulps of values up to ~5), moments to
    1e-6; inactive / skipped parameters stay bit-identical."""
    from gigagan_pytorch_amd.optimizer import FlatAdamW
    torch.manual_seed(0)
    d = dev()
    shapes = [(512, 512, 3, 3), (512,), (1024, 257), (3, 64, 7, 7), (77,)]
    ps = [torch.nn.Parameter(torch.randn(*s, device=d)) for s in shapes]
    frozen = torch.nn.Parameter(torch.randn(33, 9, device=d))
    extra = torch.nn.Parameter(torch.randn(300, 5, device=d))        # skipped in one step (its gradient is None there)
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    fo = FlatAdamW([*ps, frozen, extra], lr=2e-4, betas=(0.5, 0.9), inactive=[frozen])
    to = torch.optim.AdamW([{'params': [r for r in rs if r.ndim >= 2]}, {'params': [r for r in rs if r.ndim < 2], 'weight_decay': 0.}],
                           lr=2e-4, betas=(0.5, 0.9), weight_decay=1e-2)
    frozen0 = frozen.detach().clone()
    for step in range(5):
        fo.zero_grad()
        for p, r in zip(ps, rs):
            g = torch.randn_like(p) * (10.0 ** (step - 2))
            p.grad.add_(g)
            r.grad = g.clone()
        frozen.grad.add_(1.0)
        extra.grad.add_(1.0)
        before = extra.detach().clone()
        fo.step(skip=[extra] if step == 2 else [])
        to.step()
        assert torch.equal(extra.detach(), before) == (step == 2)      # no decay, no moment-driven move when skipped
    for p, r in zip(ps, rs):
        assert rel_err(p, r) < 2e-7 and float((p - r).abs().max()) < 4e-6, (p.shape, rel_err(p, r))
        assert rel_err(fo.state[p]['exp_avg'], to.state[r]['exp_avg']) < 1e-6
        assert rel_err(fo.state[p]['exp_avg_sq'], to.state[r]['exp_avg_sq']) < 1e-6
    assert torch.equal(frozen.detach(), frozen0)
    # data-parallel mean folded into the update: grad_scale = 1 / world
    fo.zero_grad()
    for p, r in zip(ps, rs):
        g = torch.randn_like(p)
        p.grad.add_(4 * g)
        r.grad = g.clone()
    fo.step(grad_scale=0.25)
    to.step()
    for p, r in zip(ps, rs):
        assert rel_err(p, r) < 2e-7


def test_fused_ema_matches_lerp_on_the_gpu():
    """gg_ema_flat_f32: ema += (1 - decay) * (p - ema) over a flat buffer; against torch.lerp_ in fp32: <= 1 ulp apart."""
    from gigagan_pytorch_amd import _C
    from gigagan_pytorch_amd._C import ptr
    torch.manual_seed(0)
    d = dev()
    n = 3_000_064
    ema, p = torch.randn(n, device=d), torch.randn(n, device=d)
    ref = ema.clone()
    L = _C.lib()
    for decay in (0.798, 0.995):
        L.check(L.lib.gg_ema_flat_f32(ptr(ema), ptr(p), n, 1. - decay, L.stream(ema)), 'gg_ema_flat_f32')
        ref.lerp_(p, 1. - decay)
    assert float((ema - ref).abs().max()) < 5e-7 and rel_err(ema, ref) < 1e-7


def test_softmax_passes_first_and_second_order_at_attention_shapes():
    """gg_softmax_fwd/_bwd/_bwd2 at the discriminator's 32x32 attention width (1025 keys incl. the null key, padded to 1032)
    against fp32 tensor algebra on the same bf16 inputs: bf16 outputs, 4e-3 relative L2."""
    torch.manual_seed(0)
    d = dev()
    nb, n, m, ld = 16, 1024, 1025, 1032
    x = torch.randn(nb, n, ld, device=d)
    bias = torch.randn(nb, ld, device=d) * 0.3
    alpha = 0.25
    S = K.softmax_fwd(x, bias, alpha, m)
    ref = (alpha * x[..., :m] + bias[:, None, :m]).softmax(-1)
    assert rel_err(S[..., :m], ref) < 4e-3 and float(S[..., m:].float().abs().max()) == 0
    dS, g_dx = bf(torch.randn(nb, n, ld, device=d)), bf(torch.randn(nb, n, ld, device=d))
    g_db = torch.randn(nb, ld, device=d)
    Sf, dSf = S.float()[..., :m], dS.float()[..., :m]
    r = (Sf * dSf).sum(-1, keepdim=True)
    dx, dbias = K.softmax_bwd(S, dS, alpha, m, True)
    u = Sf * (dSf - r)
    assert rel_err(dx[..., :m], alpha * u) < 4e-3 and rel_err(dbias[:, :m], u.sum(1)) < 1e-3
    gt = alpha * g_dx.float()[..., :m] + g_db[:, None, :m]
    gs = (gt * Sf).sum(-1, keepdim=True)
    g_S, g_dS = K.softmax_bwd2(S, dS, g_dx, g_db, alpha, m)
    assert rel_err(g_S[..., :m], gt * (dSf - r) - dSf * gs) < 4e-3 and rel_err(g_dS[..., :m], Sf * (gt - gs)) < 4e-3
    assert float(g_S[..., m:].float().abs().max()) == 0 and float(g_dS[..., m:].float().abs().max()) == 0


@pytest.mark.parametrize('cfg', [(4, 256, 32, 32), (2, 512, 16, 16), (2, 64, 64, 64)])
def test_channel_rmsnorm_first_and_second_order_vs_fp32_autograd(cfg):
    """ChannelRMSNorm (gp.py:224-232) forward, backward and the backward of the backward at model widths against fp32 autograd
    of the oracle on the same bf16 input: 6e-3 forward, 2e-2 first order, 5e-2 second order (bf16 intermediates)."""
    b, c, h, w = cfg
    torch.manual_seed(0)
    H_, O_ = ops.HipOps(), OracleOps()
    x0 = bf(torch.randn(b, c, h, w)).float()
    gamma0 = torch.rand(c, 1, 1) + 0.5
    wgt = torch.randn(b, c, h, w)

    def run(I, d):
        x = x0.to(d).requires_grad_()
        gamma = gamma0.to(d).requires_grad_()
        y = I.channel_rmsnorm(x, gamma).float()
        gx, = torch.autograd.grad((y * wgt.to(d)).sum(), x, create_graph=True)
        gg = torch.autograd.grad(gx.float().pow(2).sum(), [x, gamma])
        g1 = torch.autograd.grad((I.channel_rmsnorm(x, gamma).float() * wgt.to(d)).sum(), [x, gamma])
        return [t.detach().float().cpu() for t in (y, gx, *gg, *g1)]

    yh, gxh, ggxh, gggh, g1xh, g1gh = run(H_, dev())
    yo, gxo, ggxo, gggo, g1xo, g1go = run(O_, torch.device('cpu'))
    assert rel_err(yh, yo) < 6e-3 and rel_err(gxh, gxo) < 2e-2
    assert rel_err(g1xh, g1xo) < 2e-2 and rel_err(g1gh, g1go) < 2e-2
    assert rel_err(ggxh, ggxo) < 5e-2 and rel_err(gggh, gggo) < 5e-2


@pytest.mark.parametrize('cfg', [(4, 64, 32, 32), (2, 8, 128, 128), (3, 24, 20, 12)])
def test_resample_stencil_forward_backward_and_adjoint(cfg):
    """gg_resample_nhwc_bf16: bilinear x2 + binomial blur and bilinear resize against F.interpolate / the oracle's blur on the
    same bf16 input (4e-3: bf16 output rounding), their backward against fp32 autograd (1e-2), and the adjoint identity
    <R x, y> = <x, R^T y> evaluated in fp64 on the kernel's own outputs (2e-3: two bf16-rounded outputs)."""
    b, c, h, w = cfg
    torch.manual_seed(0)
    d = dev()
    H_, O_ = ops.HipOps(), OracleOps()
    x0 = bf(torch.randn(b, c, h, w)).float()
    for name, fn_h, fn_o in (('upsample_blur', lambda t: H_.upsample_blur(t), lambda t: O_.upsample_blur(t)),
                             ('resize', lambda t: H_.resize_bilinear(t, (h // 2, w // 2)),
                              lambda t: F.interpolate(t, (h // 2, w // 2), mode='bilinear'))):
        xh = x0.to(d).requires_grad_()
        xo = x0.clone().requires_grad_()
        yh, yo = fn_h(xh), fn_o(xo)
        assert rel_err(yh.float().cpu(), yo) < 4e-3, name
        g = bf(torch.randn_like(yo)).float()
        gh, = torch.autograd.grad(yh.float(), xh, g.to(d))
        go, = torch.autograd.grad(yo, xo, g)
        assert rel_err(gh.float().cpu(), go) < 1e-2, name
        xp = bf(torch.rand(b, c, h, w)).float().to(d).requires_grad_()      # positive operands: no cancellation in <.,.>
        yp = fn_h(xp)
        gp_ = bf(torch.rand(*yp.shape)).float().to(d)
        gxp, = torch.autograd.grad(yp.float(), xp, gp_)
        lhs = (yp.detach().double() * gp_.double()).sum()
        rhs = (gxp.double() * xp.detach().double()).sum()
        assert abs(lhs - rhs) / abs(lhs) < 2e-3, (name, float(lhs), float(rhs))


def test_squeeze_excite_pool_and_channel_scale_vs_fp32():
    """SqueezeExcite's global mean (gp.py:300) and the excitation multiply (gp.py:1023-1024, :1812-1813) with their gradients
    at the generator's 64x64 / 128-channel stage, against fp32 math on the same bf16 input."""
    torch.manual_seed(0)
    d = dev()
    H_ = ops.HipOps()
    x0 = bf(torch.randn(8, 128, 64, 64)).float()
    s0 = torch.rand(8, 128) + 0.5
    xh, sh = x0.to(d).requires_grad_(), s0.to(d).requires_grad_()
    xo, so = x0.clone().requires_grad_(), s0.clone().requires_grad_()
    mh, mo = H_.global_mean(xh), xo.mean(dim=(2, 3))
    assert rel_err(mh.cpu(), mo) < 1e-5
    yh = H_.channel_scale(xh, sh)
    yo = xo * so[:, :, None, None]
    assert rel_err(yh.float().cpu(), yo) < 4e-3
    g = bf(torch.randn_like(yo)).float()
    gm = torch.randn_like(mo)
    ghx, ghs = torch.autograd.grad([yh.float(), mh], [xh, sh], [g.to(d), gm.to(d)])
    gox, gos = torch.autograd.grad([yo, mo], [xo, so], [g, gm])
    assert rel_err(ghx.float().cpu(), gox) < 1e-2 and rel_err(ghs.cpu(), gos) < 1e-3
    # forked form: the trunk's gradient and the pool's broadcast gradient meet in one in-place pass; a predictor's rows join
    # the trunk gradient in place (RowsForkFn)
    xf = x0.to(d).requires_grad_()
    xa = H_.prepare(xf)
    mf, xt = H_.global_mean(xa, fork=True)
    rows, xt = H_.take_rows(xt, 3, fork=True)
    gr = bf(torch.randn(3, 128, 64, 64)).float()
    gf, = torch.autograd.grad([xt.float(), mf, rows.float()], [xf], [g.to(d), gm.to(d), gr.to(d)])
    want = g + (gm / (64 * 64))[:, :, None, None]
    want[:3] += gr
    assert rel_err(mf.cpu(), mo) < 1e-5 and rel_err(gf.float().cpu(), want) < 1e-2


def test_native_rccl_communicator_on_one_gpu():
    """gg_comm_* (RCCL bound at run time from the librccl PyTorch carries) with a one-rank communicator: init from a unique id,
    in-place all-reduce on the side stream with event fences (identity for world 1), all-gather, ncclCommCount, destroy — the
    same calls a data-parallel step issues (gp.py:1898-1908 replaced). The two-rank case is tests/test_rccl_two_gpus.py."""
    from gigagan_pytorch_amd import distributed as gdist, _C
    d = dev()
    comm = gdist.NativeComm().init(d, rank=0, world=1)
    try:
        assert _C.lib().lib.gg_comm_world() == 1
        g = torch.randn(3_000_000, device=d)
        want = g.clone()
        comm.timing = True
        h = comm.all_reduce_(g, n_slices=4)
        h.wait()
        torch.cuda.synchronize()
        assert torch.equal(g, want)
        assert len(comm.exposed_ms) == 1 and comm.exposed_ms[0][0].elapsed_time(comm.exposed_ms[0][1]) >= 0
        x = torch.randn(5, 7, device=d).to(torch.bfloat16)
        out = comm.all_gather(x)
        torch.cuda.synchronize()
        assert torch.equal(out, x)
        with pytest.raises(RuntimeError, match='already live'):
            gdist.NativeComm().init(d, rank=0, world=1)
    finally:
        comm.destroy()
    assert _C.lib().lib.gg_comm_world() == 0


def test_input_pipeline_feeds_graph_replayed_steps_from_pinned_batches(tmp_path):
    """f4 on the GPU (reference data.py:48-85 + `accelerator.prepare(dl)`, gp.py:2155-2161): a real torch DataLoader of fp32 host
    batches -> GigaGAN.set_dataloader -> DevicePrefetcher (pinned staging, H2D on a copy stream one batch ahead, event fence,
    record_stream) -> `GigaGAN(...)(steps=...)` with hipGraph replay on (each batch is copied into the captured step's static
    buffer). Checks that the batches that reach the step ARE the loader's (a marker value per batch), that the steps train
    (finite losses, weights move), that the prefetcher ran on its own stream, and that graphs stayed on."""
    from torch.utils.data import DataLoader, Dataset
    from gigagan_pytorch_amd import GigaGAN
    from gigagan_pytorch_amd.data import DevicePrefetcher
    from helpers import C1_G, C1_D
    d = dev()

    class Marked(Dataset):
        def __len__(self):
            return 64

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(i)
            x = torch.rand(3, 64, 64, generator=g)
            x[0, 0, 0] = i / 64.                       # marker: which sample this is
            return x

    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=d,
                  create_ema_generator_at_init=False, log_steps_every=10 ** 9, save_and_sample_every=10 ** 9,
                  early_save_and_sample_every=10 ** 9, model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    gan.set_dataloader(DataLoader(Marked(), batch_size=4, shuffle=False, drop_last=True))
    assert isinstance(gan.train_dl, DevicePrefetcher)
    seen = []
    stage = gan._stage

    def spy(name, src):
        if name == 'd_real':
            assert src.is_cuda and src.dtype == torch.float32 and tuple(src.shape) == (4, 3, 64, 64)
            seen.append(src[:, 0, 0, 0].clone())
        return stage(name, src)
    gan._stage = spy
    gan.save_sample = lambda *a, **k: None             # (the reference samples + checkpoints at step 1: not under test here)
    w0 = gan.D_opt.flat_p.clone()
    gan(steps=4)
    torch.cuda.synchronize()
    assert gan.use_hip_graphs and len(gan._graphs) > 0
    marks = torch.stack(seen).cpu() * 64
    want = torch.arange(16.).view(4, 4)                # D-step real batches: loader batches 0..3 in order (G steps draw none)
    assert torch.allclose(marks, want, atol=1e-3), marks
    assert not torch.equal(w0, gan.D_opt.flat_p) and torch.isfinite(gan.D_opt.flat_p).all() and torch.isfinite(gan.G_opt.flat_p).all()


def test_in_backward_gradient_exchange_is_captured_with_the_step(tmp_path):
    """data-parallel step plumbing on the one GPU this box has: with a (one-rank) gg_comm communicator up, the trainer's
    GradReducer issues the sliced all-reduce from inside the backward pass - forked onto the communicator's side stream behind
    an event, joined before the optimizer - and the whole thing is captured into the step's hipGraph (distributed.GradReducer,
    gigagan.py `_run_graphed`). Checks: the captures succeed, slices did go out during the backward, and four steps (plain and
    gradient-penalty, replayed) leave the parameters of the same run with the exchange issued after the backward (to run-to-run noise and
    what AdamW makes of it: see the bound at the end)."""
    from gigagan_pytorch_amd import GigaGAN, distributed as gdist
    from gigagan_pytorch_amd.data import SyntheticImages
    from gigagan_pytorch_amd.gigagan import cycle
    from helpers import C1_G, C1_D
    d = dev()
    comm = gdist.enable_native_comm(d)
    assert comm is not None and gdist.comm_backend() == 'gg_comm/rccl'
    try:
        digests = []
        for overlap in (True, False):
            torch.manual_seed(0)
            gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=d,
                          create_ema_generator_at_init=False, model_folder=str(tmp_path / f'm{overlap}'),
                          results_folder=str(tmp_path / f'r{overlap}'))
            assert gan.D_red is not None and gan.G_red is not None
            gan.overlap_grad_reduce = overlap
            torch.manual_seed(10)
            it = cycle(SyntheticImages(2, 64, device=d, seed=3))
            for _ in range(4):
                gan.train_step(it, 2)
            torch.cuda.synchronize()
            assert gan.use_hip_graphs, 'a capture with the gradient exchange inside was refused'
            flat = torch.cat([gan.D_opt.flat_p, gan.G_opt.flat_p])
            assert torch.isfinite(flat).all()
            if overlap:     # the capture pass of each step kind ran with learned slice counts: slices left during the backward
                assert gan.D_red.in_backward_launches >= gan.D_red.n - 1 and gan.G_red.in_backward_launches >= 1, \
                    (gan.D_red.in_backward_launches, gan.G_red.in_backward_launches, gan.D_red.n, gan.G_red.n)
            digests.append(flat.clone())
            del gan, it, flat
            import gc
            gc.collect()        # the captured graphs (RCCL nodes inside) go before the communicator does
        # identical up to run-to-run noise: two runs of the SAME configuration differ in one entry by 9e-10 (a bias gradient fed by
        # fp32 atomics, tests/gpu_overlap_probe.py -> profiles/r03_overlap_probe.log); a slice exchanged before its last gradient
        # write, or skipped, moves parameters by ~lr = 2e-4 per step
        diff = (digests[0] - digests[1]).abs()
        # (round 5: the old bound - max <= 1e-6, <= 64 entries - failed 2 runs in 16, on the round's first tree as on its last, always with
        # the same figures: max 1.117e-4 on one entry, every other entry a hair off. Not the exchange: gg_colsum_finish folds a bias
        # gradient's partial rows with one fp32 atomic add per workgroup (<= 32 of them), so its last bit depends on their arrival order;
        # AdamW divides by sqrt(v) + 1e-8, which turns a +-1e-8 gradient of a parameter whose true gradient is zero into a step of up to
        # lr, and from there the two trajectories differ everywhere by second-order amounts. A slice that missed a gradient write, or was
        # skipped, moves a sixth of the parameters by ~lr per step: that is what the bounds below still reject. Since then the finish
        # folds in ONE workgroup per column block by default - fixed order, no atomics race - and 12 runs in 12 printed (0.0, 0.0, 0.0, 0)
        # here; the bounds stay loose for GG_COLSUM_GROUPS=32.)
        stats = (float(diff.max()), float(diff.mean()), float((diff > 1e-5).float().mean()), int((diff > 0).sum()))
        print('overlap-vs-post parameter differences (max, mean, fraction > 1e-5, entries > 0):', stats)
        if os.environ.get('GG_COLSUM_GROUPS', '1') in ('', '1'):
            # the default (one workgroup per column block, fixed order) step is bit-reproducible: the exchange issued inside the backward
            # must leave EXACTLY the parameters of the exchange issued behind it (round 6: the loose bound below could hide a late slice
            # that moves few parameters a little)
            assert torch.equal(digests[0], digests[1]), stats
        else:
            assert stats[0] <= 1e-3 and stats[1] <= 5e-6 and stats[2] <= 2e-2, stats
    finally:
        gdist.shutdown()


def test_captured_collectives_never_outlive_the_communicator(tmp_path):
    """a trainer whose captured steps hold RCCL nodes may be dropped - or merely become unreachable
    (parameters <-> reducer hooks form cycles) - at any time, also AFTER the communicator is gone; RCCL aborts the process from a
    runtime thread when a graph with its nodes is destroyed behind a destroyed communicator. The communicator therefore owns the
    order (NativeComm.destroy: collect, drop every live trainer's captured graphs, drain, destroy): here the trainer is still alive and
    still holds its graphs when the communicator is shut down, is then used again (it re-captures, without collectives), and is
    finally dropped and collected - none of which may take the process down."""
    import gc
    from gigagan_pytorch_amd import GigaGAN, distributed as gdist
    from gigagan_pytorch_amd.data import SyntheticImages
    from gigagan_pytorch_amd.gigagan import cycle
    from helpers import C1_G, C1_D
    d = dev()
    comm = gdist.enable_native_comm(d)
    assert comm is not None
    torch.manual_seed(0)
    gan = GigaGAN(generator=dict(C1_G), discriminator=dict(C1_D), apply_gradient_penalty_every=2, device=d,
                  create_ema_generator_at_init=False, model_folder=str(tmp_path / 'm'), results_folder=str(tmp_path / 'r'))
    it = cycle(SyntheticImages(2, 64, device=d, seed=3))
    for _ in range(2):
        gan.train_step(it, 2)
    torch.cuda.synchronize()
    assert gan.use_hip_graphs and len(gan._graphs) >= 2, 'the steps with the exchange inside were not captured'
    gdist.shutdown()                                     # the trainer is alive and holds graphs with RCCL nodes
    assert gdist.native_comm() is None and not any(isinstance(v, tuple) and isinstance(v[0], torch.cuda.CUDAGraph)
                                                   for v in gan._graphs.values()), 'captured graphs survived the communicator'
    for _ in range(2):                                   # the same trainer keeps working: it captures again, on what exists now
        gan.train_step(it, 2)
    torch.cuda.synchronize()
    assert torch.isfinite(torch.cat([gan.D_opt.flat_p, gan.G_opt.flat_p])).all()
    del gan, it
    gc.collect()
    torch.cuda.synchronize()


@pytest.mark.parametrize('cfg', [(64, 32, 32, 128), (32, 16, 16, 256)])
def test_fused_block_pair_at_config2_shapes_batch_32(cfg):
    """the generator's 128x128 / 256x256 blocks at the bench batch (gp.py:1219-1229, no-grad pass): (a) gg_spair_fwd is bit-identical
    to gg_sconv_fwd(gg_sconv_fwd(x)) on the same per-sample banks, noise maps and skip-layer excitation (torch.equal: same
    accumulation order, same epilogue expressions, the intermediate map rounded to bf16 exactly as the unfused path stores it);
    (b) ops.modconv_pair, announced as the generator announces it, against the oracle's two layers with bf16-rounded operands."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    from gigagan_pytorch_amd import kernels as K
    C0, C1, C2, R = cfg
    b, d = 32, dev()
    torch.manual_seed(0)
    c1, c2 = AdaptiveConv2DMod(C0, C1, 3, num_conv_kernels=2), AdaptiveConv2DMod(C1, C2, 3, num_conv_kernels=2)
    x = torch.randn(b, C0, R, R)
    m1, k1, m2, k2 = torch.randn(b, C0) * 0.3, torch.randn(b, 2), torch.randn(b, C1) * 0.3, torch.randn(b, 2)
    n1, n2 = torch.randn(b, 1, R, R), torch.randn(b, 1, R, R)
    nw1, nw2 = torch.randn(C1, 1, 1) * 0.1, torch.randn(C2, 1, 1) * 0.1
    exc = torch.rand(b, C0, 1, 1) + 0.5
    with torch.no_grad():
        with ops.use_impl(OracleOps(bf16_operands=True)):
            y0 = c2(c1(x * exc, m1, k1, noise=n1, noise_weight=nw1, act='lrelu'), m2, k2, noise=n2, noise_weight=nw2, act='lrelu')
        c1, c2 = c1.to(d), c2.to(d)
        g = lambda t: t.to(d)
        impl = ops.HipOps()
        first = dict(weights=c1.weights, mod=g(m1), kernel_mod=g(k1), demod=True, eps=1e-8, noise=g(n1), noise_weight=g(nw1), act='lrelu',
                     in_excite=g(exc))
        second = dict(weights=c2.weights, mod=g(m2), kernel_mod=g(k2), demod=True, eps=1e-8, noise=g(n2), noise_weight=g(nw2), act='lrelu',
                      in_excite=None)
        specs = [(c1.weights, first['mod'], first['kernel_mod'], R, R, True, True, 1e-8),
                 (c2.weights, second['mod'], second['kernel_mod'], R, R, False, True, 1e-8)]
        xd = g(x)
        with ops.use_impl(impl):
            assert impl.modconv_prepare(specs) == 2
            mid = c1(xd, first['mod'], first['kernel_mod'], noise=first['noise'], noise_weight=first['noise_weight'], act='lrelu',
                     in_excite=first['in_excite'])
            two = c2(mid, second['mod'], second['kernel_mod'], noise=second['noise'], noise_weight=second['noise_weight'], act='lrelu')
            assert impl.modconv_prepare(specs) == 2
            K.plan_log = []
            try:
                one = impl.modconv_pair(xd, first, second)
            finally:
                plans, K.plan_log = K.plan_log, None
            impl.modconv_release()
    assert one is not None and not plans
    if C0 == 32:        # (the 256x256 block: 16x16x32 MFMAs, conv2's taps paired - the same products, another summation order)
        assert rel_err(one, two) < 4e-3
    else:
        assert torch.equal(one, two)
    assert rel_err(one.float().cpu(), y0) < 1e-2


@pytest.mark.parametrize('cfg', [(512, 512, 4, 32, 'aconv'), (512, 512, 8, 32, 'aconv'), (512, 256, 16, 32, 'aconv'),
                                 (256, 256, 16, 32, 'aconv'), (256, 128, 32, 32, 'pimg'), (128, 128, 32, 32, 'pimg'),
                                 (256, 128, 32, 16, 'aconv'), (128, 64, 64, 32, 'pimg'), (64, 64, 64, 32, 'pimg'),
                                 (64, 32, 128, 32, 'sconv'), (32, 32, 128, 32, 'sconv'), (32, 16, 256, 32, 'sconv'), (16, 16, 256, 32, 'sconv')])
def test_no_grad_adaptive_conv_at_config2_layer_shapes(cfg):
    """the generator's demodulated 3x3 adaptive convs (gp.py:344-409 + noise + leaky-relu) at EVERY BASELINE config-2 layer shape,
    no-grad path, against the oracle with bf16-rounded operands: 1e-2 relative L2 (bf16 output rounding; the oracle rounds the
    per-sample weights to bf16 AFTER modulation / demodulation, as the reference's autocast conv does, the kernels round the bank and
    the modulated activation). The last entry of a case is the formulation it must run in: 'aconv' = gg_aconv_fwd (4x4 .. 16x16, and
    32x32 where the per-image grid would not fill the chip - the batch-16 case: one launch on the fragment-ordered shared bank,
    round 5), 'pimg' = per-sample weights through gg_conv3's 64-column tile, two workgroups per CU (64x64: plan tile 8; 32x32 at batch
    32: tile 12, round 6), 'sconv' = gg_sconv on per-sample weights; aconv / sconv do not go through gg_gemm_bf16 (no contraction
    plan is logged)."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    from gigagan_pytorch_amd import kernels as K
    I, O, R, b, want_path = cfg
    want_tile = (8 if O <= 64 else 12) if want_path == 'pimg' else 0          # (per-sample weights on gg_conv3's 64-column tile)
    assert ops.HipOps._modconv_path(b, 2, O, I, R, R) == want_path
    torch.manual_seed(0)
    conv = AdaptiveConv2DMod(I, O, 3, num_conv_kernels=2)
    x, mod, km = torch.randn(b, I, R, R), torch.randn(b, I) * 0.3, torch.randn(b, 2)
    nz, nw = torch.randn(b, 1, R, R), torch.randn(O, 1, 1) * 0.1
    with torch.no_grad():
        with ops.use_impl(OracleOps(bf16_operands=True)):
            y0 = conv(x, mod, km, noise=nz, noise_weight=nw, act='lrelu')
        d = dev()
        conv = conv.to(d)
        K.plan_log = []
        try:
            with ops.use_impl(ops.HipOps()):
                y1 = conv(x.to(d), mod.to(d), km.to(d), noise=nz.to(d), noise_weight=nw.to(d), act='lrelu')
        finally:
            plans, K.plan_log = K.plan_log, None
    assert rel_err(y1.float().cpu(), y0) < 1e-2
    assert ([t for t, _ in plans] == [want_tile]) if want_tile else not plans, plans


@pytest.mark.parametrize('cfg', [(128, 64, 64, 32), (64, 64, 64, 8), (128, 128, 32, 32)])
def test_excited_per_image_adaptive_conv_takes_the_scale_on_its_operand_staging(cfg):
    """a 64x64 adaptive conv behind a skip-layer excitation (gp.py:1023-1024: x * excitation, then the block's first conv), announced
    as the generator announces it: per-sample weights WITHOUT the excitation from the batched modulation launch, the excitation as the
    per-(image, channel) scale of gg_conv3's halo staging (64-column tile, two workgroups per CU; round 6) - or, for more than 64 output
    channels, folded into the weights by one gg_modulate launch. Against the oracle with bf16-rounded operands, 1e-2 relative L2; the
    launches are counted."""
    from gigagan_pytorch_amd.modules import AdaptiveConv2DMod
    from gigagan_pytorch_amd import kernels as K
    I, O, R, b = cfg
    torch.manual_seed(1)
    conv = AdaptiveConv2DMod(I, O, 3, num_conv_kernels=2)
    x, mod, km = torch.randn(b, I, R, R), torch.randn(b, I) * 0.3, torch.randn(b, 2)
    exc = torch.rand(b, I, 1, 1) + 0.5
    nz, nw = torch.randn(b, 1, R, R), torch.randn(O, 1, 1) * 0.1
    with torch.no_grad():
        with ops.use_impl(OracleOps(bf16_operands=True)):
            y0 = conv(x * exc, mod, km, noise=nz, noise_weight=nw, act='lrelu')
        d = dev()
        conv = conv.to(d)
        impl = ops.HipOps()
        calls = []
        real_mod = K.modulate
        K.modulate = lambda *a, **kw: (calls.append('modulate'), real_mod(*a, **kw))[1]
        K.plan_log = []
        try:
            with ops.use_impl(impl):
                md, kd = mod.to(d), km.to(d)
                assert impl.modconv_prepare([(conv.weights, md, kd, R, R, True, conv.demod, conv.eps)]) == 1
                y1 = conv(x.to(d), md, kd, noise=nz.to(d), noise_weight=nw.to(d), act='lrelu', in_excite=exc.to(d))
                impl.modconv_release()
        finally:
            K.modulate = real_mod
            plans, K.plan_log = K.plan_log, None
    assert rel_err(y1.float().cpu(), y0) < 1e-2
    assert ops.HipOps._modconv_path(b, 2, O, I, R, R) == 'pimg'
    assert len(plans) == 1 and plans[0][0] in (8, 12), plans
    assert calls == ([] if O <= 64 else ['modulate']), calls


def test_training_steps_never_read_uninitialised_memory():
    """torch.empty() poisoned with NaN (deterministic-mode fill): four replayed config-2 steps (plain D, gradient-penalty D, G step
    kinds; every workspace, partial-sum and padded buffer the kernels are handed) leave losses and both flat parameter buffers
    finite — a kernel reading an element nobody wrote would not."""
    import bench
    from gigagan_pytorch_amd.data import SyntheticImages
    from gigagan_pytorch_amd.gigagan import cycle
    prev = (torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled(),
            torch.utils.deterministic.fill_uninitialized_memory)
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    try:
        torch.manual_seed(0)
        gan = bench.build_gan(256, dev(), use_hip_graphs=True)
        it = cycle(SyntheticImages(32, 256, device=dev()))
        for step in range(4):
            d, g = gan.train_step(it, 32)
            vals = [float(d.divergence), float(g.divergence), float(d.gradient_penalty)]
            assert all(math.isfinite(v) for v in vals), (step, vals)
        assert bool(torch.isfinite(gan.G_opt.flat_p).all()) and bool(torch.isfinite(gan.D_opt.flat_p).all())
        assert float(d.gradient_penalty) > 0          # step 4 is the gradient-penalty step
        del gan
    finally:
        torch.use_deterministic_algorithms(prev[0], warn_only=prev[1])
        torch.utils.deterministic.fill_uninitialized_memory = prev[2]
        torch.cuda.empty_cache()


def test_hinge_ticket_on_64_workgroups_is_deterministic_over_a_thousand_launches():
    """gg_hinge's forward adds its per-workgroup partial sums in slot order by the LAST workgroup to arrive (a ticket taken with
    release / acquire fences, put back to zero by its taker). The fiber emulator cannot expose a memory-ordering bug, so here: 64
    workgroups x 1000 eager launches and 200 replays of a hipGraph holding 5 launches - every result bit-identical to the first and
    equal to the fp32 formula (gp.py:157-163); both modes; a second stream gets its own scratch."""
    from gigagan_pytorch_amd import kernels as K
    d = dev()
    torch.manual_seed(0)
    x = torch.randn(4, 64, 8192, device=d).to(torch.bfloat16)          # 2 M elements: 64 workgroups (GG_HINGE_MAXB)
    xf = x.float()
    want = {0: xf.mean(), 1: (torch.relu(1 + xf[:, 32:]) + torch.relu(1 - xf[:, :32])).mean()}
    for mode, split in ((0, 0), (1, 32)):
        first = K.hinge(x, 64, split, mode)
        vals = torch.stack([K.hinge(x, 64, split, mode) for _ in range(1000)])
        torch.cuda.synchronize()
        assert bool((vals == first).all()), (mode, vals.unique())
        assert abs(float(first) - float(want[mode])) < 2e-5 * max(1.0, abs(float(want[mode]))), (mode, float(first), float(want[mode]))
        # captured: five launches per graph, replayed back to back (the graphs share the device's capture scratch)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                outs = [K.hinge(x, 64, split, mode) for _ in range(5)]
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(200):
            g.replay()
            for o in outs:
                assert bool(o == first), (mode, float(o), float(first))
    # an eager launch on another stream does not share the first stream's ticket
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        other = K.hinge(x, 64, 0, 0)
    s2.synchronize()
    assert bool(other == K.hinge(x, 64, 0, 0))
    assert len([k for k in K._hinge_scratch if k[0] == x.device and k[1] != 'capture']) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [(32, 512, 128, 512, True), (64, 256, 64, 256, True), (32, 64, 32, 128, True), (7, 300, 70, 130, False)])
def test_squeeze_excite_mlp_kernels_vs_fp32_autograd(cfg):
    """gg_se_mlp_fwd / _bwd at the widths of config 2's skip-layer excitations (gp.py:297-307, fp32 throughout) and on ragged ones:
    excitation, pooled-row gradient and the four parameter gradients against fp32 autograd through the reference's module stack."""
    from helpers import check_se_mlp
    b, C, H, O, bias = cfg
    check_se_mlp(dev(), b, C, H, O, bias)
