"""bench.py — images/sec of one full GigaGAN G+D training step (BASELINE.json metric), unconditional 256x256,
per-GPU batch 32, bf16 operands / fp32 accumulation, synthetic data resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one iteration of `GigaGAN.forward`'s loop body: D-step (G forward under no_grad, D(fake), D(real),
hinge + multi-scale + aux-recon losses, gradient penalty with double backward on every 4th step, fused AdamW)
then G-step (G forward, D forward, backward to G, fused AdamW, EMA). K is rounded up to a multiple of 4 so
the timed region always holds whole gradient-penalty cycles. Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

GF_PER_IMG = 1192.1       # algorithmic-minimum GFLOP per image per step, 4-step-cycle mean
MFMA_PEAK_TF = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0     # HBM3E peak, MI355X_MICROARCH.md (achievable: ~6.3 TB/s read, ~7.7 TB/s fill)
C2_G = dict(dim_capacity=8, dim_max=512, style_network=dict(dim=64, depth=4), num_skip_layers_excite=4, unconditional=True)
C2_D = dict(dim_capacity=16, dim_max=512, num_skip_layers_excite=4, unconditional=True)


# secondary workloads (BASELINE configs 5 and 4; reference README.md:104-125 and :68-95): same trainer, eager launches
C5_G = dict(style_network=dict(dim=64, depth=4), dim=32, input_image_size=64, unconditional=True)
C5_D = dict(dim_capacity=16, dim_max=512, num_skip_layers_excite=4, multiscale_input_resolutions=(128,), unconditional=True)
C4_TEXT = dict(dim=64, depth=4, clip_dim_latent=512)      # CLIP itself is external: pre-computed (b, 77, 512) encodings
C4_G = dict(dim_capacity=8, dim_max=512, style_network=dict(dim=64, depth=4, dim_text_latent=64), num_skip_layers_excite=4,
            unconditional=False)
C4_D = dict(dim_capacity=16, dim_max=512, num_skip_layers_excite=4, unconditional=False)


def build_gan(image_size, device, g_over=None, d_over=None, use_hip_graphs=None, workload='uncond'):
    from gigagan_pytorch_amd import GigaGAN
    torch.manual_seed(0)
    kw = dict(amp=True, mixed_precision_type='bf16', apply_gradient_penalty_every=4, calc_multiscale_loss_every=1,
              device=device, model_folder='/tmp/gg-bench-models', results_folder='/tmp/gg-bench-results',
              use_hip_graphs=use_hip_graphs)
    if workload == 'upsampler':
        return GigaGAN(train_upsampler=True, generator=dict(C5_G, image_size=image_size),
                       discriminator=dict(C5_D, image_size=image_size), **kw)
    if workload == 'text':
        return GigaGAN(generator=dict(C4_G, image_size=image_size, text_encoder=dict(C4_TEXT)),
                       discriminator=dict(C4_D, image_size=image_size, text_encoder=dict(C4_TEXT)),
                       generator_contrastive_loss_weight=0., **kw)
    g = dict(C2_G, image_size=image_size, **(g_over or {}))
    d = dict(C2_D, image_size=image_size, **(d_over or {}))
    return GigaGAN(generator=g, discriminator=d, **kw)


class SyntheticTextImages:
    """endless (images, token encodings) batches resident on the device: uniform images, normal (b, 77, 512) encodings with
    ragged zero padding (what a frozen CLIP text tower hands the TextEncoder, gp.py:843-853)."""

    def __init__(self, batch, image_size, device, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.batch_size = batch
        self.images = torch.rand(batch, 3, image_size, image_size, generator=g).to(device)
        enc = torch.randn(batch, 77, 512, generator=g)
        for i in range(batch):
            enc[i, 16 + (5 * i) % 60:] = 0
        self.enc = enc.to(device)

    def __iter__(self):
        while True:
            yield self.images, self.enc


def cpu_reference_leg(threads, timeout_s=600.0):
    """the UNMODIFIED reference trainer on this host's cores: `oracle/time_reference.py` in its own process with
    the GPUs hidden, importing the reference as byte code from `oracle/_ref` (compiled by `__graft_entry__.build()` from the files
    under /root/reference where they lie; git-ignored, it travels with the built .so files). None when oracle/_ref is absent."""
    import subprocess
    from oracle.build_ref import available
    if not available() and not Path('/root/reference/gigagan_pytorch').is_dir():
        return None
    try:
        out = subprocess.run([sys.executable, str(ROOT / 'oracle' / 'time_reference.py'), '--threads', str(threads)],
                             capture_output=True, text=True, timeout=timeout_s)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith('{'):
                return json.loads(ln)
        return dict(error=(out.stderr or out.stdout)[-400:])
    except Exception as e:      # noqa: BLE001 - a baseline leg must not take the bench line down
        return dict(error=f'{type(e).__name__}: {e}')


def cpu_baseline(budget_s=150.0):
    """SURVEY.md §8(d) / BASELINE.md §3 protocol on a bounded sample: config-2 dims at 256x256, fp32, batch 4, same synthetic
    uniform images. kind="reference": the unmodified reference trainer itself (`GigaGAN(...)(steps=...)`: one warm-up step, then
    one whole 4-step cycle = 3 plain + 1 gradient-penalty G+D step incl. optimizer updates and EMA) timed on THIS host by
    `cpu_reference_leg`. Next to it, as `port`, the same cycle through `train_step` of OUR trainer on the fp32 CPU oracle (what
    rounds 1-4 reported, scaled by a ratio measured in the build container); when oracle/_ref is absent the port is the value and
    kind="port". Thread count: best of a 3-point sweep on one discriminator forward of the port, used for both legs."""
    from gigagan_pytorch_amd import GigaGAN, ops
    from gigagan_pytorch_amd.data import SyntheticImages
    from gigagan_pytorch_amd.gigagan import cycle
    from oracle.torch_ops import OracleOps
    from oracle.cpu_trainer import install_cpu_adamw
    t_start = time.time()
    cores = os.cpu_count() or 1
    bs, S = 4, 256
    torch.manual_seed(0)
    with ops.use_impl(OracleOps()):
        gan = GigaGAN(generator=dict(C2_G, image_size=S), discriminator=dict(C2_D, image_size=S), device='cpu',
                      apply_gradient_penalty_every=4, create_ema_generator_at_init=True, use_hip_graphs=False,
                      model_folder='/tmp/gg-bench-cpu-models', results_folder='/tmp/gg-bench-cpu-results')
        install_cpu_adamw(gan.G_opt)
        install_cpu_adamw(gan.D_opt)
        it = cycle(SyntheticImages(bs, S, device='cpu'))
        real = next(it)
        sweep = {}
        for n in sorted({min(cores, c) for c in (8, 32, 96)}):
            torch.set_num_threads(n)
            with torch.no_grad():
                gan.D(real, gan.D.real_images_to_rgbs(real), calc_aux_loss=False)       # warm the allocator at this width
                t0 = time.time()
                gan.D(real, gan.D.real_images_to_rgbs(real), calc_aux_loss=False)
            sweep[n] = time.time() - t0
        threads = min(sweep, key=sweep.get)
        torch.set_num_threads(threads)
        times = []
        for step in range(4):                       # trainer steps 1..4: step 4 carries the gradient penalty
            t0 = time.time()
            gan.train_step(it, bs)
            times.append(time.time() - t0)
            if step == 0 and time.time() - t_start + 4.5 * times[0] > budget_s:
                break           # a slow host: one plain step + one penalty step instead of the whole cycle
        if len(times) == 4:
            per_step = sum(times) / 4
            what = 'one whole 4-step cycle through train_step: ' + ' + '.join(f'{t:.1f}' for t in times) + ' s (the 4th carries the gradient penalty)'
        else:
            t0 = time.time()
            gan.train_discriminator_step(dl_iter=it, apply_gradient_penalty=True)
            gan.train_generator_step(batch_size=bs, dl_iter=it)
            gp = time.time() - t0
            per_step = (3 * times[0] + gp) / 4
            what = f'plain step {times[0]:.1f} s + gradient-penalty step {gp:.1f} s, cycle mean (3 plain + 1 GP) / 4 (time budget)'
    del gan
    port = dict(value=bs / per_step, unit='images/sec', cores=threads,
                sample=f'our trainer on the fp32 CPU oracle, config-2 dims 256x256, batch {bs}: {what}')
    sweep_note = (f'{threads} threads = best of the sweep {({k: round(v, 2) for k, v in sweep.items()})} (seconds per D forward of the '
                  f'port) on a {cores}-core host')
    ref = cpu_reference_leg(threads)
    if ref is not None and 'images_per_sec' in ref:
        return dict(value=ref['images_per_sec'], unit='images/sec', cores=threads, kind='reference',
                    sample=(f"the unmodified reference trainer ({ref['origin']}), GigaGAN(...)(steps=...) on the CPU, fp32 (amp=False), "
                            f"config-2 dims 256x256, batch {ref['batch']}: {ref['warmup_steps']} warm-up step ({ref['warmup_s']:.1f} s), then "
                            f"trainer steps {ref['timed_steps'][0]}-{ref['timed_steps'][1]} = one whole 4-step cycle (3 plain + 1 gradient-"
                            f"penalty G+D step) in {ref['timed_cycle_s']:.1f} s; {sweep_note}"),
                    port=port, port_vs_reference=port['value'] / ref['images_per_sec'])
    rec = dict(port, kind='port', sample=port['sample'] + '; ' + sweep_note)
    if ref is not None:
        rec['reference_leg_error'] = ref.get('error')
    return rec


def modconv_forward_roofline(gan, batch, dev):
    """north-star sub-target: the style-modulated (demodulated 3x3) adaptive convolutions of ONE generator forward at the
    bench batch, no-grad path (what the D-step runs): the forward's ONE batched modulation launch (gg_modw_multi_fwd: softmax over
    the kernels, demodulation coefficients, per-sample weights of every layer that is not behind a skip-layer excitation) plus
    everything the 15 layer calls launch (excited layers' own weight launch, the convolution, split-K finish). HIP events on the
    launch stream around EVERY C-ABI launch (kernels.LaunchProfiler), executed eagerly; `achieved` = algorithmic flops
    2*b*O*I*9*H*W per layer (171.5 GF at batch 32) / the time of one hipGraph replay of exactly these launches."""
    from gigagan_pytorch_amd import ops, kernels as K
    rec, calls, prep = [], [], []
    orig, orig_prep, orig_pair = ops.HipOps.modconv2d, ops.HipOps.modconv_prepare, ops.HipOps.modconv_pair

    with K.LaunchProfiler() as prof:
        def timed(self, x, weights, mod, kernel_mod=None, demod=True, **kw):
            if not demod or weights.shape[-1] != 3:
                return orig(self, x, weights, mod, kernel_mod, demod=demod, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n0 = len(prof.records)
            e0.record()
            y = orig(self, x, weights, mod, kernel_mod, demod=demod, **kw)
            e1.record()
            b, _, H, W = x.shape
            O, I = weights.shape[1], weights.shape[2]
            rec.append((e0, e1, 2.0 * b * O * I * 9 * H * W, f'{I}->{O}@{H}x{W}', n0, len(prof.records)))
            calls.append((x, weights, mod, kernel_mod, dict(kw, demod=demod)))
            return y

        def timed_pair(self, x, first, second):
            # a block's two layers as one launch (gg_spair_fwd); None = not fused: the two modconv2d calls follow and are recorded there
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n0 = len(prof.records)
            e0.record()
            y = orig_pair(self, x, first, second)
            e1.record()
            if y is None:
                return None
            b, _, H, W = x.shape
            fl, chain = 0., []
            for d in (first, second):
                O, I = d['weights'].shape[1], d['weights'].shape[2]
                fl += 2.0 * b * O * I * 9 * H * W
                chain += [I, O] if not chain else [O]
            rec.append((e0, e1, fl, '->'.join(map(str, chain)) + f'@{H}x{W} (fused pair)', n0, len(prof.records)))
            calls.append(('pair', x, first, second))
            return y

        def timed_prep(self, specs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n0 = len(prof.records)
            e0.record()
            n = orig_prep(self, specs)
            e1.record()
            prep.append((specs, e0, e1, n0, len(prof.records), n))
            return n
        ops.HipOps.modconv2d, ops.HipOps.modconv_prepare, ops.HipOps.modconv_pair = timed, timed_prep, timed_pair
        try:
            with torch.no_grad():
                for _ in range(3):
                    rec.clear()
                    calls.clear()
                    prep.clear()
                    gan.G(noise=torch.randn(batch, gan.G.style_network_dim, device=dev))
            torch.cuda.synchronize()
        finally:
            ops.HipOps.modconv2d, ops.HipOps.modconv_prepare, ops.HipOps.modconv_pair = orig, orig_prep, orig_pair
    # the same launches, on the inputs they saw, replayed as ONE hipGraph: their GPU time as the training step executes them
    # (back to back, kernel boundaries included, no host launch gaps - an eager split-K launch pair is ~10 us apart)
    graph_ms = None
    try:
        impl = ops.HipOps()

        def run_all():
            for specs, *_ in prep:
                orig_prep(impl, specs)
            for c in calls:
                if isinstance(c[0], str):
                    orig_pair(impl, *c[1:])
                else:
                    orig(impl, *c[:4], **c[4])
            impl.modconv_release()
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                run_all()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            g, _, _ = K.capture_graph(run_all)
        for _ in range(3):
            g.replay()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for _ in range(20):
            g.replay()
        g1.record()
        torch.cuda.synchronize()
        graph_ms = g0.elapsed_time(g1) / 20
        del g
    except Exception as e:       # noqa: BLE001
        graph_ms = None
        graph_err = f'{type(e).__name__}: {e}'
    layers, call_ms, kern_ms, fl = [], 0., 0., 0.
    for specs, e0, e1, n0, n1, n in prep:
        launches = [(nm, a.elapsed_time(b)) for nm, a, b in prof.records[n0:n1]]
        t_kern = sum(t for _, t in launches)
        layers.append(dict(layer=f'batched modulation of {n} layers', call_us=e0.elapsed_time(e1) * 1e3, kernel_us=t_kern * 1e3,
                           launches={nm: round(t * 1e3, 1) for nm, t in launches}))
        call_ms, kern_ms = call_ms + e0.elapsed_time(e1), kern_ms + t_kern
    for e0, e1, f, name, n0, n1 in rec:
        t_call = e0.elapsed_time(e1)
        launches = [(nm, a.elapsed_time(b)) for nm, a, b in prof.records[n0:n1]]
        t_kern = sum(t for _, t in launches)
        layers.append(dict(layer=name, call_us=t_call * 1e3, kernel_us=t_kern * 1e3, kernel_tflops=f / max(t_kern, 1e-9) / 1e9,
                           launches={nm: round(t * 1e3, 1) for nm, t in launches}))
        call_ms, kern_ms, fl = call_ms + t_call, kern_ms + t_kern, fl + f
    t_ms = graph_ms if graph_ms else kern_ms
    return dict(achieved=fl / t_ms / 1e9, peak=MFMA_PEAK_TF, unit='TFLOP/s', frac=fl / t_ms / 1e9 / MFMA_PEAK_TF,
                graph_ms=graph_ms, kernel_ms=kern_ms, call_ms=call_ms, gflop=fl / 1e9, batch=batch, layers=layers,
                note='the 15 demodulated 3x3 adaptive convs of one no-grad generator forward: `achieved` = algorithmic flops '
                     '(2*b*O*I*9*H*W) / graph_ms, the time of one hipGraph replay of exactly these launches (the batched modulation '
                     'launch + every kernel the 15 calls launch: excited layers\' per-sample-weight kernel, convolution, split-K '
                     'reduction; kernel boundaries included). kernel_ms / the per-layer table = HIP events around every C-ABI launch '
                     'issued eagerly (a launch that enqueues two kernels includes the host gap between them); call_ms brackets the '
                     'eager calls')


# ---- the multi-rank half of the contract, as functions a world-2 gloo test can drive (tests/test_distributed_cpu.py) ------------
def operand_activity_probe(dev):
    """How much of the dominant kernel's distance to the MFMA peak is the chip's power management rather than the kernel: ONE launch shape
    of gg_conv3<256> (the discriminator's stage-4 second conv at the bench batch: 256 images x 16x16, 512 -> 512 channels, 3x3 = 309.2
    GFLOP) timed from a hipGraph on operands of different bit activity - N(0,1) activations / N(0, 0.05) weights as in training, the
    constant 1.0, zeros. Same instruction stream, same bytes moved, same launch; only the switching in the operand paths and the
    multiplier arrays differs (round 6: 1250 / 1663 / 1753 TFLOP/s, profiles/r06_power_probe.log)."""
    from gigagan_pytorch_amd import kernels as K
    n, R, ci, co = 256, 16, 512, 512
    flops = 2.0 * n * R * R * co * ci * 9
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    cases = (('training_like', lambda *sh: torch.randn(*sh, device=dev, generator=g), 0.05),
             ('constant_one', lambda *sh: torch.ones(*sh, device=dev), 1.0),
             ('zeros', lambda *sh: torch.zeros(*sh, device=dev), 1.0))
    out = dict(kernel='gg_conv3_kernel<256, 2, 4, ...>', shape=f'{n} x {R}x{R}, {ci} -> {co}, 3x3', gflop=flops / 1e9, peak=MFMA_PEAK_TF,
               note='the same launch on operands of different bit activity (hipGraph of 10 launches, HIP events); frac = TFLOP/s / peak')
    for name, make, wscale in cases:
        x = make(n, R, R, ci).to(torch.bfloat16)
        w = (make(co, 9 * ci) * wscale).to(torch.bfloat16)
        fn = lambda: K.conv2d_nhwc(x, w, ksize=3, force_tile=7)     # noqa: E731
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for _ in range(10):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        out[name] = dict(launch_us=round(us, 1), tflops=round(flops / us / 1e6, 1), frac=round(flops / us / 1e6 / MFMA_PEAK_TF, 4))
    return out


def check_comm_for_measurement(world: int, dry: bool, backend: str, comm_world, reducers: bool):
    """None when the run may be reported, else why not: at N > 1 on GPUs the gradient exchange must be gg_comm/rccl with N ranks in the
    communicator and both flat-gradient reducers wired - never the torch.distributed fallback. (The CPU dry run states its transport in
    the line instead: gloo, or the gloo-backed double of the native communicator.)"""
    if world <= 1 or dry:
        return None
    if backend != 'gg_comm/rccl':
        return f'--gpus {world}: the gradient exchange is {backend!r}, not the native RCCL path (gg_comm/rccl): refusing to report a number'
    if comm_world != world:
        return f'--gpus {world}: the RCCL communicator holds {comm_world} rank(s): refusing to report a number'
    if not reducers:
        return f'--gpus {world}: the trainer has no in-backward gradient reducers on its flat buffers: refusing to report a number'
    return None


def max_over_ranks(dt: float, world: int, dev) -> float:
    """the step time that counts is the slowest rank's: MAX-reduce over the default process group (RCCL on GPUs, gloo in the dry run)."""
    if world <= 1:
        return dt
    import torch.distributed as dist
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_per_rank(mine: dict, world: int):
    """every rank's own record (step time on its clock, exposed communication time) on every rank; None on one rank."""
    if world <= 1:
        return None
    import torch.distributed as dist
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    return per_rank


# (the toy dims of tests/helpers.py TINY_G / TINY_D: the 2-rank gloo trainer test runs the same model)
DRY_G = dict(dim_capacity=8, dim_max=32, dim_latent=32, style_network=dict(dim=32, depth=2), num_skip_layers_excite=1,
             self_attn_resolutions=(8,), self_attn_heads=2, self_attn_dim_head=16)
DRY_D = dict(dim_capacity=8, dim_max=32, num_skip_layers_excite=1, attn_resolutions=(8,), attn_heads=2, attn_dim_head=16,
             multiscale_input_resolutions=(8,))


def dry_run_setup(world):
    """--dry-run-cpu: bind the C ABI built for the host-side kernel emulator (test infrastructure) and, with GG_BENCH_FAKE_NATIVE=1,
    install the gloo-backed double of the RCCL communicator (tests/test_distributed_cpu.py::GlooBackedNativeComm) so that the
    `gg_comm/rccl` branches of this file (comm_world, exposed-communication timing, shutdown) execute at world size > 1."""
    from gigagan_pytorch_amd import _C, distributed as gdist
    _C.bind(ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so')
    if world > 1 and os.environ.get('GG_BENCH_FAKE_NATIVE'):
        sys.path.insert(0, str(ROOT / 'tests'))
        from test_distributed_cpu import GlooBackedNativeComm
        gdist._native = GlooBackedNativeComm(world)


def dry_run_register(gan):
    from gigagan_pytorch_amd import distributed as gdist
    comm = gdist.native_comm()
    if comm is not None and hasattr(comm, 'buffers'):
        comm.buffers += [gan.D_opt.flat_g, gan.G_opt.flat_g]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 32; 16 for the secondary workloads)')
    ap.add_argument('--workload', choices=['uncond', 'upsampler', 'text'], default='uncond',
                    help='uncond = the headline config 2/3; upsampler = config 5 (UnetUpsampler 64->256); text = config 4')
    ap.add_argument('--image-size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile-cycle', action='store_true')
    ap.add_argument('--no-graphs', action='store_true', help='issue every launch eagerly instead of replaying hipGraphs')
    ap.add_argument('--data', choices=['resident', 'loader'], default='resident',
                    help='resident = synthetic batches already in HBM (the contract\'s timed region); loader = the same synthetic '
                         'images as fp32 HOST tensors through a torch DataLoader + data.DevicePrefetcher (pinned staging, H2D copy '
                         'on a side stream one batch ahead): shows whether the input leg is hidden')
    ap.add_argument('--loader-workers', type=int, default=2, help='--data loader: DataLoader worker processes (collation off the main thread)')
    ap.add_argument('--no-restore', action='store_true',
                    help='let the trajectory run on (it diverges on synthetic uniform images, here as in the reference)')
    ap.add_argument('--dry-run-cpu', action='store_true',
                    help='(tests/test_distributed_cpu.py) run THIS file\'s multi-rank control flow - rank bring-up, barriers, the MAX-'
                         'reduce of the step time, per-rank gather, exposed-communication timing, the one JSON line from rank 0 - on '
                         'CPU ranks over gloo with a toy model on the host-side kernel emulator; the line it prints is not a measurement')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher (replaces `accelerate launch`, reference README.md:168-182) -
        # one rank per GPU under torch.distributed.run on this node; rank 0 of the children prints the JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        sys.exit(subprocess.call(cmd, env=env))

    from gigagan_pytorch_amd import distributed as gdist, kernels as K
    from gigagan_pytorch_amd.data import SyntheticImages
    from gigagan_pytorch_amd.gigagan import cycle
    import torch.distributed as dist

    dry = args.dry_run_cpu
    rank, local, world = gdist.init_from_env('cpu' if dry else 'cuda')
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU (or let bench.py do it)'
    if dry:
        dev = torch.device('cpu')
        dry_run_setup(world)
    else:
        assert torch.cuda.is_available(), 'bench.py needs a GPU'
        assert torch.cuda.device_count() > local, f'rank {rank}: no GPU {local} on this node ({torch.cuda.device_count()} visible)'
        dev = torch.device('cuda', local)
        torch.cuda.set_device(dev)

    def sync():
        if dev.type == 'cuda':
            torch.cuda.synchronize()

    if args.batch is None:
        args.batch = (2 if dry else 32) if args.workload == 'uncond' else 16
    steps = (args.steps + 3) // 4 * 4
    warmup = args.warmup
    if dry:
        args.image_size = 16
        gan = build_gan(16, dev, g_over=DRY_G, d_over=DRY_D, use_hip_graphs=False)
        dry_run_register(gan)
    else:
        gan = build_gan(args.image_size, dev, use_hip_graphs=False if args.no_graphs else None, workload=args.workload)
    torch.manual_seed(1 + rank)          # identical initial weights (seed 0 in build_gan), per-rank latent / noise streams
    if args.workload == 'text':
        it = iter(SyntheticTextImages(args.batch, args.image_size, dev, seed=rank))
    elif args.data == 'loader':
        from torch.utils.data import DataLoader, TensorDataset
        from gigagan_pytorch_amd.data import DevicePrefetcher
        host = torch.rand(args.batch * 8, 3, args.image_size, args.image_size, generator=torch.Generator().manual_seed(rank))

        class _Images(TensorDataset):
            def __getitem__(self, i):
                return super().__getitem__(i)[0]
        it = cycle(DevicePrefetcher(DataLoader(_Images(host), batch_size=args.batch, shuffle=True, drop_last=True, pin_memory=True,
                                               num_workers=args.loader_workers, persistent_workers=args.loader_workers > 0,
                                               prefetch_factor=4 if args.loader_workers > 0 else None), dev, depth=3))
    else:
        it = cycle(SyntheticImages(args.batch, args.image_size, device=dev, seed=rank))

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    # The trajectory diverges within ~10 steps on synthetic uniform "images" (the reference does too: G loss 59k at step 3,
    # SURVEY.md §7.3; at config 2 the step-1 gradient penalty alone is 4.4e4, tests/golden/c2_step1.pt) and then runs on
    # inf/NaN operands, which a power-limited matrix pipe multiplies faster than real data. The timed region therefore
    # restarts from the initial weights / Adam moments at the start of every 4-step gradient-penalty cycle: three in-place
    # device copies and one weight-pack launch per model (~1 ms per cycle), INSIDE the timed region - extra work, nothing
    # skipped; every step still runs its full forward / backward / optimizer / EMA update. `--no-restore` switches it off.
    snap = None if args.no_restore else gan.state_snapshot()

    def run_steps(n):
        out = None
        for _ in range(n):
            if snap is not None and (gan._steps_host - 1) % 4 == 0:
                gan.state_restore(snap)
            out = gan.train_step(it, args.batch)
        return out

    # warm-up starts at the trainer's step 1; keep the timed region aligned to whole GP cycles
    run_steps(warmup)
    # align so that the K timed steps contain exactly K/4 GP steps; with hipGraphs on, also make sure one whole
    # 4-step cycle has run untimed (the three step kinds are captured on first use). Extra steps count as warm-up.
    while (gan._steps_host - 1) % 4 != 0 or (gan.use_hip_graphs and gan._steps_host < 5):
        run_steps(1)
        warmup += 1
    comm = gdist.native_comm()
    graphs_on = bool(gan._graphable(1))
    # a multi-GPU number must be THE path it claims: the native RCCL exchange (gg_comm_*) with every rank in the communicator. The
    # trainer keeps a torch.distributed fallback for bring-up; a measurement never takes it silently
    comm_problem = check_comm_for_measurement(world, dry, gdist.comm_backend(), None if comm is None else comm.world,
                                              gan.D_red is not None and gan.G_red is not None)
    if comm_problem:
        print(f'bench.py: {comm_problem}', file=sys.stderr, flush=True)
        sys.exit(3)
    # exposed share of the gradient exchange = how long the compute stream waits at the join behind the in-backward slices.
    # HIP events cannot be recorded inside a hipGraph replay, so with graphs on it is taken from the eager cycle further down.
    time_comm_here = comm is not None and world > 1 and not graphs_on
    if time_comm_here:
        comm.timing, comm.exposed_ms = True, []
    barrier()
    t0 = time.perf_counter()
    d_losses, g_losses = run_steps(steps)
    sync()
    dt_local = time.perf_counter() - t0         # this rank's own clock, before the closing barrier
    barrier()
    dt = time.perf_counter() - t0
    mine = dict(rank=rank, ms_per_step=dt_local / steps * 1e3, exposed_comm_ms_per_step=None)
    dt = max_over_ranks(dt, world, dev)
    if time_comm_here:
        comm.timing = False
        mine.update(exposed_comm_ms_per_step=sum(a.elapsed_time(b) for a, b in comm.exposed_ms) / steps,
                    exposed_comm_measured='timed region (eager launches)')
    nonfinite = {k: int((~torch.isfinite(v)).sum()) for k, v in (('g_params', gan.G_opt.flat_p), ('d_params', gan.D_opt.flat_p),
                                                                 ('g_grads', gan.G_opt.flat_g), ('d_grads', gan.D_opt.flat_g))}
    finite = not any(nonfinite.values())
    loss_vals = [float(v) for v in (*d_losses, *g_losses) if v is not None]
    finite = finite and all(v == v and abs(v) != float('inf') for v in loss_vals)

    ms_per_step = dt / steps * 1e3
    value = args.batch * world * steps / dt

    roofline = None
    if not args.no_profile_cycle:
        # one extra GP cycle, executed eagerly on EVERY rank (the steps contain the gradient all-reduce, so all ranks
        # must take part); rank 0 brackets each contraction launch with HIP events on the launch stream
        graphs_were_on, gan.use_hip_graphs = gan.use_hip_graphs, False   # HIP events cannot be recorded inside a replay
        if rank == 0 and not dry:
            K.profiler = K.GemmProfiler()
        time_comm_cycle = comm is not None and world > 1 and mine['exposed_comm_ms_per_step'] is None
        if time_comm_cycle:
            comm.timing, comm.exposed_ms = True, []
        run_steps(4)
        if time_comm_cycle:
            comm.timing = False
            sync()
            mine.update(exposed_comm_ms_per_step=sum(a.elapsed_time(b) for a, b in comm.exposed_ms) / 4,
                        exposed_comm_measured='eager 4-step cycle after the timed region (the timed steps are hipGraph replays)')
        agg = shapes = None
        if rank == 0 and K.profiler is not None:
            agg = K.profiler.summary()
            shapes = K.profiler.shape_summary()
            K.profiler = None
        gan.use_hip_graphs = graphs_were_on
        if rank == 0 and agg:
            Path(ROOT / 'gpurun_out').mkdir(exist_ok=True)
            (ROOT / 'gpurun_out' / 'bench_gemm_shapes.json').write_text(json.dumps(shapes, indent=1))
            (ROOT / 'gpurun_out' / 'bench_gemm_breakdown.json').write_text(json.dumps(agg, indent=1))
            name, a = max(agg.items(), key=lambda kv: kv[1]['ms'])
            tot_ms = sum(v['ms'] for v in agg.values())
            tot_fl = sum(v['flops'] for v in agg.values())
            achieved = a['flops'] / a['ms'] / 1e9
            # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the
            # gfx950 correction + WRITE_SIZE, KiB -> bytes), when the summary for this kernel is present
            traffic = traffic_shape = None
            for pmc in sorted((ROOT / 'profiles').glob('r0*_pmc_traffic*.json')):
                try:
                    rec = json.loads(pmc.read_text())
                    if rec.get('kernel') == name:
                        traffic, traffic_shape = rec.get('bytes_per_launch'), rec.get('shape')
                except Exception:
                    pass
            roofline = dict(bound='mfma', kernel=name, achieved=achieved, peak=MFMA_PEAK_TF, unit='TFLOP/s',
                            frac=achieved / MFMA_PEAK_TF, traffic=traffic, traffic_shape=traffic_shape,
                            avg_launch_us=a['ms'] / a['launches'] * 1e3, launches_per_step=a['launches'] / 4,
                            # everything `achieved` is made of, so that the fraction can be recomputed from this line alone:
                            # achieved = flops_per_launch / avg_launch_us (algorithmic 2*M*N*K of the kernel's launches in one
                            # eager 4-step cycle, HIP events on the launch stream); the per-shape table behind it is written to
                            # gpurun_out/bench_gemm_shapes.json and committed under profiles/ for the closing run
                            flops_per_launch=a['flops'] / a['launches'], launches_in_cycle=a['launches'],
                            gemm_kernel_table={k: dict(launches=v['launches'], ms=round(v['ms'], 4), gflop=round(v['flops'] / 1e9, 3))
                                               for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])},
                            all_gemm_kernels=dict(tflops=tot_fl / tot_ms / 1e9, ms_per_step=tot_ms / 4,
                                                  frac_of_step=tot_ms / 4 / ms_per_step),
                            step=dict(achieved=value / world * GF_PER_IMG / 1e3, peak=MFMA_PEAK_TF,
                                      frac=value / world * GF_PER_IMG / 1e3 / MFMA_PEAK_TF,
                                      note='whole-step algorithmic-minimum 1192.1 GF/img vs dense bf16 MFMA peak'))

    if rank == 0 and roofline is not None and shapes:
        # the persistent short-K contraction (gg_pgemm, plan tile 15) is an HBM problem: algorithmic bytes of its launches in the eager
        # cycle (operands + output: 2 * M * (K + N) + 2 * N * K; residual / GELU-aux operands are not in the launch key, so this is a
        # lower bound on the bytes and on the fraction) over their HIP-event time
        import re
        pg_bytes = pg_ms = 0.
        pg_n = 0
        for key, v in shapes.items():
            m = re.match(r'gg_pgemm_kernel M=(\d+) N=(\d+) K=(\d+)', key)
            if m:
                M_, N_, K_ = map(int, m.groups())
                pg_bytes += v['launches'] * (2. * M_ * (K_ + N_) + 2. * N_ * K_)
                pg_ms += v['ms']
                pg_n += v['launches']
        if pg_n:
            gbs = pg_bytes / pg_ms / 1e6
            roofline['short_k'] = dict(bound='hbm', kernel='gg_pgemm_kernel', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s',
                                       frac=gbs / HBM_PEAK_GBS, launches_in_cycle=pg_n, ms_in_cycle=round(pg_ms, 3),
                                       bytes='2*M*(K+N) + 2*N*K per launch (residual / GELU-aux operands not counted)')

    if rank == 0 and roofline is not None and args.workload == 'uncond':
        try:
            roofline['modconv_forward'] = modconv_forward_roofline(gan, args.batch, dev)
        except Exception as e:    # noqa: BLE001 - a secondary measurement must not take the bench line down
            roofline['modconv_forward'] = dict(error=f'{type(e).__name__}: {e}')

    if rank == 0 and roofline is not None and args.workload == 'uncond' and not dry:
        try:
            roofline['operand_activity'] = operand_activity_probe(dev)
        except Exception as e:    # noqa: BLE001 - a secondary measurement must not take the bench line down
            roofline['operand_activity'] = dict(error=f'{type(e).__name__}: {e}')

    if rank == 0 and roofline is not None:
        # the secondary fractions where the driver's parser looks (scalars at the top level of `roofline`)
        mf, sk = roofline.get('modconv_forward') or {}, roofline.get('short_k') or {}
        roofline['modconv_forward_frac'] = mf.get('frac')
        roofline['short_k_frac'] = sk.get('frac')

    per_rank = gather_per_rank(mine, world)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'uncond' and not dry:
        cpu = cpu_baseline()

    if rank == 0:
        metric, what = {
            'uncond': ('images/sec G+D step, uncond 256x256 bs32',
                       f'Unconditional GigaGAN image_size={args.image_size} dim_max=512 (G cap 8, D cap 16)'),
            'upsampler': ('images/sec G+D step, UnetUpsampler 64->256 bs16',
                          f'UnetUpsampler dim=32 64->{args.image_size} (train_upsampler=True), D cap 16 dim_max=512'),
            'text': ('images/sec G+D step, text-conditional 256x256 bs16',
                     f'Text-conditional GigaGAN image_size={args.image_size} dim_max=512, TextEncoder dim 64 depth 4 on '
                     'pre-computed (77, 512) token encodings, cross attention, matching-aware loss (no CLIP contrastive loss)'),
        }[args.workload]
        if dry:
            metric = 'DRY RUN of bench.py\'s control flow on CPU ranks (gloo, toy model, kernel emulator): not a measurement'
        line = dict(
            metric=metric, value=value, unit='images/sec', n_gpus=world,
            steps=steps, warmup=warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling='weak',
            vs_baseline=None, dtype='bf16',
            data='synthetic' if args.data == 'resident' else 'synthetic (fp32 host batches: DataLoader + pinned prefetch to the device)',
            config=dict(workload=f'{what} bf16 bs={args.batch}/GPU, GP every 4th step', global_batch=args.batch * world,
                        parallelism=f'dp{world}',
                        scaling_curve=('this line is one point; no 1 -> N curve of this code has been measured by its builder (1-GPU '
                                       'boxes only): efficiency is for the driver to compute from its own per-N runs'),
                        hip_graphs=bool(gan._graphable(1)), graph_memset_nodes_repaired=sum(gan._graph_memsets.values()), comm=gdist.comm_backend(),
                        comm_world=(comm.world if comm is not None else (world if world > 1 else 0)),
                        rccl_ranks=(comm.world if (comm is not None and gdist.comm_backend() == 'gg_comm/rccl') else None),
                        exposed_comm_ms_per_step_by_rank=(None if per_rank is None else
                                                          [r.get('exposed_comm_ms_per_step') for r in per_rank]),
                        comm_overlap=('in-backward slices: D %d, G %d' % (gan.D_red.n, gan.G_red.n)
                                      if (gan.D_red is not None and gan.overlap_grad_reduce) else 'none')),
            roofline=roofline, cpu_baseline=cpu,
            finite=finite, nonfinite=nonfinite, state_restored_every_cycle=snap is not None, per_rank=per_rank,
            last_losses=dict(d=float(d_losses.divergence), g=float(g_losses.divergence),
                             gp=float(d_losses.gradient_penalty), msd=float(d_losses.multiscale_divergence)))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        gdist.shutdown()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
