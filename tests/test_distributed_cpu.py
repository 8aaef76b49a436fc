"""CPU suite: the N>1 data-parallel path with world_size 2 over gloo (RCCL on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gigagan_pytorch_amd import distributed as gdist
    gdist.init_from_env('cpu')
    assert gdist.world_size() == world and gdist.rank() == rank
    # flat gradient all-reduce: sum across ranks, mean folded into the optimizer's grad_scale
    g = torch.full((1000,), float(rank + 1))
    gdist.wait_all(gdist.all_reduce_flat_grads(g, n_slices=3))
    ok = torch.allclose(g, torch.full((1000,), 3.0))
    # parameter broadcast
    p = torch.full((10,), float(rank))
    gdist.broadcast_flat_params(p)
    ok = ok and torch.equal(p, torch.zeros(10))
    # differentiable equal-shard all_gather: backward keeps the local slice (reference distributed.py:62-68)
    x = torch.full((2, 3), float(rank + 1), requires_grad=True)
    out, sizes = gdist.all_gather(x)
    (out * torch.arange(4.)[:, None]).sum().backward()
    ok = ok and out.shape == (4, 3) and torch.equal(x.grad, torch.arange(4.)[rank * 2:(rank + 1) * 2, None].expand(2, 3))
    # the generator's CLIP contrastive loss (gp.py:174-188): images and caption embeddings of every rank are gathered, each rank
    # keeps the gradient of ITS images; equals the single-process loss / gradient on the concatenated batch
    import torch.nn.functional as F
    from gigagan_pytorch_amd.gigagan import aux_clip_loss

    class Clip:
        proj = torch.randn(3 * 2 * 2, 6, generator=torch.Generator().manual_seed(5))
        table = F.normalize(torch.randn(16, 6, generator=torch.Generator().manual_seed(6)), dim=-1)

        def embed_texts(self, texts):
            return self.table[[int(t) for t in texts]], None

        def contrastive_loss(self, images, text_embeds):
            emb = F.normalize(F.adaptive_avg_pool2d(images, 2).flatten(1) @ self.proj, dim=-1)
            sim = text_embeds @ emb.t() * 10.
            labels = torch.arange(sim.shape[0])
            return (F.cross_entropy(sim, labels) + F.cross_entropy(sim.t(), labels)) / 2

    imgs_all = torch.rand(4, 3, 8, 8, generator=torch.Generator().manual_seed(3))
    mine = imgs_all[rank * 2:(rank + 1) * 2].clone().requires_grad_()
    loss = aux_clip_loss(Clip(), mine, texts=[str(2 * rank), str(2 * rank + 1)])
    loss.backward()
    full = imgs_all.clone().requires_grad_()
    ref = Clip().contrastive_loss(full, Clip().embed_texts(['0', '1', '2', '3'])[0])
    ref.backward()
    ok = ok and torch.allclose(loss, ref, atol=1e-6) and torch.allclose(mine.grad, full.grad[rank * 2:(rank + 1) * 2], atol=1e-6)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_data_parallel_equals_large_batch_gradient():
    """splitting a batch over 2 'ranks' and averaging the flat gradients equals the full-batch gradient — what the
    all-reduce + grad_scale=1/world computes (single process, no collective)."""
    from gigagan_pytorch_amd import ops
    from gigagan_pytorch_amd.discriminator import Discriminator
    from oracle.torch_ops import OracleOps
    from helpers import SMALL_D
    torch.manual_seed(0)
    D = Discriminator(**SMALL_D).eval()
    imgs = torch.rand(4, 3, 32, 32)
    with ops.use_impl(OracleOps()):
        def grads(x):
            l, ms, _ = D(x, D.real_images_to_rgbs(x), calc_aux_loss=False)
            return torch.autograd.grad(l.mean() + sum(m.mean() for m in ms), [p for p in D.parameters()], allow_unused=True)
        full = grads(imgs)
        a, b = grads(imgs[:2]), grads(imgs[2:])
    for f, x, y in zip(full, a, b):
        if f is not None:
            assert torch.allclose(f, (x + y) / 2, rtol=1e-3, atol=1e-5)


class GlooBackedNativeComm:
    """test double for distributed.NativeComm: the SAME class with its three stream-side primitives (fork / allreduce_ptr /
    allgather_ptr / join) re-implemented on host memory + gloo, so that every line of the native branches of
    GradReducer._launch / finish, NativeComm.all_reduce_ / all_gather runs at world size 2 without a GPU: slice bounds, byte
    offsets, element counts and the fork -> collective -> join ordering are what the RCCL entry points would be handed."""

    def __new__(cls, world):
        from gigagan_pytorch_amd import distributed as gdist

        class _Double(gdist.NativeComm):
            def __init__(self):
                super().__init__()
                self.accepts_host, self.world, self.lib = True, world, None
                self.log, self.buffers = [], []

            def _view(self, ptr, nbytes):
                for t in self.buffers:
                    base = t.data_ptr()
                    if base <= ptr and ptr + nbytes <= base + t.numel() * t.element_size():
                        assert (ptr - base) % t.element_size() == 0
                        lo = (ptr - base) // t.element_size()
                        return t.view(-1)[lo:lo + nbytes // t.element_size()], t, lo
                raise AssertionError(f'collective on {nbytes} bytes at {ptr:#x}: not inside any registered buffer')

            def fork(self, like):
                self.log.append(('fork',))

            def allreduce_ptr(self, ptr, count, dtype_code):
                assert self.log and self.log[-1][0] in ('fork', 'allreduce'), 'a collective without a fork edge in front of it'
                v, t, lo = self._view(ptr, count * (4 if dtype_code == 0 else 2))
                assert v.dtype == (torch.float32 if dtype_code == 0 else torch.bfloat16)
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
                self.log.append(('allreduce', id(t), lo, count))

            def allgather_ptr(self, src, dst, nbytes):
                vs, *_ = self._view(src, nbytes)
                vd, *_ = self._view(dst, nbytes * self.world)
                dist.all_gather_into_tensor(vd, vs.contiguous())
                self.log.append(('allgather', nbytes))

            def join(self, like):
                self.log.append(('join',))

            def _mark(self, like):
                """(timing only) a host timestamp with the `elapsed_time` of a HIP event, so that bench.py's exposed-communication
                bookkeeping (pairs of marks around the join) runs in the dry run"""
                if not self.timing:
                    return None
                import time

                class _HostMark:
                    t = time.perf_counter()

                    def elapsed_time(self, other):
                        return (other.t - self.t) * 1e3
                return _HostMark()

            def _keep(self, *tensors):
                pass

            def all_gather(self, x):
                x = x.contiguous()
                self.buffers.append(x)
                orig_empty = torch.empty

                def grab(*a, **k):
                    t = orig_empty(*a, **k)
                    self.buffers.append(t)
                    return t
                torch.empty = grab
                try:
                    return super().all_gather(x)
                finally:
                    torch.empty = orig_empty
                    del self.buffers[-2:]

            def destroy(self):
                pass
        return _Double()


def _train_worker(rank, world, port, ret, tmp, overlap=True, native=False):
    if not overlap:
        os.environ['GG_NO_COMM_OVERLAP'] = '1'
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root)); sys.path.insert(0, str(root / 'tests'))
    torch.set_num_threads(2)
    from gigagan_pytorch_amd import _C, GigaGAN, distributed as gdist
    from gigagan_pytorch_amd.data import SyntheticImages
    from gigagan_pytorch_amd.gigagan import cycle
    from helpers import TINY_G, TINY_D
    _C.bind(root / 'tests' / 'emu' / 'libgigagan_amd_emu.so')
    gdist.init_from_env('cpu')
    fake = None
    if native:
        fake = gdist._native = GlooBackedNativeComm(world)
    torch.manual_seed(0)                     # identical initial replicas (and a broadcast on top)
    gan = GigaGAN(generator=dict(TINY_G), discriminator=dict(TINY_D), apply_gradient_penalty_every=2, device='cpu',
                  model_folder=f'{tmp}/m{rank}', results_folder=f'{tmp}/r{rank}')
    torch.manual_seed(10 + rank)             # per-rank latents / noise / data
    it = cycle(SyntheticImages(2, 16, seed=rank))
    d0 = gan.D_opt.flat_p.clone()
    in_bwd = []
    if fake is not None:
        fake.buffers += [gan.D_opt.flat_g, gan.G_opt.flat_g]
        assert gdist.comm_backend() == 'gg_comm/rccl' and gan.D_red is not None and gan.G_red is not None
    native_log = []
    for _ in range(4):                       # steps 2 and 4 carry the gradient penalty; 3 and 4 re-use learned slice counts
        gan.train_step(it, 2)
        if gan.D_red is not None:
            in_bwd.append((gan.D_red.in_backward_launches, gan.G_red.in_backward_launches))
        if fake is not None:
            native_log.append(list(fake.log))
            fake.log.clear()
    assert gan.overlap_grad_reduce == overlap
    if not overlap:                          # GG_NO_COMM_OVERLAP: the reducers exist but are bypassed - one exchange after the backward
        assert gan.D_red.launched == 0 and gan.G_red.launched == 0 and gan.D_red.sig is None
    if fake is not None:
        # every step: each model's flat gradient buffer went through the communicator exactly once, as its reducer's slices, last
        # slice first, each collective behind a fork edge, and the optimizer launch behind a join
        for step_log in native_log:
            for opt, red in ((gan.D_opt, gan.D_red), (gan.G_opt, gan.G_red)):
                mine = [e for e in step_log if e[0] == 'allreduce' and e[1] == id(opt.flat_g)]
                assert [(lo, lo + n) for _, _, lo, n in mine] == [(red.bounds[k], red.bounds[k + 1]) for k in reversed(range(red.n))], \
                    (mine, red.bounds)
            kinds = [e[0] for e in step_log]
            assert kinds.count('join') == 2 and kinds[-1] == 'join', kinds
            for i, k in enumerate(kinds):
                if k == 'allreduce':
                    assert kinds[i - 1] == 'fork', kinds
        # the stand-alone entry points on the same primitives
        g = torch.full((1000,), float(rank + 1))
        fake.buffers.append(g)
        gdist.wait_all(gdist.all_reduce_flat_grads(g, n_slices=3))
        assert torch.equal(g, torch.full((1000,), 3.0)), g[:4]
        assert [e[2:] for e in fake.log if e[0] == 'allreduce'] == [(0, 512), (512, 488)], fake.log
        x = torch.full((2, 3), float(rank + 1), requires_grad=True)
        out, _ = gdist.all_gather(x)
        assert out.shape == (4, 3) and torch.equal(out.detach(), torch.tensor([1., 1., 2., 2.])[:, None].expand(4, 3)), out
        (out * torch.arange(4.)[:, None]).sum().backward()
        assert torch.equal(x.grad, torch.arange(4.)[rank * 2:(rank + 1) * 2, None].expand(2, 3))
    flat = torch.cat([gan.D_opt.flat_p, gan.G_opt.flat_p])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    moved = not torch.equal(d0, gan.D_opt.flat_p)
    import hashlib
    ret[rank] = dict(ok=bool(same and moved and torch.isfinite(flat).all()), in_bwd=in_bwd, slices=(gan.D_red.n if gan.D_red else 0),
                     overlap_on=bool(gan.overlap_grad_reduce), native_steps=len(native_log),
                     digest=hashlib.sha256(flat.numpy().tobytes()).hexdigest())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_keeps_replicas_identical(tmp_path):
    """the whole trainer on 2 gloo ranks (emulator build): different data and noise per rank, summed flat gradients,
    grad_scale = 1/world in the fused AdamW -> bit-identical replicas after plain and gradient-penalty steps."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_train_worker, args=(world, port, ret, str(tmp_path)), nprocs=world, join=True)
    assert all(ret.get(r) and ret[r]['ok'] for r in range(world)), dict(ret)
    # the gradient exchange ran INSIDE the backward passes once the slice counts were learned (steps 3 and 4): most slices of both
    # models went out before the backward returned ...
    r0 = ret[0]
    assert r0['overlap_on'] and r0['slices'] >= 3, r0
    assert r0['in_bwd'][0] == (0, 0), r0                   # learning pass: everything flushed after the backward
    assert all(d >= r0['slices'] - 1 and g >= 1 for d, g in r0['in_bwd'][2:]), r0
    # ... and gives bit-identical parameters to one exchange after the backward (summation order inside a slice is the same)
    ret2 = mgr.dict()
    mp.spawn(_train_worker, args=(world, _free_port(), ret2, str(tmp_path / 'b'), False), nprocs=world, join=True)
    assert ret2[0]['ok'] and not ret2[0]['overlap_on'] and ret2[0]['digest'] == r0['digest'], (dict(ret2), r0)
    # `gg_comm_allreduce` / `gg_comm_allgather` at world > 1 cannot run without GPUs, but everything AROUND them
    # can: with a NativeComm whose stream primitives are backed by gloo (GlooBackedNativeComm), the native branches of GradReducer
    # (in-backward sliced exchange: fork edge, byte offsets, counts, join) and of all_reduce_flat_grads / all_gather execute on 2 ranks
    # and give the same bit-identical replicas - and the same parameters - as the torch.distributed path
    ret3 = mgr.dict()
    mp.spawn(_train_worker, args=(world, _free_port(), ret3, str(tmp_path / 'c'), True, True), nprocs=world, join=True)
    assert all(ret3.get(r) and ret3[r]['ok'] for r in range(world)), dict(ret3)
    assert ret3[0]['native_steps'] == 4 and ret3[0]['digest'] == ret3[1]['digest'] == r0['digest'], (dict(ret3), r0)
    assert all(d >= ret3[0]['slices'] - 1 and g >= 1 for d, g in ret3[0]['in_bwd'][2:]), dict(ret3[0])



def test_bench_multi_rank_control_flow_runs_on_two_gloo_ranks():
    """bench.py's multi-rank half (self-launch through torch.distributed.run, rank bring-up, barriers, the MAX-
    reduce of the step time, per-rank gather, exposed-communication timing on the native-communicator branch, shutdown, ONE JSON
    line from rank 0) has never run on > 1 GPU; `--dry-run-cpu` executes exactly that code on 2 CPU ranks over gloo (toy model,
    kernel emulator, the gloo-backed double of the RCCL communicator)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, GG_BENCH_FAKE_NATIVE='1', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(root / 'bench.py'), '--gpus', '2', '--dry-run-cpu', '--steps', '4', '--warmup', '1',
                          '--no-profile-cycle'],        # (the eager extra cycle only feeds the roofline block, which a dry run does not have)
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines                                   # rank 0 only, one line
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 4 and rec['scaling'] == 'weak' and rec['higher_is_better'] is True
    assert rec['config']['parallelism'] == 'dp2' and rec['config']['global_batch'] == 2 * 2
    assert rec['config']['comm'] == 'gg_comm/rccl' and rec['config']['comm_world'] == 2
    assert rec['config']['comm_overlap'].startswith('in-backward slices')
    assert 'no 1 -> N curve' in rec['config']['scaling_curve']
    assert len(rec['per_rank']) == 2 and sorted(r['rank'] for r in rec['per_rank']) == [0, 1]
    for r in rec['per_rank']:
        assert r['ms_per_step'] > 0 and r['exposed_comm_ms_per_step'] is not None and r['exposed_comm_ms_per_step'] >= 0
    # whole-job value = global batch * steps / the SLOWEST rank's time
    assert abs(rec['value'] - 4 * rec['steps'] / (rec['ms_per_step'] * rec['steps'] / 1e3)) < 1e-6 * rec['value']
    assert rec['ms_per_step'] >= max(r['ms_per_step'] for r in rec['per_rank']) - 1e-6
    assert rec['finite'] and rec['cpu_baseline'] is None and 'DRY RUN' in rec['metric']


def test_bench_refuses_a_multi_gpu_number_on_the_fallback_exchange():
    """bench.check_comm_for_measurement: at N > 1 on GPUs the line is only printed when the gradient exchange is gg_comm/rccl with all N
    ranks in the communicator and the in-backward reducers wired; the torch.distributed fallback (or a short communicator) makes
    bench.py exit non-zero instead of reporting a number that is not the path it names."""
    import bench
    ok = bench.check_comm_for_measurement
    assert ok(1, False, 'none', None, False) is None                       # one GPU: nothing to exchange
    assert ok(2, True, 'torch.distributed/gloo', None, True) is None       # the CPU dry run names its transport in the line
    assert ok(8, False, 'gg_comm/rccl', 8, True) is None
    assert 'not the native RCCL path' in ok(8, False, 'torch.distributed/nccl', None, True)
    assert 'holds 4 rank' in ok(8, False, 'gg_comm/rccl', 4, True)
    assert 'no in-backward gradient reducers' in ok(2, False, 'gg_comm/rccl', 2, False)

