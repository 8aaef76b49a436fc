"""Data-parallel plumbing: one process per GPU, RCCL over xGMI. Replaces the reference's accelerate/DDP path
(gp.py:1898-1908, :1987) and its hand-written variable-size all_gather (distributed.py:20-68).

Rendezvous, the parameter broadcast and barriers go through `torch.distributed` (backend "nccl" is RCCL on ROCm, gloo on the
CPU); the per-step gradient exchange on a GPU goes through the library's own RCCL entry points (`gg_comm_*`,
include/gigagan_amd.h): a second communicator bootstrapped from a unique id that rank 0 broadcasts, driven on a dedicated side
stream fenced by events against the compute stream, so the host never blocks and the optimizer launch simply queues behind
the last slice.

Gradient exchange: the fused optimizer keeps every gradient of a model in ONE flat fp32 buffer, so the
all-reduce is issued on a few large contiguous slices (ring all-reduce over xGMI is per-link bound; large
messages amortise latency) instead of DDP's 25 MB buckets + copies. The discriminator's gradients are NOT
reduced (or even computed) in the generator step, where the reference computes, reduces and then discards
them.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch.autograd import Function


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(device_type: str = 'cuda') -> tuple[int, int, int]:
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if ws > 1 and not (dist.is_available() and dist.is_initialized()):
        backend = 'nccl' if device_type == 'cuda' else 'gloo'
        if device_type == 'cuda':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rk, world_size=ws)
    return rk, local, ws


# ---- the library's own RCCL communicator (gg_comm_*) -----------------------------------------------------------------------
class NativeComm:
    """one RCCL communicator per process behind the C ABI, with its side stream. `init()` is collective over the default
    process group (the unique id travels through it); world 1 is allowed (a 1-GPU box exercises the same code path)."""

    def __init__(self):
        self.lib = None
        self.stream = None
        self.world = 0
        self.exposed_ms = []        # filled when `timing` is on: how long the compute stream waited per exchange
        self.timing = False

    def init(self, device, rank=None, world=None):
        from . import _C
        L = _C.lib()
        if L.is_emulator:
            raise RuntimeError('gg_comm needs the gfx950 build (RCCL runs on GPUs)')
        rank = dist.get_rank() if rank is None else rank
        world = dist.get_world_size() if world is None else world
        import ctypes as C
        rccl = [p for p in (torch.__path__[0] + '/lib/librccl.so',) if os.path.exists(p)]
        L.check(L.lib.gg_comm_load(rccl[0].encode() if rccl else None), 'gg_comm_load')
        uid = (C.c_char * 128)()
        if rank == 0:
            L.check(L.lib.gg_comm_unique_id(uid), 'gg_comm_unique_id')
        if world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        torch.cuda.set_device(device)
        L.check(L.lib.gg_comm_init(rank, world, uid), 'gg_comm_init')
        assert L.lib.gg_comm_world() == world, (L.lib.gg_comm_world(), world)
        self.lib, self.world = L, world
        self.stream = torch.cuda.Stream(device=device)
        return self

    def all_reduce_(self, flat: torch.Tensor, n_slices: int = 4):
        """in-place sum over ranks of a contiguous buffer, issued as `n_slices` large collectives on the side stream after
        everything the compute stream has queued so far; returns a handle whose wait() fences the compute stream."""
        assert flat.is_cuda and flat.is_contiguous() and flat.dtype in (torch.float32, torch.bfloat16)
        L = self.lib
        cur = torch.cuda.current_stream(flat.device)
        t0 = None
        if self.timing:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(cur)
        self.stream.wait_stream(cur)
        n = flat.numel()
        step = ((n + n_slices - 1) // n_slices + 255) // 256 * 256
        dt = 0 if flat.dtype == torch.float32 else 1
        for s in range(0, n, step):
            cnt = min(step, n - s)
            L.check(L.lib.gg_comm_allreduce(flat.data_ptr() + s * flat.element_size(), cnt, dt, self.stream.cuda_stream),
                    'gg_comm_allreduce')
        done = torch.cuda.Event()
        done.record(self.stream)
        return _Fence(self, done, cur, t0)

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        out = torch.empty((self.world * x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        cur = torch.cuda.current_stream(x.device)
        self.stream.wait_stream(cur)
        self.lib.check(self.lib.lib.gg_comm_allgather(x.data_ptr(), out.data_ptr(), x.numel() * x.element_size(), 2,
                                                      self.stream.cuda_stream), 'gg_comm_allgather')
        cur.wait_stream(self.stream)
        x.record_stream(self.stream)
        out.record_stream(self.stream)
        return out

    def destroy(self):
        if self.lib is not None:
            torch.cuda.synchronize()
            self.lib.lib.gg_comm_destroy()
            self.lib = None


class _Fence:
    def __init__(self, comm, event, stream, t0):
        self.comm, self.event, self.stream, self.t0 = comm, event, stream, t0

    def wait(self):
        self.stream.wait_event(self.event)
        if self.t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(self.stream)
            self.comm.exposed_ms.append((self.t0, t1))


_native: NativeComm | None = None


def native_comm() -> NativeComm | None:
    return _native


def enable_native_comm(device):
    """create the gg_comm_* communicator for this process (collective: every rank calls it). The ranks then agree (a MIN
    all-reduce over the bootstrap process group) on whether EVERY rank succeeded; if one did not (no librccl to bind, a refused
    communicator), all of them drop the native communicator and the gradient exchange stays on torch.distributed's RCCL backend —
    a replica must never wait in a collective its peers do not issue. Returns the communicator or None."""
    global _native
    if _native is not None:
        return _native
    comm, err = None, None
    try:
        comm = NativeComm().init(device)
    except Exception as e:     # noqa: BLE001 - any failure means "use the process group instead", decided collectively below
        err = e
    if is_distributed():
        ok = torch.tensor([1 if comm is not None else 0], device=device, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.destroy()
            if dist.get_rank() == 0:
                print(f'gigagan_pytorch_amd: native RCCL communicator unavailable on some rank ({err}); '
                      'gradient exchange through torch.distributed', flush=True)
            return None
    elif comm is None:
        raise err
    _native = comm
    return _native


def shutdown():
    global _native
    if _native is not None:
        _native.destroy()
        _native = None


def all_reduce_flat_grads(flat_grad: torch.Tensor, n_slices: int = 4):
    """sum-reduce a flat gradient buffer across ranks (the mean is folded into the optimizer's grad_scale). On a GPU with the
    native communicator up: gg_comm_allreduce on the side stream; otherwise torch.distributed (gloo on the CPU). Returns
    handles with .wait()."""
    if not is_distributed() and not (_native is not None and flat_grad.is_cuda):
        return []
    if _native is not None and flat_grad.is_cuda:
        return [_native.all_reduce_(flat_grad, n_slices)]
    n = flat_grad.numel()
    step = (n + n_slices - 1) // n_slices
    step = (step + 255) // 256 * 256
    works = []
    for s in range(0, n, step):
        works.append(dist.all_reduce(flat_grad[s:min(s + step, n)], op=dist.ReduceOp.SUM, async_op=True))
    return works


def wait_all(works):
    for w in works:
        w.wait()


def broadcast_flat_params(flat_p: torch.Tensor, src: int = 0):
    """make replicas bit-identical at start (DDP does this at wrap time)."""
    if is_distributed():
        dist.broadcast(flat_p, src=src)


class _AllGather(Function):
    """differentiable equal-shard all_gather along dim 0 (reference distributed.py:47-68: backward keeps the
    local slice). Per-rank batches are equal by construction here, so no size exchange / padding is needed."""

    @staticmethod
    def forward(ctx, x):
        ws = dist.get_world_size()
        ctx.b = x.shape[0]
        if _native is not None and x.is_cuda:
            return _native.all_gather(x)
        out = torch.empty((ws * x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        r = dist.get_rank()
        return g[r * ctx.b:(r + 1) * ctx.b]


def all_gather(x, dim=0, sizes=None):
    """reference signature `all_gather(t, dim, sizes) -> (gathered, sizes)`."""
    assert dim == 0
    if not is_distributed():
        return x, None
    out = _AllGather.apply(x)
    return out, torch.full((dist.get_world_size(),), x.shape[0], dtype=torch.long)
