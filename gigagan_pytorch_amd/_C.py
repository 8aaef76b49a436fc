"""ctypes binding of libgigagan_amd.so — the C ABI declared in include/gigagan_amd.h.

The product path loads exactly one library: the hipcc-built gfx950 shared object that sits next to this
file (built by `__graft_entry__.build()` / `make hip`). If it is missing, or a tensor handed to a kernel
is not a CUDA(ROCm) tensor, the call raises — there is no CPU or eager fallback.

`bind(path)` exists so that tests can bind the *same* ABI compiled for the host-side kernel emulator
(tests/emu); that library reports `gg_is_emulator() == 1` and is the only case in which CPU tensors are
accepted.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
DEFAULT_LIB = _HERE / 'libgigagan_amd.so'

ROWK, KROW = 0, 1
ACT_NONE, ACT_LRELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3


class GemmDesc(C.Structure):
    _fields_ = [
        ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('batch', C.c_int32),
        ('A', C.c_void_p), ('a_batch_stride', C.c_int64), ('lda', C.c_int32), ('a_layout', C.c_int32),
        ('a_conv', C.c_int32),
        ('B', C.c_void_p), ('b_batch_stride', C.c_int64), ('ldb', C.c_int32), ('b_layout', C.c_int32),
        ('H', C.c_int32), ('W', C.c_int32), ('C', C.c_int32), ('CV', C.c_int32), ('R', C.c_int32),
        ('S', C.c_int32),
        ('in_scale', C.c_void_p),
        ('C_out', C.c_void_p), ('c_batch_stride', C.c_int64), ('ldc', C.c_int32), ('c_is_f32', C.c_int32),
        ('alpha', C.c_float),
        ('bias', C.c_void_p),
        ('out_scale', C.c_void_p), ('rows_per_group', C.c_int32),
        ('noise', C.c_void_p), ('noise_w', C.c_void_p),
        ('act', C.c_int32), ('act_slope', C.c_float),
        ('force_splitk', C.c_int32), ('force_tile', C.c_int32),
        ('conv_stride', C.c_int32), ('conv_pad', C.c_int32), ('bias_scale', C.c_float),
        ('residual', C.c_void_p), ('ldr', C.c_int32), ('res_scale', C.c_float),
        ('d2s', C.c_int32), ('d2s_taps', C.c_int32), ('d2s_c', C.c_int32), ('d2s_oh', C.c_int32),
        ('d2s_ow', C.c_int32),
        ('b_image_stride', C.c_int64),
        ('bank_mix', C.c_void_p),
        ('keep_partials', C.c_int32), ('gelu_mode', C.c_int32),
        ('gelu_aux', C.c_void_p), ('ld_aux', C.c_int32), ('reserved1', C.c_int32),
    ]


class PlanEntry(C.Structure):        # mirrors gg_plan_entry
    FIELDS = ('M', 'N', 'K', 'batch', 'a_layout', 'b_layout', 'a_conv', 'H', 'W', 'C', 'CV', 'R', 'conv_stride', 'conv_pad',
              'c_is_f32', 'd2s', 'epi', 'scaled', 'tile', 'splitk')
    _fields_ = [(f, C.c_int32) for f in FIELDS]


PLAN_TABLE = Path(os.environ.get('GG_PLAN_TABLE') or _HERE / 'plans' / 'gfx950.json')      # (GG_PLAN_TABLE: A/B runs against another tuning cache)


class Library:
    """One loaded libgigagan_amd.so with typed entry points."""

    def __init__(self, path: os.PathLike | str):
        path = Path(path)
        if not path.exists():
            raise RuntimeError(
                f'gigagan_pytorch_amd: native library {path} not found. Build it with '
                f'`python -c "import __graft_entry__ as g; g.build()"` or `make hip`; there is no fallback path.')
        self.path = path
        self.lib = C.CDLL(str(path))
        L = self.lib
        L.gg_version.restype = C.c_int
        L.gg_last_error.restype = C.c_char_p
        L.gg_is_emulator.restype = C.c_int
        L.gg_gemm_workspace_bytes.restype = C.c_size_t
        L.gg_gemm_workspace_bytes.argtypes = [C.POINTER(GemmDesc)]
        L.gg_gemm_bf16.restype = C.c_int
        L.gg_gemm_bf16.argtypes = [C.POINTER(GemmDesc), C.c_void_p, C.c_size_t, C.c_void_p]
        L.gg_gemm_plan.restype = C.c_int
        L.gg_gemm_plan.argtypes = [C.POINTER(GemmDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self._declare_elementwise()
        self.is_emulator = bool(L.gg_is_emulator())
        if L.gg_version() != 12:
            raise RuntimeError(f'gigagan_pytorch_amd: ABI version mismatch in {path}')
        L.gg_gemm_plan_table.restype = C.c_int
        L.gg_gemm_plan_table.argtypes = [C.POINTER(PlanEntry), C.c_int32]
        self.plan_entries = 0
        if not self.is_emulator and PLAN_TABLE.exists() and not os.environ.get('GG_NO_PLAN_TABLE'):
            self.load_plan_table(PLAN_TABLE)

    def load_plan_table(self, path_or_entries):
        """install the tuning cache: measured-best (tile, split-K) per exact problem geometry (tests/gpu_plan_sweep.py)."""
        import json
        entries = path_or_entries
        if not isinstance(entries, (list, tuple)):
            entries = json.loads(Path(path_or_entries).read_text())['entries']
        arr = (PlanEntry * max(len(entries), 1))()
        for i, e in enumerate(entries):
            for f in PlanEntry.FIELDS:
                setattr(arr[i], f, int(e[f]))
        self.check(self.lib.gg_gemm_plan_table(arr, len(entries)), 'gg_gemm_plan_table')
        self.plan_entries = len(entries)
        import sys
        ops = sys.modules.get(__package__ + '.ops')
        if ops is not None:        # planner answers cached on the Python side belong to the previous table
            ops._ff_plan_cache.clear()

    # filled in by _elementwise_signatures (kept separate so the table reads like the header)
    def _declare_elementwise(self):
        from ._signatures import declare
        declare(self.lib)

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.gg_last_error().decode('utf-8', 'replace')
            raise RuntimeError(f'{what} failed (rc={rc}): {msg}')

    def stream(self, t: torch.Tensor) -> int:
        if self.is_emulator:
            return 0
        return torch.cuda.current_stream(t.device).cuda_stream

    def require(self, *tensors):
        """Every tensor handed to a kernel must live on the GPU (or on the CPU for the emulator build)."""
        for t in tensors:
            if t is None:
                continue
            if self.is_emulator:
                if t.device.type != 'cpu':
                    raise RuntimeError('emulator library bound: tensors must be CPU tensors')
            elif t.device.type != 'cuda':
                raise RuntimeError(
                    'gigagan_pytorch_amd kernels need CUDA/ROCm tensors; got a tensor on '
                    f'{t.device}. There is no CPU fallback in the product path.')


_bound: Library | None = None


def bind(path: os.PathLike | str) -> Library:
    """Bind a specific build of the C ABI (tests use this for the emulator build)."""
    global _bound
    _bound = Library(path)
    return _bound


def lib() -> Library:
    """The bound library; binds the in-tree gfx950 build on first use and fails loudly if it is absent."""
    global _bound
    if _bound is None:
        _bound = Library(DEFAULT_LIB)
    return _bound


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()
