"""Raw (non-autograd) launches of the C-ABI kernels on torch tensors. Layout glue only — no arithmetic."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _C
from ._C import GemmDesc, ROWK, KROW, ptr

_ACTS = {None: 0, 'none': 0, 'lrelu': 1, 'gelu': 2, 'silu': 3}
def _workspace(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    """split-K scratch for ONE launch, owned by the caller side (PyTorch's stream-ordered caching allocator). It is
    deliberately not cached across launches: a hipGraph bakes the pointer in, and a cached buffer that is later
    re-allocated (a bigger request while another step kind is being captured) would leave earlier graphs writing
    to freed memory. Inside a capture the allocation comes from that graph's private pool."""
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=like.device)


class GemmProfiler:
    """optional per-launch HIP-event timing of the contraction kernel (bench.py's roofline leg). Events are
    recorded on the stream the kernel is launched on (torch's current stream) and only read after a sync."""

    def __init__(self):
        self.records = []   # (key, flops, start_event, end_event)
        self.shapes = []    # same, keyed by kernel + problem shape

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, e0, e1 in self.records:
            a = agg.setdefault(key, dict(launches=0, ms=0., flops=0.))
            a['launches'] += 1
            a['ms'] += e0.elapsed_time(e1)
            a['flops'] += flops
        return agg

    def shape_summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, e0, e1 in self.shapes:
            a = agg.setdefault(key, dict(launches=0, ms=0., flops=0.))
            a['launches'] += 1
            a['ms'] += e0.elapsed_time(e1)
            a['flops'] += flops
        return agg


class LaunchProfiler:
    """context manager: HIP events (on the launch stream) around EVERY kernel launch that goes through the C ABI while it is
    active - contraction, coefficient, modulation, mixing, resampling ... - so that an op's time is the time of everything it
    launches. Implemented as a proxy in front of the ctypes library object; eager execution only."""
    _NOT_LAUNCHES = ('gg_gemm_plan', 'gg_gemm_workspace_bytes', 'gg_last_error', 'gg_version', 'gg_is_emulator',
                     'gg_bias_act_bwd_partials', 'gg_rmsnorm_blocks', 'gg_gemm_plan_table', 'gg_comm_', 'gg_graph_')

    def __init__(self):
        self.records = []       # (entry point, start event, end event)

    def __enter__(self):
        L = _C.lib()
        self._L, self._real = L, L.lib
        prof, real = self, L.lib

        class Proxy:
            def __getattr__(self, name):
                fn = getattr(real, name)
                if not name.startswith('gg_') or any(name.startswith(x) for x in LaunchProfiler._NOT_LAUNCHES):
                    return fn

                def timed(*args):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = fn(*args)
                    e1.record()
                    prof.records.append((name, e0, e1))
                    return rc
                return timed
        L.lib = Proxy()
        return self

    def __exit__(self, *exc):
        self._L.lib = self._real
        return False


profiler: GemmProfiler | None = None
plan_log: list | None = None    # when a list: (tile, splitk) of every launch is appended (tuning scripts)
desc_log: list | None = None    # when a list: a raw copy of every launch descriptor is appended (tests/gpu_plan_sweep.py)


def _kernel_key(d: GemmDesc, L) -> str:
    tile, sk = C.c_int32(0), C.c_int32(0)
    L.lib.gg_gemm_plan(C.byref(d), C.byref(tile), C.byref(sk))
    if tile.value == 9:
        name = f'gg_dconv_kernel<C={d.C},TN={1 if d.N <= 32 else 2}>'
    elif tile.value == 10:
        name = 'gg_wgrad9_kernel'
    elif tile.value == 15:
        name = 'gg_pgemm_kernel'
    elif tile.value == 14:
        name = f'gg_sfwd_kernel<C={d.C}>'
    elif tile.value == 13:
        name = 'gg_wgrads_kernel'
    elif tile.value == 11:
        name = 'gg_lrconv_kernel'
    elif tile.value in (7, 8, 12):
        name = f'gg_conv3_kernel<{256 if tile.value == 7 else (64 if tile.value == 12 or d.N <= 64 else 128)}>'
    elif tile.value >= 4:
        bm, bn = {4: (256, 256), 5: (256, 128), 6: (128, 128)}[tile.value]
        name = (f'gg_gemm2_kernel<{bm},{bn},2,4,A_KROW={int(d.a_layout == KROW)},B_KROW={int(d.b_layout == KROW)},'
                f'A_CONV={int(bool(d.a_conv))}>')
    else:
        bn = {1: 128, 2: 64, 3: 32}[tile.value]
        wm, wn = (4, 1) if tile.value == 3 else (2, 2)
        name = (f'gg_gemm_kernel<128,{bn},{wm},{wn},A_KROW={int(d.a_layout == KROW)},B_KROW={int(d.b_layout == KROW)},'
                f'A_CONV={int(bool(d.a_conv))}>')
    return name + ('+splitk' if sk.value > 1 else ''), sk.value


def _run_gemm(d: GemmDesc, like: torch.Tensor):
    """launches the contraction (the split-K workspace, if any, comes from the caching allocator for this one launch)."""
    L = _C.lib()
    need = L.lib.gg_gemm_workspace_bytes(C.byref(d))
    ws = _workspace(need, like) if need else None
    if desc_log is not None:
        desc_log.append(bytes(d))
    if plan_log is not None:
        tile, sk = C.c_int32(0), C.c_int32(0)
        L.lib.gg_gemm_plan(C.byref(d), C.byref(tile), C.byref(sk))
        plan_log.append((tile.value, sk.value))
    if profiler is not None and not L.is_emulator:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.lib.gg_gemm_bf16(C.byref(d), ptr(ws), need, L.stream(like))
        e1.record()
        key, sk = _kernel_key(d, L)
        profiler.records.append((key, 2. * d.M * d.N * d.K * d.batch, e0, e1))
        profiler.shapes.append((f'{key} M={d.M} N={d.N} K={d.K} b={d.batch} sk={sk} sc={int(bool(d.in_scale))}',
                                2. * d.M * d.N * d.K * d.batch, e0, e1))
    else:
        rc = L.lib.gg_gemm_bf16(C.byref(d), ptr(ws), need, L.stream(like))
    L.check(rc, 'gg_gemm_bf16')
    return ws


def _epilogue(d: GemmDesc, alpha, bias, out_scale, rows_per_group, noise, noise_w, act, act_slope, keep,
              bias_scale=1.0, residual=None, res_scale=1.0):
    d.alpha = float(alpha)
    d.bias_scale = float(bias_scale)
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.is_contiguous()
        keep.append(residual)
        d.residual, d.ldr, d.res_scale = ptr(residual), residual.shape[-1], float(res_scale)
    for name, t in (('bias', bias), ('out_scale', out_scale), ('noise', noise), ('noise_w', noise_w)):
        if t is not None:
            assert t.dtype == torch.float32 and t.is_contiguous(), name
            keep.append(t)
            setattr(d, name, ptr(t))
    d.rows_per_group = int(rows_per_group or 0)
    d.act = _ACTS[act]
    d.act_slope = float(act_slope)


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a=False, trans_b=True, out_dtype=torch.bfloat16,
         alpha=1.0, bias=None, act=None, act_slope=0.2, out_scale=None, rows_per_group=0,
         k_valid=None, m_valid=None, n_valid=None, out=None, force_splitk=0, force_tile=0, bias_scale=1.0):
    """C[b] = act(alpha * op(A[b]) @ op(B[b])^T ...) for bf16 operands of shape ([batch,] rows, cols).

    `trans_a=False`: A is stored (M, K) (k contiguous) -> ROWK; `trans_a=True`: stored (K, M) -> KROW.
    `trans_b=True`: B is stored (N, K) -> ROWK (the nn.Linear weight layout); False: stored (K, N) -> KROW.
    Row pitches must be multiples of 8 elements; `*_valid` give logical extents smaller than the storage.
    """
    L = _C.lib()
    L.require(a, b, bias, out_scale)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    if a.dim() == 2:
        a = a.unsqueeze(0)
    if b.dim() == 2:
        b = b.unsqueeze(0)
    assert a.stride(-1) == 1 and b.stride(-1) == 1
    batch = max(a.shape[0], b.shape[0])
    M = m_valid if m_valid is not None else (a.shape[2] if trans_a else a.shape[1])
    Ka = a.shape[1] if trans_a else a.shape[2]
    N = n_valid if n_valid is not None else (b.shape[1] if trans_b else b.shape[2])
    Kb = b.shape[2] if trans_b else b.shape[1]
    K = k_valid if k_valid is not None else Ka
    assert min(Ka, Kb) >= K, (Ka, Kb, K)
    if out is None:
        out = torch.empty((batch, M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    keep = [a, b, out]
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.A, d.lda, d.a_layout = ptr(a), a.stride(1), (KROW if trans_a else ROWK)
    d.a_batch_stride = a.stride(0) if a.shape[0] > 1 else 0
    d.B, d.ldb, d.b_layout = ptr(b), b.stride(1), (ROWK if trans_b else KROW)
    d.b_batch_stride = b.stride(0) if b.shape[0] > 1 else 0
    d.C_out, d.ldc, d.c_is_f32 = ptr(out), out.stride(-2), int(out.dtype == torch.float32)
    d.c_batch_stride = out.stride(0) if out.dim() == 3 else 0
    d.force_splitk, d.force_tile = force_splitk, force_tile
    _epilogue(d, alpha, bias, out_scale, rows_per_group, None, None, act, act_slope, keep, bias_scale=bias_scale)
    _run_gemm(d, a)
    return out


def _conv_out(size: int, ksize: int, stride: int, pad: int) -> int:
    return (size + 2 * pad - ksize) // stride + 1


def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, *, ksize: int, stride: int = 1, pad: int | None = None,
                cv: int | None = None, in_scale=None, out_dtype=torch.bfloat16, alpha=1.0, bias=None, bias_scale=1.0,
                act=None, act_slope=0.2, out_scale=None, noise=None, noise_w=None, residual=None, res_scale=1.0,
                force_splitk=0, force_tile=0, per_image_weights=False, bank_mix=None, gelu_aux=None, gelu_mode=0, plan_only=False):
    """Convolution of an NHWC bf16 activation x (n, H, W, C) with weights w (Cout, ksize*ksize*CV) bf16 laid out
    [co][kh][kw][cv]; window stride `stride`, zero padding `pad` (default: 'same', ksize//2); returns
    (n, OH, OW, Cout) = act(alpha*conv*out_scale + bias*bias_scale + noise) + residual*res_scale. `bank_mix` (n, CV // C) fp32: the
    banks stacked along cv are mixed per image while they are staged (w_img = sum_j bank_mix[img, j] * W_j) and `in_scale` is
    (n, C) - 16x16 images, two banks (gg_lrconv MIX)."""
    L = _C.lib()
    L.require(x, w, in_scale, bias, out_scale, noise, noise_w, residual, bank_mix)
    # (weight rows may carry a pitch larger than their length: `ldb` = w.stride(-2), a multiple of 8 elements)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_contiguous() and w.stride(-1) == 1 and w.stride(-2) % 8 == 0
    n, H, Wd, Cc = x.shape
    cv = cv or Cc
    pad = ksize // 2 if pad is None else pad
    OH, OW = _conv_out(H, ksize, stride, pad), _conv_out(Wd, ksize, stride, pad)
    if per_image_weights:        # w (n, Cout, ksize*ksize*CV): image i is convolved with w[i] (per-sample weights)
        assert w.dim() == 3 and w.shape[0] == n and stride == 1
        cout, wrow = w.shape[1], w.shape[2]
    else:
        cout, wrow = w.shape[0], w.shape[1]
    assert wrow == ksize * ksize * cv, (w.shape, ksize, cv)
    out = x if plan_only else torch.empty((n, OH, OW, cout), dtype=out_dtype, device=x.device)      # (plan_only: a placeholder pointer)
    keep = [x, w, out]
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = n * OH * OW, cout, ksize * ksize * cv, 1
    d.A, d.a_layout, d.a_conv = ptr(x), ROWK, 1
    d.B, d.ldb, d.b_layout = ptr(w), w.stride(-2), ROWK
    if per_image_weights:
        d.b_image_stride = w.stride(0)
    d.H, d.W, d.C, d.CV, d.R, d.S = H, Wd, Cc, cv, ksize, ksize
    d.conv_stride, d.conv_pad = stride, pad
    if in_scale is not None:
        assert in_scale.dtype == torch.float32 and in_scale.shape == (n, Cc if bank_mix is not None else cv) and in_scale.is_contiguous()
        d.in_scale = ptr(in_scale)
        keep.append(in_scale)
    if bank_mix is not None:
        assert bank_mix.dtype == torch.float32 and bank_mix.shape == (n, cv // Cc) and bank_mix.is_contiguous() and in_scale is not None
        d.bank_mix = ptr(bank_mix)
        keep.append(bank_mix)
    d.C_out, d.ldc, d.c_is_f32 = ptr(out), cout, int(out_dtype == torch.float32)
    d.force_splitk, d.force_tile = force_splitk, force_tile
    if gelu_mode and not plan_only:       # GELU fused around a 1x1 pair (gg_gemm2's staged epilogue): 1 = gelu_aux receives the pre-activation, 2 = holds it
        assert gelu_aux is not None and gelu_aux.dtype == torch.bfloat16 and gelu_aux.shape == out.shape and gelu_aux.is_contiguous()
        L.require(gelu_aux)
        d.gelu_aux, d.gelu_mode, d.ld_aux = ptr(gelu_aux), int(gelu_mode), cout
        keep.append(gelu_aux)
    if residual is not None and not plan_only:
        assert residual.shape == out.shape
    _epilogue(d, alpha, bias, out_scale, OH * OW if out_scale is not None else 0, noise, noise_w, act,
              act_slope, keep, bias_scale=bias_scale, residual=residual, res_scale=res_scale)
    if plan_only:           # (tile, split-K) the launch WOULD take: same descriptor, same planner call as the launch itself
        tile, sk = C.c_int32(0), C.c_int32(0)
        L.check(L.lib.gg_gemm_plan(C.byref(d), C.byref(tile), C.byref(sk)), 'gg_gemm_plan')
        return tile.value, sk.value
    _run_gemm(d, x)
    return out


FOLD_MAX_SLICES = 16      # split-K weight gradients of up to this many slices hand their slices to the queued finish (no reduce launch)


def conv2d_wgrad_nhwc(x: torch.Tensor, dy: torch.Tensor, *, ksize: int, stride: int = 1, pad: int | None = None,
                      cv: int | None = None, in_scale=None, force_splitk=0, force_tile=0, keep_slices=False):
    """Weight gradient of conv2d_nhwc: returns fp32 (ksize*ksize*CV, Cout) = sum over output pixels of
    gather(x)[pixel][(tap, cv)] * dy[pixel][co]. `keep_slices`: returns (tensor, nsplit) instead - when the launch is split over
    k-slices, the tensor is the slice stack (nsplit, ksize*ksize*CV, Cout) and NO reduction was launched (the caller's finish pass
    sums them: FinishQueue.add_wgrad(nsplit=)); otherwise the result and 1."""
    L = _C.lib()
    L.require(x, dy, in_scale)
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.is_contiguous() and dy.is_contiguous()
    n, H, Wd, Cc = x.shape
    cv = cv or Cc
    pad = ksize // 2 if pad is None else pad
    OH, OW = _conv_out(H, ksize, stride, pad), _conv_out(Wd, ksize, stride, pad)
    cout = dy.shape[-1]
    assert tuple(dy.shape[:3]) == (n, OH, OW), (dy.shape, (n, OH, OW))
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = ksize * ksize * cv, cout, n * OH * OW, 1
    d.A, d.a_layout, d.a_conv = ptr(x), KROW, 1
    d.B, d.ldb, d.b_layout = ptr(dy), cout, KROW
    d.H, d.W, d.C, d.CV, d.R, d.S = H, Wd, Cc, cv, ksize, ksize
    d.conv_stride, d.conv_pad = stride, pad
    if in_scale is not None:
        assert in_scale.dtype == torch.float32 and in_scale.shape == (n, cv) and in_scale.is_contiguous()
        d.in_scale = ptr(in_scale)
    d.ldc, d.c_is_f32 = cout, 1
    d.alpha = 1.0
    d.bias_scale = 1.0
    d.force_splitk, d.force_tile = force_splitk, force_tile
    nsplit = 1
    if keep_slices:
        tile, sk = C.c_int32(0), C.c_int32(0)
        d.C_out = ptr(x)            # (a placeholder that passes validation: the planner does not look at it)
        L.check(L.lib.gg_gemm_plan(C.byref(d), C.byref(tile), C.byref(sk)), 'gg_gemm_plan')
        if sk.value > 1:       # (more than FOLD_MAX_SLICES slices: the queue folds the stack with its batched reduce before the finish)
            nsplit, d.keep_partials = sk.value, 1
    out = torch.empty((1, 1) if nsplit > 1 else (ksize * ksize * cv, cout), dtype=torch.float32, device=x.device)
    d.C_out = ptr(out)
    ws = _run_gemm(d, x)
    if keep_slices:
        return (ws[:nsplit * d.M * d.N * 4].view(torch.float32).view(nsplit, d.M, d.N), nsplit) if nsplit > 1 else (out, 1)
    return out


def conv2d_dgrad_d2s(dy: torch.Tensor, w: torch.Tensor, *, cell: int, taps: int, alpha=1.0, force_splitk=0,
                     force_tile=0):
    """Data gradient of a stride-`cell` convolution whose windows do not overlap (taps <= cell: the stride-2 1x1
    residual conv and space-to-depth + 1x1): dy (n, oh, ow, Cout) bf16, w (Cout, taps*taps*C) bf16 laid out
    [co][ty][tx][c]; returns dx (n, oh*cell, ow*cell, C) bf16 with dx[pixel(oh*cell+ty, ow*cell+tx)][c] =
    alpha * sum_co dy[oh, ow][co] * w[co][(ty, tx, c)] and zeros at pixels no tap reaches."""
    L = _C.lib()
    L.require(dy, w)
    assert dy.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and dy.is_contiguous() and w.is_contiguous()
    n, oh, ow, cout = dy.shape
    assert w.shape[0] == cout and w.shape[1] % (taps * taps) == 0
    c = w.shape[1] // (taps * taps)
    alloc = torch.empty if taps == cell else torch.zeros
    out = alloc((n, oh * cell, ow * cell, c), dtype=torch.bfloat16, device=dy.device)
    d = GemmDesc()
    d.M, d.N, d.K, d.batch = n * oh * ow, taps * taps * c, cout, 1
    d.A, d.lda, d.a_layout = ptr(dy), cout, ROWK
    d.B, d.ldb, d.b_layout = ptr(w), w.stride(0), KROW
    d.C_out, d.ldc, d.c_is_f32 = ptr(out), taps * taps * c, 0
    d.alpha, d.bias_scale = float(alpha), 1.0
    d.d2s, d.d2s_taps, d.d2s_c, d.d2s_oh, d.d2s_ow = cell, taps, c, oh, ow
    d.force_splitk, d.force_tile = force_splitk, force_tile
    _run_gemm(d, dy)
    return out


# --------------------------------------------------------------------------------------------------
# separable banded resampling
# --------------------------------------------------------------------------------------------------

def _up2_matrix(L: int) -> torch.Tensor:
    """1-D x2 bilinear upsampling, align_corners=False (nn.Upsample, gp.py:259): (2L, L)."""
    m = torch.zeros(2 * L, L, dtype=torch.float64)
    for i in range(L):
        m[2 * i, max(i - 1, 0)] += 0.25
        m[2 * i, i] += 0.75
        m[2 * i + 1, i] += 0.75
        m[2 * i + 1, min(i + 1, L - 1)] += 0.25
    return m


def _blur_matrix(L: int) -> torch.Tensor:
    """1-D [1,2,1]/4 blur with reflect padding (kornia filter2d default border, gp.py:255): (L, L)."""
    m = torch.zeros(L, L, dtype=torch.float64)
    for j in range(L):
        lo = j - 1 if j - 1 >= 0 else 1
        hi = j + 1 if j + 1 < L else L - 2
        m[j, lo] += 0.25
        m[j, j] += 0.5
        m[j, hi] += 0.25
    return m


def _bilinear_matrix(Lin: int, Lout: int) -> torch.Tensor:
    """1-D F.interpolate(mode='bilinear', align_corners=False, antialias=False): (Lout, Lin)."""
    m = torch.zeros(Lout, Lin, dtype=torch.float64)
    scale = Lin / Lout
    for j in range(Lout):
        src = max((j + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(math.floor(src)), Lin - 1)
        i1 = min(i0 + 1, Lin - 1)
        lam = src - i0
        m[j, i0] += 1.0 - lam
        m[j, i1] += lam
    return m


def _nearest_matrix(Lin: int, Lout: int) -> torch.Tensor:
    """1-D F.interpolate(mode='nearest'): src = floor(j * Lin / Lout)."""
    m = torch.zeros(Lout, Lin, dtype=torch.float64)
    scale = Lin / Lout
    for j in range(Lout):
        m[j, min(int(math.floor(j * scale)), Lin - 1)] = 1.0
    return m


def _band_tables(m: torch.Tensor):
    """dense (out, in) matrix -> (taps, idx0 int32[out], weights fp32[out][taps])."""
    nz = m != 0
    out_len, in_len = m.shape
    first = torch.where(nz.any(1), nz.float().argmax(1), torch.zeros(out_len, dtype=torch.long))
    last = torch.where(nz.any(1), in_len - 1 - nz.flip(1).float().argmax(1), first)
    taps = int((last - first).max().item()) + 1
    idx = first[:, None] + torch.arange(taps)[None, :]
    w = torch.where(idx < in_len, m.gather(1, idx.clamp(max=in_len - 1)), torch.zeros((), dtype=m.dtype))
    return taps, first.to(torch.int32), w.to(torch.float32).contiguous()


class ResampleSpec:
    """Host-built tap tables for one 2-D separable resampling operator (and, lazily, its transpose)."""

    _cache: dict = {}

    def __init__(self, my: torch.Tensor, mx: torch.Tensor, key):
        self.my, self.mx, self.key = my, mx, key
        self.oh, self.ih = my.shape
        self.ow, self.iw = mx.shape
        self.ty, self.iy0, self.wy = _band_tables(my)
        self.tx, self.ix0, self.wx = _band_tables(mx)
        self._dev: dict = {}
        self._t = None

    def tables(self, device):
        k = (device.type, device.index)
        if k not in self._dev:
            self._dev[k] = tuple(t.to(device) for t in (self.iy0, self.ix0, self.wy, self.wx))
        return self._dev[k]

    def transposed(self) -> 'ResampleSpec':
        if self._t is None:
            self._t = ResampleSpec._get(('T',) + self.key, lambda: (self.my.t().contiguous(), self.mx.t().contiguous()))
            self._t._t = self
        return self._t

    @classmethod
    def _get(cls, key, build):
        if key not in cls._cache:
            my, mx = build()
            cls._cache[key] = ResampleSpec(my, mx, key)
        return cls._cache[key]

    @classmethod
    def upsample_blur(cls, H, W):
        return cls._get(('upblur', H, W), lambda: (_blur_matrix(2 * H) @ _up2_matrix(H), _blur_matrix(2 * W) @ _up2_matrix(W)))

    @classmethod
    def blur(cls, H, W):
        return cls._get(('blur', H, W), lambda: (_blur_matrix(H), _blur_matrix(W)))

    @classmethod
    def bilinear(cls, H, W, OH, OW):
        return cls._get(('bilinear', H, W, OH, OW), lambda: (_bilinear_matrix(H, OH), _bilinear_matrix(W, OW)))

    @classmethod
    def nearest(cls, H, W, OH, OW):
        return cls._get(('nearest', H, W, OH, OW), lambda: (_nearest_matrix(H, OH), _nearest_matrix(W, OW)))


def resample_nhwc(x: torch.Tensor, spec: ResampleSpec) -> torch.Tensor:
    L = _C.lib()
    L.require(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    n, H, W, Cc = x.shape
    assert (H, W) == (spec.ih, spec.iw), ((H, W), (spec.ih, spec.iw))
    iy0, ix0, wy, wx = spec.tables(x.device)
    out = torch.empty((n, spec.oh, spec.ow, Cc), dtype=x.dtype, device=x.device)
    rc = L.lib.gg_resample_nhwc_bf16(ptr(x), ptr(out), n, H, W, spec.oh, spec.ow, Cc, spec.ty, spec.tx,
                                     ptr(iy0), ptr(ix0), ptr(wy), ptr(wx), L.stream(x))
    L.check(rc, 'gg_resample_nhwc_bf16')
    return out


# --------------------------------------------------------------------------------------------------
# attention softmax, bias / activation backward
# --------------------------------------------------------------------------------------------------

def softmax_fwd(x: torch.Tensor, bias, alpha: float, n_valid: int) -> torch.Tensor:
    """x fp32 (batch, rows, ld) -> bf16 softmax(alpha*x + bias[batch]) over the first n_valid columns."""
    L = _C.lib()
    L.require(x, bias)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    nb, n, ld = x.shape
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.shape == (nb, ld) and bias.is_contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    rc = L.lib.gg_softmax_fwd(ptr(x), ptr(out), ptr(bias), nb * n, n, n_valid, ld, alpha, L.stream(x))
    L.check(rc, 'gg_softmax_fwd')
    return out


def softmax_bwd(S: torch.Tensor, dS: torch.Tensor, alpha: float, n_valid: int, want_dbias: bool):
    """returns (dx bf16 = alpha*u, dbias fp32 (batch, ld) or None) with u = S*(dS - rowsum(S*dS))."""
    L = _C.lib()
    L.require(S, dS)
    assert S.dtype == torch.bfloat16 and dS.dtype == torch.bfloat16 and S.is_contiguous() and dS.is_contiguous()
    nb, n, ld = S.shape
    dx = torch.empty_like(S)
    dbias = torch.zeros((nb, ld), dtype=torch.float32, device=S.device) if want_dbias else None
    rc = L.lib.gg_softmax_bwd(ptr(S), ptr(dS), ptr(dx), ptr(dbias), nb * n, n, n_valid, ld, alpha, L.stream(S))
    L.check(rc, 'gg_softmax_bwd')
    return dx, dbias


def softmax_bwd2(S: torch.Tensor, dS: torch.Tensor, g_dx, g_dbias, alpha: float, n_valid: int):
    """second-order softmax pass: returns (g_S, g_dS) bf16 for incoming gradients g_dx (bf16) / g_dbias (fp32)."""
    L = _C.lib()
    L.require(S, dS, g_dx, g_dbias)
    assert S.dtype == torch.bfloat16 and dS.dtype == torch.bfloat16 and S.is_contiguous() and dS.is_contiguous()
    nb, n, ld = S.shape
    if g_dx is not None:
        assert g_dx.dtype == torch.bfloat16 and g_dx.is_contiguous() and g_dx.shape == S.shape
    if g_dbias is not None:
        assert g_dbias.dtype == torch.float32 and g_dbias.is_contiguous() and g_dbias.shape == (nb, ld)
    g_S, g_dS = torch.empty_like(S), torch.empty_like(S)
    rc = L.lib.gg_softmax_bwd2(ptr(S), ptr(dS), ptr(g_dx), ptr(g_dbias), ptr(g_S), ptr(g_dS), nb * n, n, n_valid, ld, alpha,
                               L.stream(S))
    L.check(rc, 'gg_softmax_bwd2')
    return g_S, g_dS


def gelu(x: torch.Tensor, dy: torch.Tensor | None = None, g: torch.Tensor | None = None):
    """exact GELU passes over contiguous bf16 tensors of equal shape (numel % 8 == 0): gelu(x) | dy * gelu'(x) (with dy) |
    (g * gelu'(x), g * dy * gelu''(x)) (with dy and g)."""
    L = _C.lib()
    L.require(x, dy, g)
    mode = 0 if dy is None else (1 if g is None else 2)
    for t in (x, dy, g):
        assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous() and t.shape == x.shape)
    out0 = torch.empty_like(x)
    out1 = torch.empty_like(x) if mode == 2 else None
    rc = L.lib.gg_gelu(ptr(x), ptr(dy), ptr(g), ptr(out0), ptr(out1), x.numel(), mode, L.stream(x))
    L.check(rc, 'gg_gelu')
    return out0 if mode < 2 else (out0, out1)


def colsum_finish(part: torch.Tensor, n: int, alpha: float = 1.0, out: torch.Tensor | None = None,
                  accumulate: bool = False) -> torch.Tensor:
    """part (P, C) fp32 -> (n,) fp32 = alpha * column sums (first n columns); with `out` (+ accumulate) added in place."""
    L = _C.lib()
    L.require(part, out)
    assert part.dtype == torch.float32 and part.dim() == 2 and part.is_contiguous() and n <= part.shape[1]
    if out is None:
        assert not accumulate
        out = torch.zeros(n, dtype=torch.float32, device=part.device)
    elif not accumulate:
        out.zero_()
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == n
    rc = L.lib.gg_colsum_finish(ptr(part), ptr(out), part.shape[0], part.shape[1], n, float(alpha), L.stream(part))
    L.check(rc, 'gg_colsum_finish')
    return out


class GraphApiMissing(RuntimeError):
    """this torch has no CUDAGraph(keep_graph=True) / raw_cuda_graph() / instantiate() (older than 2.8): the memset repair a captured
    step needs cannot be applied, so steps run eagerly."""


def capture_graph(fn, capture_error_mode: str = 'global'):
    """capture `fn()` into a hipGraph, repair it (gg_graph_patch_memsets: this HIP runtime replays captured memset nodes with a
    corrupted value, which breaks every PyTorch split reduction inside the graph from the second replay on), instantiate it.
    Returns (graph, fn's outputs, number of memset nodes repaired). The caller has warmed `fn` up on a side stream."""
    L = _C.lib()
    if not (hasattr(torch.cuda.CUDAGraph, 'raw_cuda_graph') and hasattr(torch.cuda.CUDAGraph, 'instantiate')):
        raise GraphApiMissing('torch.cuda.CUDAGraph lacks raw_cuda_graph() / instantiate() (torch < 2.8)')
    try:
        graph = torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError as e:      # (the constructor of an older torch: the ONLY TypeError that means "no graph API" - fn's own errors propagate)
        raise GraphApiMissing(f'torch.cuda.CUDAGraph(keep_graph=True) is not available: {e}') from e
    with torch.cuda.graph(graph, capture_error_mode=capture_error_mode):
        outs = fn()
    n = C.c_int32(0)
    L.check(L.lib.gg_graph_patch_memsets(C.c_void_p(graph.raw_cuda_graph()), C.byref(n)), 'gg_graph_patch_memsets')
    graph.instantiate()
    return graph, outs, int(n.value)


_hinge_scratch: dict = {}


def hinge(x: torch.Tensor, nb: int, split: int, mode: int, gscale: torch.Tensor | None = None):
    """gg_hinge: x dense (outer, nb, inner...) bf16 / fp32. gscale None: returns the loss (fp32 scalar tensor); else the gradient
    gscale * d loss / d x (x's dtype and shape)."""
    L = _C.lib()
    L.require(x, gscale)
    assert x.is_contiguous() and x.dtype in (torch.bfloat16, torch.float32) and x.dim() >= 2 and x.shape[1] == nb
    inner = x.numel() // (x.shape[0] * nb)
    if gscale is None:
        out = torch.empty((), dtype=torch.float32, device=x.device)
        dx = None
    else:
        assert gscale.dtype == torch.float32 and gscale.numel() == 1
        out = dx = torch.empty_like(x)
    scratch = None
    if gscale is None:
        # ticket + 64 partials. Launches that are ordered among themselves share one scratch (each leaves the ticket at zero): one per
        # (device, stream) for eager launches, one per device for everything captured into hipGraphs (replays are issued on one stream;
        # it is allocated by the first EAGER call - the warm-up run in front of a capture - so that it does not live in a graph's
        # private pool). A capture that was never warmed up zeroes a scratch of its own inside the graph.
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            key = (x.device, L.stream(x))
            scratch = _hinge_scratch.get(key)
            if scratch is None:
                scratch = _hinge_scratch[key] = torch.zeros(65, dtype=torch.float32, device=x.device)
            if x.is_cuda and (x.device, 'capture') not in _hinge_scratch:
                _hinge_scratch[(x.device, 'capture')] = torch.zeros(65, dtype=torch.float32, device=x.device)
        else:
            scratch = _hinge_scratch.get((x.device, 'capture'))
            if scratch is None:
                scratch = torch.zeros(65, dtype=torch.float32, device=x.device)
    rc = L.lib.gg_hinge(ptr(x), ptr(dx), ptr(gscale), ptr(out) if gscale is None else None, ptr(scratch), x.numel(), inner, nb, split,
                        mode, int(x.dtype == torch.float32), L.stream(x))
    L.check(rc, 'gg_hinge')
    return out


class ReduceItem(C.Structure):       # mirrors gg_reduce_item (include/gigagan_amd.h)
    _fields_ = [('src', C.c_void_p), ('n', C.c_int64), ('nsplit', C.c_int32), ('reserved', C.c_int32)]


def reduce_multi(stacks):
    """stacks: [(tensor, n, nsplit)]: fp32 slice stacks (nsplit, n) folded IN PLACE into their slice 0 by one gg_reduce_multi launch."""
    if not stacks:
        return
    L = _C.lib()
    L.require(*[t for t, _, _ in stacks])
    for t, n, ns in stacks:
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n * ns and ns >= 2
    arr = (ReduceItem * len(stacks))(*[ReduceItem(ptr(t), n, ns, 0) for t, n, ns in stacks])
    L.check(L.lib.gg_reduce_multi(C.cast(arr, C.c_void_p), len(stacks), L.stream(stacks[0][0])), 'gg_reduce_multi')


class FinishItem(C.Structure):       # mirrors gg_finish_item (include/gigagan_amd.h)
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('kind', C.c_int32), ('O', C.c_int32), ('I', C.c_int32), ('T', C.c_int32),
                ('C8', C.c_int32), ('O8', C.c_int32), ('accumulate', C.c_int32), ('alpha', C.c_float), ('nsplit', C.c_int32),
                ('reserved0', C.c_int32)]


class FinishQueue:
    """weight-gradient / bias-gradient finishes that write into parameters' .grad (the flat gradient buffer) are collected during a
    backward pass and executed by gg_finish_multi in batches: 221 launches of 4-8 us per step become a handful. `add_*` keeps the
    source tensors alive until the flush; an item whose destination is already queued flushes first (the batch's items run
    concurrently); `notify` callbacks (the in-backward gradient exchange's readiness reports) run after the launch that wrote them."""
    LIMIT = 40                      # GG_FM_MAX: one launch per flush

    def __init__(self):
        self.items, self.keep, self.notify, self.dsts, self.reduces = [], [], [], set(), []

    def _add(self, it, dst, keep, notify):
        if dst.data_ptr() in self.dsts:
            self.flush()
        self.items.append(it)
        self.keep.append(keep)
        self.dsts.add(dst.data_ptr())
        if notify is not None:
            self.notify.append(notify)
        if len(self.items) >= self.LIMIT:
            self.flush()

    def add_wgrad(self, g: torch.Tensor, O: int, I: int, T: int, alpha: float, out: torch.Tensor, notify=None, nsplit: int = 1):
        """g: the weight-gradient GEMM's (T*C8, O8) fp32 output, or with nsplit > 1 its split-K slices (nsplit, T*C8, O8)."""
        L = _C.lib()
        L.require(g, out)
        rows = g.shape[-2]
        C8 = rows // T
        assert g.dtype == torch.float32 and g.is_contiguous() and rows == T * C8 and (g.dim() == 2 if nsplit == 1 else g.shape[0] == nsplit)
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == O * I * T
        if nsplit > FOLD_MAX_SLICES:      # a deep stack: folded into its slice 0 by the flush's batched reduce, the finish reads that
            self.reduces.append(ReduceItem(ptr(g), g[0].numel(), nsplit, 0))
            nsplit = 1
        self._add(FinishItem(ptr(g), ptr(out), 0, O, I, T, C8, g.shape[-1], 1, float(alpha), nsplit, 0), out, (g, out), notify)

    def add_colsum(self, part: torch.Tensor, n: int, alpha: float, out: torch.Tensor, notify=None):
        L = _C.lib()
        L.require(part, out)
        assert part.dtype == torch.float32 and part.dim() == 2 and part.is_contiguous() and n <= part.shape[1]
        assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == n
        self._add(FinishItem(ptr(part), ptr(out), 1, part.shape[0], part.shape[1], n, 0, 0, 1, float(alpha), 1, 0), out, (part, out), notify)

    def add_axpy(self, src: torch.Tensor, alpha: float, out: torch.Tensor, notify=None):
        """out += alpha * src (fp32, same number of elements: a dense (O, I) linear-layer gradient)."""
        L = _C.lib()
        L.require(src, out)
        assert src.dtype == torch.float32 and src.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous()
        assert src.numel() == out.numel()
        self._add(FinishItem(ptr(src), ptr(out), 2, src.numel(), 0, 0, 0, 0, 1, float(alpha), 1, 0), out, (src, out), notify)

    def clear(self):
        self.items, self.keep, self.notify, self.dsts, self.reduces = [], [], [], set(), []

    def flush(self):
        if not self.items:
            return
        L = _C.lib()
        like = self.keep[0][0]
        rc = 0
        if self.reduces:
            red = (ReduceItem * len(self.reduces))(*self.reduces)
            rc = L.lib.gg_reduce_multi(C.cast(red, C.c_void_p), len(self.reduces), L.stream(like))
        arr = (FinishItem * len(self.items))(*self.items)
        rc2 = L.lib.gg_finish_multi(C.cast(arr, C.c_void_p), len(self.items), L.stream(like)) if rc == 0 else 0
        notify = self.notify
        self.items, self.keep, self.notify, self.dsts, self.reduces = [], [], [], set(), []
        L.check(rc, 'gg_reduce_multi')
        L.check(rc2, 'gg_finish_multi')
        for fn in notify:
            fn()


finish_queue = FinishQueue()


def bias_act_bwd(dy: torch.Tensor, y, want_db: bool, slope: float = 0.2, partials: bool = False):
    """dz = dy * lrelu'(y) (dz is dy itself when y is None) and db = column sums of dz (fp32) in one pass; with
    `partials` the per-workgroup partial sums (P, C) are returned for colsum_finish instead of db."""
    L = _C.lib()
    L.require(dy, y)
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous()
    Cc = dy.shape[-1]
    rows = dy.numel() // Cc
    dz = torch.empty_like(dy) if y is not None else None
    if y is not None:
        assert y.dtype == torch.bfloat16 and y.is_contiguous() and y.shape == dy.shape
    part = None
    if want_db:
        part = torch.empty((L.lib.gg_bias_act_bwd_partials(rows, Cc), Cc), dtype=torch.float32, device=dy.device)
    rc = L.lib.gg_bias_act_bwd(ptr(dy), ptr(y), ptr(dz), ptr(part), rows, Cc, slope, L.stream(dy))
    L.check(rc, 'gg_bias_act_bwd')
    if partials:
        return (dz if dz is not None else dy), part
    return (dz if dz is not None else dy), (colsum_finish(part, Cc) if part is not None else None)


# --------------------------------------------------------------------------------------------------
# the bf16 passes around the adaptive convolution (gg_modconv.h)
# --------------------------------------------------------------------------------------------------

def _chunks(P: int, C: int) -> int:
    """workgroups per image for the per-image reductions: ~1024 16-byte vectors each (four per thread: the low-resolution layers were
    latency-bound on 1-2 workgroups per image), at most 64."""
    return max(1, min(64, (P * (C // 8) + 1023) // 1024))


def modulate(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """x (b, H, W, C) bf16 * s (b, C) fp32 -> bf16."""
    L = _C.lib()
    L.require(x, s)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and s.dtype == torch.float32 and s.is_contiguous()
    b, H, W, Cc = x.shape
    assert s.shape == (b, Cc)
    out = torch.empty_like(x)
    rc = L.lib.gg_modulate_fwd(ptr(x), ptr(s), ptr(out), b, H * W, Cc, L.stream(x))
    L.check(rc, 'gg_modulate_fwd')
    return out


def modulate_bank(x: torch.Tensor, s: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """x (b, H, W, Cin) bf16, s (b, Cin) fp32, a (b, N) fp32 -> (b, H, W, N*Cin) bf16 with out[..., n*Cin + i] = x[..., i] * s[b, i] *
    a[b, n]: the activation pre-scaled for each of the N kernels of a bank, channels laid out like the stacked reduction."""
    L = _C.lib()
    L.require(x, s, a)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    for t in (s, a):
        assert t.dtype == torch.float32 and t.is_contiguous()
    b, H, W, Cin = x.shape
    N = a.shape[1]
    assert s.shape == (b, Cin) and a.shape[0] == b
    out = torch.empty((b, H, W, N * Cin), dtype=torch.bfloat16, device=x.device)
    rc = L.lib.gg_modulate_bank_fwd(ptr(x), ptr(s), ptr(a), ptr(out), b, H * W, Cin, N * Cin, L.stream(x))
    L.check(rc, 'gg_modulate_bank_fwd')
    return out


MODW_MAX_B, MODW_MAX_N, MODW_MAX_W, MODW_MAX_G = 64, 4, 9216, 1536


def modw_eligible(b: int, N: int, I: int, T: int) -> bool:
    return (b <= MODW_MAX_B and N <= MODW_MAX_N and I % 4 == 0 and N * I * T <= MODW_MAX_W
            and (N * (N + 1) // 2) * I <= MODW_MAX_G)


def modw_fwd(w: torch.Tensor, mod: torch.Tensor, kmod, demod: bool, eps: float, Ip: int, Op: int, coef: bool = True,
             wmix: torch.Tensor | None = None, layout: int = 0, xs: torch.Tensor | None = None):
    """one launch per adaptive-conv layer (gg_modfwd.h): returns (s (b, Ip), a (b, N), d (b, Op)) when `coef`, and fills `wmix`
    (per-sample weights, layout 1 = (b, O, T*I) rows / layout 2 = (b, T, I/16, 32, 16)) when given."""
    L = _C.lib()
    L.require(w, mod, kmod, wmix, xs)
    N, O, I = w.shape[:3]
    T = w.shape[3] * w.shape[4]
    b = mod.shape[0]
    assert w.dtype == torch.float32 and w.is_contiguous() and mod.dtype == torch.float32 and mod.stride(1) == 1
    assert mod.shape == (b, I) and (kmod is None or (kmod.shape == (b, N) and kmod.dtype == torch.float32 and kmod.stride(1) == 1))
    s = a = d = None
    if coef:
        s = torch.empty((b, Ip), dtype=torch.float32, device=w.device)
        a = torch.empty((b, N), dtype=torch.float32, device=w.device)
        d = torch.empty((b, Op), dtype=torch.float32, device=w.device)
    if wmix is not None:
        assert wmix.dtype == torch.bfloat16 and wmix.is_contiguous()
    if xs is not None:
        assert xs.dtype == torch.float32 and xs.shape == (b, I) and xs.stride(1) == 1
    rc = L.lib.gg_modw_fwd(ptr(w), ptr(mod), mod.stride(0), ptr(kmod), 0 if kmod is None else kmod.stride(0), ptr(xs),
                           0 if xs is None else xs.stride(0), ptr(s), ptr(a), ptr(d),
                           ptr(wmix), layout, b, N, O, I, T, Ip, Op, int(bool(demod)), float(eps), L.stream(w))
    L.check(rc, 'gg_modw_fwd')
    return s, a, d


class ModwItem(C.Structure):         # mirrors gg_modw_item (include/gigagan_amd.h)
    _fields_ = ([(f, C.c_void_p) for f in ('w', 'mod', 'kmod', 'xs', 'gram', 's', 'a', 'd', 'insc', 'wmix')] +
                [(f, C.c_int32) for f in ('mod_ld', 'kmod_ld', 'xs_ld', 'layout', 'b', 'N', 'O', 'I', 'T', 'Ip', 'Op', 'demod')] +
                [('eps', C.c_float), ('reserved', C.c_int32)])


def modw_multi(layers):
    """gg_modw_multi_fwd: the coefficient / per-sample-weight work of MANY adaptive-conv layers in one launch (per 16 layers).
    `layers`: dicts with w (N, O, I, k, k) fp32, mod (b, I), kmod (b, N) | None, demod, eps, Ip, Op and the wanted outputs:
    coef=True -> s (b, Ip), a (b, N), d (b, Op), insc (b, N*Ip) are allocated and returned; wmix / layout as in modw_fwd. Returns a
    list of dicts(s, a, d, insc) (None where not requested)."""
    L = _C.lib()
    arr = (ModwItem * len(layers))()
    outs, keep = [], []
    for it, ly in zip(arr, layers):
        w, mod, kmod, wmix, gram = ly['w'], ly['mod'], ly.get('kmod'), ly.get('wmix'), ly.get('gram')
        L.require(w, mod, kmod, wmix, gram)
        N, O, I = w.shape[:3]
        T = w.shape[3] * w.shape[4]
        b = mod.shape[0]
        assert w.dtype == torch.float32 and w.is_contiguous() and mod.dtype == torch.float32 and mod.stride(1) == 1
        assert mod.shape == (b, I) and (kmod is None or (kmod.shape == (b, N) and kmod.dtype == torch.float32 and kmod.stride(1) == 1))
        Ip, Op = ly['Ip'], ly['Op']
        o = dict(s=None, a=None, d=None, insc=None)
        if ly.get('coef', True):
            o['s'] = torch.empty((b, Ip), dtype=torch.float32, device=w.device)
            o['a'] = torch.empty((b, N), dtype=torch.float32, device=w.device)
            o['d'] = torch.empty((b, Op), dtype=torch.float32, device=w.device)
            o['insc'] = torch.empty((b, N * Ip), dtype=torch.float32, device=w.device)
        if wmix is not None:
            assert wmix.dtype == torch.bfloat16 and wmix.is_contiguous()
        if gram is not None:
            assert gram.dtype == torch.float32 and gram.is_contiguous() and gram.shape == (N * (N + 1) // 2, O, I)
        it.w, it.mod, it.kmod, it.xs, it.gram = ptr(w), ptr(mod), ptr(kmod), None, ptr(gram)
        it.s, it.a, it.d, it.insc, it.wmix = ptr(o['s']), ptr(o['a']), ptr(o['d']), ptr(o['insc']), ptr(wmix)
        it.mod_ld, it.kmod_ld, it.xs_ld = mod.stride(0), (0 if kmod is None else kmod.stride(0)), 0
        it.layout = int(ly.get('layout', 0))
        it.b, it.N, it.O, it.I, it.T, it.Ip, it.Op = b, N, O, I, T, Ip, Op
        it.demod, it.eps = int(bool(ly.get('demod', True))), float(ly.get('eps', 1e-8))
        outs.append(o)
        keep.append((w, mod, kmod, wmix, gram))
    rc = L.lib.gg_modw_multi_fwd(C.cast(arr, C.c_void_p), len(layers), L.stream(layers[0]['w']))
    L.check(rc, 'gg_modw_multi_fwd')
    return outs


def sconv(x: torch.Tensor, wmix: torch.Tensor, O: int, noise=None, noise_w=None, act=None, slope: float = 0.2, xs=None) -> torch.Tensor:
    """streaming 3x3 convolution with per-image filter banks (gg_sconv_fwd): x (b, H, W, C) bf16, wmix (b, 9, C/16, 32, 16) bf16
    (or (1, ...) shared) -> (b, H, W, O) bf16 = act(conv(x * xs) + noise * noise_w); xs (b, C) fp32 optional."""
    L = _C.lib()
    L.require(x, wmix, noise, noise_w, xs)
    if xs is not None:
        assert xs.dtype == torch.float32 and xs.is_contiguous() and xs.shape == (x.shape[0], x.shape[3])
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and wmix.dtype == torch.bfloat16 and wmix.is_contiguous()
    b, H, W, Cc = x.shape
    assert wmix.shape[1:] == (9, Cc // 16, 32, 16) and wmix.shape[0] in (1, b)
    y = torch.empty((b, H, W, O), dtype=torch.bfloat16, device=x.device)
    w_bs = wmix.stride(0) if wmix.shape[0] > 1 else 0
    rc = L.lib.gg_sconv_fwd(ptr(x), ptr(wmix), w_bs, ptr(y), ptr(noise), ptr(noise_w), ptr(xs), b, H, W, Cc, O,
                            1 if act == 'lrelu' else 0, float(slope), L.stream(x))
    L.check(rc, 'gg_sconv_fwd')
    return y


_spair_ok: dict = {}


def spair_supported(H: int, W: int, C0: int, C1: int, C2: int) -> bool:
    """gg_spair_supported, asked once per geometry and library (a pure function of its arguments)."""
    L = _C.lib()
    key = (id(L), H, W, C0, C1, C2)
    ok = _spair_ok.get(key)
    if ok is None:
        ok = _spair_ok[key] = bool(L.lib.gg_spair_supported(H, W, C0, C1, C2))
    return ok


def spair(x: torch.Tensor, wmix1: torch.Tensor, wmix2: torch.Tensor, C1: int, C2: int, noise1=None, noise_w1=None, noise2=None,
          noise_w2=None, act1=None, act2=None, slope: float = 0.2, xs=None) -> torch.Tensor:
    """two streaming 3x3 convolutions with per-image filter banks in one launch, the intermediate map kept in LDS (gg_spair_fwd):
    y = act2(conv(act1(conv(x * xs, w1) + noise1 * nw1), w2) + noise2 * nw2); x (b, H, W, C0) bf16, wmix1 (b, 9, C0/16, 32, 16),
    wmix2 (b, 9, C1/16, 32, 16) bf16 (or (1, ...) shared) -> (b, H, W, C2) bf16. Bit-identical to sconv(sconv(x)) on the 32x32x16 form
    (W 128; W 256 with C2 > 16), equal to bf16 rounding on the 16x16x32 form (W 256, C2 <= 16)."""
    L = _C.lib()
    L.require(x, wmix1, wmix2, noise1, noise_w1, noise2, noise_w2, xs)
    b, H, W, C0 = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    for wm, ci in ((wmix1, C0), (wmix2, C1)):
        assert wm.dtype == torch.bfloat16 and wm.is_contiguous() and wm.shape[1:] == (9, ci // 16, 32, 16) and wm.shape[0] in (1, b)
    if xs is not None:
        assert xs.dtype == torch.float32 and xs.is_contiguous() and xs.shape == (b, C0)
    y = torch.empty((b, H, W, C2), dtype=torch.bfloat16, device=x.device)
    rc = L.lib.gg_spair_fwd(ptr(x), ptr(wmix1), wmix1.stride(0) if wmix1.shape[0] > 1 else 0, ptr(wmix2),
                            wmix2.stride(0) if wmix2.shape[0] > 1 else 0, ptr(y), ptr(noise1), ptr(noise_w1), ptr(noise2),
                            ptr(noise_w2), ptr(xs), b, H, W, C0, C1, C2, 1 if act1 == 'lrelu' else 0, 1 if act2 == 'lrelu' else 0,
                            float(slope), L.stream(x))
    L.check(rc, 'gg_spair_fwd')
    return y


class AconvDesc(C.Structure):        # mirrors gg_aconv_desc (include/gigagan_amd.h)
    _fields_ = ([(f, C.c_void_p) for f in ('x', 'wf', 'y', 's', 'xs', 'a', 'd', 'noise', 'noise_w')] +
                [(f, C.c_int32) for f in ('b', 'H', 'W', 'C', 'O', 'NB', 'act')] + [('slope', C.c_float)] +
                [(f, C.c_int32) for f in ('force_tm', 'force_nwn')] + [('next_wf', C.c_void_p)] +
                [(f, C.c_int32) for f in ('next_b', 'next_H', 'next_C', 'next_O', 'next_NB', 'reserved')])


def _aconv_desc(b, H, W, Cc, O, NB, force_tm=0, force_nwn=0):
    d = AconvDesc()
    d.b, d.H, d.W, d.C, d.O, d.NB = b, H, W, Cc, O, NB
    d.force_tm, d.force_nwn = force_tm, force_nwn
    return d


def aconv_plan(b: int, H: int, W: int, Cc: int, O: int, NB: int, force_tm: int = 0, force_nwn: int = 0):
    """(TM, NWN, NWK, LDS bytes, workgroups) gg_aconv_fwd would take for this layer, or None when the layer cannot run on it."""
    L = _C.lib()
    d = _aconv_desc(b, H, W, Cc, O, NB, force_tm, force_nwn)
    d.x = d.wf = d.y = d.s = d.a = 16          # (placeholders that pass the null checks: the planner does not touch them)
    vals = [C.c_int32(0) for _ in range(5)]
    if L.lib.gg_aconv_plan(C.byref(d), *[C.byref(v) for v in vals]) != 0:
        return None
    return tuple(v.value for v in vals)


def aconv(x: torch.Tensor, wf: torch.Tensor, s: torch.Tensor, a, d, O: int, noise=None, noise_w=None, act=None, slope: float = 0.2,
          xs=None, force_tm: int = 0, force_nwn: int = 0, next_bank=None, _dbg: int = 0) -> torch.Tensor:
    """gg_aconv_fwd: the no-grad adaptive 3x3 convolution on a shared bank in fragment order (PackTable.register_frag /
    frag_pack): x (b, H, W, C) bf16, wf (O/32, NB, 9, C/16, 64, 8) bf16, s (b, C), a (b, NB) | None, d (b, O) | None, xs (b, C) | None,
    noise (b*H*W,) with noise_w (O,) fp32 -> (b, H, W, O) bf16. `next_bank` = (wf', b', H') of the NEXT aconv launch on this stream: its
    bank is requested into the L2 by the wavefronts that finish early (a hint; None = no prefetch)."""
    L = _C.lib()
    L.require(x, wf, s, a, d, noise, noise_w, xs)
    b, H, W, Cc = x.shape
    NB = wf.shape[1]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and wf.dtype == torch.bfloat16 and wf.is_contiguous()
    assert tuple(wf.shape) == (O // 32, NB, 9, Cc // 16, 64, 8), (tuple(wf.shape), O, Cc)
    for t, shp in ((s, (b, Cc)), (xs, (b, Cc)), (a, (b, NB)), (d, (b, O))):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shp), (None if t is None else t.shape, shp)
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.numel() == b * H * W
        assert noise_w is not None and noise_w.dtype == torch.float32 and noise_w.is_contiguous() and noise_w.numel() == O
    y = torch.empty((b, H, W, O), dtype=torch.bfloat16, device=x.device)
    dsc = _aconv_desc(b, H, W, Cc, O, NB, force_tm, force_nwn)
    dsc.x, dsc.wf, dsc.y, dsc.s, dsc.xs, dsc.a, dsc.d = ptr(x), ptr(wf), ptr(y), ptr(s), ptr(xs), ptr(a), ptr(d)
    dsc.noise, dsc.noise_w = ptr(noise), ptr(noise_w) if noise is not None else None
    dsc.act, dsc.slope = (1 if act == 'lrelu' else 0), float(slope)
    dsc.reserved = int(_dbg)            # (timing probes: phases switched off, results then meaningless)
    if next_bank is not None:
        nwf, nb_, nh_ = next_bank
        L.require(nwf)
        assert nwf.dtype == torch.bfloat16 and nwf.is_contiguous() and nwf.dim() == 6 and tuple(nwf.shape[4:]) == (64, 8) and nwf.shape[2] == 9
        dsc.next_wf, dsc.next_b, dsc.next_H = ptr(nwf), int(nb_), int(nh_)
        dsc.next_O, dsc.next_NB, dsc.next_C = nwf.shape[0] * 32, nwf.shape[1], nwf.shape[3] * 16
    L.check(L.lib.gg_aconv_fwd(C.byref(dsc), L.stream(x)), 'gg_aconv_fwd')
    return y


def frag_pack(w: torch.Tensor) -> torch.Tensor:
    """(N, O, I, 3, 3) fp32 bank -> the bf16 MFMA-fragment order gg_aconv_fwd reads, (O/32, N, 9, I/16, 64, 8): lane l of a block holds
    output channel l & 31, input channels 16 cb + 8 (l >> 5) + 0..7. Tensor algebra (tests, parameters outside a pack table); the
    trainer's banks are kept in this order by gg_pack_weights (PackTable.register_frag)."""
    N, O, I, kh, kw = w.shape
    assert O % 32 == 0 and I % 16 == 0
    t = w.detach().reshape(N, O // 32, 32, I // 16, 2, 8, kh * kw)             # n, ot, ol, cb, half, e, t
    t = t.permute(1, 0, 6, 3, 4, 2, 5)                                          # ot, n, t, cb, half, ol, e
    return t.reshape(O // 32, N, kh * kw, I // 16, 64, 8).to(torch.bfloat16).contiguous()


def modulate_bwd(g: torch.Tensor, x: torch.Tensor, s: torch.Tensor):
    """returns (dx = g*s bf16, ds (b, C) fp32 = sum over pixels of g*x)."""
    L = _C.lib()
    L.require(g, x, s)
    assert g.dtype == torch.bfloat16 and g.is_contiguous() and x.is_contiguous() and g.shape == x.shape
    b, H, W, Cc = x.shape
    ch = _chunks(H * W, Cc)
    dx = torch.empty_like(x)
    part = torch.empty((b, ch, Cc), dtype=torch.float32, device=x.device)
    rc = L.lib.gg_modulate_bwd(ptr(g), ptr(x), ptr(s), ptr(dx), ptr(part), b, H * W, Cc, ch, L.stream(x))
    L.check(rc, 'gg_modulate_bwd')
    return dx, (part.sum(1) if ch > 1 else part[:, 0])


def modmix_fwd(Y: torch.Tensor, a: torch.Tensor, d, noise, noise_w, O: int, N: int, act):
    """Y (b, H, W, N*Os) bf16 -> y (b, H, W, O) bf16 = act(d * sum_n a_n Y_n + noise_w * noise)."""
    L = _C.lib()
    L.require(Y, a, d, noise, noise_w)
    assert Y.dtype == torch.bfloat16 and Y.is_contiguous()
    b, H, W, tot = Y.shape
    Os = tot // N
    y = torch.empty((b, H, W, O), dtype=torch.bfloat16, device=Y.device)
    rc = L.lib.gg_modmix_fwd(ptr(Y), ptr(a), ptr(d), ptr(noise), ptr(noise_w), ptr(y), b, H * W, O, Os, N,
                             1 if act == 'lrelu' else 0, 0.2, L.stream(Y))
    L.check(rc, 'gg_modmix_fwd')
    return y


def modmix_bwd(dy: torch.Tensor, y, Y: torch.Tensor, a: torch.Tensor, d, noise, O: int, N: int, act):
    """returns (dY like Y, da (b, N) or None, dd (b, O) or None, dnw (O,) or None)."""
    L = _C.lib()
    L.require(dy, y, Y, a, d, noise)
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and Y.is_contiguous()
    b, H, W, tot = Y.shape
    Os = tot // N
    ch = _chunks(H * W, O)
    dev = Y.device
    dY = torch.empty_like(Y) if Os == O else torch.zeros_like(Y)
    da = torch.empty((ch, b, N), dtype=torch.float32, device=dev) if N > 1 else None
    dd = torch.empty((ch, b, O), dtype=torch.float32, device=dev) if d is not None else None
    dnw = torch.empty((ch, b, O), dtype=torch.float32, device=dev) if noise is not None else None
    rc = L.lib.gg_modmix_bwd(ptr(dy), ptr(y), ptr(Y), ptr(a), ptr(d), ptr(noise), ptr(dY), ptr(da), ptr(dd), ptr(dnw),
                             b, H * W, O, Os, N, ch, 1 if act == 'lrelu' else 0, 0.2, L.stream(Y))
    L.check(rc, 'gg_modmix_bwd')
    # the partials are chunk-major slice stacks: ONE gg_reduce_multi launch folds the chunks of all three into their first slices (three
    # torch reductions per layer before); the noise weights' gradient is then summed over the images (a (b, O) table: a stack of b * ch
    # slices of O values would give the fold one workgroup)
    reduce_multi([(t, t[0].numel(), ch) for t in (da, dd, dnw) if t is not None and ch > 1])
    return (dY, None if da is None else da[0], None if dd is None else dd[0], None if dnw is None else dnw[0].sum(0))


# --------------------------------------------------------------------------------------------------
# fused self-attention (gg_attention.h)
# --------------------------------------------------------------------------------------------------

def attn_fwd(q, k, v, k0, v0, heads: int, alpha: float, beta: float):
    """q, k, v (B, n, heads*64) bf16; k0, v0 (heads, 64) bf16 -> (o (B, n, heads*64) bf16, lse (B*heads, n) fp32)."""
    L = _C.lib()
    L.require(q, k, v, k0, v0)
    for t in (q, k, v, k0, v0):
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
    B, n, hd = q.shape
    assert hd == heads * 64 and k.shape == q.shape and v.shape == q.shape and k0.shape == (heads, 64)
    o = torch.empty_like(q)
    lse = torch.empty((B * heads, n), dtype=torch.float32, device=q.device)
    rc = L.lib.gg_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(k0), ptr(v0), ptr(o), ptr(lse), B, n, heads, alpha, beta, L.stream(q))
    L.check(rc, 'gg_attn_fwd')
    return o, lse


def attn_bwd(q, k, v, k0, v0, o, lse, d_o, heads: int, alpha: float, beta: float, return_dvec: bool = False,
             tied: bool = False):
    """returns (dq, dk, dv bf16 like q; dk0_q (heads, 64) fp32 = alpha * sum dS_i0 q_i; dv0 (heads, 64) fp32;
    dbias0 (heads,) fp32 = sum dS_i0). `tied` (k is q): dq and dk are ONE tensor holding dq + dk."""
    L = _C.lib()
    L.require(q, k, v, k0, v0, o, lse, d_o)
    B, n, hd = q.shape
    assert d_o.dtype == torch.bfloat16 and d_o.is_contiguous() and d_o.shape == q.shape
    dq, dv = torch.empty_like(q), torch.empty_like(q)
    dk = dq if tied else torch.empty_like(q)
    dvec = torch.empty_like(lse)
    nblk = n // 128
    part = torch.empty((B, heads, nblk, 3, 64), dtype=torch.float32, device=q.device)
    rc = L.lib.gg_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(k0), ptr(v0), ptr(o), ptr(lse), ptr(d_o), ptr(dvec), ptr(dq), ptr(dk),
                           ptr(dv), ptr(part), B, n, heads, alpha, beta, L.stream(q))
    L.check(rc, 'gg_attn_bwd')
    s = part.sum(dim=(0, 2))                 # (heads, 3, 64)
    if return_dvec:
        return dq, dk, dv, s[:, 0] * alpha, s[:, 1], s[:, 2, 0], dvec
    return dq, dk, dv, s[:, 0] * alpha, s[:, 1], s[:, 2, 0]


def _attn_gen_check(q, k, v, heads):
    for t in (q, k, v):
        assert t.dtype == torch.bfloat16 and t.dim() == 3 and t.stride(2) == 1 and t.shape[2] == heads * 64
        assert t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0
    assert k.shape[:2] == v.shape[:2] and k.shape[0] == q.shape[0]


def attn_gen_fwd(q, k, v, k0, v0, kbias, heads: int, alpha: float, beta: float = 0.0):
    """general fused attention (gg_attn_gen_fwd): q (B, n, heads*64), k / v (B, m, heads*64) bf16 with a free row pitch (channel
    slices of a fused projection; stride(0) = len * stride(1)), optional null key / value k0, v0 (heads, 64) bf16, optional
    per-key bias kbias (B, m) fp32 -> (o (B, n, heads*64) bf16 dense, lse (B*heads, n) fp32)."""
    L = _C.lib()
    L.require(q, k, v, k0, v0, kbias)
    _attn_gen_check(q, k, v, heads)
    B, n, m = q.shape[0], q.shape[1], k.shape[1]
    if kbias is not None:
        assert kbias.dtype == torch.float32 and kbias.is_contiguous() and kbias.shape == (B, m)
    o = torch.empty((B, n, heads * 64), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((B * heads, n), dtype=torch.float32, device=q.device)
    rc = L.lib.gg_attn_gen_fwd(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(k0), ptr(v0), ptr(kbias), ptr(o),
                               ptr(lse), B, n, m, heads, alpha, beta, L.stream(q))
    L.check(rc, 'gg_attn_gen_fwd')
    return o, lse


def attn_gen_bwd(q, k, v, k0, v0, kbias, o, lse, d_o, heads: int, alpha: float, beta: float = 0.0):
    """-> (dq (B, n, heads*64), dk, dv (B, m, heads*64) bf16 dense; dk0_q, dv0 (heads, 64) fp32 and dbias0 (heads,) fp32 when the null
    token is present, else None)."""
    L = _C.lib()
    L.require(q, k, v, k0, v0, kbias, o, lse, d_o)
    _attn_gen_check(q, k, v, heads)
    B, n, m = q.shape[0], q.shape[1], k.shape[1]
    assert d_o.dtype == torch.bfloat16 and d_o.is_contiguous() and d_o.shape == o.shape and o.is_contiguous()
    dq = torch.empty_like(o)
    dk = torch.empty((B, m, heads * 64), dtype=torch.bfloat16, device=q.device)
    dv = torch.empty_like(dk)
    dvec = torch.empty_like(lse)
    nblk = (n + 127) // 128
    part = torch.empty((B, heads, nblk, 3, 64), dtype=torch.float32, device=q.device) if k0 is not None else None
    rc = L.lib.gg_attn_gen_bwd(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(k0), ptr(v0), ptr(kbias), ptr(o),
                               ptr(lse), ptr(d_o), ptr(dvec), ptr(dq), ptr(dk), ptr(dv), ptr(part), B, n, m, heads, alpha, beta,
                               L.stream(q))
    L.check(rc, 'gg_attn_gen_bwd')
    if part is None:
        return dq, dk, dv, None, None, None
    s = part.sum(dim=(0, 2))
    return dq, dk, dv, s[:, 0] * alpha, s[:, 1], s[:, 2, 0]


def attn_bwd2(q, k, v, k0, v0, d_o, lse, dvec, aq, ak, av, ak0, av0, heads: int, alpha: float, beta: float):
    """second-order pass: incoming gradients (aq, ak, av like q; ak0, av0 (heads, 64) bf16) w.r.t. attn_bwd's outputs ->
    (gq, gk, gv, gdo bf16 like q; gk0, gv0 (heads, 64) fp32)."""
    L = _C.lib()
    L.require(q, k, v, k0, v0, d_o, lse, dvec, aq, ak, av, ak0, av0)
    B, n, hd = q.shape
    for t in (d_o, aq, ak, av):
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.shape == q.shape
    for t in (ak0, av0):
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.shape == (heads, 64)
    gq, gk, gv, gdo = (torch.empty_like(q) for _ in range(4))
    mu, gi = torch.empty_like(lse), torch.empty_like(lse)
    nblk = n // 128
    part = torch.empty((B, heads, nblk, 3, 64), dtype=torch.float32, device=q.device)
    rc = L.lib.gg_attn_bwd2(ptr(q), ptr(k), ptr(v), ptr(k0), ptr(v0), ptr(d_o), ptr(aq), ptr(ak), ptr(av), ptr(ak0), ptr(av0),
                            ptr(lse), ptr(dvec), ptr(mu), ptr(gi), ptr(gq), ptr(gk), ptr(gv), ptr(gdo), ptr(part), B, n, heads,
                            alpha, beta, L.stream(q))
    L.check(rc, 'gg_attn_bwd2')
    s = part.sum(dim=(0, 2))                 # (heads, 3, 64)
    gk0 = alpha * s[:, 0] + 2.0 * beta * (s[:, 2, 0:1] * k0.float() + s[:, 2, 1:2] * ak0.float())
    return gq, gk, gv, gdo, gk0, s[:, 1]


# --------------------------------------------------------------------------------------------------
# ChannelRMSNorm passes
# --------------------------------------------------------------------------------------------------
RMS_EPS = 1e-12     # F.normalize's eps (gp.py:230)


def _rows3(t: torch.Tensor):
    """(b, n, C) bf16 view with unit channel stride and b * n uniformly pitched rows -> (ptr, row pitch)."""
    assert t.dtype == torch.bfloat16 and t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1)
    return ptr(t), t.stride(1)


def linattn_q_fwd(q: torch.Tensor, scale: float) -> torch.Tensor:
    """q (b, n, C) channel slice -> qs (b, n, C) contiguous = scale * softmax over each head's 64 features."""
    L = _C.lib()
    L.require(q)
    b, n, Cc = q.shape
    qs = torch.empty((b, n, Cc), dtype=torch.bfloat16, device=q.device)
    pq, ldq = _rows3(q)
    rc = L.lib.gg_linattn_q_fwd(pq, ldq, ptr(qs), Cc, b * n, Cc, float(scale), L.stream(q))
    L.check(rc, 'gg_linattn_q_fwd')
    return qs


def linattn_q_bwd(qs: torch.Tensor, dqs: torch.Tensor, dq_out: torch.Tensor, scale: float) -> None:
    L = _C.lib()
    L.require(qs, dqs, dq_out)
    b, n, Cc = qs.shape
    (p0, l0), (p1, l1), (p2, l2) = _rows3(qs), _rows3(dqs), _rows3(dq_out)
    rc = L.lib.gg_linattn_q_bwd(p0, l0, p1, l1, p2, l2, b * n, Cc, float(scale), L.stream(qs))
    L.check(rc, 'gg_linattn_q_bwd')


def linattn_k_fwd(k: torch.Tensor) -> torch.Tensor:
    """k (b, n, C) channel slice -> eks (b, n, C) contiguous = softmax over the n positions of every channel."""
    L = _C.lib()
    L.require(k)
    b, n, Cc = k.shape
    eks = torch.empty((b, n, Cc), dtype=torch.bfloat16, device=k.device)
    chunks = L.lib.gg_linattn_chunks(b, n)
    part = torch.empty((b, chunks, Cc, 2), dtype=torch.float32, device=k.device)
    stat = torch.empty((b, Cc, 2), dtype=torch.float32, device=k.device)
    pk, ldk = _rows3(k)
    rc = L.lib.gg_linattn_k_fwd(pk, ldk, ptr(eks), Cc, ptr(part), ptr(stat), b, n, Cc, L.stream(k))
    L.check(rc, 'gg_linattn_k_fwd')
    return eks


def linattn_k_bwd(eks: torch.Tensor, deks: torch.Tensor, dk_out: torch.Tensor) -> None:
    L = _C.lib()
    L.require(eks, deks, dk_out)
    b, n, Cc = eks.shape
    chunks = L.lib.gg_linattn_chunks(b, n)
    part = torch.empty((b, chunks, Cc), dtype=torch.float32, device=eks.device)
    stat = torch.empty((b, Cc), dtype=torch.float32, device=eks.device)
    (p0, l0), (p1, l1), (p2, l2) = _rows3(eks), _rows3(deks), _rows3(dk_out)
    rc = L.lib.gg_linattn_k_bwd(p0, l0, p1, l1, p2, l2, ptr(part), ptr(stat), b, n, Cc, L.stream(eks))
    L.check(rc, 'gg_linattn_k_bwd')


def scaled_add(a: torch.Tensor, b, c: float, d=None) -> torch.Tensor:
    """(a + b) * c + d (b, d optional) over dense bf16 tensors of one shape and ONE memory layout (any dimension order: the
    kernel walks the storage)."""
    L = _C.lib()
    L.require(a, b, d)
    assert a.dtype == torch.bfloat16 and a.numel() % 8 == 0
    for t in (b, d):
        assert t is None or (t.dtype == torch.bfloat16 and t.shape == a.shape and t.stride() == a.stride())
    y = torch.empty_like(a)
    assert y.stride() == a.stride()
    rc = L.lib.gg_scaled_add(ptr(a), ptr(b), ptr(d), ptr(y), a.numel(), float(c), L.stream(a))
    L.check(rc, 'gg_scaled_add')
    return y


def addcat(x: torch.Tensor, feats: torch.Tensor) -> torch.Tensor:
    """cat((x + tile(feats), tile(feats)), 0) for dense bf16 x (B, ...) and feats (f, ...) with B % f == 0 (same trailing shape)."""
    L = _C.lib()
    L.require(x, feats)
    assert x.dtype == torch.bfloat16 and feats.dtype == torch.bfloat16 and x.is_contiguous() and feats.is_contiguous()
    B, f = x.shape[0], feats.shape[0]
    n = x[0].numel()
    assert B % f == 0 and feats[0].numel() == n and n % 8 == 0
    out = torch.empty((2 * B,) + tuple(x.shape[1:]), dtype=torch.bfloat16, device=x.device)
    rc = L.lib.gg_addcat_fwd(ptr(x), ptr(feats), ptr(out), B, f, n, L.stream(x))
    L.check(rc, 'gg_addcat_fwd')
    return out


def addcat_bwd(g: torch.Tensor, f: int) -> torch.Tensor:
    """gradient w.r.t. feats of `addcat`: g (2B, ...) bf16 dense -> (f, ...)."""
    L = _C.lib()
    L.require(g)
    assert g.dtype == torch.bfloat16 and g.is_contiguous() and g.shape[0] % 2 == 0
    B = g.shape[0] // 2
    n = g[0].numel()
    out = torch.empty((f,) + tuple(g.shape[1:]), dtype=torch.bfloat16, device=g.device)
    rc = L.lib.gg_addcat_bwd(ptr(g), ptr(out), B, f, n, L.stream(g))
    L.check(rc, 'gg_addcat_bwd')
    return out


def pool_mean(x: torch.Tensor) -> torch.Tensor:
    """x (b, H, W, C) bf16 contiguous -> (b, C) fp32 mean over the pixels (SqueezeExcite's pool)."""
    L = _C.lib()
    L.require(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4
    b, H, W, Cc = x.shape
    P = H * W
    part = torch.empty((b, L.lib.gg_pool_chunks(b, P), Cc), dtype=torch.float32, device=x.device)
    out = torch.empty((b, Cc), dtype=torch.float32, device=x.device)
    rc = L.lib.gg_pool_mean_fwd(ptr(x), ptr(part), ptr(out), b, P, Cc, L.stream(x))
    L.check(rc, 'gg_pool_mean_fwd')
    return out


def pool_mean_bwd(gs: torch.Tensor, shape, g=None, inplace: bool = False) -> torch.Tensor:
    """y (b, H, W, C) bf16 = g + gs[b, c] (gs fp32 (b, C), already divided by H*W); g None: the plain broadcast."""
    L = _C.lib()
    L.require(gs, g)
    b, H, W, Cc = shape
    assert gs.dtype == torch.float32 and gs.is_contiguous() and gs.shape == (b, Cc)
    if g is not None:
        assert g.dtype == torch.bfloat16 and g.is_contiguous() and tuple(g.shape) == tuple(shape)
    y = g if (inplace and g is not None) else torch.empty(shape, dtype=torch.bfloat16, device=gs.device)
    rc = L.lib.gg_pool_mean_bwd(ptr(g), ptr(gs), ptr(y), b, H * W, Cc, L.stream(gs))
    L.check(rc, 'gg_pool_mean_bwd')
    return y


SEMLP_MAX_C, SEMLP_MAX_H = 2048, 512


def _f32c(*ts):
    for t in ts:
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous()), 'fp32 contiguous tensors'


def se_mlp_fwd(m: torch.Tensor, w1: torch.Tensor, b1, w2: torch.Tensor, b2):
    """SqueezeExcite's excitation MLP in one launch (gg_se_mlp_fwd): m (b, C), w1 (H, C), w2 (O, H), biases or None, all fp32 ->
    (h (b, H) pre-activation, hs = silu(h), e (b, O) = sigmoid(w2 hs + b2))."""
    L = _C.lib()
    L.require(m, w1, b1, w2, b2)
    _f32c(m, w1, b1, w2, b2)
    b, Cc = m.shape
    H, O = w1.shape[0], w2.shape[0]
    assert w1.shape == (H, Cc) and w2.shape == (O, H) and (b1 is None or b1.numel() == H) and (b2 is None or b2.numel() == O)
    hh = torch.empty((2, b, H), dtype=torch.float32, device=m.device)
    e = torch.empty((b, O), dtype=torch.float32, device=m.device)
    rc = L.lib.gg_se_mlp_fwd(ptr(m), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(hh[0]), ptr(hh[1]), ptr(e), b, Cc, H, O, L.stream(m))
    L.check(rc, 'gg_se_mlp_fwd')
    return hh[0], hh[1], e


def se_mlp_bwd(de: torch.Tensor, e, h, hs, m, w1, w2, want_dm: bool = True, want_gw: bool = True):
    """backward of se_mlp_fwd (gg_se_mlp_bwd): de (b, O) -> (dm (b, C) or None, (gW1 (H, C), gb1 (H), gW2 (O, H), gb2 (O)) or None);
    the parameter gradients are views into one buffer, summed over the samples in sample order."""
    L = _C.lib()
    L.require(de, e, h, hs, m, w1, w2)
    _f32c(de, e, h, hs, m, w1, w2)
    b, Cc = m.shape
    H, O = w1.shape[0], w2.shape[0]
    assert de.shape == (b, O) and e.shape == (b, O) and h.shape == (b, H) and hs.shape == (b, H)
    dz = torch.empty(b * (O + H), dtype=torch.float32, device=m.device)
    dm = torch.empty((b, Cc), dtype=torch.float32, device=m.device) if want_dm else None
    gw = torch.empty(H * Cc + H + O * H + O, dtype=torch.float32, device=m.device) if want_gw else None
    rc = L.lib.gg_se_mlp_bwd(ptr(de), ptr(e), ptr(h), ptr(hs), ptr(m), ptr(w1), ptr(w2), ptr(dz), ptr(dz[b * O:]), ptr(dm), ptr(gw),
                             b, Cc, H, O, L.stream(m))
    L.check(rc, 'gg_se_mlp_bwd')
    if gw is None:
        return dm, None
    n1, n2, n3 = H * Cc, H * Cc + H, H * Cc + H + O * H
    return dm, (gw[:n1].view(H, Cc), gw[n1:n2], gw[n2:n3].view(O, H), gw[n3:])


def poolhf_fwd(x: torch.Tensor):
    """x (b, H, W, C) bf16 NHWC -> (max_pool2d(x, 2) (b, H/2, W/2, C), x - blur(x) (b, H, W, C)) in one pass (gg_poolhf_fwd)."""
    L = _C.lib()
    L.require(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    b, H, W, Cc = x.shape
    pool = torch.empty((b, H // 2, W // 2, Cc), dtype=torch.bfloat16, device=x.device)
    hf = torch.empty_like(x)
    L.check(L.lib.gg_poolhf_fwd(ptr(x), ptr(pool), ptr(hf), b, H, W, Cc, L.stream(x)), 'gg_poolhf_fwd')
    return pool, hf


def poolhf_bwd(x: torch.Tensor, g_pool, g_hf) -> torch.Tensor:
    """gradient of poolhf_fwd w.r.t. x from the gradients of its two outputs (either may be None)."""
    L = _C.lib()
    L.require(x, g_pool, g_hf)
    b, H, W, Cc = x.shape
    for g in (g_pool, g_hf):
        assert g is None or (g.dtype == torch.bfloat16 and g.is_contiguous())
    dx = torch.empty_like(x)
    L.check(L.lib.gg_poolhf_bwd(ptr(x), ptr(g_pool), ptr(g_hf), ptr(dx), b, H, W, Cc, L.stream(x)), 'gg_poolhf_bwd')
    return dx


def rmsnorm_fwd(x: torch.Tensor, gamma: torch.Tensor, silu: bool = False) -> torch.Tensor:
    """x (..., C) bf16 contiguous, gamma (C,) fp32 -> y bf16; `silu`: y = silu(norm(x)) in the same pass."""
    L = _C.lib()
    L.require(x, gamma)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and gamma.dtype == torch.float32 and gamma.is_contiguous()
    Cc = x.shape[-1]
    y = torch.empty_like(x)
    rc = L.lib.gg_rmsnorm_fwd(ptr(x), ptr(gamma), ptr(y), x.numel() // Cc, Cc, RMS_EPS, int(silu), L.stream(x))
    L.check(rc, 'gg_rmsnorm_fwd')
    return y


def rmsnorm_bwd(x, g, gamma, want_dgamma: bool, carry=None, silu: bool = False, partials: bool = False):
    """(dx [+ carry], dgamma or None); `carry` (x's shape, bf16) is the skip branch's gradient, added inside the pass. `partials`: the
    per-workgroup partial sums (P, C) are returned instead of dgamma (for FinishQueue.add_colsum)."""
    L = _C.lib()
    L.require(x, g, gamma, carry)
    assert g.dtype == torch.bfloat16 and g.is_contiguous() and g.shape == x.shape
    assert carry is None or (carry.dtype == torch.bfloat16 and carry.is_contiguous() and carry.shape == x.shape)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dx = torch.empty_like(x)
    part = torch.empty((L.lib.gg_rmsnorm_blocks(rows), Cc), dtype=torch.float32, device=x.device) if want_dgamma else None
    rc = L.lib.gg_rmsnorm_bwd(ptr(x), ptr(g), ptr(gamma), ptr(carry), ptr(dx), ptr(part), rows, Cc, RMS_EPS, int(silu), L.stream(x))
    L.check(rc, 'gg_rmsnorm_bwd')
    if partials:
        return dx, part
    return dx, (part.sum(0) if part is not None else None)


def rmsnorm_bwd2(x, g, v, gamma, want_dgamma: bool):
    """second-order pass: (gx w.r.t. x, gg w.r.t. g, dgamma of this pass or None) for incoming v (gradient w.r.t. dx)."""
    L = _C.lib()
    L.require(x, g, v, gamma)
    assert v.dtype == torch.bfloat16 and v.is_contiguous() and v.shape == x.shape
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    gx, gg = torch.empty_like(x), torch.empty_like(x)
    part = torch.empty((L.lib.gg_rmsnorm_blocks(rows), Cc), dtype=torch.float32, device=x.device) if want_dgamma else None
    rc = L.lib.gg_rmsnorm_bwd2(ptr(x), ptr(g), ptr(v), ptr(gamma), ptr(gx), ptr(gg), ptr(part), rows, Cc, RMS_EPS, L.stream(x))
    L.check(rc, 'gg_rmsnorm_bwd2')
    return gx, gg, (part.sum(0) if part is not None else None)


# --------------------------------------------------------------------------------------------------
# weights: fp32 parameter layout <-> GEMM operand layouts (gg_weights.h)
# --------------------------------------------------------------------------------------------------

def wgrad_finish(g: torch.Tensor, O: int, I: int, T: int, alpha: float = 1.0, out: torch.Tensor | None = None,
                 accumulate: bool = False) -> torch.Tensor:
    """g: (T*C8, O8) fp32 from conv2d_wgrad_nhwc -> (O, I, T) fp32 = alpha * g transposed; with `out` (+ accumulate)
    the result is written / added in place (e.g. into a parameter's .grad view of the flat gradient buffer)."""
    L = _C.lib()
    L.require(g, out)
    assert g.dtype == torch.float32 and g.is_contiguous() and g.dim() == 2 and g.shape[0] % T == 0
    c8, o8 = g.shape[0] // T, g.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty((O, I, T), dtype=torch.float32, device=g.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == O * I * T
    rc = L.lib.gg_wgrad_finish(ptr(g), ptr(out), O, I, T, c8, o8, float(alpha), int(accumulate), L.stream(g))
    L.check(rc, 'gg_wgrad_finish')
    return out


class PackEntry(C.Structure):       # mirrors gg_pack_entry
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('first_item', C.c_int64), ('O', C.c_int32), ('I', C.c_int32),
                ('T', C.c_int32), ('O8', C.c_int32), ('I8', C.c_int32), ('kind', C.c_int32), ('dst_row', C.c_int32),
                ('dst_tap', C.c_int32)]


class PackTable:
    """The device-resident work table of gg_pack_weights for one model: every (weight, kind) registered here is
    re-packed by ONE launch of `refresh()`. Table and header are fixed device buffers that are only appended to, so
    graph-captured refresh launches follow later registrations."""
    KINDS = {'fwd': 0, 'bwd': 1}

    def __init__(self, device, capacity: int = 2048):
        self.device = torch.device(device)
        self.capacity = capacity
        self.table = torch.zeros(capacity * C.sizeof(PackEntry) // 8, dtype=torch.int64, device=self.device)
        self.header = torch.zeros(2, dtype=torch.int64, device=self.device)
        self.n = 0
        self.items = 0
        self.keep = []          # (src, dst) tensors kept alive
        self.dirty = True

    def register_bank(self, src: torch.Tensor, N: int, O: int, I: int, T: int) -> torch.Tensor:
        """src: fp32 contiguous (N, O, I, T) kernel bank -> persistent bf16 (O8, T*N*I8) laid out [co][tap][n][ci]: the
        fused adaptive conv's B operand (the N kernels interleaved along the reduction), one table entry per kernel."""
        assert src.dtype == torch.float32 and src.is_contiguous() and src.numel() == N * O * I * T
        o8, i8 = (O + 7) // 8 * 8, (I + 7) // 8 * 8
        dst = torch.zeros((o8, T * N * i8), dtype=torch.bfloat16, device=self.device)
        for n in range(N):
            self.register(src[n], O, I, T, 'fwd', into=(dst, n * i8, T * N * i8, N * i8))
        return dst

    def register_frag(self, src: torch.Tensor, N: int, O: int, I: int, T: int) -> torch.Tensor:
        """src: fp32 contiguous (N, O, I, T) kernel bank -> persistent bf16 (O/32, N, T, I/16, 64, 8): the bank in MFMA-fragment order
        (gg_aconv_fwd's weight stream), one table entry (kind 3) per kernel."""
        assert src.dtype == torch.float32 and src.is_contiguous() and src.numel() == N * O * I * T and O % 32 == 0 and I % 16 == 0 and T <= 16
        dst = torch.zeros((O // 32, N, T, I // 16, 64, 8), dtype=torch.bfloat16, device=self.device)
        for n in range(N):
            assert self.n < self.capacity, 'PackTable capacity exceeded'
            e = PackEntry(src[n].data_ptr(), dst.data_ptr(), self.items, O, I, T, O, I, 3, N, n)
            words = torch.frombuffer(bytearray(bytes(e)), dtype=torch.int64)
            w = words.numel()
            self.table[self.n * w:(self.n + 1) * w].copy_(words)
            self.n += 1
            self.items += O * ((I + 255) // 256)
            self.header.copy_(torch.tensor([self.n, self.items], dtype=torch.int64))
        self.keep.append((src, dst))
        return dst

    def register_gram(self, src: torch.Tensor, N: int, O: int, I: int, T: int) -> torch.Tensor:
        """src: fp32 contiguous (N, O, I, T) kernel bank -> persistent fp32 (N(N+1)/2, O, I) Gram rows sum_t W_n W_m (off-diagonal
        pairs doubled): what the adaptive conv's demodulation needs of the bank (gg_modfwd.h). One table entry (kind 2), one work
        item per output channel."""
        assert src.dtype == torch.float32 and src.is_contiguous() and src.numel() == N * O * I * T
        assert self.n < self.capacity, 'PackTable capacity exceeded'
        dst = torch.zeros((N * (N + 1) // 2, O, I), dtype=torch.float32, device=self.device)
        e = PackEntry(src.data_ptr(), dst.data_ptr(), self.items, O, I, T, O, I, 2, N, 0)
        words = torch.frombuffer(bytearray(bytes(e)), dtype=torch.int64)
        w = words.numel()
        self.table[self.n * w:(self.n + 1) * w].copy_(words)
        self.n += 1
        self.items += O
        self.header.copy_(torch.tensor([self.n, self.items], dtype=torch.int64))
        self.keep.append((src, dst))
        return dst

    def register(self, src: torch.Tensor, O: int, I: int, T: int, kind: str, into=None) -> torch.Tensor:
        """src: fp32 contiguous storage of (O, I, T); returns the persistent bf16 operand (rows, T*cols8). `into` =
        (tensor, element offset, row pitch, tap pitch) writes kind 'fwd' into a strided window of an existing operand."""
        assert src.dtype == torch.float32 and src.is_contiguous() and src.numel() == O * I * T
        assert self.n < self.capacity, 'PackTable capacity exceeded'
        o8, i8 = (O + 7) // 8 * 8, (I + 7) // 8 * 8
        k = self.KINDS[kind]
        if into is None:
            dst = torch.empty((o8, T * i8) if k == 0 else (i8, T * o8), dtype=torch.bfloat16, device=self.device)
            e = PackEntry(src.data_ptr(), dst.data_ptr(), self.items, O, I, T, o8, i8, k, 0, 0)
        else:
            assert k == 0
            dst, off, row, tap = into
            e = PackEntry(src.data_ptr(), dst.data_ptr() + 2 * off, self.items, O, I, T, o8, i8, k, row, tap)
        words = torch.frombuffer(bytearray(bytes(e)), dtype=torch.int64)
        w = words.numel()
        self.table[self.n * w:(self.n + 1) * w].copy_(words)
        self.n += 1
        if T <= 16:     # work items per entry: the formulas of gg_weights.h
            self.items += o8 * ((i8 + 255) // 256) if k == 0 else ((o8 + 63) // 64) * ((i8 + 15) // 16)
        else:
            self.items += (o8 * i8 // 8 * T + 255) // 256
        self.header.copy_(torch.tensor([self.n, self.items], dtype=torch.int64))
        self.keep.append((src, dst))
        return dst

    def refresh(self):
        if self.n == 0:
            return
        L = _C.lib()
        L.require(self.table)
        rc = L.lib.gg_pack_weights(ptr(self.table), ptr(self.header), 2 if L.is_emulator else 0, L.stream(self.table))
        L.check(rc, 'gg_pack_weights')


# --------------------------------------------------------------------------------------------------
# adaptive-conv coefficients (gg_modcoef.h)
# --------------------------------------------------------------------------------------------------

MODCOEF_MAX_N, MODCOEF_MAX_C = 4, 1024


def modcoef_fwd(w: torch.Tensor, mod: torch.Tensor, kmod, demod: bool, eps: float, Ip: int, Op: int):
    """w (N, O, I, k, k) fp32, mod (b, I) fp32, kmod (b, N) fp32 or None -> s (b, Ip), a (b, N), d (b, Op) or None."""
    L = _C.lib()
    L.require(w, mod, kmod)
    N, O, I = w.shape[:3]
    T = w.shape[3] * w.shape[4]
    b = mod.shape[0]
    assert w.dtype == torch.float32 and w.is_contiguous() and mod.dtype == torch.float32 and mod.is_contiguous()
    assert mod.shape == (b, I) and (kmod is None or (kmod.shape == (b, N) and kmod.dtype == torch.float32 and kmod.is_contiguous()))
    s = torch.empty((b, Ip), dtype=torch.float32, device=w.device)
    a = torch.empty((b, N), dtype=torch.float32, device=w.device)
    d = torch.empty((b, Op), dtype=torch.float32, device=w.device) if demod else None
    rc = L.lib.gg_modcoef_fwd(ptr(w), ptr(mod), ptr(kmod), ptr(s), ptr(a), ptr(d), b, N, O, I, T, Ip, Op, float(eps),
                              L.stream(w))
    L.check(rc, 'gg_modcoef_fwd')
    return s, a, d


MODGRAM_MAX_B = 64


def modgram(w: torch.Tensor) -> torch.Tensor:
    """w (N, O, I, k, k) fp32 -> the bank's Gram rows (P, O, I) fp32, P = N (N + 1) / 2 pairs n <= m (factor 2 off the diagonal)."""
    L = _C.lib()
    L.require(w)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 5
    N, O, I = w.shape[:3]
    gram = torch.empty((N * (N + 1) // 2, O, I), dtype=torch.float32, device=w.device)
    L.check(L.lib.gg_modgram(ptr(w), ptr(gram), N, O, I, w.shape[3] * w.shape[4], L.stream(w)), 'gg_modgram')
    return gram


def modcoef_gram_fwd(gram: torch.Tensor, N: int, mod: torch.Tensor, kmod, eps: float, Ip: int, Op: int):
    """gram (P, O, I), mod (b, I), kmod (b, N) or None -> s (b, Ip), a (b, N), d (b, Op), tsum (b, P, O) (kept for the backward)."""
    L = _C.lib()
    L.require(gram, mod, kmod)
    P, O, I = gram.shape
    b = mod.shape[0]
    assert P == N * (N + 1) // 2 and mod.dtype == torch.float32 and mod.is_contiguous() and mod.shape == (b, I) and b <= MODGRAM_MAX_B
    assert kmod is None or (kmod.shape == (b, N) and kmod.dtype == torch.float32 and kmod.is_contiguous())
    s = torch.empty((b, Ip), dtype=torch.float32, device=gram.device)
    a = torch.empty((b, N), dtype=torch.float32, device=gram.device)
    d = torch.empty((b, Op), dtype=torch.float32, device=gram.device)
    tsum = torch.empty((b, P, O), dtype=torch.float32, device=gram.device)
    rc = L.lib.gg_modcoef_gram_fwd(ptr(gram), ptr(mod), ptr(kmod), ptr(s), ptr(a), ptr(d), ptr(tsum), b, N, O, I, Ip, Op, float(eps),
                                   L.stream(gram))
    L.check(rc, 'gg_modcoef_gram_fwd')
    return s, a, d, tsum


def modcoef_gram_bwd(w, gram, kmod, s, d, tsum, gs, ga, gd, gw, eps: float):
    """-> (gmod (b, I), gkmod (b, N) or None); adds the demodulation path's weight gradient into gw (w-shaped) if given."""
    L = _C.lib()
    L.require(w, gram, kmod, s, d, tsum, gs, ga, gd, gw)
    N, O, I = w.shape[:3]
    T = w.shape[3] * w.shape[4]
    b, Ip = s.shape
    Op = d.shape[1]
    for t in (gs, ga, gd, gw):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    gmod = torch.empty((b, I), dtype=torch.float32, device=w.device)
    gkmod = torch.empty((b, N), dtype=torch.float32, device=w.device) if N > 1 else None
    slots = torch.empty((O, b, N), dtype=torch.float32, device=w.device) if N > 1 else None
    rc = L.lib.gg_modcoef_gram_bwd(ptr(w), ptr(gram), ptr(kmod), ptr(s), ptr(d), ptr(tsum), ptr(gs), ptr(ga), ptr(gd), ptr(gmod),
                                   ptr(gkmod), ptr(slots), ptr(gw), b, N, O, I, T, Ip, Op, float(eps), L.stream(w))
    L.check(rc, 'gg_modcoef_gram_bwd')
    return gmod, gkmod


def modcoef_bwd(w, kmod, s, d, gs, ga, gd, gw, eps: float):
    """-> (gmod (b, I), gkmod (b, N) or None); adds the demodulation path's weight gradient into gw (w-shaped) if given."""
    L = _C.lib()
    L.require(w, kmod, s, d, gs, ga, gd, gw)
    N, O, I = w.shape[:3]
    T = w.shape[3] * w.shape[4]
    b, Ip = s.shape
    Op = d.shape[1]
    for t in (gs, ga, gd, gw):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    gmod = torch.empty((b, I), dtype=torch.float32, device=w.device)
    gkmod = torch.empty((b, N), dtype=torch.float32, device=w.device) if N > 1 else None
    da_acc = torch.zeros((b, N), dtype=torch.float32, device=w.device)
    rc = L.lib.gg_modcoef_bwd(ptr(w), ptr(kmod), ptr(s), ptr(d), ptr(gs), ptr(ga), ptr(gd), ptr(gmod), ptr(gkmod),
                              ptr(da_acc), ptr(gw), b, N, O, I, T, Ip, Op, float(eps), L.stream(w))
    L.check(rc, 'gg_modcoef_bwd')
    return gmod, gkmod
