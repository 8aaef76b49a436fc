"""Builds the tuning cache gigagan_pytorch_amd/plans/gfx950.json: every distinct contraction geometry of a 4-step training cycle
at BASELINE config 2 (batch 32) is replayed on random operands with each eligible tile and a ladder of split-K factors; the
fastest plan is kept when it beats the cost model's choice by more than 4 %. Test / tuning infrastructure (run on the GPU box):
    python tests/gpu_plan_sweep.py [--workload uncond|upsampler|text] [--min-us 20]
"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402
from gigagan_pytorch_amd import _C, kernels as K   # noqa: E402
from gigagan_pytorch_amd._C import GemmDesc, PlanEntry, ROWK, KROW   # noqa: E402
from gigagan_pytorch_amd.data import SyntheticImages   # noqa: E402
from gigagan_pytorch_amd.gigagan import cycle   # noqa: E402


def key_of(d):
    full = bool(d.bias or d.out_scale or d.noise or d.residual or d.act)
    conv = bool(d.a_conv)
    return (d.M, d.N, d.K, d.batch, d.a_layout, d.b_layout, int(conv), d.H if conv else 0, d.W if conv else 0, d.C if conv else 0,
            d.CV if conv else 0, d.R if conv else 0, d.conv_stride if conv else 0, d.conv_pad if conv else 0, int(bool(d.c_is_f32)),
            d.d2s, int(full), int(bool(d.in_scale)))


def materialise(d, dev):
    """fresh random operands for a captured descriptor (pointer fields are replaced); returns the tensors to keep alive."""
    keep = []

    def buf(n, dtype, scale=1.0):
        t = (torch.randn(max(int(n), 8), device=dev) * scale).to(dtype)
        keep.append(t)
        return t.data_ptr()
    bf, f32 = torch.bfloat16, torch.float32
    if d.a_conv:
        oh = (d.H + 2 * d.conv_pad - d.R) // d.conv_stride + 1
        ow = (d.W + 2 * d.conv_pad - d.S) // d.conv_stride + 1
        n_img = (d.M if d.a_layout == ROWK else d.K) // (oh * ow)
        d.A = buf(n_img * d.H * d.W * d.C, bf)
        if d.in_scale:
            d.in_scale = buf(n_img * d.CV, f32)
    else:
        d.A = buf((d.batch - 1) * d.a_batch_stride + (d.M if d.a_layout == ROWK else d.K) * d.lda, bf)
    b_elems = (d.batch - 1) * d.b_batch_stride + (d.N if d.b_layout == ROWK else d.K) * d.ldb
    if d.b_image_stride:          # per-image weights: image i reads B + i * b_image_stride
        b_elems += (d.M // (oh * ow) - 1) * d.b_image_stride
    d.B = buf(b_elems, bf, 0.05)
    if d.d2s:
        n_img = d.M // (d.d2s_oh * d.d2s_ow)
        celems = n_img * d.d2s_oh * d.d2s * d.d2s_ow * d.d2s * d.d2s_c
    else:
        celems = (d.batch - 1) * d.c_batch_stride + d.M * d.ldc
    d.C_out = buf(celems, f32 if d.c_is_f32 else bf)
    d._out_tensor = keep[-1]
    if d.bias:
        d.bias = buf(d.N, f32)
    if d.out_scale:
        d.out_scale = buf((d.M // d.rows_per_group + 1) * d.N, f32)
    if d.noise:
        d.noise = buf(d.M, f32)
        d.noise_w = buf(d.N, f32)
    if d.residual:
        d.residual = buf(d.M * d.ldr, bf)
    d.keep_partials = 0              # (round 4: the step's weight gradients leave their split-K slices to the queued finish; here every plan reduces)
    if d.gelu_mode:                  # (round 4: GELU fused around a 1x1 pair: a pre-activation buffer of the output's shape)
        d.gelu_aux = buf(d.M * d.ld_aux, bf)
    return keep


def output_of(L, d, ws):
    """run the plan once and return a copy of what it wrote (the whole C_out allocation)."""
    stream = torch.cuda.current_stream().cuda_stream
    rc = L.lib.gg_gemm_bf16(C.byref(d), ws.data_ptr(), ws.numel(), stream)
    if rc:
        return None
    torch.cuda.synchronize()
    return d._out_tensor.clone()


def time_plan(L, d, ws, iters=3):
    need = L.lib.gg_gemm_workspace_bytes(C.byref(d))
    if need > ws.numel():
        return None
    stream = torch.cuda.current_stream().cuda_stream
    rc = L.lib.gg_gemm_bf16(C.byref(d), ws.data_ptr(), ws.numel(), stream)
    if rc:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.lib.gg_gemm_bf16(C.byref(d), ws.data_ptr(), ws.numel(), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='uncond')
    ap.add_argument('--min-us', type=float, default=15.0, help='geometries whose planned launch is shorter are left to the cost model')
    ap.add_argument('--gain', type=float, default=0.04)
    ap.add_argument('--tiles', default='', help='comma-separated candidate tiles (default: 1-10); e.g. 11,12 for the round-3 kernels')
    ap.add_argument('--keep-table', action='store_true', help='time candidates against the INSTALLED plan table instead of the bare cost model')
    args = ap.parse_args()
    cand_tiles = tuple(int(t) for t in args.tiles.split(',') if t) or (1, 2, 3, 4, 5, 6, 7, 8, 9, 10)
    dev = torch.device('cuda', 0)
    L = _C.lib()
    if not args.keep_table:
        L.load_plan_table([])                   # capture and time against the bare cost model
    batch = 32 if args.workload == 'uncond' else 16
    gan = bench.build_gan(256, dev, use_hip_graphs=False, workload=args.workload)
    if args.workload == 'text':
        it = iter(bench.SyntheticTextImages(batch, 256, dev))
    else:
        it = cycle(SyntheticImages(batch, 256, device=dev))
    gan.train_step(it, batch)
    K.desc_log = []
    for _ in range(4):
        gan.train_step(it, batch)
    with torch.no_grad():
        gan.G(noise=torch.randn(batch, gan.G.style_network.dim, device=dev)) if args.workload == 'uncond' else None
    raw, K.desc_log = K.desc_log, None
    del gan
    torch.cuda.empty_cache()
    uniq = {}
    for b in raw:
        d = GemmDesc.from_buffer_copy(b)
        if d.b_image_stride:
            continue            # per-image weights are always planned by the cost model (the table does not apply)
        k = key_of(d)
        if k not in uniq:
            uniq[k] = [d, 0]
        uniq[k][1] += 1
    print(f'{len(raw)} launches per cycle, {len(uniq)} distinct geometries', flush=True)
    ws = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
    ladder = [1, 2, 3, 4, 6, 7, 8, 12, 14, 16, 24, 28, 32, 48, 64, 96, 128, 256, 512, 1024, 2048, 4096]
    entries, report = [], []
    saved_total = planned_total = 0.0
    for k, (d, count) in sorted(uniq.items(), key=lambda kv: -kv[1][1] * kv[0][0] * kv[0][1] * kv[0][2]):
        keep = materialise(d, dev)
        d.force_tile = d.force_splitk = 0
        tile, sk = C.c_int32(0), C.c_int32(0)
        L.lib.gg_gemm_plan(C.byref(d), C.byref(tile), C.byref(sk))
        t_plan = time_plan(L, d, ws)
        if t_plan is None:
            continue
        planned_total += t_plan * count
        best = (t_plan, tile.value, sk.value)
        # numerics yardstick: the 4-wave kernel, unsplit (every candidate - and the planner's own choice - must reproduce it:
        # a fast plan that computes something else does not enter the table)
        d.force_tile, d.force_splitk = (3 if d.N <= 32 else (2 if d.N <= 64 else 1)), 1
        d._out_tensor.zero_()
        ref_out = output_of(L, d, ws)
        d.force_tile = d.force_splitk = 0
        tol = 1e-4 if d.c_is_f32 else 2e-2

        def agrees(what):
            if ref_out is None:
                return True
            d._out_tensor.zero_()
            out = output_of(L, d, ws)
            if out is None:
                return False
            err = float((out.float() - ref_out.float()).norm() / ref_out.float().norm().clamp(min=1e-12))
            if not (err < tol):
                print(f'!! WRONG RESULT {what}: rel err {err:.3g} for M={d.M} N={d.N} K={d.K} lay={d.a_layout}{d.b_layout} conv={d.a_conv} '
                      f'H={d.H} C={d.C} CV={d.CV} R={d.R} s={d.conv_stride} f32={d.c_is_f32} d2s={d.d2s}', flush=True)
                return False
            return True
        agrees(f'planner choice {(tile.value, sk.value)}')
        if t_plan >= args.min_us:
            ktiles = (d.K + 31) // 32
            for ft in cand_tiles:
                seen = set()
                for fs in (ladder if ft not in (7, 8, 11, 12) else [1, 2, 3, 4, 6, 8, 12, 16]):       # (gg_conv3 / gg_lrconv split over channel chunks)
                    if fs > ktiles:
                        break
                    d.force_tile, d.force_splitk = ft, fs
                    t2, s2 = C.c_int32(0), C.c_int32(0)
                    if L.lib.gg_gemm_plan(C.byref(d), C.byref(t2), C.byref(s2)) or t2.value != ft or s2.value in seen:
                        continue
                    seen.add(s2.value)
                    t = time_plan(L, d, ws)
                    if t is not None and t < best[0] and agrees(f'forced {(ft, s2.value)}'):
                        best = (t, ft, s2.value)
                    if ft == 9:
                        break               # the direct convolution does not split K
                    if t is not None and t > 3 * best[0] and fs >= 8:
                        break                   # far off already: deeper splits only add reduction traffic
        d.force_tile = d.force_splitk = 0
        gain = 1 - best[0] / t_plan
        row = dict(zip(PlanEntry.FIELDS[:18], k))
        report.append(dict(row, launches=count, planned_us=t_plan, planned=(tile.value, sk.value), best_us=best[0], best=(best[1], best[2])))
        if gain > args.gain and (best[1], best[2]) != (tile.value, sk.value):
            entries.append(dict(row, tile=best[1], splitk=best[2]))
            saved_total += (t_plan - best[0]) * count
            print(f'M={k[0]:7d} N={k[1]:5d} K={k[2]:7d} lay={k[4]}{k[5]} conv={k[6]} R={k[11]} s={k[12]} f32={k[14]} epi={k[16]} x{count:3d}: '
                  f'{t_plan:8.1f} us {(tile.value, sk.value)} -> {best[0]:8.1f} us {(best[1], best[2])}  (-{100 * gain:.0f} %)', flush=True)
        del keep
    print(f'planned contraction time per cycle {planned_total / 1e3:.1f} ms; table saves {saved_total / 1e3:.2f} ms per cycle '
          f'({saved_total / 4e3:.2f} ms per step) with {len(entries)} entries', flush=True)
    out = ROOT / 'gpurun_out'
    out.mkdir(exist_ok=True)
    (out / f'plan_sweep_{args.workload}.json').write_text(json.dumps(dict(
        note='measured on one MI355X by tests/gpu_plan_sweep.py: (tile, split-K) that beat the cost model by > 4 % on random operands',
        workload=args.workload, entries=entries), indent=0))
    (out / f'plan_sweep_{args.workload}_report.json').write_text(json.dumps(report))


if __name__ == '__main__':
    main()
