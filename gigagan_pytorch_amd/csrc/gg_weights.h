// gg_weights.h — the two passes that sit between the fp32 master parameters (reference layout (O, I, kh, kw), flat
// AdamW buffers) and the GEMM kernels' operand layouts.
//
//  gg_pack_weights_kernel: ONE launch re-packs every registered conv weight of a model into its bf16 GEMM operand(s)
//      kind 0 ('fwd'): dst[o][t][i8]        = src[o][i][t]            (forward B operand, [co][kh][kw][ci])
//      kind 1 ('bwd'): dst[i][T-1-t][o8]    = src[o][i][t]            (data-gradient B operand: flipped, in/out swapped)
//      kind 2 ('gram'): src is a kernel bank (N, O, I, T); dst (fp32) [pair][o][i] = c * sum_t W_n[o][i][t] W_m[o][i][t] for the
//                      pairs n <= m (c = 1 on the diagonal, 2 off it): the Gram rows from which the adaptive convolution's
//                      demodulation d[b,o] = rsqrt(sum_i s_i^2 a^T G[o][i] a) is formed (gp.py:390-400) - they only change when the
//                      optimizer steps, so they are refreshed here instead of in every forward (gg_modfwd.h reads them)
//      kind 3 ('frag'): src is kernel n of a bank (N, O, I, T) with O % 32 == 0, I % 16 == 0; dst is the whole bank in MFMA-FRAGMENT
//                      order [O/32][N][T][I/16][lane 64][8]: lane l of block (o >> 5, n, t, i >> 4) holds output channel (o & 31) =
//                      l & 31, input channels 16 (i >> 4) + 8 (l >> 5) + 0..7 - the A operand of one v_mfma_f32_32x32x16_bf16 as ONE
//                      coalesced 1 KB load (gg_aconv.h streams the bank straight into registers). dst_row = N, dst_tap = n.
//    channel counts zero-padded to multiples of 8. It replaces the per-weight permute / flip / pad / cast chain the
//    reference gets from cuDNN's internal filter transforms (gp.py:402-409, :1608-1621 F.conv2d call sites) - ~1200
//    tiny launches per step - by a table walk. The table and its header live in device memory, so a captured hipGraph
//    sees entries registered after capture. HBM-bound: 4 B read + 2 B written per weight element and kind.
//
//  gg_wgrad_finish_kernel: dst[o][i][t] (+)= alpha * g[(t*C8 + i)*O8 + o] - the weight-gradient GEMM's fp32 output
//    ([tap][ci][co], what the MFMA kernel produces with pixels as the reduction) scattered into the parameter layout
//    through an LDS transpose, scaled, and accumulated straight into the flat gradient buffer (.grad view) when asked.
#pragma once
#include "gg_device.h"

struct GgPackEntry {
    const float* src;       // (O, I, T) fp32, contiguous
    bf16_t* dst;            // kind 0: (O8, T, I8) ; kind 1: (I8, T, O8)
    long long first_item;   // prefix sum of work items over the table
    int O, I, T, O8, I8, kind;
    int dst_row, dst_tap;   // kind 0: element pitch of a dst row / of a tap within it (0 = dense: T * I8, I8); kind 2: dst_row = N;
                            // kind 3: dst_row = N (kernels in the bank), dst_tap = n (which of them src is)
};

// Work items (one workgroup each; `first_item` is their prefix sum, computed by the host with the same formulas):
//   T <= 16: kind 0: one output row o x 256 input channels      -> O8 * ceil(I8 / 256) items
//            kind 1: 64 output channels x 16 input channels     -> ceil(O8 / 64) * ceil(I8 / 16) items
//            both stream contiguous fp32 runs in, transpose through LDS, and write contiguous bf16 rows out
//   T  > 16: 256 (eight-channel unit, tap) pairs per item, one per thread (strided scalar gathers; 7x7 stems only)
//            -> ceil(O8 * I8 / 8 * T / 256) items
//   kind 2:  one output channel of the bank (all pairs, all input channels) -> O items
#define GG_PK_TMAX 16
#define GG_PK_P0 264      // kind 0 LDS pitch (bf16): 256 + 8
#define GG_PK_P1 66       // kind 1 LDS pitch (bf16): 64 + 2 -> 33-word row shift, conflict-free transposed writes

// header[0] = number of entries, header[1] = total work items
// A workgroup searches the table once, then only advances, and issues all fp32 loads of an item
// before the first use (kind 0: up to 16 in flight per thread, kind 1: batches of 12): the first version searched the table per
// item and looped load -> convert -> LDS store, i.e. ~12 us of exposed round trips per 9 KB item (1.8 TB/s on the D model).
#define GG_PK_B1 12       // kind 1 loads in flight per thread

GG_DEVICE int gg_pk_div(int r, int inv) { return (int)(((unsigned)r * (unsigned)inv) >> 16); }     // r / T for r < 4096, inv = 65536 / T + 1

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_pack_weights_kernel(const GgPackEntry* table, const long long* header) {
    GG_SHARED __attribute__((aligned(16))) bf16_t lds[16 * GG_PK_TMAX * GG_PK_P1];
    const int n = (int)header[0];
    const long long total = header[1];
    const int tid = threadIdx.x;
    // items are dealt round-robin (block b takes b, b + G, b + 2G, ...: heavy and light entries spread evenly over the workgroups;
    // contiguous ranges left the blocks that drew the 9 KB transposing items running 2x longer than the launch's mean); a block's
    // items increase, so the table position only ever advances
    long long item = blockIdx.x;
    if (item >= total) return;
    int lo = 0, hi = n - 1;                           // last entry with first_item <= item
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_item <= item) lo = mid; else hi = mid - 1;
    }
    GgPackEntry e = table[lo];
    long long next_first = lo + 1 < n ? table[lo + 1].first_item : total;
    for (; item < total; item += gridDim.x) {
        while (item >= next_first) {
            ++lo;
            next_first = lo + 1 < n ? table[lo + 1].first_item : total;
            if (item < next_first) e = table[lo];
        }
        const long long local = item - e.first_item;
        const int T = e.T;
        if (e.kind == 2) {          // Gram rows of output channel o = local (threads run along the input channels)
            const int N = e.dst_row, o = (int)local;
            float* g = (float*)e.dst;
            for (int i = tid; i < e.I; i += 256) {
                int pair = 0;
                for (int n = 0; n < N; ++n)
                    for (int m = n; m < N; ++m, ++pair) {
                        const float* wn = e.src + (((long long)n * e.O + o) * e.I + i) * T;
                        const float* wm = e.src + (((long long)m * e.O + o) * e.I + i) * T;
                        float acc = 0.f;
                        for (int t = 0; t < T; ++t) acc += wn[t] * wm[t];
                        g[((long long)pair * e.O + o) * e.I + i] = (n == m ? 1.f : 2.f) * acc;
                    }
            }
            continue;
        }
        const int tinv = 65536 / T + 1;
        const long long drow = e.dst_row ? e.dst_row : (long long)T * e.I8;    // kind 0 destination pitches
        const long long dtap = e.dst_tap ? e.dst_tap : e.I8;
        if (T > GG_PK_TMAX) {      // one (8-channel unit, tap) per thread: 8 strided scalar loads, one 16-byte store
            const long long w = local * 256 + tid;
            const long long unit = w / T;
            const int t = (int)(w - unit * T);
            if (e.kind == 0) {
                const int chunks = e.I8 >> 3;
                if (unit >= (long long)e.O8 * chunks) continue;
                const int o = (int)(unit / chunks), i0 = (int)(unit % chunks) * 8;
                const float* s = e.src + ((long long)o * e.I + i0) * T + t;
                const int valid = (o < e.O) ? (e.I - i0 < 8 ? e.I - i0 : 8) : 0;
                u16x8 v;
                for (int j = 0; j < 8; ++j) v[j] = gg_f2bf(j < valid ? s[(long long)j * T] : 0.f);
                *(u16x8*)(e.dst + (long long)o * drow + i0 + (long long)t * dtap) = v;
            } else {
                if (unit >= (long long)e.I8 * (e.O8 >> 3)) continue;
                const int i = (int)(unit % e.I8), o0 = (int)(unit / e.I8) * 8;
                const float* s = e.src + ((long long)o0 * e.I + i) * T + t;
                const int valid = (i < e.I) ? (e.O - o0 < 8 ? e.O - o0 : 8) : 0;
                const long long ostride = (long long)e.I * T;
                u16x8 v;
                for (int j = 0; j < 8; ++j) v[j] = gg_f2bf(j < valid ? s[j * ostride] : 0.f);
                *(u16x8*)(e.dst + (long long)i * T * e.O8 + o0 + (long long)(T - 1 - t) * e.O8) = v;
            }
            continue;
        }
        if (e.kind == 0 || e.kind == 3) {
            const int chunks = (e.I8 + 255) >> 8;
            const int o = (int)(local / chunks), i0 = (int)(local % chunks) * 256;
            const int nI8 = e.I8 - i0 < 256 ? e.I8 - i0 : 256;
            int nI = (o < e.O) ? e.I - i0 : 0;
            nI = nI < 0 ? 0 : (nI > 256 ? 256 : nI);
            const float* s = nI ? e.src + ((long long)o * e.I + i0) * T : e.src;     // (a padding row reads nothing)
            const int cnt = nI8 * T, real = nI * T;
            float v[GG_PK_TMAX];
#pragma unroll
            for (int k = 0; k < GG_PK_TMAX; ++k) {
                const int idx = tid + 256 * k;
                v[k] = s[idx < real ? idx : 0];
            }
#pragma unroll
            for (int k = 0; k < GG_PK_TMAX; ++k) {
                const int idx = tid + 256 * k;
                if (idx < cnt) {
                    const int il = gg_pk_div(idx, tinv), t = idx - il * T;
                    lds[t * GG_PK_P0 + il] = gg_f2bf(idx < real ? v[k] : 0.f);
                }
            }
            gg_sync();
            const int cpr = nI8 >> 3;
            if (e.kind == 3) {      // fragment order: the 8-channel vector g = (i0 >> 3) + j of (o, t) is lane (o & 31) + 32 (g & 1) of block g >> 1
                const int NBk = e.dst_row, nb = e.dst_tap, CB = e.I8 >> 4;
                for (int c = tid; c < T * cpr; c += 256) {
                    const int t = c / cpr, j = c - t * cpr;
                    const int g = (i0 >> 3) + j;
                    const long long blk = (((long long)(o >> 5) * NBk + nb) * T + t) * CB + (g >> 1);
                    *(u16x8*)(e.dst + blk * 512 + ((o & 31) + 32 * (g & 1)) * 8) = *(const u16x8*)&lds[t * GG_PK_P0 + 8 * j];
                }
                gg_sync();
                continue;
            }
            bf16_t* d = e.dst + (long long)o * drow + i0;
            for (int c = tid; c < T * cpr; c += 256) {
                const int t = c / cpr, j = c - t * cpr;
                *(u16x8*)(d + (long long)t * dtap + 8 * j) = *(const u16x8*)&lds[t * GG_PK_P0 + 8 * j];
            }
            gg_sync();
        } else {
            const int ibs = (e.I8 + 15) >> 4;
            const int o0 = (int)(local / ibs) * 64, i0 = (int)(local % ibs) * 16;
            int nI = e.I - i0;
            nI = nI < 0 ? 0 : (nI > 16 ? 16 : nI);
            const int run = 16 * T, cnt = 64 * run, real = nI * T;
            // element idx = tid + 256 k -> (output channel ol = idx / run, position r = idx % run inside its 16-channel run)
            int ol = 0, r = tid;
            while (r >= run) { r -= run; ++ol; }
            for (int kb = 0; kb * 256 < cnt; kb += GG_PK_B1) {
                float v[GG_PK_B1];
                int lpos[GG_PK_B1];
#pragma unroll
                for (int u = 0; u < GG_PK_B1; ++u) {
                    const bool inb = (kb + u) * 256 + tid < cnt;
                    const bool ok = inb && o0 + ol < e.O && r < real;
                    v[u] = e.src[ok ? ((long long)(o0 + ol) * e.I + i0) * T + r : 0];
                    const int il = gg_pk_div(r, tinv), t = r - il * T;
                    lpos[u] = inb ? (il * T + (T - 1 - t)) * GG_PK_P1 + ol : -1;
                    if (!ok) v[u] = 0.f;
                    r += 256;
                    while (r >= run) { r -= run; ++ol; }
                }
#pragma unroll
                for (int u = 0; u < GG_PK_B1; ++u)
                    if (lpos[u] >= 0) lds[lpos[u]] = gg_f2bf(v[u]);
            }
            gg_sync();
            const int nI8 = e.I8 - i0 < 16 ? e.I8 - i0 : 16;
            bf16_t* d = e.dst + (long long)i0 * T * e.O8 + o0;
            for (int idx = tid; idx < nI8 * T * 32; idx += 256) {
                const int w = idx & 31, row = idx >> 5;
                if (o0 + 2 * w < e.O8)
                    *(unsigned int*)(d + (long long)row * e.O8 + 2 * w) = *(const unsigned int*)&lds[row * GG_PK_P1 + 2 * w];
            }
            gg_sync();
        }
    }
}

#define GG_WF_TG 9        // taps staged per pass (a 3x3 kernel in one pass)
#define GG_WF_IB 8        // input channels per workgroup (x 32 output channels): 256 threads = one element per thread and tap

struct GgWgradFinishParams {
    const float* g;       // (T*C8, O8) fp32
    float* dst;           // (O, I, T) fp32
    int O, I, T, C8, O8, accumulate;
    float alpha;
    int nsplit;           // split-K slices [nsplit][T*C8][O8] behind g, summed while they are read (<= 1: one)
};

// Workgroup (x, y): output channels 32 x .. + 31, input channels 8 y .. + 7. Thread t reads, for every tap of the pass, the
// element (o = t & 31, i = (t >> 5) & 7): all of a pass's loads are issued before the first use (at most 9 in flight per thread;
// the first version looped load -> LDS store and every one of its 36 round trips was exposed: ~22 us for ANY layer size), the
// tile is transposed through LDS, and each output channel's 8 x T run is written (or accumulated) contiguously.
typedef float GgWfTile[32][GG_WF_IB * GG_WF_TG + 1];
GG_DEVICE void gg_wgrad_finish_body(const GgWgradFinishParams& p, GgWfTile& tile, int bx, int by) {
    const int o0 = bx * 32, i0 = by * GG_WF_IB;
    const int t = threadIdx.x;
    const int ol = t & 31, il = (t >> 5) & (GG_WF_IB - 1);
    const bool ok = o0 + ol < p.O && i0 + il < p.I;
    const int oc = o0 + ol < p.O ? o0 + ol : p.O - 1, ic = i0 + il < p.I ? i0 + il : p.I - 1;     // clamped: unconditional loads
    for (int t0 = 0; t0 < p.T; t0 += GG_WF_TG) {
        const int tg = p.T - t0 < GG_WF_TG ? p.T - t0 : GG_WF_TG;
        float v[GG_WF_TG];
#pragma unroll
        for (int k = 0; k < GG_WF_TG; ++k) {
            const int tc = k < tg ? t0 + k : t0;
            v[k] = p.g[((long long)tc * p.C8 + ic) * p.O8 + oc];
        }
        if (p.nsplit > 1) {           // the split-K reduction folded in: slice s sits s * T * C8 * O8 floats further; fixed order
            const long long slice = (long long)p.T * p.C8 * p.O8;
            int s = 1;
            for (; s + 2 < p.nsplit; s += 3) {            // three slices (27 loads) in flight per thread
                float a[3][GG_WF_TG];
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int k = 0; k < GG_WF_TG; ++k) {
                        const int tc = k < tg ? t0 + k : t0;
                        a[j][k] = p.g[(s + j) * slice + ((long long)tc * p.C8 + ic) * p.O8 + oc];
                    }
#pragma unroll
                for (int k = 0; k < GG_WF_TG; ++k) v[k] += (a[0][k] + a[1][k]) + a[2][k];
            }
            for (; s < p.nsplit; ++s) {
#pragma unroll
                for (int k = 0; k < GG_WF_TG; ++k) {
                    const int tc = k < tg ? t0 + k : t0;
                    v[k] += p.g[s * slice + ((long long)tc * p.C8 + ic) * p.O8 + oc];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < GG_WF_TG; ++k)
            if (k < tg) tile[ol][il * tg + k] = ok ? v[k] : 0.f;
        gg_sync();
        const int cols = GG_WF_IB * tg;               // 32 rows x cols elements, `tg` per thread
        float* d[GG_WF_TG];
        float w[GG_WF_TG], old[GG_WF_TG];
        bool live[GG_WF_TG];
#pragma unroll
        for (int k = 0; k < GG_WF_TG; ++k) {
            const int idx = t + 256 * k;
            const int orow = idx / cols, col = idx - orow * cols;
            const int icol = col / tg, tl = col - icol * tg;
            live[k] = k < tg && o0 + orow < p.O && i0 + icol < p.I;
            const int orc = live[k] ? orow : ol, icc = live[k] ? icol : 0, tlc = live[k] ? tl : 0;
            d[k] = p.dst + ((long long)(live[k] ? o0 + orc : oc) * p.I + (live[k] ? i0 + icc : ic)) * p.T + t0 + tlc;
            w[k] = live[k] ? tile[orc][icc * tg + tlc] : 0.f;
            old[k] = 0.f;
        }
        if (p.accumulate) {
#pragma unroll
            for (int k = 0; k < GG_WF_TG; ++k) old[k] = *d[k];           // (clamped to a valid element where not live)
        }
#pragma unroll
        for (int k = 0; k < GG_WF_TG; ++k)
            if (live[k]) *d[k] = old[k] + p.alpha * w[k];
        gg_sync();
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_wgrad_finish_kernel(GgWgradFinishParams p) {
    GG_SHARED GgWfTile tile;
    gg_wgrad_finish_body(p, tile, blockIdx.x, blockIdx.y);
}

// dst[c] += alpha * sum_p part[p][c], c < n: the per-workgroup partial column sums of gg_bias_act_bwd (or of any
// [P][C] fp32 partial buffer) folded, scaled and accumulated into a bias gradient in one launch - replaces the
// sum / slice / scale / AccumulateGrad chain (4 launches per bias). grid (ceil(n/64), G): workgroup (x, y) folds the
// partial rows y, y+G, ... (4 waves interleaved) of 64 channels and issues one atomic add per channel; dst holds the
// running gradient (or zeros).
GG_DEVICE void gg_colsum_finish_body(const float* part, float* dst, int P, int C, int n, float alpha, float (&red)[4][64], int bx, int by,
                                     int groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = bx * 64 + lane;
    float s = 0.f;
    if (c < n) {
        // (eight rows in flight per wavefront: with one workgroup per column block - GG_COLSUM_GROUPS=1 - a wavefront folds up to 256 rows,
        // and one load at a time made that a chain of round trips)
        const int step = 4 * groups;
        int q = by * 4 + wave;
        for (; q + 7 * step < P; q += 8 * step) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(long long)(q + k * step) * C + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; q < P; q += step) s += part[(long long)q * C + c];
    }
    red[wave][lane] = s;
    gg_sync();
    if (wave == 0 && c < n) gg_atomic_add(dst + c, alpha * ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])));
}
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_colsum_finish_kernel(const float* part, float* dst, int P, int C, int n, float alpha) {
    GG_SHARED float red[4][64];
    gg_colsum_finish_body(part, dst, P, C, n, alpha, red, blockIdx.x, blockIdx.y, gridDim.y);
}

// ---- many finishes in ONE launch -------------------------------------------------------------------------------------------------
// A backward pass ends every convolution with a weight-gradient finish (and a bias column-sum finish): 221 launches per step of
// 4-8 us each, every one a handful of dependent round trips on a few KB - 1.5 ms of latency for 0.1 ms of traffic. The host queues
// them (kernels.FinishQueue) and hands batches of up to GG_FM_MAX items over by value (kernel arguments); a workgroup finds its item
// in the prefix sums of the items' workgroup counts and runs the single-launch body on it.
#define GG_FM_MAX 40
struct GgFinishItem {
    const float* src;     // kind 0: (T*C8, O8) fp32 weight-gradient GEMM output ; kind 1: (P, C) fp32 partial column sums
    float* dst;           // kind 0: (O, I, T) fp32 ; kind 1: (n,) fp32, accumulated by atomics
    int nsplit;
    int kind, O, I, T, C8, O8, accumulate;      // kind 1: O = P, I = C, T = n, C8 = row groups; kind 2 (dst += alpha * src): O = elements
    float alpha;
};
struct GgFinishBatch {
    GgFinishItem item[GG_FM_MAX];
    int first_wg[GG_FM_MAX + 1];      // prefix sums of the items' workgroup counts
    int n;
};
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_finish_multi_kernel(GgFinishBatch b) {
    GG_SHARED GgWfTile tile;
    const int wg = blockIdx.x;
    int lo = 0, hi = b.n - 1;                          // last item with first_wg <= wg
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.first_wg[mid] <= wg) lo = mid;
        else hi = mid - 1;
    }
    const GgFinishItem& it = b.item[lo];
    const int local = wg - b.first_wg[lo];
    if (it.kind == 0) {
        GgWgradFinishParams p;
        p.g = it.src; p.dst = it.dst; p.O = it.O; p.I = it.I; p.T = it.T; p.C8 = it.C8; p.O8 = it.O8; p.accumulate = it.accumulate;
        p.alpha = it.alpha; p.nsplit = it.nsplit;
        const int gx = (it.O + 31) / 32;
        gg_wgrad_finish_body(p, tile, local % gx, local / gx);
    } else if (it.kind == 2) {                         // dst += alpha * src, 1024 elements per workgroup
        const long long e0 = (long long)local * 1024 + threadIdx.x * 4;
        if (e0 + 3 < it.O && !(((unsigned long long)it.src | (unsigned long long)it.dst) & 15)) {
            const f32x4 a = *(const f32x4*)(it.src + e0);
            f32x4 d = *(const f32x4*)(it.dst + e0);
            d[0] += it.alpha * a[0]; d[1] += it.alpha * a[1]; d[2] += it.alpha * a[2]; d[3] += it.alpha * a[3];
            *(f32x4*)(it.dst + e0) = d;
        } else {
            for (int e = 0; e < 4; ++e)
                if (e0 + e < it.O) it.dst[e0 + e] += it.alpha * it.src[e0 + e];
        }
    } else {
        const int gx = (it.T + 63) / 64;
        gg_colsum_finish_body(it.src, it.dst, it.O, it.I, it.T, it.alpha, *(float (*)[4][64])&tile[0][0], local % gx, local / gx, it.C8);
    }
}

// ---- many split-K slice reductions in ONE launch ----------------------------------------------------------------------------------------
// Weight gradients split over more than 16 k-slices (thin layers: a few thousand outputs, 32..512 slices) ended in one
// gg_splitk_reduce launch each (~70 per step) before their finish was queued. The slice stacks are kept instead (kernels.FinishQueue) and
// a flush first folds all of them here - slice 0 += slices 1.. in place, fixed order: a workgroup per 64 consecutive outputs (256-byte rows
// of the stack), its four wavefronts interleaved over the slices and combined through LDS, as in gg_splitk_reduce_kernel - then runs the
// finishes on slice 0.
struct GgReduceItem {
    float* src;              // (nsplit, n) fp32 slice stack; the sum is left in slice 0
    long long n;             // elements per slice
    int nsplit;
    int reserved;
};
struct GgReduceBatch {
    GgReduceItem item[GG_FM_MAX];
    int first_wg[GG_FM_MAX + 1];
    int n;
};
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_reduce_multi_kernel(GgReduceBatch b) {
    GG_SHARED float part[3][64];
    const int wg = blockIdx.x;
    int lo = 0, hi = b.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.first_wg[mid] <= wg) lo = mid;
        else hi = mid - 1;
    }
    const GgReduceItem& it = b.item[lo];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long idx = (long long)(wg - b.first_wg[lo]) * 64 + lane;
    const bool valid = idx < it.n;
    float s = 0.f;
    if (valid) {
        const float* src = it.src + idx;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int ks = wave;
        for (; ks + 12 < it.nsplit; ks += 16) {
            s0 += src[(long long)ks * it.n];
            s1 += src[(long long)(ks + 4) * it.n];
            s2 += src[(long long)(ks + 8) * it.n];
            s3 += src[(long long)(ks + 12) * it.n];
        }
        for (; ks < it.nsplit; ks += 4) s0 += src[(long long)ks * it.n];
        s = (s0 + s1) + (s2 + s3);
    }
    if (wave) part[wave - 1][lane] = s;
    gg_sync();
    if (wave == 0 && valid) it.src[idx] = s + part[0][lane] + part[1][lane] + part[2][lane];
}

