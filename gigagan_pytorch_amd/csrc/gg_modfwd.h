// gg_modfwd.h — the no-grad forward of the adaptive / style-modulated convolution (reference AdaptiveConv2DMod.forward,
// gp.py:344-409, + Noise gp.py:925-940 + leaky_relu gp.py:109): what the discriminator step's generator pass and `generate()` run.
//
//   gg_modw_kernel     per layer, ONE launch: s = mod + 1, a = softmax(kernel_mod), the demodulation coefficients
//                      d[b,o] = rsqrt(max(sum_{i,k} (sum_n a_n W_n[o,i,k] s_i)^2, eps)) through the Gram matrix of the bank
//                      (sum_k W_n W_m per (o, i): computed once per workgroup, so the per-sample cost is 3 I multiply-adds
//                      instead of N I k^2), and - where the bank is small next to the activation - the reference's own
//                      per-sample weights d[b,o] s[b,i] sum_n a[b,n] W_n[o,i,k] in bf16, laid out for the consumer kernel.
//   gg_sconv_kernel    the narrow high-resolution layers (16 / 32 / 64 input channels, <= 32 output channels, 128x128 and
//                      256x256) as a streaming direct convolution on per-sample weights: HBM-bound work (2 C + 2 O bytes per
//                      pixel against 18 C O flops), so no im2col and no LDS round trip for the activation: a wavefront owns 32
//                      consecutive pixels of a row, its MFMA B fragments (pixels x 16 channels) are 16-byte global loads
//                      straight into registers for each of the 9 taps (neighbouring taps hit the same lines in the vector
//                      L1), the per-sample filter bank (<= 36 KiB) sits in LDS as ready-made A fragments, noise + leaky-relu
//                      run on the accumulators and each lane stores 8-byte channel quads of its pixel.
// Algorithmic work: 2 b O I 9 H W flops; bytes: (I + O) * 2 per pixel + the bank.
#pragma once
#include "gg_device.h"

#define GG_MW_NMAX 4
#define GG_MW_BMAX 64          // samples per launch
#define GG_MW_WMAX 9216        // N * I * T floats staged per workgroup (36 KiB: two 512-channel 3x3 kernels)
#define GG_MW_GMAX 1536        // pairs * I floats (Gram rows)
#define GG_MW_SMAX 4608        // samples * I floats of weight scales staged per pass (the upper half of the bank's LDS area)

struct GgModWParams {
    const float* w;        // (N, O, I, T) fp32 parameter layout
    const float* mod;      // (b, I)
    const float* kmod;     // (b, N) or null (N == 1)
    float* s;              // (b, Ip) out, optional
    float* a;              // (b, N) out, optional
    float* d;              // (b, Op) out, optional (ones when demod == 0)
    bf16_t* wmix;          // per-sample weights out, optional
    int layout;            // 1: [b][O][T][I] rows of T*I (implicit-GEMM weight operand per image); 2: [b][T][I/16][32][16]
    int b, N, O, I, T, Ip, Op;
    int demod;
    float eps;
    int mod_ld, kmod_ld;   // row pitches of mod / kmod in floats (they are column slices of the style network's output)
    const float* xs;       // optional (b, I) extra scale of the INPUT activation (skip-layer excitation, gp.py:1023-1024) folded into
    int xs_ld;             // s and the per-sample weights - not into the demodulation, which the reference computes from mod + 1 alone
    int bc;                // samples per workgroup: grid = (O, ceil(b / bc)); every workgroup re-derives the Gram rows of its channel
    const float* gram;     // optional [pair][O][I]: the bank's Gram rows, refreshed with the packed operands after each optimizer step
                           // (gg_weights.h kind 2); a coefficient-only workgroup then never touches the bank itself
    int fast;              // set by the host: coefficient-only item with cached Gram rows and I <= 512 -> gg_modw_coef_body (a workgroup per
                           // channel pair, Gram rows in registers, no LDS, no workgroup barriers): grid ceil(O / 2) workgroups
    float* insc;           // optional (b, N * Ip) out: a[b,n] * s[b,i] - the per-(sample, stacked channel) input scale of the shared-
                           // bank convolution with the N kernels stacked along the reduction (gg_conv3_kernel SCALED)
};

// one launch for MANY layers (the generator's no-grad forward: every layer's modulation comes out of ONE style vector, gp.py:1160-
// 1175, so all coefficient / per-sample-weight work of a forward is known before its first convolution): workgroup -> (item, channel,
// sample chunk) through the prefix table
#define GG_MW_MAX_ITEMS 16
struct GgModWMulti {
    int n;
    int first_block[GG_MW_MAX_ITEMS + 1];
    GgModWParams item[GG_MW_MAX_ITEMS];
};

GG_DEVICE float gg_mw_wave_sum(float v) {
    v += gg_shfl_xor(v, 1); v += gg_shfl_xor(v, 2); v += gg_shfl_xor(v, 4);
    v += gg_shfl_xor(v, 8); v += gg_shfl_xor(v, 16); v += gg_shfl_xor(v, 32);
    return v;
}

// grid: (O, ceil(b / bc)) workgroups of 256 threads: workgroup (o, c) owns output channel o for samples c*bc .. c*bc + bc - 1.
// N (kernels in the bank) is a template parameter and every global load is unconditional (indices clamped, results weighted by
// 0 / 1): with run-time loop bounds and loads under conditions the compiler emitted one branch + `s_waitcnt vmcnt(0)` per load,
// i.e. a chain of ~60 dependent memory round trips per workgroup (50 us for a 512-channel layer; measured, profiles/).
template <int N>
GG_DEVICE void gg_modw_body(const GgModWParams& p, int o, int chunk, float* wl, float* gram, float (*a_s)[GG_MW_NMAX], float* d_s,
                            float* ssc = nullptr) {
    constexpr int NP = N * (N + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int IT = p.I * p.T;
    const int b_lo = chunk * p.bc;
    const int b_hi = b_lo + p.bc < p.b ? b_lo + p.bc : p.b;
    // s / insc / the zero padding of d of this chunk's samples: workgroup o writes the rows b_lo + o, b_lo + o + O, ... (spread over
    // the channels' workgroups: one workgroup writing all rows was a 32-deep chain of dependent round trips - the launch's tail)
    if (p.s || p.insc || (p.d && p.Op > p.O))
        for (int row = b_lo + o; row < b_hi; row += p.O) {
            if (p.s)
                for (int i = tid; i < p.Ip; i += 256) {
                    const int ic = i < p.I ? i : p.I - 1;
                    const float v = (p.mod[(long long)row * p.mod_ld + ic] + 1.f) * (p.xs ? p.xs[(long long)row * p.xs_ld + ic] : 1.f);
                    p.s[(long long)row * p.Ip + i] = i < p.I ? v : 0.f;
                }
            if (p.insc) {       // a[row, n] * s[row, i] over the stacked channel axis (n, i): the softmax is re-derived here
                float av[GG_MW_NMAX] = {1.f, 0.f, 0.f, 0.f};
                if (N > 1) {
                    float kv[N], mx = -3.0e38f, sum = 0.f;
#pragma unroll
                    for (int n = 0; n < N; ++n) { kv[n] = p.kmod[(long long)row * p.kmod_ld + n]; mx = kv[n] > mx ? kv[n] : mx; }
#pragma unroll
                    for (int n = 0; n < N; ++n) { kv[n] = gg_expf(kv[n] - mx); sum += kv[n]; }
#pragma unroll
                    for (int n = 0; n < N; ++n) av[n] = kv[n] / sum;
                }
                for (int i = tid; i < p.Ip; i += 256) {
                    const int ic = i < p.I ? i : p.I - 1;
                    const float v = (p.mod[(long long)row * p.mod_ld + ic] + 1.f) * (p.xs ? p.xs[(long long)row * p.xs_ld + ic] : 1.f);
#pragma unroll
                    for (int n = 0; n < N; ++n) p.insc[((long long)row * N + n) * p.Ip + i] = i < p.I ? av[n] * v : 0.f;
                }
            }
            if (p.d)
                for (int c = p.O + tid; c < p.Op; c += 256) p.d[(long long)row * p.Op + c] = 0.f;
        }
    // cached Gram rows of this channel: fetched into registers FIRST, so that they travel with the bank rows instead of costing a
    // round trip of their own behind the first barrier (a workgroup of this kernel is a chain of dependent memory round trips)
    constexpr int GPT = (GG_MW_GMAX + 255) / 256;
    float greg[GPT];
    const bool gcached = p.demod && p.gram != nullptr;
    const int gtot = NP * p.I;
    if (gcached) {
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int f = tid + 256 * j < gtot ? tid + 256 * j : gtot - 1;
            const int pair = f / p.I, i = f - pair * p.I;
            greg[j] = p.gram[((long long)pair * p.O + o) * p.I + i];
        }
    }
    // the bank rows of this channel: 16-byte loads, four in flight per thread (I % 4 == 0: rows are 16-byte aligned). Not needed
    // when the Gram rows are cached and no per-sample weights are asked for (workgroup-uniform)
    if (p.wmix || !p.gram) {
        const int ivn = IT >> 2;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float* src = p.w + ((long long)n * p.O + o) * IT;
            for (int v0 = tid; v0 < ivn; v0 += 1024) {
                f32x4 r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int v = v0 + 256 * u < ivn ? v0 + 256 * u : ivn - 1;
                    r[u] = *(const f32x4*)(src + v * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (v0 + 256 * u < ivn) *(f32x4*)(wl + n * IT + (v0 + 256 * u) * 4) = r[u];
            }
        }
    }
    if (tid >= b_lo && tid < b_hi) {
        float av[GG_MW_NMAX] = {1.f, 0.f, 0.f, 0.f};
        if (N > 1) {
            float kv[N], mx = -3.0e38f, sum = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) { kv[n] = p.kmod[(long long)tid * p.kmod_ld + n]; mx = kv[n] > mx ? kv[n] : mx; }
#pragma unroll
            for (int n = 0; n < N; ++n) { kv[n] = gg_expf(kv[n] - mx); sum += kv[n]; }
#pragma unroll
            for (int n = 0; n < N; ++n) av[n] = kv[n] / sum;
        }
#pragma unroll
        for (int n = 0; n < GG_MW_NMAX; ++n) a_s[tid][n] = av[n];
        if (p.a && o == 0)
#pragma unroll
            for (int n = 0; n < N; ++n) p.a[tid * N + n] = av[n];
    }
    if (gcached) {
#pragma unroll
        for (int j = 0; j < GPT; ++j)
            if (tid + 256 * j < gtot) gram[tid + 256 * j] = greg[j];
    }
    gg_sync();
    if (p.demod) {
        if (!p.gram) {
            int pair = 0;
#pragma unroll
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int m = n; m < N; ++m, ++pair)
                    for (int i = tid; i < p.I; i += 256) {
                        float acc = 0.f;
                        for (int t = 0; t < p.T; ++t) acc += wl[n * IT + i * p.T + t] * wl[m * IT + i * p.T + t];
                        gram[pair * p.I + i] = (n == m ? 1.f : 2.f) * acc;       // the symmetric pair counted twice
                    }
        }
        gg_sync();
        // d[b] = rsqrt(sum_i s_i^2 * (a^T G_i a)): a wave takes samples b_lo + wave, +4, ...; lanes run along i; the modulation
        // rows of four samples are fetched back to back before any value is used
        for (int bb0 = b_lo + wave; bb0 < b_hi; bb0 += 16) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i0 = 0; i0 < p.I; i0 += 512) {
                float mv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int bc = bb0 + 4 * u < b_hi ? bb0 + 4 * u : b_hi - 1;
                    const float* mrow = p.mod + (long long)bc * p.mod_ld;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ic = i0 + lane + 64 * j < p.I ? i0 + lane + 64 * j : p.I - 1;
                        mv[u][j] = mrow[ic];
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + lane + 64 * j;
                    const int ic = i < p.I ? i : p.I - 1;
                    const float live = i < p.I ? 1.f : 0.f;
                    float g[NP];
#pragma unroll
                    for (int pr = 0; pr < NP; ++pr) g[pr] = gram[pr * p.I + ic] * live;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int bc = bb0 + 4 * u < b_hi ? bb0 + 4 * u : b_hi - 1;
                        float q = 0.f;
                        int pr = 0;
#pragma unroll
                        for (int n = 0; n < N; ++n)
#pragma unroll
                            for (int m = n; m < N; ++m, ++pr) q = gg_fmaf(a_s[bc][n] * a_s[bc][m], g[pr], q);     // (explicit fmas: every
                        const float sv = mv[u][j] + 1.f;                                                        //  unrolled copy rounds alike)
                        acc[u] = gg_fmaf(sv * sv, q, acc[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bb = bb0 + 4 * u;
                const float tot = gg_mw_wave_sum(acc[u]);
                if (lane == 0 && bb < b_hi) {
                    const float dv = gg_rsqrtf(tot > p.eps ? tot : p.eps);
                    d_s[bb] = dv;
                    if (p.d) p.d[(long long)bb * p.Op + o] = dv;
                }
            }
        }
    } else {
        if (tid >= b_lo && tid < b_hi) {
            d_s[tid] = 1.f;
            if (p.d) p.d[(long long)tid * p.Op + o] = 1.f;
        }
    }
    if (!p.wmix) return;
    gg_sync();
    // per-sample weights. The scalar form (one bf16 store and ~40 instructions per element: division, 64-bit offset, two LDS
    // broadcasts, a global load) was INSTRUCTION-bound: 19 M elements x 40 / 64 lanes = 12 M wave instructions = 45 us for config 2's
    // four per-image-weight layers. Vector form: a thread owns 8 consecutive input channels of one tap (one 16-byte store per
    // sample), the samples' scales (mod + 1) * xs are staged in LDS in passes of SB samples (one batch of independent loads per
    // pass: no global load inside the loop), the bank values of the unit are read once for all samples.
    const bool vec = ssc && (p.I & 7) == 0 && N * IT <= GG_MW_WMAX - GG_MW_SMAX && p.I <= GG_MW_SMAX;
    if (vec) {
        const int i8n = p.I >> 3, units = p.T * i8n;
        const int SB = GG_MW_SMAX / p.I;                                   // samples per staging pass
        for (int s0 = b_lo; s0 < b_hi; s0 += SB) {
            const int s1 = s0 + SB < b_hi ? s0 + SB : b_hi;
            gg_sync();                                                    // (the previous pass is done with ssc)
            const int cnt = (s1 - s0) * p.I;
            for (int e0 = tid; e0 < cnt; e0 += 256 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + 256 * u < cnt ? e0 + 256 * u : cnt - 1;
                    const int bb = s0 + e / p.I, i = e % p.I;
                    v[u] = (p.mod[(long long)bb * p.mod_ld + i] + 1.f) * (p.xs ? p.xs[(long long)bb * p.xs_ld + i] : 1.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + 256 * u < cnt) ssc[e0 + 256 * u] = v[u];
            }
            gg_sync();
            for (int un = tid; un < units; un += 256) {
                const int t = un / i8n, i0 = (un - t * i8n) * 8;
                float wv[N][8];
#pragma unroll
                for (int n = 0; n < N; ++n)
#pragma unroll
                    for (int e = 0; e < 8; ++e) wv[n][e] = wl[n * IT + (i0 + e) * p.T + t];
                for (int bb = s0; bb < s1; ++bb) {
                    const float* sp = ssc + (bb - s0) * p.I + i0;
                    const f32x4 sa = *(const f32x4*)sp, sb = *(const f32x4*)(sp + 4);
                    const float dv = d_s[bb];
                    u16x8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float m = 0.f;
#pragma unroll
                        for (int n = 0; n < N; ++n) m += a_s[bb][n] * wv[n][e];
                        o8[e] = gg_f2bf(dv * (e < 4 ? sa[e] : sb[e - 4]) * m);
                    }
                    long long off;
                    if (p.layout == 1) off = (((long long)bb * p.O + o) * p.T + t) * p.I + i0;
                    else off = ((((long long)bb * p.T + t) * (p.I >> 4) + (i0 >> 4)) * 32 + o) * 16 + (i0 & 15);
                    *(u16x8*)(p.wmix + off) = o8;
                }
            }
        }
        return;
    }
    for (int e = tid; e < IT; e += 256) {
        const int t = e / p.I, i = e - t * p.I;
        float wv[N];
#pragma unroll
        for (int n = 0; n < N; ++n) wv[n] = wl[n * IT + i * p.T + t];
        for (int bb0 = b_lo; bb0 < b_hi; bb0 += 8) {
            float sc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bc = bb0 + u < b_hi ? bb0 + u : b_hi - 1;
                sc[u] = p.mod[(long long)bc * p.mod_ld + i] + 1.f;
            }
            if (p.xs) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bc = bb0 + u < b_hi ? bb0 + u : b_hi - 1;
                    sc[u] *= p.xs[(long long)bc * p.xs_ld + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = bb0 + u;
                if (bb < b_hi) {
                    float m = 0.f;
#pragma unroll
                    for (int n = 0; n < N; ++n) m += a_s[bb][n] * wv[n];
                    const float v = d_s[bb] * sc[u] * m;
                    long long off;
                    if (p.layout == 1) off = (((long long)bb * p.O + o) * p.T + t) * p.I + i;
                    else off = ((((long long)bb * p.T + t) * (p.I >> 4) + (i >> 4)) * 32 + o) * 16 + (i & 15);
                    p.wmix[off] = gg_f2bf(v);
                }
            }
        }
    }
}

// Coefficient-only items with cached Gram rows: one WAVE per output channel. The per-workgroup chain of gg_modw_body (Gram rows ->
// LDS -> barrier -> samples) ran ~6 us per channel and a 512-channel layer is 512 workgroups: 54 us for config 2's seven wide
// layers even with the Gram rows cached. Here a lane keeps the 3 x 8 Gram values of its channel in registers (i = lane + 64 j,
// I <= 512), broadcasts a sample's softmax weights with readlane, and the wave walks the samples four at a time (their modulation
// rows fetched back to back). Workgroup wg of the item's nwg also writes the rows wg, wg + nwg, ... of s / a / insc.
template <int N>
GG_DEVICE void gg_modw_coef_body(const GgModWParams& p, int wg, int nwg) {
    constexpr int NP = N * (N + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int row = wg; row < p.b; row += nwg) {
        float av[GG_MW_NMAX] = {1.f, 0.f, 0.f, 0.f};
        if (N > 1) {
            float kv[N], mx = -3.0e38f, sum = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) { kv[n] = p.kmod[(long long)row * p.kmod_ld + n]; mx = kv[n] > mx ? kv[n] : mx; }
#pragma unroll
            for (int n = 0; n < N; ++n) { kv[n] = gg_expf(kv[n] - mx); sum += kv[n]; }
#pragma unroll
            for (int n = 0; n < N; ++n) av[n] = kv[n] / sum;
        }
        if (p.a && tid < N) p.a[row * N + tid] = av[tid < GG_MW_NMAX ? tid : 0];
        for (int i = tid; i < p.Ip; i += 256) {
            const int ic = i < p.I ? i : p.I - 1;
            const float v = (p.mod[(long long)row * p.mod_ld + ic] + 1.f) * (p.xs ? p.xs[(long long)row * p.xs_ld + ic] : 1.f);
            if (p.s) p.s[(long long)row * p.Ip + i] = i < p.I ? v : 0.f;
            if (p.insc) {
#pragma unroll
                for (int n = 0; n < N; ++n) p.insc[((long long)row * N + n) * p.Ip + i] = i < p.I ? av[n] * v : 0.f;
            }
        }
        if (p.d)
            for (int c = p.O + tid; c < p.Op; c += 256) p.d[(long long)row * p.Op + c] = 0.f;
    }
    // the workgroup owns the channel pair (2 wg, 2 wg + 1); its four waves split the samples (a wave walking all 32 samples of a
    // channel was a ~30 us serial chain). The pair shares every modulation value and rides on packed fp32 FMAs (v_pk_fma_f32).
    const int o0 = wg * 2;
    if (o0 >= p.O || !p.d) return;
    const bool two = o0 + 1 < p.O;
    const int per = ((p.b + 3) / 4 + 3) & ~3;                  // samples per wave, a multiple of the batch of four
    const int s_lo = wave * per, s_hi = s_lo + per < p.b ? s_lo + per : p.b;
    if (s_lo >= p.b) return;
    if (!p.demod) {
        for (int bb = s_lo + lane; bb < s_hi; bb += 64) {
            p.d[(long long)bb * p.Op + o0] = 1.f;
            if (two) p.d[(long long)bb * p.Op + o0 + 1] = 1.f;
        }
        return;
    }
    // this lane's sample (lane < b <= 64): softmax over the kernels; other lanes' values are fetched with readlane below
    float av[GG_MW_NMAX] = {1.f, 0.f, 0.f, 0.f};
    if (N > 1) {
        const int sb = lane < p.b ? lane : p.b - 1;
        float kv[N], mx = -3.0e38f, sum = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) { kv[n] = p.kmod[(long long)sb * p.kmod_ld + n]; mx = kv[n] > mx ? kv[n] : mx; }
#pragma unroll
        for (int n = 0; n < N; ++n) { kv[n] = gg_expf(kv[n] - mx); sum += kv[n]; }
#pragma unroll
        for (int n = 0; n < N; ++n) av[n] = kv[n] / sum;
    }
    f32x2 g[NP][8];                                             // (channel o0, channel o0 + 1) pairs
#pragma unroll
    for (int pr = 0; pr < NP; ++pr)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane + 64 * j;
            const long long base = ((long long)pr * p.O + o0) * p.I + (i < p.I ? i : p.I - 1);
            const float v0 = p.gram[base], v1 = p.gram[base + (two ? p.I : 0)];
            g[pr][j] = (f32x2){i < p.I ? v0 : 0.f, i < p.I ? v1 : 0.f};
        }
    auto load_batch = [&](float (&mv)[4][8], int bb0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int bc = bb0 + u < p.b ? bb0 + u : p.b - 1;
            const float* mrow = p.mod + (long long)bc * p.mod_ld;
#pragma unroll
            for (int j = 0; j < 8; ++j) mv[u][j] = mrow[lane + 64 * j < p.I ? lane + 64 * j : p.I - 1];
        }
    };
    float cur[4][8], nxt[4][8];
    load_batch(cur, s_lo);
    for (int bb0 = s_lo; bb0 < s_hi; bb0 += 4) {
        load_batch(nxt, bb0 + 4 < s_hi ? bb0 + 4 : bb0);        // the next four rows fly while these are reduced
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int bc = bb0 + u < p.b ? bb0 + u : p.b - 1;
            f32x2 cp[NP];
            int pr = 0;
#pragma unroll
            for (int n = 0; n < N; ++n)
#pragma unroll
                for (int m = n; m < N; ++m, ++pr) {
                    const float c = gg_readlane(av[n], bc) * gg_readlane(av[m], bc);
                    cp[pr] = (f32x2){c, c};
                }
            f32x2 acc = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sv = cur[u][j] + 1.f;
                const float s2 = sv * sv;
                f32x2 q = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NP; ++k) q = (f32x2){gg_fmaf(cp[k][0], g[k][j][0], q[0]), gg_fmaf(cp[k][1], g[k][j][1], q[1])};
                acc = (f32x2){gg_fmaf(s2, q[0], acc[0]), gg_fmaf(s2, q[1], acc[1])};     // (explicit fmas: every unrolled copy rounds alike)
            }
            const float t0 = gg_wave_sum_all(acc[0]), t1 = gg_wave_sum_all(acc[1]);
            if (lane == 0 && bb0 + u < s_hi) {
                p.d[(long long)(bb0 + u) * p.Op + o0] = gg_rsqrtf(t0 > p.eps ? t0 : p.eps);
                if (two) p.d[(long long)(bb0 + u) * p.Op + o0 + 1] = gg_rsqrtf(t1 > p.eps ? t1 : p.eps);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[u][j] = nxt[u][j];
    }
}

template <int N>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modw_kernel(GgModWParams p) {
    GG_SHARED __attribute__((aligned(16))) float wl[GG_MW_WMAX];   // [n][i*T + t]
    GG_SHARED float gram[GG_MW_GMAX];                     // [pair][i], pair = (n, m >= n) in row-major upper-triangle order
    GG_SHARED float a_s[GG_MW_BMAX][GG_MW_NMAX];
    GG_SHARED float d_s[GG_MW_BMAX];
    gg_modw_body<N>(p, blockIdx.x, blockIdx.y, wl, gram, a_s, d_s);
}

// ONE launch for all items (N <= 2 here: the host sends banks of more kernels to gg_modw_kernel, whose instantiations need > 256
// registers): coefficient-only items with cached Gram rows run gg_modw_coef_body, the items that build per-sample weights
// gg_modw_body with their scales staged in LDS. Both kinds run side by side in the one grid.
GG_KERNEL GG_LAUNCH_BOUNDS2(256, 3) void gg_modw_multi_kernel(GgModWMulti m) {
    GG_SHARED __attribute__((aligned(16))) float wl[GG_MW_WMAX];
    GG_SHARED float gram[GG_MW_GMAX];
    GG_SHARED float a_s[GG_MW_BMAX][GG_MW_NMAX];
    GG_SHARED float d_s[GG_MW_BMAX];
    float* const ssc = wl + (GG_MW_WMAX - GG_MW_SMAX);       // staged weight scales share the bank area (banks <= half of it)
    // (the table is read through the kernarg pointer: run-time indexing of the by-value struct would copy it to scratch)
    const GgModWMulti* mp = gg_late_params(m);
    int it = 0;
    while (it + 1 < mp->n && (int)blockIdx.x >= mp->first_block[it + 1]) ++it;
    const GgModWParams p = mp->item[it];
    const int rel = blockIdx.x - mp->first_block[it];
    if (p.fast) {           // (item-uniform: every workgroup of the item takes this branch)
        const int nwg = mp->first_block[it + 1] - mp->first_block[it];
        if (p.N == 1) gg_modw_coef_body<1>(p, rel, nwg);
        else gg_modw_coef_body<2>(p, rel, nwg);
        return;
    }
    const int o = rel % p.O, chunk = rel / p.O;
    if (p.N == 1) gg_modw_body<1>(p, o, chunk, wl, gram, a_s, d_s, ssc);
    else gg_modw_body<2>(p, o, chunk, wl, gram, a_s, d_s, ssc);
}

// ---- streaming direct convolution on per-sample weights ------------------------------------------------------------------

struct GgSconvParams {
    const bf16_t* x;        // [b][H][W][C]
    const bf16_t* w;        // [b][9][C/16][32][16] (gg_modw layout 2; rows >= O are zero)
    long long w_bs;         // elements between the banks of consecutive images (0: one shared bank)
    bf16_t* y;              // [b][H][W][O]
    const float* noise;     // [b][H*W] or null
    const float* noise_w;   // [O] (with noise)
    int b, H, W, O;
    int act;                // 0 none, 1 leaky relu
    float slope;
    const float* xs;        // optional [b][C]: a per-sample scale of the INPUT channels (the skip-layer excitation, gp.py:1023-1024),
                            // applied to the bank as it is parked in LDS - the convolution is linear in (x * xs) = weights * xs
    int rows_per_item;      // a wavefront's work item: a 32-pixel-wide strip of this many rows
    int items_per_wave;     // items a wavefront walks through (a workgroup of 4 wavefronts stays inside one image)
};

// one image row of a strip as MFMA B fragments: (dx = -1, 0, +1) x (C / 16) 16-byte loads per lane, ALWAYS issued and from a
// clamped address (a load under a per-lane condition is branched around and waited for on the spot); what lies outside the image
// is zeroed when the fragment is used (gg_sc_row_mfma), many instructions later, so the loads stay in flight meanwhile
template <int C>
GG_DEVICE void gg_sc_load_row(u16x8 (&row)[3 * (C / 16)], const bf16_t* xi, int iy, int x0, int H, int W, int pl, int hi) {
    constexpr int KC = C / 16;
    const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int ix = x0 + pl + dx - 1;
        const int cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        const bf16_t* src = xi + ((long long)cy * W + cx) * C + hi * 8;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) row[dx * KC + kc] = *(const u16x8*)(src + kc * 16);
    }
}

// the 3 * C/16 MFMAs of one kernel row; `iy` / x0 say which fragments lie outside the image (zeroed here)
template <int C>
GG_DEVICE f32x16 gg_sc_row_mfma(f32x16 acc, const u16x8 (&row)[3 * (C / 16)], const bf16_t* wl, int ky, int iy, int x0,
                                int H, int W, int pl, int hi) {
    constexpr int KC = C / 16;
    // the bank is loop invariant, and the compiler would park all 9 * C/16 fragments in registers for the whole strip (72 of
    // them at C = 32: one wave per SIMD): an OFFSET it cannot see through keeps the ds_read_b128 in the loop. (Laundering the
    // pointer itself loses the LDS address space: the reads become flat loads, which count on vmcnt AND lgkmcnt out of order,
    // and every wait turns into vmcnt(0): no prefetch left.)
    int opaque = 0;
#if !defined(GG_HOST_EMULATION)
    asm volatile("" : "+v"(opaque));
#endif
    wl += opaque;
    const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool row_in = iy >= 0 && iy < H;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int ix = x0 + pl + dx - 1;
        const bool in = row_in && ix >= 0 && ix < W;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
        {
            const u16x8 wf = *(const u16x8*)(wl + (((ky * 3 + dx) * KC + kc) * 32 + pl) * 16 + hi * 8);   // A fragment from the bank in LDS
            acc = gg_mfma_32x32x16_bf16(wf, in ? row[dx * KC + kc] : z, acc);                              // D[channel][pixel]
        }
    }
    return acc;
}

// one output row yy: the row TWO below is fetched first (two rows of loads in flight per wavefront: the memory latency is ~8x the
// 9 * C/16 MFMAs of a row), then the three rows already in registers are multiplied, noise and activation applied, stored
template <int C, int AHEAD>
GG_DEVICE void gg_sc_step(const GgSconvParams& p, const bf16_t* wl, const bf16_t* xi, long long img_pix0, int yy,
                          int x0, const u16x8 (&top)[3 * (C / 16)], const u16x8 (&mid)[3 * (C / 16)], const u16x8 (&bot)[3 * (C / 16)],
                          u16x8 (&next)[3 * (C / 16)], int pl, int hi, const float* nw, bool live) {
    // `live` (wave uniform) is false for the padding steps of the unrolled ring past the strip's last row: their loads are
    // clamped into the image and nothing is stored (no control flow around the register rotation: no copies between rows)
    const int yc = live ? yy : p.H - 1;
    const long long pix = img_pix0 + (long long)yc * p.W + x0 + pl;
    // BEFORE the row loads: waiting for it must not drain them (vmcnt is in order). Always issued (without a noise map it reads
    // the activations, in range: a pixel is >= 32 bytes there) so that no branch splits the step
    const float* nsrc = p.noise ? p.noise : (const float*)p.x;
    float nz = nsrc[pix];
    nz = p.noise ? nz : 0.f;
    gg_sc_load_row<C>(next, xi, yy + AHEAD, x0, p.H, p.W, pl, hi);      // AHEAD = 2: `next` is a fourth buffer; 1: `next` IS `bot`
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = gg_sc_row_mfma<C>(acc, top, wl, 0, yy - 1, x0, p.H, p.W, pl, hi);
    acc = gg_sc_row_mfma<C>(acc, mid, wl, 1, yy, x0, p.H, p.W, pl, hi);
    acc = gg_sc_row_mfma<C>(acc, bot, wl, 2, yy + 1, x0, p.H, p.W, pl, hi);
    int opaque = 0;       // as for the bank: keep the four ds_read_b128 of the noise weights in the loop instead of 16 registers
#if !defined(GG_HOST_EMULATION)
    asm volatile("" : "+v"(opaque));
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch0 = 8 * q + 4 * hi;
        if (live && ch0 < p.O) {
            u16x4 o;
            const f32x4 w4 = *(const f32x4*)(nw + opaque + ch0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[q * 4 + e] + nz * w4[e];
                if (p.act == 1) v = v > 0.f ? v : v * p.slope;
                o[e] = gg_f2bf(v);
            }
            *(u16x4*)(p.y + pix * p.O + ch0) = o;
        }
    }
}

template <int C>
GG_KERNEL GG_LAUNCH_BOUNDS2(256, (C <= 32 ? 3 : 2)) void gg_sconv_kernel(GgSconvParams p) {
    constexpr int KC = C / 16;
    constexpr int NV = 9 * KC * 32 * 2;                                  // 16-byte vectors of one filter bank
    GG_SHARED __attribute__((aligned(16))) bf16_t wl[9 * KC * 32 * 16];
    GG_SHARED __attribute__((aligned(16))) float nw[32];                 // noise weights by output channel (0 without noise / beyond O)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 32) {
        const float v = p.noise_w ? p.noise_w[tid < p.O ? tid : 0] : 0.f;
        nw[tid] = (p.noise_w && tid < p.O) ? v : 0.f;
    }
    const int strips = p.W >> 5;
    const int chunks = (p.H + p.rows_per_item - 1) / p.rows_per_item;
    const int items = strips * chunks;                                   // per image; strip index fastest
    const int per_wg = 4 * p.items_per_wave;
    const int wgs_per_img = (items + per_wg - 1) / per_wg;
    const int img = blockIdx.x / wgs_per_img, first = (blockIdx.x - img * wgs_per_img) * per_wg;
    {   // this image's bank -> LDS, all of a thread's loads in flight together
        const u16x8* src = (const u16x8*)(p.w + (long long)img * p.w_bs);
        constexpr int PER = (NV + 255) / 256;
        u16x8 r[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) r[u] = src[tid + 256 * u < NV ? tid + 256 * u : NV - 1];
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + 256 * u < NV) {
                u16x8 v = r[u];
                if (p.xs) {         // vector index -> ((tap * KC + kc) * 32 + o) * 2 + half: input channels kc * 16 + half * 8 + 0..7
                    const int vi = tid + 256 * u;
                    const float* sp = p.xs + (long long)img * C + ((vi >> 6) % KC) * 16 + (vi & 1) * 8;
                    const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = gg_f2bf(gg_bf2f(v[e]) * s0[e]);
                        v[e + 4] = gg_f2bf(gg_bf2f(v[e + 4]) * s1[e]);
                    }
                }
                ((u16x8*)wl)[tid + 256 * u] = v;
            }
    }
    gg_sync();
    const int pl = lane & 31, hi = lane >> 5;
    const bf16_t* xi = p.x + (long long)img * p.H * p.W * C;
    const long long img_pix0 = (long long)img * p.H * p.W;
    for (int it = 0; it < p.items_per_wave; ++it) {
        const int item = first + it * 4 + wave;
        if (item >= items) break;
        const int cy = item / strips, x0 = (item - cy * strips) << 5;
        const int y_lo = cy * p.rows_per_item;
        const int y_hi = y_lo + p.rows_per_item < p.H ? y_lo + p.rows_per_item : p.H;
        if constexpr (C <= 32) {
            // four register rows rotate through the roles (above, centre, below, in flight): the loop is unrolled by four so
            // that every access is statically indexed; each input row is fetched once per strip, two rows ahead of its use
            u16x8 r0[3 * KC], r1[3 * KC], r2[3 * KC], r3[3 * KC];
            gg_sc_load_row<C>(r0, xi, y_lo - 1, x0, p.H, p.W, pl, hi);
            gg_sc_load_row<C>(r1, xi, y_lo, x0, p.H, p.W, pl, hi);
            gg_sc_load_row<C>(r2, xi, y_lo + 1, x0, p.H, p.W, pl, hi);
            for (int yy = y_lo; yy < y_hi; yy += 4) {
                gg_sc_step<C, 2>(p, wl, xi, img_pix0, yy, x0, r0, r1, r2, r3, pl, hi, nw, true);
                gg_sc_step<C, 2>(p, wl, xi, img_pix0, yy + 1, x0, r1, r2, r3, r0, pl, hi, nw, yy + 1 < y_hi);
                gg_sc_step<C, 2>(p, wl, xi, img_pix0, yy + 2, x0, r2, r3, r0, r1, pl, hi, nw, yy + 2 < y_hi);
                gg_sc_step<C, 2>(p, wl, xi, img_pix0, yy + 3, x0, r3, r0, r1, r2, pl, hi, nw, yy + 3 < y_hi);
            }
        } else {
            // 64 channels: a row is 12 fragments (48 registers), so three rows rotate and the row below is fetched at the top
            // of its own step (its loads fly during the MFMAs of the two rows above)
            u16x8 r0[3 * KC], r1[3 * KC], r2[3 * KC];
            gg_sc_load_row<C>(r0, xi, y_lo - 1, x0, p.H, p.W, pl, hi);
            gg_sc_load_row<C>(r1, xi, y_lo, x0, p.H, p.W, pl, hi);
            for (int yy = y_lo; yy < y_hi; yy += 3) {
                gg_sc_step<C, 1>(p, wl, xi, img_pix0, yy, x0, r0, r1, r2, r2, pl, hi, nw, true);
                gg_sc_step<C, 1>(p, wl, xi, img_pix0, yy + 1, x0, r1, r2, r0, r0, pl, hi, nw, yy + 1 < y_hi);
                gg_sc_step<C, 1>(p, wl, xi, img_pix0, yy + 2, x0, r2, r0, r1, r1, pl, hi, nw, yy + 2 < y_hi);
            }
        }
    }
}
