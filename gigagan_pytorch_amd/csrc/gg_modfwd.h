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
    int bc;                // samples per workgroup: grid = (O, ceil(b / bc)); every workgroup re-derives the Gram rows of its channel
};

GG_DEVICE float gg_mw_wave_sum(float v) {
    v += gg_shfl_xor(v, 1); v += gg_shfl_xor(v, 2); v += gg_shfl_xor(v, 4);
    v += gg_shfl_xor(v, 8); v += gg_shfl_xor(v, 16); v += gg_shfl_xor(v, 32);
    return v;
}

// grid: (O, ceil(b / bc)) workgroups of 256 threads: workgroup (o, c) owns output channel o for samples c*bc .. c*bc + bc - 1
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modw_kernel(GgModWParams p) {
    GG_SHARED float wl[GG_MW_WMAX];                       // [n][i*T + t]
    GG_SHARED float gram[GG_MW_GMAX];                     // [pair][i], pair = (n, m >= n) in row-major upper-triangle order
    GG_SHARED float a_s[GG_MW_BMAX][GG_MW_NMAX];
    GG_SHARED float d_s[GG_MW_BMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x;
    const int IT = p.I * p.T;
    const int b_lo = blockIdx.y * p.bc;
    const int b_hi = b_lo + p.bc < p.b ? b_lo + p.bc : p.b;
    // s / a / the zero padding of d of this chunk's samples: written by the workgroups of channel 0
    if (o == 0)
        for (int row = b_lo; row < b_hi; ++row) {
            if (p.s)
                for (int i = tid; i < p.Ip; i += 256) p.s[(long long)row * p.Ip + i] = i < p.I ? p.mod[(long long)row * p.mod_ld + i] + 1.f : 0.f;
            if (p.d)
                for (int c = p.O + tid; c < p.Op; c += 256) p.d[(long long)row * p.Op + c] = 0.f;
        }
    {   // the bank rows of this channel: 16-byte loads, several in flight per thread (I % 4 == 0: rows are 16-byte aligned)
        const int nv = (p.N * IT) >> 2;
        const int ivn = IT >> 2;
        for (int v0 = tid; v0 < nv; v0 += 256 * 4) {
            f32x4 r[4];      // loads are unconditional (clamped index): a load under a per-lane condition is branched around and
#pragma unroll       // waited for on its own - a chain of dependent round trips instead of four in flight
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + u * 256 < nv ? v0 + u * 256 : nv - 1;
                const int n = v / ivn, e4 = v - n * ivn;
                r[u] = *(const f32x4*)(p.w + ((long long)n * p.O + o) * IT + e4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + u * 256;
                if (v < nv) *(f32x4*)(wl + v * 4) = r[u];
            }
        }
    }
    if (tid >= b_lo && tid < b_hi) {
        float a0 = 1.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (p.kmod && p.N > 1) {
            const float* km = p.kmod + (long long)tid * p.kmod_ld;
            const float k0 = km[0], k1 = km[1], k2 = p.N > 2 ? km[2] : -3.0e38f, k3 = p.N > 3 ? km[3] : -3.0e38f;
            float mx = k0 > k1 ? k0 : k1;
            mx = k2 > mx ? k2 : mx;
            mx = k3 > mx ? k3 : mx;
            a0 = gg_expf(k0 - mx); a1 = gg_expf(k1 - mx);
            a2 = p.N > 2 ? gg_expf(k2 - mx) : 0.f; a3 = p.N > 3 ? gg_expf(k3 - mx) : 0.f;
            const float inv = 1.f / (a0 + a1 + a2 + a3);
            a0 *= inv; a1 *= inv; a2 *= inv; a3 *= inv;
        }
        a_s[tid][0] = a0; a_s[tid][1] = a1; a_s[tid][2] = a2; a_s[tid][3] = a3;
        if (p.a && o == 0) {
            p.a[tid * p.N] = a0;
            if (p.N > 1) p.a[tid * p.N + 1] = a1;
            if (p.N > 2) p.a[tid * p.N + 2] = a2;
            if (p.N > 3) p.a[tid * p.N + 3] = a3;
        }
    }
    gg_sync();
    if (p.demod) {
        int pair = 0;
        for (int n = 0; n < p.N; ++n)
            for (int m = n; m < p.N; ++m, ++pair)
                for (int i = tid; i < p.I; i += 256) {
                    float acc = 0.f;
                    for (int t = 0; t < p.T; ++t) acc += wl[n * IT + i * p.T + t] * wl[m * IT + i * p.T + t];
                    gram[pair * p.I + i] = acc;
                }
        gg_sync();
        // d[b] = rsqrt(sum_i s_i^2 * (a^T G_i a)): a wave takes samples b_lo + wave, +4, ...; lanes run along i. The modulation
        // values of FOUR samples are fetched before any of them is used (independent loads in flight, not one round trip each)
        for (int bb0 = b_lo + wave; bb0 < b_hi; bb0 += 16) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i0 = 0; i0 < p.I; i0 += 512) {
                float mv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int bb = bb0 + 4 * u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = i0 + lane + 64 * j;
                        const int bc = bb < b_hi ? bb : b_hi - 1, ic = i < p.I ? i : p.I - 1;      // unconditional loads
                        const float v = p.mod[(long long)bc * p.mod_ld + ic] + 1.f;
                        mv[u][j] = (bb < b_hi && i < p.I) ? v : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int bb = bb0 + 4 * u < b_hi ? bb0 + 4 * u : b_lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = i0 + lane + 64 * j < p.I ? i0 + lane + 64 * j : p.I - 1;     // (mv is 0 beyond I)
                        float q = 0.f;
                        int pr = 0;
                        for (int n = 0; n < p.N; ++n)
                            for (int m = n; m < p.N; ++m, ++pr)
                                q += (n == m ? 1.f : 2.f) * a_s[bb][n] * a_s[bb][m] * gram[pr * p.I + i];
                        acc[u] += mv[u][j] * mv[u][j] * q;
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
    // per-sample weights: threads run along (t, i) with i fastest, so the bf16 stores of a wave are contiguous
    for (int bb = b_lo; bb < b_hi; ++bb) {
        const float dv = d_s[bb];
        for (int e = tid; e < IT; e += 256) {
            const int t = e / p.I, i = e - t * p.I;
            float m = 0.f;
            for (int n = 0; n < p.N; ++n) m += a_s[bb][n] * wl[n * IT + i * p.T + t];
            const float v = dv * (p.mod[(long long)bb * p.mod_ld + i] + 1.f) * m;
            long long off;
            if (p.layout == 1) off = (((long long)bb * p.O + o) * p.T + t) * p.I + i;
            else off = ((((long long)bb * p.T + t) * (p.I >> 4) + (i >> 4)) * 32 + o) * 16 + (i & 15);
            p.wmix[off] = gg_f2bf(v);
        }
    }
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
    int rows_per_item;      // a wavefront's work item: a 32-pixel-wide strip of this many rows
    int items_per_wave;     // items a wavefront walks through (a workgroup of 4 wavefronts stays inside one image)
};

// one image row of a strip as MFMA B fragments: (dx = -1, 0, +1) x (C / 16) 16-byte loads per lane; pixels beyond the image
// borders are loaded from a clamped address and zeroed afterwards (a load under a per-lane condition is branched around and
// waited for one at a time), rows beyond the image are zeros without any load (the condition is wave-uniform)
template <int C>
GG_DEVICE void gg_sc_load_row(u16x8 (&row)[3 * (C / 16)], const bf16_t* xi, int iy, int x0, int H, int W, int pl, int hi) {
    constexpr int KC = C / 16;
    const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (iy < 0 || iy >= H) {
#pragma unroll
        for (int f = 0; f < 3 * KC; ++f) row[f] = z;
        return;
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int ix = x0 + pl + dx - 1;
        const bool in = ix >= 0 && ix < W;
        const int cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        const bf16_t* src = xi + ((long long)iy * W + cx) * C + hi * 8;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const u16x8 v = *(const u16x8*)(src + kc * 16);
            row[dx * KC + kc] = in ? v : z;
        }
    }
}

template <int C>
GG_DEVICE f32x16 gg_sc_row_mfma(f32x16 acc, const u16x8 (&row)[3 * (C / 16)], const bf16_t* wl, int ky, int pl, int hi) {
    constexpr int KC = C / 16;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const u16x8 wf = *(const u16x8*)(wl + (((ky * 3 + dx) * KC + kc) * 32 + pl) * 16 + hi * 8);
            acc = gg_mfma_32x32x16_bf16(wf, row[dx * KC + kc], acc);     // D[out channel][pixel]: registers run along channels
        }
    return acc;
}

// one output row: the row below is fetched first (its loads fly during the 6 * C/16 MFMAs of the two rows already in registers)
template <int C>
GG_DEVICE void gg_sc_step(const GgSconvParams& p, const bf16_t* wl, const bf16_t* xi, long long img_pix0, int yy, int x0,
                          const u16x8 (&top)[3 * (C / 16)], const u16x8 (&mid)[3 * (C / 16)], u16x8 (&bot)[3 * (C / 16)], int pl, int hi,
                          const float (&nw)[16]) {
    gg_sc_load_row<C>(bot, xi, yy + 1, x0, p.H, p.W, pl, hi);
    const long long pix = img_pix0 + (long long)yy * p.W + x0 + pl;
    const float nz = p.noise ? p.noise[pix] : 0.f;        // issued with the row's loads, consumed after the MFMAs
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = gg_sc_row_mfma<C>(acc, top, wl, 0, pl, hi);
    acc = gg_sc_row_mfma<C>(acc, mid, wl, 1, pl, hi);
    acc = gg_sc_row_mfma<C>(acc, bot, wl, 2, pl, hi);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch0 = 8 * q + 4 * hi;
        if (ch0 < p.O) {
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[q * 4 + e] + nz * nw[q * 4 + e];
                if (p.act == 1) v = v > 0.f ? v : v * p.slope;
                o[e] = gg_f2bf(v);
            }
            *(u16x4*)(p.y + pix * p.O + ch0) = o;
        }
    }
}

template <int C>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_sconv_kernel(GgSconvParams p) {
    constexpr int KC = C / 16;
    GG_SHARED __attribute__((aligned(16))) bf16_t wl[9 * KC * 32 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int strips = p.W >> 5;
    const int chunks = (p.H + p.rows_per_item - 1) / p.rows_per_item;
    const int items = strips * chunks;                                   // per image; strip index fastest
    const int per_wg = 4 * p.items_per_wave;
    const int wgs_per_img = (items + per_wg - 1) / per_wg;
    const int img = blockIdx.x / wgs_per_img, first = (blockIdx.x - img * wgs_per_img) * per_wg;
    {
        const u16x8* src = (const u16x8*)(p.w + (long long)img * p.w_bs);
        u16x8* dst = (u16x8*)wl;
        for (int v = tid; v < 9 * KC * 32 * 2; v += 256) dst[v] = src[v];
    }
    gg_sync();
    const int pl = lane & 31, hi = lane >> 5;
    const bf16_t* xi = p.x + (long long)img * p.H * p.W * C;
    const long long img_pix0 = (long long)img * p.H * p.W;
    float nw[16];            // the noise weights of this lane's 16 output channels, fetched once (0 without noise / beyond O)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = 8 * (r >> 2) + 4 * hi + (r & 3);
        const float v = p.noise_w ? p.noise_w[ch < p.O ? ch : 0] : 0.f;
        nw[r] = (p.noise_w && ch < p.O) ? v : 0.f;
    }
    for (int it = 0; it < p.items_per_wave; ++it) {
        const int item = first + it * 4 + wave;
        if (item >= items) break;
        const int cy = item / strips, x0 = (item - cy * strips) << 5;
        const int y_lo = cy * p.rows_per_item;
        const int y_hi = y_lo + p.rows_per_item < p.H ? y_lo + p.rows_per_item : p.H;
        // three register rows rotate through the roles (above, centre, below): the loop is unrolled by three so that every
        // access is statically indexed; each input row is fetched once per strip instead of three times
        u16x8 r0[3 * KC], r1[3 * KC], r2[3 * KC];
        gg_sc_load_row<C>(r0, xi, y_lo - 1, x0, p.H, p.W, pl, hi);
        gg_sc_load_row<C>(r1, xi, y_lo, x0, p.H, p.W, pl, hi);
        for (int yy = y_lo; yy < y_hi; yy += 3) {
            gg_sc_step<C>(p, wl, xi, img_pix0, yy, x0, r0, r1, r2, pl, hi, nw);
            if (yy + 1 < y_hi) gg_sc_step<C>(p, wl, xi, img_pix0, yy + 1, x0, r1, r2, r0, pl, hi, nw);
            if (yy + 2 < y_hi) gg_sc_step<C>(p, wl, xi, img_pix0, yy + 2, x0, r2, r0, r1, pl, hi, nw);
        }
    }
}
