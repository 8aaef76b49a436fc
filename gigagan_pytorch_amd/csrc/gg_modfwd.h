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
#define GG_MW_WMAX 18432       // N * I * T floats staged per workgroup (72 KiB)
#define GG_MW_GMAX 5120        // pairs * I floats (Gram rows)

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
};

GG_DEVICE float gg_mw_wave_sum(float v) {
    v += gg_shfl_xor(v, 1); v += gg_shfl_xor(v, 2); v += gg_shfl_xor(v, 4);
    v += gg_shfl_xor(v, 8); v += gg_shfl_xor(v, 16); v += gg_shfl_xor(v, 32);
    return v;
}

// grid: O workgroups of 256 threads, workgroup o owns output channel o for every sample
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modw_kernel(GgModWParams p) {
    GG_SHARED float wl[GG_MW_WMAX];                       // [n][i*T + t]
    GG_SHARED float gram[GG_MW_GMAX];                     // [pair][i], pair = (n, m >= n) in row-major upper-triangle order
    GG_SHARED float a_s[GG_MW_BMAX][GG_MW_NMAX];
    GG_SHARED float d_s[GG_MW_BMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x;
    const int IT = p.I * p.T;
    // s / a / the zero padding of d: written row by row by the workgroups in turn
    for (int row = blockIdx.x; row < p.b; row += gridDim.x) {
        if (p.s)
            for (int i = tid; i < p.Ip; i += 256) p.s[(long long)row * p.Ip + i] = i < p.I ? p.mod[(long long)row * p.I + i] + 1.f : 0.f;
        if (p.d)
            for (int c = p.O + tid; c < p.Op; c += 256) p.d[(long long)row * p.Op + c] = 0.f;
    }
    for (int n = 0; n < p.N; ++n)
        for (int e = tid; e < IT; e += 256) wl[n * IT + e] = p.w[((long long)n * p.O + o) * IT + e];
    if (tid < p.b) {
        float v[GG_MW_NMAX];
        for (int n = 0; n < GG_MW_NMAX; ++n) v[n] = 0.f;
        if (p.kmod && p.N > 1) {
            float mx = -3.0e38f;
            for (int n = 0; n < p.N; ++n) { v[n] = p.kmod[tid * p.N + n]; mx = v[n] > mx ? v[n] : mx; }
            float sum = 0.f;
            for (int n = 0; n < p.N; ++n) { v[n] = gg_expf(v[n] - mx); sum += v[n]; }
            for (int n = 0; n < p.N; ++n) v[n] /= sum;
        } else {
            v[0] = 1.f;
        }
        for (int n = 0; n < GG_MW_NMAX; ++n) a_s[tid][n] = v[n];
        if (p.a && blockIdx.x == 0)
            for (int n = 0; n < p.N; ++n) p.a[tid * p.N + n] = v[n];
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
        for (int bb = wave; bb < p.b; bb += 4) {
            float acc = 0.f;
            for (int i = lane; i < p.I; i += 64) {
                const float sv = p.mod[(long long)bb * p.I + i] + 1.f;
                float q = 0.f;
                int pr = 0;
                for (int n = 0; n < p.N; ++n)
                    for (int m = n; m < p.N; ++m, ++pr)
                        q += (n == m ? 1.f : 2.f) * a_s[bb][n] * a_s[bb][m] * gram[pr * p.I + i];
                acc += sv * sv * q;
            }
            acc = gg_mw_wave_sum(acc);
            if (lane == 0) {
                const float dv = gg_rsqrtf(acc > p.eps ? acc : p.eps);
                d_s[bb] = dv;
                if (p.d) p.d[(long long)bb * p.Op + o] = dv;
            }
        }
    } else {
        if (tid < p.b) {
            d_s[tid] = 1.f;
            if (p.d) p.d[(long long)tid * p.Op + o] = 1.f;
        }
    }
    if (!p.wmix) return;
    gg_sync();
    // per-sample weights: threads run along (t, i) with i fastest, so the bf16 stores of a wave are contiguous
    for (int bb = 0; bb < p.b; ++bb) {
        const float dv = d_s[bb];
        for (int e = tid; e < IT; e += 256) {
            const int t = e / p.I, i = e - t * p.I;
            float m = 0.f;
            for (int n = 0; n < p.N; ++n) m += a_s[bb][n] * wl[n * IT + i * p.T + t];
            const float v = dv * (p.mod[(long long)bb * p.I + i] + 1.f) * m;
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
    int groups_per_wg;      // 32-pixel row groups per workgroup (a workgroup stays inside one image)
};

template <int C>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_sconv_kernel(GgSconvParams p) {
    constexpr int KC = C / 16;
    GG_SHARED __attribute__((aligned(16))) bf16_t wl[9 * KC * 32 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gpi = p.H * (p.W >> 5);                           // groups per image
    const int chunks = (gpi + p.groups_per_wg - 1) / p.groups_per_wg;
    const int img = blockIdx.x / chunks, chunk = blockIdx.x - img * chunks;
    {
        const u16x8* src = (const u16x8*)(p.w + (long long)img * p.w_bs);
        u16x8* dst = (u16x8*)wl;
        for (int v = tid; v < 9 * KC * 32 * 2; v += 256) dst[v] = src[v];
    }
    gg_sync();
    const int pl = lane & 31, hi = lane >> 5;
    const int g_end = (chunk + 1) * p.groups_per_wg < gpi ? (chunk + 1) * p.groups_per_wg : gpi;
    const bf16_t* xi = p.x + (long long)img * p.H * p.W * C;
    for (int g = chunk * p.groups_per_wg + wave; g < g_end; g += 4) {
        const int yy = g / (p.W >> 5), x0 = (g - yy * (p.W >> 5)) << 5;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            u16x8 xa[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = yy + tap / 3 - 1, ix = x0 + pl + tap % 3 - 1;
                u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = *(const u16x8*)(xi + ((long long)iy * p.W + ix) * C + kc * 16 + hi * 8);
                xa[tap] = v;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const u16x8 wf = *(const u16x8*)(wl + ((tap * KC + kc) * 32 + pl) * 16 + hi * 8);
                acc = gg_mfma_32x32x16_bf16(wf, xa[tap], acc);      // D[out channel][pixel]: the lane's registers run along channels
            }
        }
        const long long pix = (long long)img * p.H * p.W + (long long)yy * p.W + x0 + pl;
        const float nz = p.noise ? p.noise[pix] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch0 = 8 * q + 4 * hi;
            if (ch0 < p.O) {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[q * 4 + e];
                    if (p.noise) v += nz * p.noise_w[ch0 + e];
                    if (p.act == 1) v = v > 0.f ? v : v * p.slope;
                    o[e] = gg_f2bf(v);
                }
                *(u16x4*)(p.y + pix * p.O + ch0) = o;
            }
        }
    }
}
