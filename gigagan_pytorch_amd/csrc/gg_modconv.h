// gg_modconv.h — the HBM-bound passes around the adaptive / style-modulated convolution (reference
// AdaptiveConv2DMod.forward gp.py:344-409, Noise gp.py:925-940, leaky_relu gp.py:109) in its batched formulation:
//
//     y[b] = lrelu( d[b,o] * sum_n a[b,n] * conv(W_n, x[b] * s[b,:]) + noise_w[o] * noise[b,p] )
//
// with s = mod + 1, a = softmax(kernel_mod), d = the demodulation coefficients. The contraction conv(W_n, .) for all n
// is ONE implicit-GEMM launch with the N kernels stacked along output channels (gg_gemm2.h); the kernels here are the
// two bf16 passes around it and their gradients:
//   gg_modulate_kernel      xs = x * s                                      (2 B read + 2 B written per element)
//   gg_modulate_bwd_kernel  dx = g * s, ds[b,c] = sum_p g*x                 (4 B read + 2 B written)
//   gg_modmix_fwd_kernel    y  = act(d * sum_n a_n Y_n + nw * noise)        (2N B read + 2 B written)
//   gg_modmix_bwd_kernel    dY_n = a_n d dz;  da, dd, dnw partial sums      (2N+4 B read + 2N B written)
// They replace ~25 fp32 tensor-algebra passes per layer (the autograd of the broadcasted products, sums and casts).
// Layout: activations [b][pixels][C] bf16 with C % 8 == 0; one thread owns one 16-byte column group of one row at a
// time; per-image reductions are written as per-workgroup partial sums (no atomics), each workgroup working inside
// ONE image.
#pragma once
#include "gg_device.h"

struct GgModulateParams {
    const bf16_t* x;      // [b][P][C]
    const bf16_t* g;      // bwd: incoming gradient [b][P][C]
    const float* s;       // [b][C]
    bf16_t* out;          // fwd: xs ; bwd: dx
    float* ds_part;       // bwd: [b][chunks][C] partial sums of g*x
    int b, P, C, chunks;  // chunks = workgroups per image
    int Cin;              // fwd only: > 0: x has Cin channels, s is [b][Cin], `a` is [b][C / Cin] and output channel n*Cin + i is
    const float* a;       // x[..., i] * s[b, i] * a[b, n] (the activation pre-scaled for each of the bank's N kernels); 0: plain x * s
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modulate_kernel(GgModulateParams p) {
    const int ncg = p.C / 8;
    const long long total = (long long)p.b * p.P * ncg;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int cg = (int)(idx % ncg);
        const long long row = idx / ncg;
        const int img = (int)(row / p.P);
        u16x8 v, o;
        if (p.Cin > 0) {
            const int c = cg * 8, n = c / p.Cin, ci = c - n * p.Cin;
            v = *(const u16x8*)(p.x + row * p.Cin + ci);
            const float* sc = p.s + (long long)img * p.Cin + ci;
            const float an = p.a[(long long)img * (p.C / p.Cin) + n];
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(v[e]) * (sc[e] * an));
        } else {
            v = *(const u16x8*)(p.x + row * p.C + cg * 8);
            const float* sc = p.s + (long long)img * p.C + cg * 8;
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(v[e]) * sc[e]);
        }
        *(u16x8*)(p.out + row * p.C + cg * 8) = o;
    }
}

// block -> (image, chunk of rows); thread -> (row lane, column group); column groups beyond 256 are looped
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modulate_bwd_kernel(GgModulateParams p) {
    GG_SHARED float red[256][8];
    const int ncg = p.C / 8;
    const int t = threadIdx.x;
    const int lanes_per_row = ncg < 256 ? ncg : 256;
    const int row_lanes = 256 / lanes_per_row;
    const int cgl = t % lanes_per_row, rl = t / lanes_per_row;
    const int img = blockIdx.x / p.chunks, chunk = blockIdx.x % p.chunks;
    const int rows_per_chunk = (p.P + p.chunks - 1) / p.chunks;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > p.P) r1 = p.P;
    for (int cg0 = 0; cg0 < ncg; cg0 += lanes_per_row) {       // (workgroup-uniform trip count: the barriers below are inside)
        const int cg = cg0 + cgl;
        const bool live = rl < row_lanes && cg < ncg;
        float acc[8];
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (live) {
            const float* sc = p.s + (long long)img * p.C + cg * 8;
            float sv[8];
            for (int e = 0; e < 8; ++e) sv[e] = sc[e];
            // four rows per trip, their loads issued together (as gg_bias_act_bwd, round 6); accumulated in the old order: same bits
            constexpr int UR = 4;
            for (int rb = r0 + rl; rb < r1; rb += row_lanes * UR) {
                u16x8 gv[UR], xv[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int r = rb + u * row_lanes;
                    const long long off = ((long long)img * p.P + (r < r1 ? r : r1 - 1)) * p.C + cg * 8;
                    gv[u] = *(const u16x8*)(p.g + off);
                    xv[u] = *(const u16x8*)(p.x + off);
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int r = rb + u * row_lanes;
                    if (r < r1) {
                        u16x8 o;
                        for (int e = 0; e < 8; ++e) {
                            float gf = gg_bf2f(gv[u][e]);
                            o[e] = gg_f2bf(gf * sv[e]);
                            acc[e] += gf * gg_bf2f(xv[u][e]);
                        }
                        *(u16x8*)(p.out + ((long long)img * p.P + r) * p.C + cg * 8) = o;
                    }
                }
            }
        }
        for (int e = 0; e < 8; ++e) red[t][e] = live ? acc[e] : 0.f;
        gg_sync();
        if (rl == 0 && cg < ncg) {
            for (int k = 1; k < row_lanes; ++k)
                for (int e = 0; e < 8; ++e) acc[e] += red[k * lanes_per_row + cgl][e];
            float* dst = p.ds_part + ((long long)img * p.chunks + chunk) * p.C + cg * 8;
            for (int e = 0; e < 8; ++e) dst[e] = acc[e];
        }
        gg_sync();
    }
}

#define GG_MIX_MAXN 4   // kernels in a bank (reference default num_conv_kernels = 2)

struct GgModMixParams {
    const bf16_t* Y;       // [b][P][N*Os]  stacked conv outputs (Os = per-kernel output pitch)
    const float* a;        // [b][N]   softmax(kernel_mod); all ones for N == 1
    const float* d;        // [b][O]   demodulation coefficients, optional
    const float* noise;    // [b][P]   optional
    const float* noise_w;  // [O]      (with noise)
    bf16_t* y;             // fwd out / bwd: the forward output (activation mask), optional in bwd
    const bf16_t* dy;      // bwd in   [b][P][O]
    bf16_t* dY;            // bwd out  [b][P][N*Os]
    float* da_part;        // bwd out  [chunks][b][N]   (only when N > 1)   chunk-major: a slice stack for gg_reduce_multi
    float* dd_part;        // bwd out  [chunks][b][O]   (only when d != null)
    float* dnw_part;       // bwd out  [chunks][b][O]   (only when noise != null)
    int b, P, O, Os, N, chunks;
    int act;               // 0 none, 1 leaky relu
    float slope;
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modmix_fwd_kernel(GgModMixParams p) {
    const int ncg = p.O / 8;
    const long long total = (long long)p.b * p.P * ncg;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int cg = (int)(idx % ncg);
        const long long row = idx / ncg;
        const int img = (int)(row / p.P);
        float acc[8];
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int n = 0; n < p.N; ++n) {
            const float an = p.a[(long long)img * p.N + n];
            u16x8 v = *(const u16x8*)(p.Y + row * ((long long)p.N * p.Os) + (long long)n * p.Os + cg * 8);
            for (int e = 0; e < 8; ++e) acc[e] += an * gg_bf2f(v[e]);
        }
        if (p.d) {
            const float* dc = p.d + (long long)img * p.O + cg * 8;
            for (int e = 0; e < 8; ++e) acc[e] *= dc[e];
        }
        if (p.noise) {
            const float nz = p.noise[row];
            for (int e = 0; e < 8; ++e) acc[e] += nz * p.noise_w[cg * 8 + e];
        }
        u16x8 o;
        for (int e = 0; e < 8; ++e) {
            float v = acc[e];
            if (p.act == 1) v = v > 0.f ? v : v * p.slope;
            o[e] = gg_f2bf(v);
        }
        *(u16x8*)(p.y + row * p.O + cg * 8) = o;
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modmix_bwd_kernel(GgModMixParams p) {
    GG_SHARED float red[256][8];
    GG_SHARED float red_a[256][GG_MIX_MAXN];
    const int ncg = p.O / 8;
    const int t = threadIdx.x;
    const int lanes_per_row = ncg < 256 ? ncg : 256;
    const int row_lanes = 256 / lanes_per_row;
    const int cgl = t % lanes_per_row, rl = t / lanes_per_row;
    const int img = blockIdx.x / p.chunks, chunk = blockIdx.x % p.chunks;
    const int rows_per_chunk = (p.P + p.chunks - 1) / p.chunks;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > p.P) r1 = p.P;
    float an[GG_MIX_MAXN];
    for (int n = 0; n < GG_MIX_MAXN; ++n) an[n] = n < p.N ? p.a[(long long)img * p.N + n] : 0.f;
    float da[GG_MIX_MAXN];
    for (int n = 0; n < GG_MIX_MAXN; ++n) da[n] = 0.f;
    for (int cg = cgl; cg < ncg; cg += lanes_per_row) {
        float dd[8], dnw[8], dv[8];
        for (int e = 0; e < 8; ++e) { dd[e] = 0.f; dnw[e] = 0.f; dv[e] = 1.f; }
        if (p.d)
            for (int e = 0; e < 8; ++e) dv[e] = p.d[(long long)img * p.O + cg * 8 + e];
        if (rl < row_lanes) {
            for (int r = r0 + rl; r < r1; r += row_lanes) {
                const long long row = (long long)img * p.P + r;
                u16x8 g = *(const u16x8*)(p.dy + row * p.O + cg * 8);
                float dz[8];
                if (p.act == 1) {
                    u16x8 yv = *(const u16x8*)(p.y + row * p.O + cg * 8);
                    for (int e = 0; e < 8; ++e) dz[e] = gg_bf2f(g[e]) * (gg_bf2f(yv[e]) > 0.f ? 1.f : p.slope);
                } else {
                    for (int e = 0; e < 8; ++e) dz[e] = gg_bf2f(g[e]);
                }
                if (p.noise) {
                    const float nz = p.noise[row];
                    for (int e = 0; e < 8; ++e) dnw[e] += dz[e] * nz;
                }
                float tmix[8];
                for (int e = 0; e < 8; ++e) tmix[e] = 0.f;
                // all kernels' Y vectors first, the dY stores last (a store between two loads serialises them: dY may alias Y as far as
                // the compiler knows)
                const long long ybase = row * ((long long)p.N * p.Os) + cg * 8;
                u16x8 yv4[GG_MIX_MAXN], o4[GG_MIX_MAXN];
#pragma unroll
                for (int n = 0; n < GG_MIX_MAXN; ++n) {
                    const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    yv4[n] = n < p.N ? *(const u16x8*)(p.Y + ybase + (long long)n * p.Os) : z;
                }
#pragma unroll
                for (int n = 0; n < GG_MIX_MAXN; ++n) {
                    float dan = 0.f;
                    for (int e = 0; e < 8; ++e) {
                        const float yv = gg_bf2f(yv4[n][e]);
                        tmix[e] += an[n] * yv;
                        dan += dz[e] * dv[e] * yv;
                        o4[n][e] = gg_f2bf(an[n] * dv[e] * dz[e]);
                    }
                    da[n] += dan;
                }
#pragma unroll
                for (int n = 0; n < GG_MIX_MAXN; ++n)
                    if (n < p.N) *(u16x8*)(p.dY + ybase + (long long)n * p.Os) = o4[n];
                for (int e = 0; e < 8; ++e) dd[e] += dz[e] * tmix[e];
            }
        }
        // column-wise partial sums of this column group
        if (p.dd_part) {
            for (int e = 0; e < 8; ++e) red[t][e] = (rl < row_lanes) ? dd[e] : 0.f;
            gg_sync();
            if (rl == 0) {
                for (int k = 1; k < row_lanes; ++k)
                    for (int e = 0; e < 8; ++e) dd[e] += red[k * lanes_per_row + cgl][e];
                float* dst = p.dd_part + ((long long)chunk * p.b + img) * p.O + cg * 8;
                for (int e = 0; e < 8; ++e) dst[e] = dd[e];
            }
            gg_sync();
        }
        if (p.dnw_part) {
            for (int e = 0; e < 8; ++e) red[t][e] = (rl < row_lanes) ? dnw[e] : 0.f;
            gg_sync();
            if (rl == 0) {
                for (int k = 1; k < row_lanes; ++k)
                    for (int e = 0; e < 8; ++e) dnw[e] += red[k * lanes_per_row + cgl][e];
                float* dst = p.dnw_part + ((long long)chunk * p.b + img) * p.O + cg * 8;
                for (int e = 0; e < 8; ++e) dst[e] = dnw[e];
            }
            gg_sync();
        }
    }
    if (p.da_part) {
        for (int n = 0; n < GG_MIX_MAXN; ++n) red_a[t][n] = da[n];
        gg_sync();
        if (t < GG_MIX_MAXN) {
            float s = 0.f;
            for (int k = 0; k < 256; ++k) s += red_a[k][t];
            if (t < p.N) p.da_part[((long long)chunk * p.b + img) * p.N + t] = s;
        }
    }
}
