// gg_semlp.h — the excitation MLP of SqueezeExcite (reference gp.py:297-307: Linear -> SiLU -> Linear -> Sigmoid on the pooled
// (b, C) rows) as ONE launch, its backward as two. The stack is a few hundred KFLOP per sample: as separate launches it was two
// GEMM launches, two pointwise passes and the casts between them per forward and a dozen launches per backward, each a few
// microseconds of fixed cost on a dependent chain. Here one workgroup owns one sample; the fp32 parameters are read where they lie
// (nn.Linear's [out][in] rows, L2-resident: every workgroup reads the same 0.1-0.6 MB), everything is fp32.
//
//   forward   (grid b):      h = W1 m + b1 ; hs = silu(h) ; e = sigmoid(W2 hs + b2)            -> h, hs, e
//   backward rows (grid b):  dz2 = de * e (1 - e) ; dh = W2^T dz2 ; dz1 = dh * silu'(h) ; dm = W1^T dz1   -> dz2, dz1, dm
//   backward weights (grid over the parameter elements): gW1 = dz1^T m, gb1 = sum_b dz1, gW2 = dz2^T hs, gb2 = sum_b dz2
//
// Deterministic (fixed summation orders, no atomics). No alignment requirement beyond 4-byte floats.
#pragma once
#include "gg_device.h"

#define GG_SEMLP_MAX_C 2048      // pooled channels / excitation channels per sample (LDS rows)
#define GG_SEMLP_MAX_H 512       // hidden width

struct GgSeMlpParams {
    const float* m;        // [b][C] pooled activation
    const float* w1;       // [H][C]
    const float* b1;       // [H] or null
    const float* w2;       // [O][H]
    const float* b2;       // [O] or null
    float* h;              // [b][H] pre-activation (forward: out, backward: in)
    float* hs;             // [b][H] silu(h)
    float* e;              // [b][O] excitation
    const float* de;       // [b][O] incoming gradient
    float* dz2;            // [b][O]
    float* dz1;            // [b][H]
    float* dm;             // [b][C] or null
    float* gw;             // backward weights: [H*C | H | O*H | O]
    int b, C, H, O;
};

GG_DEVICE float gg_semlp_sigmoid(float x) { return 1.f / (1.f + gg_expf(-x)); }

#define GG_SEMLP_THREADS 1024

// y[r] = sum_k w[r][k] * v[k] for the rows r of a row-major matrix: one 16-lane DPP row per matrix row (the 16 partial sums folded by
// gg_row16_sum), four matrix rows per 16-lane group and pass with their loads issued together (256 matrix rows in flight per workgroup:
// the chain is latency-bound, ~0.5 MB of L2-resident parameters per sample). 16-byte loads when K % 4 == 0 and the rows are 16-byte
// aligned (VEC), 4-byte loads otherwise. `v` and `y` live in LDS.
template <bool VEC>
GG_DEVICE void gg_semlp_matvec_impl(const float* __restrict__ w, const float* v, float* y, int rows, int K) {
    constexpr int R = 4;
    const int t = threadIdx.x, g = t >> 4, l = t & 15;
    for (int r0 = 0; r0 < rows; r0 += R * (GG_SEMLP_THREADS / 16)) {        // (every lane takes part in the row sums: no early exit)
        float acc[R];
        const float* wr[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = r0 + g + j * (GG_SEMLP_THREADS / 16);
            acc[j] = 0.f;
            wr[j] = w + (long long)(r < rows ? r : 0) * K;
        }
        if (VEC) {
            for (int k = 4 * l; k < K; k += 64) {
                const f32x4 vv = *(const f32x4*)(v + k);
                f32x4 ww[R];
#pragma unroll
                for (int j = 0; j < R; ++j) ww[j] = *(const f32x4*)(wr[j] + k);
#pragma unroll
                for (int j = 0; j < R; ++j) acc[j] += (ww[j][0] * vv[0] + ww[j][1] * vv[1]) + (ww[j][2] * vv[2] + ww[j][3] * vv[3]);
            }
        } else {
            for (int k = l; k < K; k += 16) {
                const float vv = v[k];
#pragma unroll
                for (int j = 0; j < R; ++j) acc[j] += wr[j][k] * vv;
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = r0 + g + j * (GG_SEMLP_THREADS / 16);
            const float s = gg_row16_sum(acc[j]);
            if (r < rows && l == 0) y[r] = s;
        }
    }
}

GG_DEVICE void gg_semlp_matvec(const float* w, const float* v, float* y, int rows, int K) {
    if ((K & 3) == 0 && ((unsigned long long)w & 15) == 0) gg_semlp_matvec_impl<true>(w, v, y, rows, K);
    else gg_semlp_matvec_impl<false>(w, v, y, rows, K);
}

// y[c] = sum_r v[r] * w[r][c] (the transposed product): adjacent threads take adjacent columns (coalesced matrix rows), `parts`
// thread groups split the r range, partial sums folded through LDS in group order. `v`, `y` and `part` live in LDS.
GG_DEVICE void gg_semlp_matvec_t(const float* __restrict__ w, const float* v, float* y, float* part, int rows, int cols) {
    const int t = threadIdx.x;
    int parts = 1;
    while (parts * 2 * cols <= GG_SEMLP_THREADS) parts *= 2;
    const int cpp = GG_SEMLP_THREADS / parts;                            // columns per pass
    const int per = (rows + parts - 1) / parts;
    for (int c0 = 0; c0 < cols; c0 += cpp) {
        const int c = c0 + t % cpp, l = t / cpp;
        float acc = 0.f;
        if (c < cols) {
            const int r0 = l * per, r1 = r0 + per < rows ? r0 + per : rows;
#pragma unroll 8
            for (int r = r0; r < r1; ++r) acc += v[r] * w[(long long)r * cols + c];
        }
        part[t] = acc;
        gg_sync();
        if (l == 0 && c < cols) {
            float s = 0.f;
            for (int k = 0; k < parts; ++k) s += part[k * cpp + (t % cpp)];
            y[c] = s;
        }
        gg_sync();
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(GG_SEMLP_THREADS) void gg_se_mlp_fwd_kernel(GgSeMlpParams p) {
    GG_SHARED __attribute__((aligned(16))) float ms[GG_SEMLP_MAX_C];          // the pooled row, then the excitation pre-activations
    GG_SHARED __attribute__((aligned(16))) float hl[GG_SEMLP_MAX_H];
    const int t = threadIdx.x, img = blockIdx.x;
    for (int c = t; c < p.C; c += GG_SEMLP_THREADS) ms[c] = p.m[(long long)img * p.C + c];
    gg_sync();
    gg_semlp_matvec(p.w1, ms, hl, p.H, p.C);
    gg_sync();
    for (int j = t; j < p.H; j += GG_SEMLP_THREADS) {
        const float s = hl[j] + (p.b1 ? p.b1[j] : 0.f);
        const float a = s * gg_semlp_sigmoid(s);
        p.h[(long long)img * p.H + j] = s;
        p.hs[(long long)img * p.H + j] = a;
        hl[j] = a;
    }
    gg_sync();
    gg_semlp_matvec(p.w2, hl, ms, p.O, p.H);
    gg_sync();
    for (int o = t; o < p.O; o += GG_SEMLP_THREADS)
        p.e[(long long)img * p.O + o] = gg_semlp_sigmoid(ms[o] + (p.b2 ? p.b2[o] : 0.f));
}

GG_KERNEL GG_LAUNCH_BOUNDS(GG_SEMLP_THREADS) void gg_se_mlp_bwd_rows_kernel(GgSeMlpParams p) {
    GG_SHARED float z2[GG_SEMLP_MAX_C];          // dz2, then dm
    GG_SHARED float z1[GG_SEMLP_MAX_H];
    GG_SHARED float part[GG_SEMLP_THREADS];
    const int t = threadIdx.x, img = blockIdx.x;
    for (int o = t; o < p.O; o += GG_SEMLP_THREADS) {
        const float e = p.e[(long long)img * p.O + o];
        const float v = p.de[(long long)img * p.O + o] * e * (1.f - e);
        z2[o] = v;
        p.dz2[(long long)img * p.O + o] = v;
    }
    gg_sync();
    gg_semlp_matvec_t(p.w2, z2, z1, part, p.O, p.H);                    // dh = W2^T dz2
    for (int j = t; j < p.H; j += GG_SEMLP_THREADS) {
        const float h = p.h[(long long)img * p.H + j];
        const float sg = gg_semlp_sigmoid(h);
        const float v = z1[j] * (sg * (1.f + h * (1.f - sg)));          // silu'(h) = s(h) (1 + h (1 - s(h)))
        z1[j] = v;
        p.dz1[(long long)img * p.H + j] = v;
    }
    gg_sync();
    if (p.dm) {
        gg_semlp_matvec_t(p.w1, z1, z2, part, p.H, p.C);                // dm = W1^T dz1
        for (int c = t; c < p.C; c += GG_SEMLP_THREADS) p.dm[(long long)img * p.C + c] = z2[c];
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_se_mlp_bwd_weights_kernel(GgSeMlpParams p) {
    const long long n1 = (long long)p.H * p.C, n2 = n1 + p.H, n3 = n2 + (long long)p.O * p.H, n = n3 + p.O;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        if (i < n1) {                   // gW1[j][c] = sum_b dz1[b][j] m[b][c]
            const int j = (int)(i / p.C), c = (int)(i - (long long)j * p.C);
#pragma unroll 4
            for (int b = 0; b < p.b; ++b) s += p.dz1[(long long)b * p.H + j] * p.m[(long long)b * p.C + c];
        } else if (i < n2) {            // gb1[j]
            const int j = (int)(i - n1);
#pragma unroll 4
            for (int b = 0; b < p.b; ++b) s += p.dz1[(long long)b * p.H + j];
        } else if (i < n3) {            // gW2[o][j] = sum_b dz2[b][o] hs[b][j]
            const long long r = i - n2;
            const int o = (int)(r / p.H), j = (int)(r - (long long)o * p.H);
#pragma unroll 4
            for (int b = 0; b < p.b; ++b) s += p.dz2[(long long)b * p.O + o] * p.hs[(long long)b * p.H + j];
        } else {                        // gb2[o]
            const int o = (int)(i - n3);
#pragma unroll 4
            for (int b = 0; b < p.b; ++b) s += p.dz2[(long long)b * p.O + o];
        }
        p.gw[i] = s;
    }
}
