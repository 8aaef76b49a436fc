// gg_elementwise.h — HBM-bound kernels of the GigaGAN step (NHWC bf16 activations, fp32 math).
#pragma once
#include "gg_device.h"

// ---- separable banded resampling ----------------------------------------------------------------------
// out[n][oy][ox][c] = sum_{a<TY} sum_{b<TX} wy[oy][a] * wx[ox][b] * in[n][iy0[oy]+a][ix0[ox]+b][c]
// Covers nn.Upsample(x2,bilinear)+Blur (gp.py:246-261) as ONE pass, F.interpolate bilinear/nearest
// (gp.py:1683-1687, :2210) and all of their adjoints (the transposed tables). Algorithmic bytes per
// launch: (in + out) * 2 B. One thread produces 8 channels of one output pixel with 16-byte accesses.
struct GgResampleParams {
    const bf16_t* in;
    bf16_t* out;
    int n, IH, IW, OH, OW, C;
    int TY, TX;
    const int* iy0;
    const int* ix0;
    const float* wy;  // [OH][TY]
    const float* wx;  // [OW][TX]
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_resample_kernel(GgResampleParams p) {
    const int cg = (p.C + 7) / 8;
    const long long total = (long long)p.n * p.OH * p.OW * cg;
    const bool vec = (p.C % 8) == 0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        int g = (int)(idx % cg);
        long long pix = idx / cg;
        int ox = (int)(pix % p.OW);
        long long t = pix / p.OW;
        int oy = (int)(t % p.OH);
        int img = (int)(t / p.OH);
        float acc[8];
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        const int y0 = p.iy0[oy], x0 = p.ix0[ox];
        const int c0 = g * 8;
        const int nc = (p.C - c0) < 8 ? (p.C - c0) : 8;
        for (int a = 0; a < p.TY; ++a) {
            float wya = p.wy[oy * p.TY + a];
            int iy = y0 + a;
            if (wya == 0.f || iy < 0 || iy >= p.IH) continue;
            for (int b = 0; b < p.TX; ++b) {
                float w = wya * p.wx[ox * p.TX + b];
                int ix = x0 + b;
                if (w == 0.f || ix < 0 || ix >= p.IW) continue;
                const bf16_t* src = p.in + (((long long)img * p.IH + iy) * p.IW + ix) * p.C + c0;
                if (vec) {
                    u16x8 v = *(const u16x8*)src;
                    for (int e = 0; e < 8; ++e) acc[e] += w * gg_bf2f(v[e]);
                } else {
                    for (int e = 0; e < nc; ++e) acc[e] += w * gg_bf2f(src[e]);
                }
            }
        }
        bf16_t* dst = p.out + (((long long)img * p.OH + oy) * p.OW + ox) * p.C + c0;
        if (vec) {
            u16x8 o;
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(acc[e]);
            *(u16x8*)dst = o;
        } else {
            for (int e = 0; e < nc; ++e) dst[e] = gg_f2bf(acc[e]);
        }
    }
}

// ---- fused multi-tensor AdamW -------------------------------------------------------------------------
// One launch updates a whole model (reference: torch.optim.AdamW built by optimizer.py:10-34, stepped at
// gp.py:2477 / :2596). Parameters, gradients and both moments live in flat fp32 buffers; every parameter
// starts on a 256-element boundary and `flags[i / 256]` says whether the chunk is stepped at all (bit 0:
// parameters whose gradient is None in the reference are skipped, not decayed — SURVEY.md B.13) and
// whether decoupled weight decay applies (bit 1: ndim >= 2, optimizer.py:3-8).
// HBM-bound: 16 B read + 12 B written per element = 28 B/param.
struct GgAdamWParams {
    float* p;
    const float* g;
    float* m;
    float* v;
    const unsigned char* flags;
    long long n;  // multiple of 4
    float lr, beta1, beta2, eps, wd;
    float bc1, bc2_sqrt;  // 1 - beta1^t, sqrt(1 - beta2^t)
    float grad_scale;     // multiplies g first (1/world for summed all-reduce, 1/accum ...)
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_adamw_kernel(GgAdamWParams a) {
    const long long n4 = a.n / 4;
    const float step_size = a.lr / a.bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const unsigned char fl = a.flags[(i * 4) >> 8];
        if (!(fl & 1)) continue;
        f32x4 p = *(const f32x4*)(a.p + i * 4);
        f32x4 g = *(const f32x4*)(a.g + i * 4);
        f32x4 m = *(const f32x4*)(a.m + i * 4);
        f32x4 v = *(const f32x4*)(a.v + i * 4);
        const float decay = (fl & 2) ? (1.f - a.lr * a.wd) : 1.f;
        for (int e = 0; e < 4; ++e) {
            float ge = g[e] * a.grad_scale;
            float pe = p[e] * decay;
            float me = a.beta1 * m[e] + (1.f - a.beta1) * ge;
            float ve = a.beta2 * v[e] + (1.f - a.beta2) * ge * ge;
            float denom = sqrtf(ve) / a.bc2_sqrt + a.eps;
            p[e] = pe - step_size * (me / denom);
            m[e] = me;
            v[e] = ve;
        }
        *(f32x4*)(a.p + i * 4) = p;
        *(f32x4*)(a.m + i * 4) = m;
        *(f32x4*)(a.v + i * 4) = v;
    }
}

// ema[i] = ema[i] + (1 - beta) * (p[i] - ema[i])  (ema_pytorch lerp; gp.py:2603), one launch per model
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_ema_kernel(float* ema, const float* p, long long n, float one_minus_beta) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 e = *(const f32x4*)(ema + i * 4);
        f32x4 q = *(const f32x4*)(p + i * 4);
        for (int k = 0; k < 4; ++k) e[k] = e[k] + one_minus_beta * (q[k] - e[k]);
        *(f32x4*)(ema + i * 4) = e;
    }
}
