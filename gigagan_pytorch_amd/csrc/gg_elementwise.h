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

// The same operator with the tap counts known at compile time and C % 8 == 0 (every call of the training step: 1x1 nearest /
// transposed bilinear, 2x2 bilinear, 3x3 upsample+blur / blur, 6x6 their adjoints): all loads of a tap row are issued
// unconditionally from clamped coordinates and weighted by zero when the tap lies outside the image. In the generic kernel above
// every load sits under a per-lane condition and feeds the accumulator at once, so the compiler branches around each one and
// waits for it before issuing the next: up to 36 dependent round trips per output vector (measured 1.26 TB/s on the 128 -> 256
// feature upsample).
template <int TY, int TX>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_resample_taps_kernel(GgResampleParams p) {
    const int cg = p.C / 8;
    const long long total = (long long)p.n * p.OH * p.OW * cg;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int g = (int)(idx % cg);
        const long long pix = idx / cg;
        const int ox = (int)(pix % p.OW);
        const long long t = pix / p.OW;
        const int oy = (int)(t % p.OH);
        const int img = (int)(t / p.OH);
        const int y0 = p.iy0[oy], x0 = p.ix0[ox];
        float wyv[TY], wxv[TX];
        int cx[TX];
#pragma unroll
        for (int a = 0; a < TY; ++a) wyv[a] = p.wy[oy * TY + a];
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            const int ix = x0 + b;
            const bool in = ix >= 0 && ix < p.IW;
            wxv[b] = in ? p.wx[ox * TX + b] : 0.f;
            cx[b] = ix < 0 ? 0 : (ix >= p.IW ? p.IW - 1 : ix);
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        const bf16_t* base = p.in + (long long)img * p.IH * p.IW * p.C + g * 8;
#pragma unroll
        for (int a = 0; a < TY; ++a) {
            const int iy = y0 + a;
            const float wya = (iy >= 0 && iy < p.IH) ? wyv[a] : 0.f;
            const int cy = iy < 0 ? 0 : (iy >= p.IH ? p.IH - 1 : iy);
            u16x8 v[TX];
#pragma unroll
            for (int b = 0; b < TX; ++b) v[b] = *(const u16x8*)(base + ((long long)cy * p.IW + cx[b]) * p.C);
#pragma unroll
            for (int b = 0; b < TX; ++b) {
                const float w = wya * wxv[b];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += w * gg_bf2f(v[b][e]);
            }
        }
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(acc[e]);
        *(u16x8*)(p.out + (((long long)img * p.OH + oy) * p.OW + ox) * p.C + g * 8) = o;
    }
}

// The same for the rgb maps (C = 3: 6-byte pixels, no 16-byte vectors): one thread per output pixel, every tap loaded unconditionally from
// clamped coordinates (three 2-byte loads) and weighted by zero outside the image - the generic kernel branches around each scalar load
// and waits for it (29 us per 256 x 256 x 32 rgb map: 0.5 TB/s).
template <int TY, int TX, int CC>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_resample_taps_small_kernel(GgResampleParams p) {
    const long long total = (long long)p.n * p.OH * p.OW;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
        const int ox = (int)(pix % p.OW);
        const long long t = pix / p.OW;
        const int oy = (int)(t % p.OH);
        const int img = (int)(t / p.OH);
        const int y0 = p.iy0[oy], x0 = p.ix0[ox];
        float wxv[TX];
        int cx[TX];
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            const int ix = x0 + b;
            wxv[b] = (ix >= 0 && ix < p.IW) ? p.wx[ox * TX + b] : 0.f;
            cx[b] = ix < 0 ? 0 : (ix >= p.IW ? p.IW - 1 : ix);
        }
        float acc[CC];
#pragma unroll
        for (int e = 0; e < CC; ++e) acc[e] = 0.f;
        const bf16_t* base = p.in + (long long)img * p.IH * p.IW * CC;
#pragma unroll
        for (int a = 0; a < TY; ++a) {
            const int iy = y0 + a;
            const float wya = (iy >= 0 && iy < p.IH) ? p.wy[oy * TY + a] : 0.f;
            const int cy = iy < 0 ? 0 : (iy >= p.IH ? p.IH - 1 : iy);
            bf16_t v[TX][CC];
#pragma unroll
            for (int b = 0; b < TX; ++b)
#pragma unroll
                for (int e = 0; e < CC; ++e) v[b][e] = base[((long long)cy * p.IW + cx[b]) * CC + e];
#pragma unroll
            for (int b = 0; b < TX; ++b) {
                const float w = wya * wxv[b];
#pragma unroll
                for (int e = 0; e < CC; ++e) acc[e] += w * gg_bf2f(v[b][e]);
            }
        }
        bf16_t* dst = p.out + pix * CC;
#pragma unroll
        for (int e = 0; e < CC; ++e) dst[e] = gg_f2bf(acc[e]);
    }
}

// Up-sampling / same-size filters (neighbouring outputs start their windows 0 or 1 input pixels apart): a thread produces a 2 x 2 block of
// output pixels from ONE (TY + 1) x (TX + 1) window - 16 loads for four outputs at 3 x 3 taps where the kernel above issues 36. That kernel
// is bound by the vector-memory pipe, not by HBM (ten 16-byte operations per 16 bytes stored: 1.4-2.1 TB/s of in + out on the generator's
// feature up-sampling, tests/gpu_bw_census.py). A block whose windows are further apart (never with the trainer's tables) falls back to
// direct loads, so the kernel is correct for any table.
template <int TY, int TX>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_resample_taps2x2_kernel(GgResampleParams p) {
    const int cg = p.C / 8;
    const int BH = p.OH / 2, BW = p.OW / 2;
    const long long total = (long long)p.n * BH * BW * cg;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int g = (int)(idx % cg);
        const long long blk = idx / cg;
        const int bx = (int)(blk % BW);
        const long long t = blk / BW;
        const int by = (int)(t % BH);
        const int img = (int)(t / BH);
        const int oy0 = 2 * by, ox0 = 2 * bx;
        const int ya = p.iy0[oy0], xa = p.ix0[ox0];
        const int dy = p.iy0[oy0 + 1] - ya, dx = p.ix0[ox0 + 1] - xa;
        const bf16_t* base = p.in + (long long)img * p.IH * p.IW * p.C + g * 8;
        float acc[2][2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][i][e] = 0.f;
        if (dy < 0 || dy > 1 || dx < 0 || dx > 1) {          // windows too far apart for one shared window: four direct evaluations
            for (int j = 0; j < 2; ++j)
                for (int i = 0; i < 2; ++i) {
                    const int y0 = p.iy0[oy0 + j], x0 = p.ix0[ox0 + i];
                    for (int a = 0; a < TY; ++a) {
                        const int iy = y0 + a;
                        if (iy < 0 || iy >= p.IH) continue;
                        for (int b = 0; b < TX; ++b) {
                            const int ix = x0 + b;
                            if (ix < 0 || ix >= p.IW) continue;
                            const float w = p.wy[(oy0 + j) * TY + a] * p.wx[(ox0 + i) * TX + b];
                            const u16x8 v = *(const u16x8*)(base + ((long long)iy * p.IW + ix) * p.C);
                            for (int e = 0; e < 8; ++e) acc[j][i][e] += w * gg_bf2f(v[e]);
                        }
                    }
                }
        } else {
            float wyj[2][TY + 1], wxj[2][TX + 1];
            int cy[TY + 1], cx[TX + 1];
#pragma unroll
            for (int r = 0; r <= TY; ++r) {
                const int iy = ya + r;
                const bool in = iy >= 0 && iy < p.IH;
                cy[r] = iy < 0 ? 0 : (iy >= p.IH ? p.IH - 1 : iy);
                wyj[0][r] = (in && r < TY) ? p.wy[oy0 * TY + r] : 0.f;
                const int r1 = r - dy;
                wyj[1][r] = (in && r1 >= 0 && r1 < TY) ? p.wy[(oy0 + 1) * TY + r1] : 0.f;
            }
#pragma unroll
            for (int c = 0; c <= TX; ++c) {
                const int ix = xa + c;
                const bool in = ix >= 0 && ix < p.IW;
                cx[c] = ix < 0 ? 0 : (ix >= p.IW ? p.IW - 1 : ix);
                wxj[0][c] = (in && c < TX) ? p.wx[ox0 * TX + c] : 0.f;
                const int c1 = c - dx;
                wxj[1][c] = (in && c1 >= 0 && c1 < TX) ? p.wx[(ox0 + 1) * TX + c1] : 0.f;
            }
#pragma unroll
            for (int r = 0; r <= TY; ++r) {
                u16x8 v[TX + 1];
#pragma unroll
                for (int c = 0; c <= TX; ++c) v[c] = *(const u16x8*)(base + ((long long)cy[r] * p.IW + cx[c]) * p.C);
#pragma unroll
                for (int c = 0; c <= TX; ++c) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = gg_bf2f(v[c][e]);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const float w = wyj[j][r] * wxj[i][c];
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[j][i][e] += w * f[e];
                        }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(acc[j][i][e]);
                *(u16x8*)(p.out + (((long long)img * p.OH + oy0 + j) * p.OW + ox0 + i) * p.C + g * 8) = o;
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

// ---- attention softmax over materialised logits (one wavefront per row) --------------------------------
// fwd:  S[r][j] = softmax_j(alpha * x[r][j] + bias[r / rows_per_batch][j]),  j < n_valid; columns in
//       [n_valid, ld) are written as 0.  x fp32 (pitch ld), S bf16 (pitch ld).
// bwd:  u = S * (dS - sum_j S*dS);  dx = alpha * u (bf16);  dbias[batch][j] += sum_rows u (fp32 atomics,
//       one per column per 16 rows).
// Replaces sim*scale / masked_fill / softmax / casts of gp.py:584-588 (and their autograd) — six fp32
// passes over the (b*h, n, n+1) similarity tensor — by one pass each way. HBM-bound:
// fwd 6 B/element, bwd 6 B/element.
#define GG_SM_MAXV 8       // register cache: up to 8 float4 per lane => rows of <= 2048 columns
#define GG_SM_ROWS 16      // rows per wavefront (amortises the dbias atomics)

struct GgSoftmaxParams {
    const float* x;      // fwd input
    const bf16_t* S;     // bwd input (fwd output)
    const bf16_t* dS;    // bwd input
    bf16_t* out;         // fwd: S ; bwd: dx
    const float* bias;   // fwd, optional [nbatch][ld]
    float* dbias;        // bwd, optional [nbatch][ld], pre-zeroed
    long long rows;
    int rows_per_batch, n_valid, ld;
    float alpha;
};

GG_DEVICE float gg_wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, gg_shfl_xor(v, o));
    return v;
}
GG_DEVICE float gg_wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += gg_shfl_xor(v, o);
    return v;
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_softmax_fwd_kernel(GgSoftmaxParams p) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nv = (p.ld + 255) / 256;
    for (int rr = 0; rr < GG_SM_ROWS; ++rr) {
        const long long r = wave * GG_SM_ROWS + rr;
        if (r >= p.rows) break;   // wave-uniform
        const float* xr = p.x + r * p.ld;
        const float* br = p.bias ? p.bias + (r / p.rows_per_batch) * p.ld : nullptr;
        f32x4 v[GG_SM_MAXV];
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            f32x4 z = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            if (j < p.ld) {
                f32x4 xv = *(const f32x4*)(xr + j);
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (br) bv = *(const f32x4*)(br + j);
                for (int e = 0; e < 4; ++e)
                    if (j + e < p.n_valid) z[e] = p.alpha * xv[e] + bv[e];
            }
            v[t] = z;
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, z[e]);
        }
        mx = gg_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            for (int e = 0; e < 4; ++e) {
                float ex = (j + e < p.n_valid) ? gg_expf(v[t][e] - mx) : 0.f;
                v[t][e] = ex;
                sum += ex;
            }
        }
        sum = gg_wave_sum(sum);
        const float inv = 1.f / sum;
        bf16_t* orow = p.out + r * p.ld;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            if (j < p.ld) {
                u16x4 o = {gg_f2bf(v[t][0] * inv), gg_f2bf(v[t][1] * inv), gg_f2bf(v[t][2] * inv), gg_f2bf(v[t][3] * inv)};
                *(u16x4*)(orow + j) = o;
            }
        }
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_softmax_bwd_kernel(GgSoftmaxParams p) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nv = (p.ld + 255) / 256;
    f32x4 colacc[GG_SM_MAXV];
#pragma unroll
    for (int t = 0; t < GG_SM_MAXV; ++t) colacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long acc_batch = -1;
    // GG_SM_ROWS divides rows_per_batch (checked by the host), so one wave never straddles two batches
    for (int rr = 0; rr < GG_SM_ROWS; ++rr) {
        const long long r = wave * GG_SM_ROWS + rr;
        if (r >= p.rows) break;
        acc_batch = r / p.rows_per_batch;
        const bf16_t* sr = p.S + r * p.ld;
        const bf16_t* dr = p.dS + r * p.ld;
        f32x4 s[GG_SM_MAXV], d[GG_SM_MAXV];
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
            if (j < p.ld) {
                u16x4 a = *(const u16x4*)(sr + j);
                u16x4 b = *(const u16x4*)(dr + j);
                for (int e = 0; e < 4; ++e) {
                    if (j + e < p.n_valid) {
                        sv[e] = gg_bf2f(a[e]);
                        dv[e] = gg_bf2f(b[e]);
                    }
                    dot += sv[e] * dv[e];
                }
            }
            s[t] = sv;
            d[t] = dv;
        }
        dot = gg_wave_sum(dot);
        bf16_t* orow = p.out + r * p.ld;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            if (j < p.ld) {
                float u[4];
                for (int e = 0; e < 4; ++e) {
                    u[e] = s[t][e] * (d[t][e] - dot);
                    colacc[t][e] += u[e];
                }
                u16x4 o = {gg_f2bf(p.alpha * u[0]), gg_f2bf(p.alpha * u[1]), gg_f2bf(p.alpha * u[2]), gg_f2bf(p.alpha * u[3])};
                *(u16x4*)(orow + j) = o;
            }
        }
    }
    if (p.dbias && acc_batch >= 0) {
        float* db = p.dbias + acc_batch * p.ld;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            for (int e = 0; e < 4; ++e)
                if (j + e < p.n_valid) gg_atomic_add(db + j + e, colacc[t][e]);
        }
    }
}

// second-order pass of the attention softmax (gradient-penalty steps differentiate the backward above):
//   given the incoming gradients g_dx (w.r.t. dx) and g_dbias (w.r.t. dbias), with gt = alpha*g_dx + g_dbias[batch]:
//     r = sum_j S*dS,  gs = sum_j gt*S,   g_dS = S*(gt - gs),   g_S = gt*(dS - r) - dS*gs
// one wavefront per row, one pass: replaces ~12 fp32 tensor-algebra passes over the (b*h, n, n+1) tensors.
struct GgSoftmaxBwd2Params {
    const bf16_t* S;
    const bf16_t* dS;
    const bf16_t* g_dx;     // optional
    const float* g_dbias;   // optional [nbatch][ld]
    bf16_t* g_S;
    bf16_t* g_dS;
    long long rows;
    int rows_per_batch, n_valid, ld;
    float alpha;
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_softmax_bwd2_kernel(GgSoftmaxBwd2Params p) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nv = (p.ld + 255) / 256;
    for (int rr = 0; rr < GG_SM_ROWS; ++rr) {
        const long long r = wave * GG_SM_ROWS + rr;
        if (r >= p.rows) break;   // wave-uniform
        const bf16_t* sr = p.S + r * p.ld;
        const bf16_t* dr = p.dS + r * p.ld;
        const bf16_t* gr = p.g_dx ? p.g_dx + r * p.ld : nullptr;
        const float* br = p.g_dbias ? p.g_dbias + (r / p.rows_per_batch) * p.ld : nullptr;
        f32x4 s[GG_SM_MAXV], d[GG_SM_MAXV], g[GG_SM_MAXV];
        float rsum = 0.f, gsum = 0.f;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dv = sv, gv = sv;
            if (j < p.ld) {
                u16x4 a = *(const u16x4*)(sr + j);
                u16x4 b = *(const u16x4*)(dr + j);
                u16x4 c = {0, 0, 0, 0};
                if (gr) c = *(const u16x4*)(gr + j);
                f32x4 bb = {0.f, 0.f, 0.f, 0.f};
                if (br) bb = *(const f32x4*)(br + j);
                for (int e = 0; e < 4; ++e) {
                    if (j + e < p.n_valid) {
                        sv[e] = gg_bf2f(a[e]);
                        dv[e] = gg_bf2f(b[e]);
                        gv[e] = p.alpha * gg_bf2f(c[e]) + bb[e];
                    }
                    rsum += sv[e] * dv[e];
                    gsum += gv[e] * sv[e];
                }
            }
            s[t] = sv; d[t] = dv; g[t] = gv;
        }
        rsum = gg_wave_sum(rsum);
        gsum = gg_wave_sum(gsum);
        bf16_t* os = p.g_S + r * p.ld;
        bf16_t* od = p.g_dS + r * p.ld;
#pragma unroll
        for (int t = 0; t < GG_SM_MAXV; ++t) {
            if (t >= nv) break;
            int j = t * 256 + lane * 4;
            if (j < p.ld) {
                u16x4 o1, o2;
                for (int e = 0; e < 4; ++e) {
                    o1[e] = gg_f2bf(g[t][e] * (d[t][e] - rsum) - d[t][e] * gsum);
                    o2[e] = gg_f2bf(s[t][e] * (g[t][e] - gsum));
                }
                *(u16x4*)(os + j) = o1;
                *(u16x4*)(od + j) = o2;
            }
        }
    }
}

// ---- bias / activation backward ---------------------------------------------------------------------------
// dz = dy * (y > 0 ? 1 : slope)  (skipped when y == null: dz aliases dy) and per-workgroup partial column sums
// db_part[wg][c] = sum over the workgroup's rows of dz[row][c] (a few hundred rows of [C] floats, summed by the caller).
// One pass instead of compare + where + cast + mul + float-cast + reduce (autograd of nn.Conv2d bias and
// nn.LeakyReLU, gp.py:109, :1608-1621). x: [rows][C] bf16, C % 8 == 0. 4 B (+2 B) per element.
struct GgBiasActBwdParams {
    const bf16_t* dy;
    const bf16_t* y;   // optional
    bf16_t* dz;        // required iff y != null
    float* db;         // optional [gridDim.x][C] per-workgroup partial sums (summed by the caller: no atomics)
    long long rows;
    int C;
    float slope;
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_bias_act_bwd_kernel(GgBiasActBwdParams p) {
    GG_SHARED float red[256][8];
    const int ncg = p.C / 8;                       // column groups of 8 channels
    const int t = threadIdx.x;
    // thread -> (row lane, column group); column groups beyond 256 are looped
    const int lanes_per_row = ncg < 256 ? ncg : 256;
    const int row_lanes = 256 / lanes_per_row;
    const int cgl = t % lanes_per_row, rl = t / lanes_per_row;
    const long long rows_per_block = (p.rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > p.rows) r1 = p.rows;
    for (int cg0 = 0; cg0 < ncg; cg0 += lanes_per_row) {       // (workgroup-uniform trip count: the barriers below are inside)
        const int cg = cg0 + cgl;
        const bool live = rl < row_lanes && cg < ncg;
        float acc[8];
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (live) {
            // four rows per trip, their loads issued together (round 6: one row per trip kept 32 bytes in flight per thread - at 1,024
            // workgroups that is ~8 MB chip-wide, i.e. the 4.1-4.8 TB/s this kernel ran at is its latency bound, not the memory's);
            // rows are still accumulated one after the other in the old order: the column sums keep their bits
            constexpr int UR = 4;
            for (long long rb = r0 + rl; rb < r1; rb += (long long)row_lanes * UR) {
                u16x8 g[UR], yv[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const long long r = rb + (long long)u * row_lanes;
                    const long long off = (r < r1 ? r : r1 - 1) * p.C + cg * 8;
                    g[u] = *(const u16x8*)(p.dy + off);
                    if (p.y) yv[u] = *(const u16x8*)(p.y + off);
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const long long r = rb + (long long)u * row_lanes;
                    if (r < r1) {
                        const long long off = r * p.C + cg * 8;
                        float f[8];
                        if (p.y) {
                            u16x8 o;
                            for (int e = 0; e < 8; ++e) {
                                f[e] = gg_bf2f(g[u][e]) * (gg_bf2f(yv[u][e]) > 0.f ? 1.f : p.slope);
                                o[e] = gg_f2bf(f[e]);
                                f[e] = gg_bf2f(o[e]);
                            }
                            *(u16x8*)(p.dz + off) = o;
                        } else {
                            for (int e = 0; e < 8; ++e) f[e] = gg_bf2f(g[u][e]);
                        }
                        for (int e = 0; e < 8; ++e) acc[e] += f[e];
                    }
                }
            }
        }
        if (p.db) {
            for (int e = 0; e < 8; ++e) red[t][e] = live ? acc[e] : 0.f;
            gg_sync();
            if (rl == 0 && cg < ncg) {
                for (int k = 1; k < row_lanes; ++k)
                    for (int e = 0; e < 8; ++e) acc[e] += red[k * lanes_per_row + cgl][e];
                for (int e = 0; e < 8; ++e) p.db[(long long)blockIdx.x * p.C + cg * 8 + e] = acc[e];
            }
            gg_sync();
        }
    }
}

// ---- exact (erf) GELU of the attention feed-forward (nn.GELU, gp.py:731; unet.py:388) and its two derivatives --------
// One pass each over contiguous bf16, fp32 math:
//   mode 0: out0 = gelu(x)
//   mode 1: out0 = dy * gelu'(x)                                     (backward)
//   mode 2: out0 = g * gelu'(x) [d/d dy],  out1 = g * dy * gelu''(x) [d/d x]   (backward of the backward: the gradient
//           penalty's double backward, for which autograd otherwise chains ~12 pointwise launches over the 4x-wide hidden)
// gelu(x) = x Phi(x), gelu'(x) = Phi(x) + x phi(x), gelu''(x) = phi(x) (2 - x^2); phi = exp(-x^2/2)/sqrt(2 pi).
struct GgGeluParams {
    const bf16_t* x;
    const bf16_t* dy;   // modes 1, 2
    const bf16_t* g;    // mode 2
    bf16_t* out0;
    bf16_t* out1;       // mode 2
    long long n8;       // number of 8-element vectors
    int mode;
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_gelu_kernel(GgGeluParams p) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < p.n8; v += stride) {
        const u16x8 xv = *(const u16x8*)(p.x + v * 8);
        u16x8 dv = xv, gv = xv, o0, o1 = xv;
        if (p.mode >= 1) dv = *(const u16x8*)(p.dy + v * 8);
        if (p.mode == 2) gv = *(const u16x8*)(p.g + v * 8);
        for (int e = 0; e < 8; ++e) {
            const float x = gg_bf2f(xv[e]);
            float cdf, pdf;
            gg_normal_cdf_pdf(x, cdf, pdf);           // (the same arithmetic as the GEMM epilogue's fused form: bit-identical results)
            if (p.mode == 0) {
                o0[e] = gg_f2bf(x * cdf);
            } else {
                const float d1 = cdf + x * pdf;
                if (p.mode == 1) {
                    o0[e] = gg_f2bf(gg_bf2f(dv[e]) * d1);
                } else {
                    const float g = gg_bf2f(gv[e]);
                    o0[e] = gg_f2bf(g * d1);
                    o1[e] = gg_f2bf(g * gg_bf2f(dv[e]) * pdf * (2.f - x * x));
                }
            }
        }
        *(u16x8*)(p.out0 + v * 8) = o0;
        if (p.mode == 2) *(u16x8*)(p.out1 + v * 8) = o1;
    }
}

// ---- ChannelRMSNorm (gp.py:224-232): y = x / max(|x|, eps) * sqrt(C) * gamma over the channel axis ---------------
// NHWC makes the reduction contiguous: one wavefront per pixel row, 8 channels (16 bytes) per lane per pass, fp32
// statistics. Three passes exist because gradient-penalty steps differentiate the backward as well:
//   fwd :  y = s/n * x * gamma                                   (n = max(|x|, eps), s = sqrt(C))
//   bwd :  h = gamma*g;  dx = s/n * (h - u (u.h)),  dgamma[c] += s/n * x_c * g_c        (u = x/n; eps branch: dx = s/eps*h)
//   bwd2:  for an incoming v (gradient w.r.t. dx):   gg = gamma * s/n * (v - u (u.v)),   dgamma[c] += g_c * s/n * (v - u (u.v))_c
//          gx = -s/n^2 * ( u (v.h - (u.v)(u.h)) + (u.h) (v - u (u.v)) + (u.v) (h - u (u.h)) )
// replacing ~10 (fwd+bwd) and ~40 (second order) fp32 tensor-algebra passes. 2 B read + 2 B written per element (fwd).
#define GG_RMS_MAXV 4   // up to 4 x 512 = 2048 channels

struct GgRmsParams {
    const bf16_t* x;      // [rows][C]
    const bf16_t* g;      // bwd/bwd2: gradient w.r.t. y
    const bf16_t* v;      // bwd2: gradient w.r.t. dx ; bwd: optional carry added to dx (the skip branch's gradient)
    const float* gamma;   // [C]
    bf16_t* out0;         // fwd: y ; bwd: dx ; bwd2: gx
    bf16_t* out1;         // bwd2: gg
    float* dgamma_part;   // bwd / bwd2: [gridDim.x][C] partial sums (optional)
    long long rows;
    int C;
    float eps;
    int act;              // 1: y = silu(norm(x)) (the unet Block's activation, unet.py:268-269); fwd and bwd only
};

template <int MODE, bool ACT = false>   // MODE 0 fwd, 1 bwd, 2 bwd2; ACT: silu after the norm (a compile-time variant: the plain
                                        // kernels keep their register budget)
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_rmsnorm_kernel(GgRmsParams p) {
    GG_SHARED float red[4][GG_RMS_MAXV * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = (p.C + 511) / 512;
    const float s = sqrtf((float)p.C);
    float dgam[GG_RMS_MAXV][8];
#pragma unroll
    for (int t = 0; t < GG_RMS_MAXV; ++t)
        for (int e = 0; e < 8; ++e) dgam[t][e] = 0.f;
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < p.rows; r += nwaves) {
        float xf[GG_RMS_MAXV][8], hf[GG_RMS_MAXV][8], vf[GG_RMS_MAXV][8], gf[GG_RMS_MAXV][8];
        float ss = 0.f, uh = 0.f, uv = 0.f, vh = 0.f;
#pragma unroll
        for (int t = 0; t < GG_RMS_MAXV; ++t) {
            if (t >= nv) break;
            const int c = t * 512 + lane * 8;
            for (int e = 0; e < 8; ++e) { xf[t][e] = 0.f; hf[t][e] = 0.f; vf[t][e] = 0.f; gf[t][e] = 0.f; }
            if (c < p.C) {
                u16x8 xv = *(const u16x8*)(p.x + r * p.C + c);
                for (int e = 0; e < 8; ++e) { xf[t][e] = gg_bf2f(xv[e]); ss += xf[t][e] * xf[t][e]; }
                if (MODE >= 1) {
                    u16x8 gv = *(const u16x8*)(p.g + r * p.C + c);
                    for (int e = 0; e < 8; ++e) {
                        gf[t][e] = gg_bf2f(gv[e]);
                        hf[t][e] = gf[t][e] * p.gamma[c + e];
                        uh += xf[t][e] * hf[t][e];
                    }
                }
                if (MODE == 2) {
                    u16x8 vv = *(const u16x8*)(p.v + r * p.C + c);
                    for (int e = 0; e < 8; ++e) {
                        vf[t][e] = gg_bf2f(vv[e]);
                        uv += xf[t][e] * vf[t][e];
                        vh += vf[t][e] * hf[t][e];
                    }
                }
            }
        }
        ss = gg_wave_sum(ss);
        const float nrm = sqrtf(ss);
        const bool clamped = nrm < p.eps;
        const float n = clamped ? p.eps : nrm;
        const float rn = s / n;                   // s / n
        if (MODE == 1 && ACT) {
            // y = silu(z), z = x * rn * gamma: the incoming gradient is first taken through silu'(z) = s (1 + z (1 - s)), then the
            // plain backward runs on it (h and u.h are rebuilt from the adjusted gradient)
            uh = 0.f;
#pragma unroll
            for (int t = 0; t < GG_RMS_MAXV; ++t) {
                if (t >= nv) break;
                const int c = t * 512 + lane * 8;
                if (c < p.C)
                    for (int e = 0; e < 8; ++e) {
                        const float z = xf[t][e] * rn * p.gamma[c + e];
                        const float sg = 1.f / (1.f + gg_expf(-z));
                        gf[t][e] *= sg * (1.f + z * (1.f - sg));
                        hf[t][e] = gf[t][e] * p.gamma[c + e];
                        uh += xf[t][e] * hf[t][e];
                    }
            }
        }
        if (MODE >= 1) { uh = gg_wave_sum(uh) / n; }            // u.h  (u = x / n)
        if (MODE == 2) { uv = gg_wave_sum(uv) / n; vh = gg_wave_sum(vh); }
        if (clamped) { uh = 0.f; uv = 0.f; }      // y is linear in x below eps: the projection terms vanish
#pragma unroll
        for (int t = 0; t < GG_RMS_MAXV; ++t) {
            if (t >= nv) break;
            const int c = t * 512 + lane * 8;
            if (c < p.C) {
                u16x8 o0, o1;
                u16x8 cv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (MODE == 1 && p.v) cv = *(const u16x8*)(p.v + r * p.C + c);
                for (int e = 0; e < 8; ++e) {
                    const float u = xf[t][e] / n;
                    if (MODE == 0) {
                        float z = xf[t][e] * rn * p.gamma[c + e];
                        if (ACT) z = z / (1.f + gg_expf(-z));
                        o0[e] = gg_f2bf(z);
                    } else if (MODE == 1) {
                        o0[e] = gg_f2bf(rn * (hf[t][e] - u * uh) + gg_bf2f(cv[e]));
                        dgam[t][e] += rn * xf[t][e] * gf[t][e];
                    } else {
                        const float pv = vf[t][e] - u * uv;          // (P v)_c
                        const float ph = hf[t][e] - u * uh;          // (P h)_c
                        o1[e] = gg_f2bf(p.gamma[c + e] * rn * pv);
                        dgam[t][e] += gf[t][e] * rn * pv;
                        float gx = clamped ? 0.f : -(rn / n) * (u * (vh - uv * uh) + uh * pv + uv * ph);
                        o0[e] = gg_f2bf(gx);
                    }
                }
                *(u16x8*)(p.out0 + r * p.C + c) = o0;
                if (MODE == 2) *(u16x8*)(p.out1 + r * p.C + c) = o1;
            }
        }
    }
    if (MODE >= 1 && p.dgamma_part) {
#pragma unroll
        for (int t = 0; t < GG_RMS_MAXV; ++t) {
            if (t >= nv) break;
            for (int e = 0; e < 8; ++e) red[wave][t * 512 + lane * 8 + e] = dgam[t][e];
        }
        gg_sync();
        for (int c = threadIdx.x; c < p.C; c += 256)
            p.dgamma_part[(long long)blockIdx.x * p.C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

// ---- the same fwd / bwd passes for C = 8 * LPR <= 512 channels: LPR lanes per pixel row, 64 / LPR rows per wavefront and pass --------
// The kernel above gives a whole wavefront to one row: at C = 256 half of its lanes idle, at C = 64 seven in eight, and every row pays
// full 64-lane butterflies - 1.1-2.2 TB/s on the trainer's shapes (tests/gpu_rmsnorm_probe.py). Here a row belongs to a group of LPR
// lanes (one 16-byte vector each), a wavefront walks 64 / LPR rows per pass and keeps TWO passes in flight (all loads of both before the
// arithmetic). The group butterflies run over LPR lanes only; they add the same values in the same order as the first log2(LPR) steps of
// the 64-lane butterfly (whose remaining steps add zeros there): y / dx differ from the kernel above only where the compiler contracts an
// fma in one and not the other (1 bf16 ulp on ~1e-5 of the elements). The gain gradient's
// per-lane partial sums are folded over the row groups of a wavefront by the remaining butterfly steps, over the four wavefronts in LDS.
template <int LPR>
GG_DEVICE float gg_group_sum(float v) {
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) v += gg_shfl_xor(v, o);
    return v;
}

template <int MODE, bool ACT, int LPR>   // MODE 0 fwd, 1 bwd, 2 bwd2 (ACT: first order only)
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_rmsnorm_rows_kernel(GgRmsParams p) {
    GG_SHARED float red[4][512];
    constexpr int G = 64 / LPR;                    // rows per wavefront and pass
    constexpr int U = 2;                           // passes in flight
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, c = (lane % LPR) * 8;
    const float s = sqrtf((float)p.C);
    float gam[8], dgam[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gam[e] = p.gamma[c + e]; dgam[e] = 0.f; }
    const long long npass = (p.rows + G - 1) / G;
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long q0 = (long long)blockIdx.x * 4 + wave; q0 < npass; q0 += nwaves * U) {
        u16x8 xv[U], gv[U], cv[U];
        bool live[U];
        long long off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = (q0 + u * nwaves) * G + sub;
            live[u] = (q0 + u * nwaves) < npass && r < p.rows;
            off[u] = r * p.C + c;
            const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            xv[u] = z; gv[u] = z; cv[u] = z;
            if (live[u]) {
                xv[u] = *(const u16x8*)(p.x + off[u]);
                if (MODE >= 1) {
                    gv[u] = *(const u16x8*)(p.g + off[u]);
                    if (MODE == 2 || p.v) cv[u] = *(const u16x8*)(p.v + off[u]);     // bwd: the carry; bwd2: the incoming v
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float xf[8], hf[8], gf[8], vf[8];
            float ss = 0.f, uh = 0.f, uv = 0.f, vh = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xf[e] = gg_bf2f(xv[u][e]);
                ss += xf[e] * xf[e];
                if (MODE >= 1) {
                    gf[e] = gg_bf2f(gv[u][e]);
                    hf[e] = gf[e] * gam[e];
                    uh += xf[e] * hf[e];
                }
                if (MODE == 2) {
                    vf[e] = gg_bf2f(cv[u][e]);
                    uv += xf[e] * vf[e];
                    vh += vf[e] * hf[e];
                }
            }
            ss = gg_group_sum<LPR>(ss);
            const float nrm = sqrtf(ss);
            const bool clamped = nrm < p.eps;
            const float n = clamped ? p.eps : nrm;
            const float rn = s / n;
            const float rinv = 1.f / n;                  // (one division per row: `x / n` per element is ten instructions each)
            if (MODE == 1 && ACT) {
                uh = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = xf[e] * rn * gam[e];
                    const float sg = 1.f / (1.f + gg_expf(-z));
                    gf[e] *= sg * (1.f + z * (1.f - sg));
                    hf[e] = gf[e] * gam[e];
                    uh += xf[e] * hf[e];
                }
            }
            if (MODE >= 1) uh = gg_group_sum<LPR>(uh) / n;
            if (MODE == 2) { uv = gg_group_sum<LPR>(uv) / n; vh = gg_group_sum<LPR>(vh); }
            if (clamped) { uh = 0.f; uv = 0.f; }
            u16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (MODE == 0) {
                    float z = xf[e] * rn * gam[e];
                    if (ACT) z = z / (1.f + gg_expf(-z));
                    o0[e] = gg_f2bf(z);
                } else if (MODE == 1) {
                    const float uu = xf[e] * rinv;
                    o0[e] = gg_f2bf(rn * (hf[e] - uu * uh) + gg_bf2f(cv[u][e]));
                    dgam[e] += rn * xf[e] * gf[e];          // (dead rows carry x = 0)
                } else {
                    const float uu = xf[e] * rinv;
                    const float pv = vf[e] - uu * uv;       // (P v)_c
                    const float ph = hf[e] - uu * uh;       // (P h)_c
                    o1[e] = gg_f2bf(gam[e] * rn * pv);
                    dgam[e] += gf[e] * rn * pv;
                    const float gx = clamped ? 0.f : -(rn / n) * (uu * (vh - uv * uh) + uh * pv + uv * ph);
                    o0[e] = gg_f2bf(gx);
                }
            }
            if (live[u]) {
                *(u16x8*)(p.out0 + off[u]) = o0;
                if (MODE == 2) *(u16x8*)(p.out1 + off[u]) = o1;
            }
        }
    }
    if (MODE >= 1 && p.dgamma_part) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = dgam[e];
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) v += gg_shfl_xor(v, o);
            if (sub == 0) red[wave][c + e] = v;
        }
        gg_sync();
        for (int cc = threadIdx.x; cc < p.C; cc += 256)
            p.dgamma_part[(long long)blockIdx.x * p.C + cc] = red[0][cc] + red[1][cc] + red[2][cc] + red[3][cc];
    }
}

// ---- SqueezeExcite's pool (gp.py:300: mean over the pixels) and its backward -------------------------------------
// forward: x [b][P][C] bf16 -> part [b][chunks][C] fp32 partial sums (grid (chunks, b); deterministic: no atomics), then
// out [b][C] = scale * sum over chunks. backward: y[b][p][c] = g[b][p][c] + gs[b][c] (g optional: the gradient that reached
// the pooled tensor over its other consumer; y may alias g), one pass instead of expand + cast + add.
struct GgPoolParams {
    const bf16_t* x;     // fwd: activation ; bwd: incoming gradient of the other branch (may be null)
    float* part;
    float* out;          // fwd: [b][C]
    const float* gs;     // bwd: [b][C] per-sample per-channel constant
    bf16_t* y;           // bwd: output
    int b, P, C, chunks;
    float scale;
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_pool_partial_kernel(GgPoolParams p) {
    GG_SHARED float red[256 * 8];
    const int cvs = p.C >> 3;                    // 16-byte vectors per pixel (<= 64)
    const int pl_n = 256 / cvs;                  // pixel lanes
    const int t = threadIdx.x, cv = t % cvs, pl = t / cvs;
    const int chunk = blockIdx.x, img = blockIdx.y;
    const int per = (p.P + p.chunks - 1) / p.chunks;
    const int p0 = chunk * per, p1 = p0 + per < p.P ? p0 + per : p.P;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (pl < pl_n) {
        const bf16_t* base = p.x + ((long long)img * p.P) * p.C + cv * 8;
        for (int q = p0 + pl; q < p1; q += pl_n) {
            const u16x8 v = *(const u16x8*)(base + (long long)q * p.C);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += gg_bf2f(v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[t * 8 + e] = acc[e];
    gg_sync();
    for (int c = t; c < p.C; c += 256) {
        float s = 0.f;
        for (int l = 0; l < pl_n; ++l) s += red[(l * cvs + (c >> 3)) * 8 + (c & 7)];
        p.part[((long long)img * p.chunks + chunk) * p.C + c] = s;
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_pool_finish_kernel(GgPoolParams p) {
    const long long n = (long long)p.b * p.C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int img = (int)(i / p.C), c = (int)(i - (long long)img * p.C);
        float s = 0.f;
        for (int k = 0; k < p.chunks; ++k) s += p.part[((long long)img * p.chunks + k) * p.C + c];
        p.out[i] = s * p.scale;
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_pool_bwd_kernel(GgPoolParams p) {
    const int cvs = p.C >> 3;
    const long long n = (long long)p.b * p.P * cvs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvs);
        const long long pix = i / cvs;
        const int img = (int)(pix / p.P);
        const float* g8 = p.gs + (long long)img * p.C + cv * 8;
        u16x8 o;
        if (p.x) {
            const u16x8 v = *(const u16x8*)(p.x + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(v[e]) + g8[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(g8[e]);
        }
        *(u16x8*)(p.y + i * 8) = o;
    }
}

// ---- multi-scale input merge of the discriminator (gp.py:1797-1803): x = cat((x + feats, feats)) with feats tiled over the
// scale-major batch. forward: out [2B][n] from x [B][n] and feats [f][n] (B % f == 0, row r of the tiled feats = feats[r % f]);
// backward w.r.t. feats: gfeats[j] = sum over r = j (mod f) of (g[r] + g[B + r]) in fp32 (the gradient w.r.t. x is the view g[:B]).
struct GgAddCatParams {
    const bf16_t* x;
    const bf16_t* feats;
    bf16_t* out;         // fwd: [2B][n] ; bwd: gfeats [f][n]
    const bf16_t* g;     // bwd: [2B][n]
    long long n;         // elements per sample (multiple of 8)
    int B, f;
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_addcat_fwd_kernel(GgAddCatParams p) {
    const long long nv = p.n >> 3, total = (long long)p.B * nv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / nv, v = i - r * nv;
        const u16x8 a = *(const u16x8*)(p.x + i * 8);
        const u16x8 fv = *(const u16x8*)(p.feats + ((r % p.f) * nv + v) * 8);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(a[e]) + gg_bf2f(fv[e]));
        *(u16x8*)(p.out + i * 8) = o;
        *(u16x8*)(p.out + (total + i) * 8) = fv;
    }
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_addcat_bwd_kernel(GgAddCatParams p) {
    const long long nv = p.n >> 3, total = (long long)p.f * nv;
    const int reps = p.B / p.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long j = i / nv, v = i - j * nv;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int half = 0; half < 2; ++half)
            for (int t = 0; t < reps; ++t) {
                const long long r = (long long)half * p.B + (long long)t * p.f + j;
                const u16x8 gv = *(const u16x8*)(p.g + (r * nv + v) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += gg_bf2f(gv[e]);
            }
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(acc[e]);
        *(u16x8*)(p.out + i * 8) = o;
    }
}

// ---- hinge losses over a logit tensor in ONE launch (gp.py:157-163: generator_hinge_loss = mean(fake), discriminator_hinge_loss =
// mean(relu(1 + real) + relu(1 - fake))) and their backward in one more. x is (outer, nb, inner) bf16 or fp32 with the batch on the
// middle axis; mode 1: rows j < split are the FAKE half, the others the real half (the merged discriminator pass evaluates both as one
// batch), loss = sum over all elements of relu(1 + sign * x) / (n / 2); mode 0: loss = sum x / n. The tensors are tiny (<= a few 10^5
// elements) and the reference's formulation is ~16 PyTorch launches per tensor: up to 64 workgroups, fixed summation order (deterministic).
// dx == null: forward, loss[0] written. dx != null: dx = gscale[0] * d loss / d x in x's dtype.
struct GgHingeParams {
    const void* x;
    void* dx;
    const float* gscale;
    float* loss;
    unsigned* scratch;      // forward with more than one workgroup: [0] ticket (zero between launches), [1 + b] partial of workgroup b
    long long n, inner;
    int nb, split, mode, x_f32;
};

#define GG_HINGE_MAXB 64

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_hinge_kernel(GgHingeParams p) {
    GG_SHARED float red[256];
    GG_SHARED int last;
    const int t = threadIdx.x;
    const float inv = p.mode == 1 ? 2.f / (float)p.n : 1.f / (float)p.n;
    const float g = p.dx ? p.gscale[0] * inv : 0.f;
    const unsigned n = (unsigned)p.n, inner = (unsigned)p.inner, nb = (unsigned)p.nb;
    float acc = 0.f;
    for (unsigned i = blockIdx.x * 256u + t; i < n; i += gridDim.x * 256u) {
        const float x = p.x_f32 ? ((const float*)p.x)[i] : gg_bf2f(((const bf16_t*)p.x)[i]);
        float f, df;
        if (p.mode == 1) {
            const unsigned j = (i / inner) % nb;
            const float sgn = j < (unsigned)p.split ? -1.f : 1.f;
            const float v = 1.f + sgn * x;
            f = v > 0.f ? v : 0.f;
            df = v > 0.f ? sgn : 0.f;
        } else {
            f = x;
            df = 1.f;
        }
        acc += f;
        if (p.dx) {
            if (p.x_f32) ((float*)p.dx)[i] = g * df;
            else ((bf16_t*)p.dx)[i] = gg_f2bf(g * df);
        }
    }
    if (p.dx) return;
    red[t] = acc;
    gg_sync();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        gg_sync();
    }
    if (gridDim.x == 1) {
        if (t == 0) p.loss[0] = red[0] * inv;
        return;
    }
    // several workgroups: partials in a fixed slot each, the last workgroup to arrive adds them in slot order (deterministic) and
    // leaves the ticket at zero for the next launch on this stream
    if (t == 0) {
        ((volatile float*)p.scratch)[1 + blockIdx.x] = red[0];
        last = gg_ticket_take(p.scratch) == gridDim.x - 1;
    }
    gg_sync();
    if (last && t == 0) {
        float sum = 0.f;
        for (unsigned b = 0; b < gridDim.x; ++b) sum += ((volatile float*)p.scratch)[1 + b];
        p.loss[0] = sum * inv;
        gg_ticket_reset(p.scratch);
    }
}

// ---- y = (a + b) * c [+ d] over dense bf16 buffers (b, d optional): the predictor's residual merges (gp.py:1493, :1495) in one
// pass; with b, d null it is the merge's backward (g * c for both inputs).
struct GgScaledAddParams {
    const bf16_t* a;
    const bf16_t* b;
    const bf16_t* d;     // optional: y = (a + b) * c + d
    bf16_t* y;
    long long n;     // multiple of 8
    float c;
};

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_scaled_add_kernel(GgScaledAddParams p) {
    const long long nv = p.n >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        const u16x8 av = *(const u16x8*)(p.a + i * 8);
        u16x8 o;
        if (p.b) {
            const u16x8 bv = *(const u16x8*)(p.b + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf((gg_bf2f(av[e]) + gg_bf2f(bv[e])) * p.c);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(av[e]) * p.c);
        }
        if (p.d) {
            const u16x8 dv = *(const u16x8*)(p.d + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(o[e]) + gg_bf2f(dv[e]));
        }
        *(u16x8*)(p.y + i * 8) = o;
    }
}

// ---- LinearAttention softmaxes (unet.py:338-348) over NHWC bf16 channel slices -------------------------------------------
// q: softmax over the 64 features of each head at every position, times `scale`; k: softmax over the positions of every
// feature. Inputs are channel slices of the fused to_qkv output ([rows][ld_in], C = heads * 64 channels from a base pointer),
// outputs dense [rows][C] (forward) or channel slices of the fused qkv gradient (backward). fp32 statistics.
struct GgLinAttnParams {
    const bf16_t* x;     // q-forward: q ; q-backward: qs ; k passes: k resp. eks
    const bf16_t* g;     // backward passes: gradient w.r.t. qs resp. eks
    bf16_t* y;
    float* part;         // k passes: [b][chunks][C][2] partial (max, sum) resp. [b][chunks][C] partial dots
    float* stat;         // k passes: [b][C][2] (max, sum) resp. [b][C] dots
    long long rows;      // q passes: b * n
    int ld_x, ld_g, ld_y, C, b, n, chunks;
    float scale;
};

// q: one 8-lane group per (position, head); MODE 0: qs = scale * softmax ; MODE 1: dq = p * (scale*dqs - sum(p * scale*dqs)), p = qs / scale
template <int MODE>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_linattn_q_kernel(GgLinAttnParams p) {
    const int heads = p.C >> 6;
    const long long groups = p.rows * heads;
    const int sub = threadIdx.x & 7;
    for (long long gidx = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3; gidx < groups; gidx += ((long long)gridDim.x * 256) >> 3) {
        const long long row = gidx / heads;
        const int h = (int)(gidx - row * heads);
        const u16x8 xv = *(const u16x8*)(p.x + row * p.ld_x + h * 64 + sub * 8);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gg_bf2f(xv[e]);
        u16x8 o;
        if (MODE == 0) {
            float mx = v[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) mx = fmaxf(mx, v[e]);
            mx = fmaxf(mx, gg_shfl_xor(mx, 1)); mx = fmaxf(mx, gg_shfl_xor(mx, 2)); mx = fmaxf(mx, gg_shfl_xor(mx, 4));
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = gg_expf(v[e] - mx); s += v[e]; }
            s += gg_shfl_xor(s, 1); s += gg_shfl_xor(s, 2); s += gg_shfl_xor(s, 4);
            const float r = p.scale / s;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(v[e] * r);
        } else {
            const u16x8 gv = *(const u16x8*)(p.g + row * p.ld_g + h * 64 + sub * 8);
            float gq[8], dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { gq[e] = gg_bf2f(gv[e]); dot += v[e] * gq[e]; }     // sum qs * dqs = scale * sum p * dqs
            dot += gg_shfl_xor(dot, 1); dot += gg_shfl_xor(dot, 2); dot += gg_shfl_xor(dot, 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(v[e] * (gq[e] - dot / p.scale));
        }
        *(u16x8*)(p.y + row * p.ld_y + h * 64 + sub * 8) = o;
    }
}

// k statistics, stage 1: grid (chunks, b); thread -> channel vector t % (C/8), position lane t / (C/8).
// MODE 0: running (max, sum exp) per channel over the chunk's positions; MODE 1: sum of x * g per channel
template <int MODE>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_linattn_kpart_kernel(GgLinAttnParams p) {
    GG_SHARED float red[256 * 8 * 2];
    const int cvs = p.C >> 3, pl_n = 256 / cvs;
    const int t = threadIdx.x, cv = t % cvs, pl = t / cvs;
    const int chunk = blockIdx.x, img = blockIdx.y;
    const int per = (p.n + p.chunks - 1) / p.chunks;
    const int p0 = chunk * per, p1 = p0 + per < p.n ? p0 + per : p.n;
    float m[8], s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -3.0e38f; s[e] = 0.f; }
    if (pl < pl_n) {
        const bf16_t* xb = p.x + (long long)img * p.n * p.ld_x + cv * 8;
        const bf16_t* gb = MODE == 1 ? p.g + (long long)img * p.n * p.ld_g + cv * 8 : nullptr;
        for (int q = p0 + pl; q < p1; q += pl_n) {
            const u16x8 xv = *(const u16x8*)(xb + (long long)q * p.ld_x);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = gg_bf2f(xv[e]);
                    const float mn = fmaxf(m[e], x);
                    s[e] = s[e] * gg_expf(m[e] - mn) + gg_expf(x - mn);
                    m[e] = mn;
                }
            } else {
                const u16x8 gv = *(const u16x8*)(gb + (long long)q * p.ld_g);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += gg_bf2f(xv[e]) * gg_bf2f(gv[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[(t * 8 + e) * 2] = m[e]; red[(t * 8 + e) * 2 + 1] = s[e]; }
    gg_sync();
    for (int c = t; c < p.C; c += 256) {
        float mm = -3.0e38f, ss = 0.f;
        for (int l = 0; l < pl_n; ++l) {
            const int idx = ((l * cvs + (c >> 3)) * 8 + (c & 7)) * 2;
            if (MODE == 0) {
                const float m2 = red[idx], s2 = red[idx + 1];
                const float mn = fmaxf(mm, m2);
                ss = ss * gg_expf(mm - mn) + s2 * gg_expf(m2 - mn);
                mm = mn;
            } else {
                ss += red[idx + 1];
            }
        }
        const long long o = ((long long)img * p.chunks + chunk) * p.C + c;
        if (MODE == 0) { p.part[o * 2] = mm; p.part[o * 2 + 1] = ss; }
        else p.part[o] = ss;
    }
}

// stage 2: combine the chunks
template <int MODE>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_linattn_kfinish_kernel(GgLinAttnParams p) {
    const long long nbc = (long long)p.b * p.C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nbc; i += (long long)gridDim.x * 256) {
        const int img = (int)(i / p.C), c = (int)(i - (long long)img * p.C);
        float mm = -3.0e38f, ss = 0.f;
        for (int k = 0; k < p.chunks; ++k) {
            const long long o = ((long long)img * p.chunks + k) * p.C + c;
            if (MODE == 0) {
                const float m2 = p.part[o * 2], s2 = p.part[o * 2 + 1];
                const float mn = fmaxf(mm, m2);
                ss = ss * gg_expf(mm - mn) + s2 * gg_expf(m2 - mn);
                mm = mn;
            } else {
                ss += p.part[o];
            }
        }
        if (MODE == 0) { p.stat[i * 2] = mm; p.stat[i * 2 + 1] = ss; }
        else p.stat[i] = ss;
    }
}

// apply: MODE 0: eks = exp(k - max) / sum ; MODE 1: dk = eks * (deks - dot)
template <int MODE>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_linattn_kapply_kernel(GgLinAttnParams p) {
    const int cvs = p.C >> 3;
    const long long total = (long long)p.b * p.n * cvs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvs);
        const long long row = i / cvs;
        const int img = (int)(row / p.n);
        const u16x8 xv = *(const u16x8*)(p.x + row * p.ld_x + cv * 8);
        u16x8 o;
        if (MODE == 0) {
            const float* st = p.stat + ((long long)img * p.C + cv * 8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_expf(gg_bf2f(xv[e]) - st[2 * e]) / st[2 * e + 1]);
        } else {
            const u16x8 gv = *(const u16x8*)(p.g + row * p.ld_g + cv * 8);
            const float* st = p.stat + (long long)img * p.C + cv * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(xv[e]) * (gg_bf2f(gv[e]) - st[e]));
        }
        *(u16x8*)(p.y + row * p.ld_y + cv * 8) = o;
    }
}

// ---- unet Downsample tail (reference unet_upsampler.py:134-160): pooled = max_pool2d(x, 2) and the high-frequency map
// hf = x - blur(x) (blur = kornia filter2d, normalised [1,2,1]^2 / 16, border 'reflect') that rides the skip connection -------
// forward: one thread per (2x2 output cell, 8-channel vector): the cell's 4x4 neighbourhood (16-byte loads, reflected at the image
// border; the overlap between neighbouring cells is served by the vector L1 / L2) -> four hf vectors + one pooled vector.
// HBM: x read once, hf + pooled written once (the stock formulation: max-pool pass + blur pass + two fp32 casts + a subtraction).
// backward: dx = scatter(g_pool to the window's arg-max, first maximum in row-major window order as torch) + g_hf - blur^T(g_hf);
// blur^T is the exact adjoint of the reflect-padded stencil (border rows / columns collect the reflected taps).
struct GgPoolHfParams {
    const bf16_t* x;        // [b][H][W][C]
    bf16_t* pool;           // fwd out [b][H/2][W/2][C]
    bf16_t* hf;             // fwd out [b][H][W][C]
    const bf16_t* g_pool;   // bwd in (may be null)
    const bf16_t* g_hf;     // bwd in (may be null)
    bf16_t* dx;             // bwd out [b][H][W][C]
    int b, H, W, C;
};

GG_DEVICE int gg_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_poolhf_fwd_kernel(GgPoolHfParams p) {
    const int cvs = p.C >> 3, OH = p.H >> 1, OW = p.W >> 1;
    const long long n = (long long)p.b * OH * OW * cvs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvs);
        long long cell = i / cvs;
        const int ox = (int)(cell % OW);
        cell /= OW;
        const int oy = (int)(cell % OH), img = (int)(cell / OH);
        const bf16_t* xi = p.x + ((long long)img * p.H * p.W) * p.C + cv * 8;
        u16x8 v[4][4];                       // rows 2oy-1 .. 2oy+2, columns 2ox-1 .. 2ox+2 (all loads issued before the first use)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int yy = gg_reflect(2 * oy - 1 + r, p.H);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xx = gg_reflect(2 * ox - 1 + c, p.W);
                v[r][c] = *(const u16x8*)(xi + ((long long)yy * p.W + xx) * p.C);
            }
        }
        u16x8 mx;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) f[r][c] = gg_bf2f(v[r][c][e]);
            float m = f[1][1];
            m = f[1][2] > m ? f[1][2] : m;
            m = f[2][1] > m ? f[2][1] : m;
            m = f[2][2] > m ? f[2][2] : m;
            mx[e] = gg_f2bf(m);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int r = 1 + dy, c = 1 + dx;
                    const float blur = (f[r - 1][c - 1] + 2.f * f[r - 1][c] + f[r - 1][c + 1] + 2.f * f[r][c - 1] + 4.f * f[r][c] +
                                        2.f * f[r][c + 1] + f[r + 1][c - 1] + 2.f * f[r + 1][c] + f[r + 1][c + 1]) * 0.0625f;
                    v[r][c][e] = gg_f2bf(f[r][c] - blur);
                }
        }
        *(u16x8*)(p.pool + (((long long)img * OH + oy) * OW + ox) * p.C + cv * 8) = mx;
        bf16_t* hi = p.hf + ((long long)img * p.H * p.W) * p.C + cv * 8;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                *(u16x8*)(hi + ((long long)(2 * oy + dy) * p.W + 2 * ox + dx) * p.C) = v[1 + dy][1 + dx];
    }
}

// weight of g[y] in dx[u] along one axis: sum of the taps w_k = {1, 2, 1} / 4 of output y that read input u (reflected)
GG_DEVICE float gg_blur_adj(int u, int y, int n) {
    float c = 0.f;
    if (y < 0 || y >= n) return 0.f;
    if (gg_reflect(y - 1, n) == u) c += 0.25f;
    if (y == u) c += 0.5f;
    if (gg_reflect(y + 1, n) == u) c += 0.25f;
    return c;
}

GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_poolhf_bwd_kernel(GgPoolHfParams p) {
    const int cvs = p.C >> 3, OH = p.H >> 1, OW = p.W >> 1;
    const long long n = (long long)p.b * OH * OW * cvs;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvs);
        long long cell = i / cvs;
        const int ox = (int)(cell % OW);
        cell /= OW;
        const int oy = (int)(cell % OH), img = (int)(cell / OH);
        const long long ibase = ((long long)img * p.H * p.W) * p.C + cv * 8;
        float acc[2][2][8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[dy][dx][e] = 0.f;
        if (p.g_hf) {
            // g_hf at rows 2oy-1 .. 2oy+2 / columns 2ox-1 .. 2ox+2 (clamped loads; out-of-image rows get weight 0 below)
            u16x8 g[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int yy = 2 * oy - 1 + r, yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int xx = 2 * ox - 1 + c, xc = xx < 0 ? 0 : (xx >= p.W ? p.W - 1 : xx);
                    g[r][c] = *(const u16x8*)(p.g_hf + ibase + ((long long)yc * p.W + xc) * p.C);
                }
            }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int u = 2 * oy + dy, w = 2 * ox + dx;
                    float wy[3], wx[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        wy[k] = gg_blur_adj(u, u - 1 + k, p.H);
                        wx[k] = gg_blur_adj(w, w - 1 + k, p.W);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float s = 0.f;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) s += wy[ky] * wx[kx] * gg_bf2f(g[dy + ky][dx + kx][e]);
                        acc[dy][dx][e] = gg_bf2f(g[1 + dy][1 + dx][e]) - s;
                    }
                }
        }
        if (p.g_pool) {
            const u16x8 gp = *(const u16x8*)(p.g_pool + (((long long)img * OH + oy) * OW + ox) * p.C + cv * 8);
            u16x8 xv[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    xv[dy][dx] = *(const u16x8*)(p.x + ibase + ((long long)(2 * oy + dy) * p.W + 2 * ox + dx) * p.C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int by = 0, bx = 0;
                float m = gg_bf2f(xv[0][0][e]);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const float f = gg_bf2f(xv[dy][dx][e]);
                        if (f > m) { m = f; by = dy; bx = dx; }
                    }
                const float gv = gg_bf2f(gp[e]);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) acc[dy][dx][e] += (dy == by && dx == bx) ? gv : 0.f;
            }
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(acc[dy][dx][e]);
                *(u16x8*)(p.dx + ibase + ((long long)(2 * oy + dy) * p.W + 2 * ox + dx) * p.C) = o;
            }
    }
}
