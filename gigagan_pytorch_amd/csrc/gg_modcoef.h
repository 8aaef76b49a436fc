// gg_modcoef.h — the per-sample coefficients of the adaptive convolution (reference AdaptiveConv2DMod.forward,
// gp.py:378-400) in one launch forward and two backward, instead of ~13 + ~20 tiny tensor-algebra launches per layer:
//     s[b,i] = mod[b,i] + 1                                   (style modulation, gp.py:394-396)
//     a[b,n] = softmax_n(kernel_mod[b,:])                     (kernel selection, gp.py:380-388)
//     d[b,o] = rsqrt(max(sum_{i,k} (sum_n a[b,n] W[n,o,i,k] s[b,i])^2, eps))     (demodulation, gp.py:398-400)
// The per-sample weights (b,O,I,k,k) of the reference are never formed: a workgroup owns one output channel o, streams
// W[:,o,:,:] once per chunk of 16 samples (s and a of the chunk staged in LDS) and reduces over (i,k).
// Backward, with g[b,o] = dL/d(sumsq) = -1/2 d^3 gd (0 where the clamp was active) and M = sum_n a W (per b,o,i,k):
//     gW[n,o,i,k] += sum_b 2 g M s^2 a[b,n]                  (kernel B1, workgroup per o; accumulated in place)
//     ga[b,n]     += sum_{o,i,k} 2 g M s^2 W[n,o,i,k]        (kernel B1, block reduction + one atomic per (b,n,o))
//     gs[b,i]      = gs_ext + 2 s sum_{o,k} g M^2            (kernel B2, workgroup per i)
//     gmod = gs,   gkernel_mod = a * (ga_tot - sum_n a ga_tot)   (softmax backward, workgroup 0 of B2)
// fp32 throughout ((b,O)-sized statistics). N <= 4 kernels, I <= 1024.
#pragma once
#include "gg_device.h"

#define GG_MC_CB 16        // samples per chunk
#define GG_MC_NMAX 4
#define GG_MC_IMAX 1024

struct GgModCoefParams {
    const float* w;        // (N, O, I, T)
    const float* mod;      // (b, I)           forward input
    const float* kmod;     // (b, N) or null   (N == 1: a = 1)
    float* s;              // (b, Ip)          forward output / backward input
    float* a;              // (b, N)
    float* d;              // (b, Op) or null: no demodulation
    const float* gs;       // (b, Ip) or null  gradient w.r.t. s from the modulate pass
    const float* ga;       // (b, N) or null   gradient w.r.t. a from the mix pass
    const float* gd;       // (b, Op)          gradient w.r.t. d from the mix pass
    float* gmod;           // (b, I)   out
    float* gkmod;          // (b, N)   out or null
    float* da_acc;         // (b, N)   zero-initialised: gradient w.r.t. a through d (B1 atomics -> B2)
    float* gw;             // (N, O, I, T) accumulated in place, or null
    int b, N, O, I, T, Ip, Op;
    float eps;
};

GG_DEVICE float gg_mc_wave_sum(float v) {
    v += gg_shfl_xor(v, 1); v += gg_shfl_xor(v, 2); v += gg_shfl_xor(v, 4);
    v += gg_shfl_xor(v, 8); v += gg_shfl_xor(v, 16); v += gg_shfl_xor(v, 32);
    return v;
}

// a_s[bb][n] = softmax(kmod[row]) for the chunk's rows (threads 0..CB-1); rows >= b get zeros
GG_DEVICE void gg_mc_stage_a(const GgModCoefParams& p, float (*a_s)[GG_MC_NMAX], int b0) {
    const int t = threadIdx.x;
    if (t < GG_MC_CB) {
        const int row = b0 + t;
        float v[GG_MC_NMAX];
        for (int n = 0; n < GG_MC_NMAX; ++n) v[n] = 0.f;
        if (row < p.b) {
            if (p.kmod && p.N > 1) {
                float mx = -3.0e38f;
                for (int n = 0; n < p.N; ++n) { v[n] = p.kmod[row * p.N + n]; mx = v[n] > mx ? v[n] : mx; }
                float sum = 0.f;
                for (int n = 0; n < p.N; ++n) { v[n] = gg_expf(v[n] - mx); sum += v[n]; }
                for (int n = 0; n < p.N; ++n) v[n] /= sum;
            } else {
                v[0] = 1.f;
            }
        }
        for (int n = 0; n < GG_MC_NMAX; ++n) a_s[t][n] = v[n];
    }
}

// grid: O workgroups when demodulating, else min(b, 256)
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modcoef_fwd_kernel(GgModCoefParams p) {
    GG_SHARED float a_s[GG_MC_CB][GG_MC_NMAX];
    GG_SHARED float red[4][GG_MC_CB];
    GG_SHARED float s_s[GG_MC_CB * GG_MC_IMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x;
    const int IT = p.I * p.T;
    // rows of s / a (and the zero padding of d) are written by the workgroups in turn
    for (int row = blockIdx.x; row < p.b; row += gridDim.x) {
        for (int i = tid; i < p.Ip; i += 256) p.s[(long long)row * p.Ip + i] = i < p.I ? p.mod[(long long)row * p.I + i] + 1.f : 0.f;
        if (p.d)
            for (int c = p.O + tid; c < p.Op; c += 256) p.d[(long long)row * p.Op + c] = 0.f;
    }
    for (int b0 = 0; b0 < p.b; b0 += GG_MC_CB) {
        gg_sync();
        gg_mc_stage_a(p, a_s, b0);
        if (p.d && o < p.O)
            for (int idx = tid; idx < GG_MC_CB * p.I; idx += 256) {
                const int bb = idx / p.I, i = idx - bb * p.I;
                s_s[bb * p.I + i] = (b0 + bb < p.b) ? p.mod[(long long)(b0 + bb) * p.I + i] + 1.f : 0.f;
            }
        gg_sync();
        if (blockIdx.x == 0 && tid < GG_MC_CB && b0 + tid < p.b)
            for (int n = 0; n < p.N; ++n) p.a[(b0 + tid) * p.N + n] = a_s[tid][n];
        if (!p.d || o >= p.O) continue;
        float acc[GG_MC_CB];
        for (int bb = 0; bb < GG_MC_CB; ++bb) acc[bb] = 0.f;
        for (int e = tid; e < IT; e += 256) {
            const int i = e / p.T;
            float wv[GG_MC_NMAX];
            for (int n = 0; n < GG_MC_NMAX; ++n) wv[n] = n < p.N ? p.w[((long long)n * p.O + o) * IT + e] : 0.f;
            for (int bb = 0; bb < GG_MC_CB; ++bb) {
                float m = 0.f;
                for (int n = 0; n < GG_MC_NMAX; ++n) m += a_s[bb][n] * wv[n];
                m *= s_s[bb * p.I + i];
                acc[bb] += m * m;
            }
        }
        for (int bb = 0; bb < GG_MC_CB; ++bb) {
            const float v = gg_mc_wave_sum(acc[bb]);
            if (lane == 0) red[wave][bb] = v;
        }
        gg_sync();
        if (tid < GG_MC_CB && b0 + tid < p.b) {
            const float sumsq = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            p.d[(long long)(b0 + tid) * p.Op + o] = gg_rsqrtf(sumsq > p.eps ? sumsq : p.eps);
        }
    }
}

// g[b,o] = dL/d(sumsq): zero where the clamp was active (d == rsqrt(eps))
GG_DEVICE float gg_mc_g(const GgModCoefParams& p, int row, int o) {
    const float dv = p.d[(long long)row * p.Op + o];
    const float lim = gg_rsqrtf(p.eps);
    return dv < lim ? -0.5f * p.gd[(long long)row * p.Op + o] * dv * dv * dv : 0.f;
}

// B1: one workgroup per output channel o: gW[:,o,:,:] += ..., da_acc[b,n] += ...
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modcoef_bwd_w_kernel(GgModCoefParams p) {
    GG_SHARED float a_s[GG_MC_CB][GG_MC_NMAX];
    GG_SHARED float g_s[GG_MC_CB];
    GG_SHARED float red[4][GG_MC_CB * GG_MC_NMAX];
    GG_SHARED float s_s[GG_MC_CB * GG_MC_IMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x;
    const int IT = p.I * p.T;
    for (int b0 = 0; b0 < p.b; b0 += GG_MC_CB) {
        gg_sync();
        gg_mc_stage_a(p, a_s, b0);
        if (tid < GG_MC_CB) g_s[tid] = (b0 + tid < p.b) ? gg_mc_g(p, b0 + tid, o) : 0.f;
        for (int idx = tid; idx < GG_MC_CB * p.I; idx += 256) {
            const int bb = idx / p.I, i = idx - bb * p.I;
            s_s[bb * p.I + i] = (b0 + bb < p.b) ? p.s[(long long)(b0 + bb) * p.Ip + i] : 0.f;
        }
        gg_sync();
        float da[GG_MC_CB][GG_MC_NMAX];
        for (int bb = 0; bb < GG_MC_CB; ++bb)
            for (int n = 0; n < GG_MC_NMAX; ++n) da[bb][n] = 0.f;
        for (int e = tid; e < IT; e += 256) {
            const int i = e / p.T;
            float wv[GG_MC_NMAX], gwv[GG_MC_NMAX];
            for (int n = 0; n < GG_MC_NMAX; ++n) { wv[n] = n < p.N ? p.w[((long long)n * p.O + o) * IT + e] : 0.f; gwv[n] = 0.f; }
            for (int bb = 0; bb < GG_MC_CB; ++bb) {
                float m = 0.f;
                for (int n = 0; n < GG_MC_NMAX; ++n) m += a_s[bb][n] * wv[n];
                const float sv = s_s[bb * p.I + i];
                const float c = 2.f * g_s[bb] * m * sv * sv;
                for (int n = 0; n < GG_MC_NMAX; ++n) { gwv[n] += c * a_s[bb][n]; da[bb][n] += c * wv[n]; }
            }
            if (p.gw)
                for (int n = 0; n < p.N; ++n) p.gw[((long long)n * p.O + o) * IT + e] += gwv[n];
        }
        if (p.N > 1) {
            for (int bb = 0; bb < GG_MC_CB; ++bb)
                for (int n = 0; n < GG_MC_NMAX; ++n) {
                    const float v = gg_mc_wave_sum(da[bb][n]);
                    if (lane == 0) red[wave][bb * GG_MC_NMAX + n] = v;
                }
            gg_sync();
            if (tid < GG_MC_CB * GG_MC_NMAX) {
                const int bb = tid / GG_MC_NMAX, n = tid - bb * GG_MC_NMAX;
                if (b0 + bb < p.b && n < p.N)
                    gg_atomic_add(p.da_acc + (b0 + bb) * p.N + n, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
            }
        }
    }
}

// B2: one workgroup per input channel i: gmod[:, i]; workgroup 0 also finishes the kernel-selection softmax backward
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modcoef_bwd_s_kernel(GgModCoefParams p) {
    GG_SHARED float a_s[GG_MC_CB][GG_MC_NMAX];
    GG_SHARED float red[4][GG_MC_CB];
    GG_SHARED float g_s[GG_MC_CB * GG_MC_IMAX];        // g[bb][o], O <= GG_MC_IMAX
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x;
    const int OT = p.O * p.T;
    for (int b0 = 0; b0 < p.b; b0 += GG_MC_CB) {
        gg_sync();
        gg_mc_stage_a(p, a_s, b0);
        for (int idx = tid; idx < GG_MC_CB * p.O; idx += 256) {
            const int bb = idx / p.O, o = idx - bb * p.O;
            g_s[bb * p.O + o] = (b0 + bb < p.b) ? gg_mc_g(p, b0 + bb, o) : 0.f;
        }
        gg_sync();
        if (blockIdx.x == 0 && p.gkmod && tid < GG_MC_CB && b0 + tid < p.b) {
            const int row = b0 + tid;
            float gt[GG_MC_NMAX], dot = 0.f;
            for (int n = 0; n < p.N; ++n) {
                gt[n] = (p.ga ? p.ga[row * p.N + n] : 0.f) + p.da_acc[row * p.N + n];
                dot += a_s[tid][n] * gt[n];
            }
            for (int n = 0; n < p.N; ++n) p.gkmod[row * p.N + n] = a_s[tid][n] * (gt[n] - dot);
        }
        float acc[GG_MC_CB];
        for (int bb = 0; bb < GG_MC_CB; ++bb) acc[bb] = 0.f;
        for (int e = tid; e < OT; e += 256) {
            const int o = e / p.T, k = e - o * p.T;
            float wv[GG_MC_NMAX];
            for (int n = 0; n < GG_MC_NMAX; ++n) wv[n] = n < p.N ? p.w[(((long long)n * p.O + o) * p.I + i) * p.T + k] : 0.f;
            for (int bb = 0; bb < GG_MC_CB; ++bb) {
                float m = 0.f;
                for (int n = 0; n < GG_MC_NMAX; ++n) m += a_s[bb][n] * wv[n];
                acc[bb] += g_s[bb * p.O + o] * m * m;
            }
        }
        for (int bb = 0; bb < GG_MC_CB; ++bb) {
            const float v = gg_mc_wave_sum(acc[bb]);
            if (lane == 0) red[wave][bb] = v;
        }
        gg_sync();
        if (tid < GG_MC_CB && b0 + tid < p.b) {
            const int row = b0 + tid;
            const float sv = p.s[(long long)row * p.Ip + i];
            const float tot = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            p.gmod[(long long)row * p.I + i] = (p.gs ? p.gs[(long long)row * p.Ip + i] : 0.f) + 2.f * sv * tot;
        }
    }
}
