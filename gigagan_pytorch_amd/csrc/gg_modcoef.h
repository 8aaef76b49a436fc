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

// ===== the same coefficients through the bank's Gram rows ===========================================================================
// The kernels above walk (b, o, i, t): 75 M (sample, weight) pairs on a 512 x 512 x 9 bank at batch 32, ~17 VALU operations each -
// 30 M wavefront instructions per backward launch, 200-290 us, VALU-bound (profiles/r04_pmc_modcoef.log), 2 ms per training step.
// With H[n,m,o,i] = sum_t W_n W_m (the tap sum does not depend on the sample):
//     sumsq[b,o] = sum_i s^2[b,i] sum_{n<=m} a_n a_m G[(n,m),o,i],        G = H on the diagonal, 2 H off it           (t is gone)
//                = sum_p A[b,p] T[b,p,o],   A_p = a_n a_m,   T[b,p,o] = sum_i s^2[b,i] G[p,o,i]
// and, with g[b,o] = dL/d sumsq as above,
//     ga[b,k]    += sum_o g[b,o] sum_p T[b,p,o] dA_p/da_k                                                    (slots per o, added by K_i)
//     gs[b,i]     = gs_ext + 2 s[b,i] sum_p A[b,p] R[b,p,i],   R[b,p,i] = sum_o g[b,o] G[p,o,i]
//     gW[n,o,i,t] += sum_m W[m,o,i,t] Q[(n,m),o,i],            Q[p,o,i] = sum_b 2 g[b,o] A[b,p] s^2[b,i]
// T, R and Q are three thin contractions of 2 * b * O * I * P flops each (0.05 GF at the shape above) and the weights are touched once,
// in the element-wise gW update. P = N (N + 1) / 2 pairs, ordered n <= m row-major (the order of gg_pack_weights' 'gram' kind).
#define GG_MG_PMAX 10          // N = 4
#define GG_MG_BMAX 64          // samples per launch (the per-sample tables of a workgroup live in LDS / registers)

struct GgModGramParams {
    const float* w;        // (N, O, I, T)
    float* gram;           // (P, O, I): written by gg_modgram_kernel, read by the others
    const float* mod;      // (b, I)
    const float* kmod;     // (b, N) or null
    float* s;              // (b, Ip)
    float* a;              // (b, N)
    float* d;              // (b, Op)
    float* tsum;           // (b, P, O): T, kept for the backward
    const float* gs;       // (b, Ip) or null
    const float* ga;       // (b, N) or null
    const float* gd;       // (b, Op)
    float* gmod;           // (b, I)
    float* gkmod;          // (b, N) or null
    float* da_slots;       // (O, b, N) scratch: ga through d, one slot per output channel (K_o -> K_i workgroup 0)
    float* gw;             // (N, O, I, T) accumulated in place, or null
    int b, N, P, O, I, T, Ip, Op;
    float eps;
};

template <int NN>
GG_DEVICE void gg_mg_softmax(const GgModGramParams& p, int row, float (&a)[NN]) {
    if (p.kmod && NN > 1) {
        float mx = -3.0e38f;
#pragma unroll
        for (int n = 0; n < NN; ++n) { a[n] = p.kmod[row * NN + n]; mx = a[n] > mx ? a[n] : mx; }
        float sum = 0.f;
#pragma unroll
        for (int n = 0; n < NN; ++n) { a[n] = gg_expf(a[n] - mx); sum += a[n]; }
#pragma unroll
        for (int n = 0; n < NN; ++n) a[n] /= sum;
    } else {
#pragma unroll
        for (int n = 0; n < NN; ++n) a[n] = n == 0 ? 1.f : 0.f;
    }
}

// G[p][o][i]: workgroup per output channel, threads along the input channels
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modgram_kernel(GgModGramParams p) {
    const int o = blockIdx.x;
    for (int i = threadIdx.x; i < p.I; i += 256) {
        float acc[GG_MG_PMAX];
        for (int q = 0; q < GG_MG_PMAX; ++q) acc[q] = 0.f;
        for (int t = 0; t < p.T; ++t) {
            float wv[GG_MC_NMAX];
            for (int n = 0; n < GG_MC_NMAX; ++n) wv[n] = n < p.N ? p.w[(((long long)n * p.O + o) * p.I + i) * p.T + t] : 0.f;
            int q = 0;
            for (int n = 0; n < p.N; ++n)
                for (int m = n; m < p.N; ++m, ++q) acc[q] += wv[n] * wv[m];
        }
        int q = 0;
        for (int n = 0; n < p.N; ++n)
            for (int m = n; m < p.N; ++m, ++q) p.gram[((long long)q * p.O + o) * p.I + i] = (n == m ? 1.f : 2.f) * acc[q];
    }
}

// forward: workgroup per output channel; its Gram rows sit in LDS, a wavefront per sample (lanes along the input channels)
template <int NN>       // kernels of the bank (compile-time: the pair tables stay in registers)
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modcoef_gram_fwd_kernel(GgModGramParams p) {
    constexpr int PP = NN * (NN + 1) / 2;
    GG_SHARED float g_s[GG_MG_PMAX * GG_MC_IMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x;
    for (int row = blockIdx.x; row < p.b; row += gridDim.x) {       // rows of s / a and the zero padding of d: the workgroups in turn
        for (int i = tid; i < p.Ip; i += 256) p.s[(long long)row * p.Ip + i] = i < p.I ? p.mod[(long long)row * p.I + i] + 1.f : 0.f;
        for (int c = p.O + tid; c < p.Op; c += 256) p.d[(long long)row * p.Op + c] = 0.f;
        if (tid == 0) {
            float av[NN];
            gg_mg_softmax<NN>(p, row, av);
            _Pragma("unroll") for (int n = 0; n < NN; ++n) p.a[row * NN + n] = av[n];
        }
    }
    for (int idx = tid; idx < PP * p.I; idx += 256) {
        const int q = idx / p.I, i = idx - q * p.I;
        g_s[idx] = p.gram[((long long)q * p.O + o) * p.I + i];
    }
    gg_sync();
    for (int row = wave; row < p.b; row += 4) {
        float acc[PP];
        _Pragma("unroll") for (int q = 0; q < PP; ++q) acc[q] = 0.f;
        for (int i = lane; i < p.I; i += 64) {
            const float sv = p.mod[(long long)row * p.I + i] + 1.f;
            const float s2 = sv * sv;
            _Pragma("unroll") for (int q = 0; q < PP; ++q) acc[q] += s2 * g_s[q * p.I + i];
        }
        float av[NN];
        gg_mg_softmax<NN>(p, row, av);
        float sumsq = 0.f;
        int q = 0;
        _Pragma("unroll") for (int n = 0; n < NN; ++n)
            _Pragma("unroll") for (int m = n; m < NN; ++m, ++q) {
                const float tv = gg_mc_wave_sum(acc[q]);
                sumsq += av[n] * av[m] * tv;
                if (lane == 0) p.tsum[((long long)row * PP + q) * p.O + o] = tv;
            }
        if (lane == 0) p.d[(long long)row * p.Op + o] = gg_rsqrtf(sumsq > p.eps ? sumsq : p.eps);
    }
}

// backward K_o: workgroup per output channel: the ga slots of this channel, Q[p,o,:] and the weights' gradient
template <int NN>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modcoef_gram_bwd_o_kernel(GgModGramParams p) {
    constexpr int PP = NN * (NN + 1) / 2;
    GG_SHARED float a_s[GG_MG_BMAX][GG_MC_NMAX];
    GG_SHARED float g2a_s[GG_MG_BMAX][GG_MG_PMAX];      // 2 g[b,o] A[b,p]
    GG_SHARED float g_s[GG_MG_BMAX];
    const int tid = threadIdx.x;
    const int o = blockIdx.x;
    if (tid < p.b) {
        float av[NN];
        gg_mg_softmax<NN>(p, tid, av);
        const float dv = p.d[(long long)tid * p.Op + o];
        const float gv = dv < gg_rsqrtf(p.eps) ? -0.5f * p.gd[(long long)tid * p.Op + o] * dv * dv * dv : 0.f;
        g_s[tid] = gv;
        _Pragma("unroll") for (int n = 0; n < NN; ++n) a_s[tid][n] = av[n];
        int q = 0;
        _Pragma("unroll") for (int n = 0; n < NN; ++n)
            _Pragma("unroll") for (int m = n; m < NN; ++m, ++q) g2a_s[tid][q] = 2.f * gv * av[n] * av[m];
    }
    gg_sync();
    if (NN > 1)
        for (int idx = tid; idx < p.b * NN; idx += 256) {
            const int row = idx / NN, k = idx - row * NN;
            float acc = 0.f;
            int q = 0;
            _Pragma("unroll") for (int n = 0; n < NN; ++n)
                _Pragma("unroll") for (int m = n; m < NN; ++m, ++q) {
                    const float tv = p.tsum[((long long)row * PP + q) * p.O + o];
                    const float c = (n == k ? a_s[row][m] : 0.f) + (m == k ? a_s[row][n] : 0.f);      // dA_p / da_k
                    acc += tv * c;
                }
            p.da_slots[((long long)o * p.b + row) * NN + k] = g_s[row] * acc;
        }
    if (!p.gw) return;
    for (int i = tid; i < p.I; i += 256) {
        float qv[PP];
        _Pragma("unroll") for (int q = 0; q < PP; ++q) qv[q] = 0.f;
        for (int row = 0; row < p.b; ++row) {
            const float sv = p.s[(long long)row * p.Ip + i];
            const float s2 = sv * sv;
            _Pragma("unroll") for (int q = 0; q < PP; ++q) qv[q] += g2a_s[row][q] * s2;
        }
        // nine taps per pass: every load of the pass (weights and the old gradient) is issued before the first store (gw may alias w as
        // far as the compiler knows; one tap at a time every tap paid two dependent round trips)
        for (int t0 = 0; t0 < p.T; t0 += 9) {
            float wv[9][NN], old[9][NN];
#pragma unroll
            for (int tt = 0; tt < 9; ++tt)
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    const bool on = n < NN && t0 + tt < p.T;
                    const long long off = (((long long)n * p.O + o) * p.I + i) * p.T + t0 + tt;
                    wv[tt][n] = on ? p.w[off] : 0.f;
                    old[tt][n] = on ? p.gw[off] : 0.f;
                }
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) {
                if (t0 + tt >= p.T) break;
                _Pragma("unroll") for (int n = 0; n < NN; ++n) {
                    float acc = old[tt][n];
                    _Pragma("unroll") for (int m = 0; m < NN; ++m) {
                        const int lo = n < m ? n : m, hi = n < m ? m : n;
                        acc += wv[tt][m] * qv[lo * NN - lo * (lo - 1) / 2 + (hi - lo)];
                    }
                    p.gw[(((long long)n * p.O + o) * p.I + i) * p.T + t0 + tt] = acc;
                }
            }
        }
    }
}

// backward K_i: workgroup per (64 input channels, sample): lane = input channel, the four wavefronts split the output channels; R and
// gs of that sample. The workgroups of the first channel block also add the sample's ga slots up and finish its kernel-selection softmax
// backward. (A first version - 16 channels x all samples per workgroup, I / 16 workgroups, one dependent Gram load per output channel -
// took 250 us at O = I = 512: 32 workgroups on 256 CUs and 512 serial round trips each.)
template <int NN>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_modcoef_gram_bwd_i_kernel(GgModGramParams p) {
    constexpr int PP = NN * (NN + 1) / 2;
    GG_SHARED float g_s[GG_MC_IMAX];
    GG_SHARED float red[4][GG_MG_PMAX][64];
    GG_SHARED float dsum[4][GG_MC_NMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.y;
    const int i = blockIdx.x * 64 + lane;
    const float lim = gg_rsqrtf(p.eps);
    for (int oo = tid; oo < p.O; oo += 256) {
        const float dv = p.d[(long long)row * p.Op + oo];
        g_s[oo] = dv < lim ? -0.5f * p.gd[(long long)row * p.Op + oo] * dv * dv * dv : 0.f;
    }
    gg_sync();
    float r[PP];
    _Pragma("unroll") for (int q = 0; q < PP; ++q) r[q] = 0.f;
    const int oq = (p.O + 3) / 4;
    const int o_begin = wave * oq, o_end = (o_begin + oq < p.O) ? o_begin + oq : p.O;
    if (i < p.I) {
        const float* gcol = p.gram + i;
        int oo = o_begin;
        for (; oo + 8 <= o_end; oo += 8) {             // eight output channels per pass: their Gram loads first
            float gq[8][PP];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                _Pragma("unroll") for (int q = 0; q < PP; ++q) gq[u][q] = gcol[((long long)q * p.O + oo + u) * p.I];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float gv = g_s[oo + u];
                _Pragma("unroll") for (int q = 0; q < PP; ++q) r[q] += gv * gq[u][q];
            }
        }
        for (; oo < o_end; ++oo) {
            const float gv = g_s[oo];
            _Pragma("unroll") for (int q = 0; q < PP; ++q) r[q] += gv * gcol[((long long)q * p.O + oo) * p.I];
        }
    }
    _Pragma("unroll") for (int q = 0; q < PP; ++q) red[wave][q][lane] = r[q];
    gg_sync();
    float av[NN];
    gg_mg_softmax<NN>(p, row, av);
    if (wave == 0 && i < p.I) {
        float tot = 0.f;
        int q = 0;
        _Pragma("unroll") for (int n = 0; n < NN; ++n)
            _Pragma("unroll") for (int m = n; m < NN; ++m, ++q)
                tot += av[n] * av[m] * ((red[0][q][lane] + red[1][q][lane]) + (red[2][q][lane] + red[3][q][lane]));
        const float sv = p.s[(long long)row * p.Ip + i];
        p.gmod[(long long)row * p.I + i] = (p.gs ? p.gs[(long long)row * p.Ip + i] : 0.f) + 2.f * sv * tot;
    }
    if (blockIdx.x == 0 && p.gkmod) {
        // ga through d for this sample: the per-output-channel slots, added in a fixed order
        float part[NN];
        _Pragma("unroll") for (int n = 0; n < NN; ++n) part[n] = 0.f;
        for (int oo = tid; oo < p.O; oo += 256)
            _Pragma("unroll") for (int n = 0; n < NN; ++n) part[n] += p.da_slots[((long long)oo * p.b + row) * NN + n];
        _Pragma("unroll") for (int n = 0; n < NN; ++n) {
            const float v = gg_mc_wave_sum(part[n]);
            if (lane == 0) dsum[wave][n] = v;
        }
        gg_sync();
        if (tid == 0) {
            float gt[NN], dot = 0.f;
            _Pragma("unroll") for (int n = 0; n < NN; ++n) {
                gt[n] = (p.ga ? p.ga[row * NN + n] : 0.f) + ((dsum[0][n] + dsum[1][n]) + (dsum[2][n] + dsum[3][n]));
                dot += av[n] * gt[n];
            }
            _Pragma("unroll") for (int n = 0; n < NN; ++n) p.gkmod[row * NN + n] = av[n] * (gt[n] - dot);
        }
    }
}
