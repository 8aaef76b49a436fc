// gg_attention2.h — the SECOND-ORDER pass of the fused self-attention (gg_attention.h): what the gradient penalty's
// double backward (reference gradient_penalty gp.py:120-155, `create_graph=True` through SelfAttention gp.py:538-594)
// needs from the attention block, again without materialising any (n x n) tensor.
//
// Setting.  O_i = sum_j P_ij v_j,  P = softmax_j(x_ij),  x_ij = alpha q_i.k_j + beta |k_j|^2  (j runs over the learned
// null key/value and the n tokens).  The first backward maps a cotangent dO to (dq, dk, dv, dk0, dv0).  When that
// backward is itself differentiated, the incoming gradients (A_q, A_k, A_v, A_k0, A_v0) w.r.t. its outputs define the
// scalar  F = <A_q,dq> + <A_k,dk> + <A_v,dv> + ... = <dO, JVP_A(O)>,  and the pass must return dF/d(q, k, v, k0, v0, dO).
// With, per pair (i, j):
//     T = alpha (A_q,i.k_j + q_i.A_k,j) + 2 beta k_j.A_k,j     (tangent of the logit)
//     e = dO_i.v_j,   u = dO_i.A_v,j,   D_i = dO_i.O_i,   mu_i = sum_j P T,   g_i = sum_j P (T e + u) - 2 mu_i D_i
//     dS = P (e - D_i),   Pd = P (T - mu_i),   R = P (T e - T D_i - mu_i e + u - g_i)
// the gradients are
//     gq_i  = alpha sum_j (R k_j + dS A_k,j)                          gdO_i = sum_j (Pd v_j + P A_v,j)
//     gk_j  = alpha sum_i (R q_i + dS A_q,i) + 2 beta ((sum_i R) k_j + (sum_i dS) A_k,j)        gv_j = sum_i Pd dO_i
// (derivation in DESIGN.md §4). Three kernels, same tiling and lane mapping as the first-order kernels:
//   gg_attn_bwd2_q_kernel<true>   per query tile, loop over keys: the row statistics mu_i, g_i
//   gg_attn_bwd2_q_kernel<false>  per query tile, loop over keys: gq, gdO (+ the null token's partial sums)
//   gg_attn_bwd2_kv_kernel        per key tile,  loop over queries: gk, gv
// Each (32 x 32) block costs 5 score contractions (S, A_q.k, q.A_k, e, u) and 4 (q side) / 3 (kv side) output
// contractions; scores stay in registers as in gg_attention.h (lane = query resp. key, transposed MFMA issue, packed
// bf16 B operands with re-labelled reduction slots).
#pragma once
#include "gg_attention.h"

struct GgAttn2Params {
    const bf16_t *q, *k, *v, *k0, *v0;   // primal inputs ([B][n][h*64]; [h][64])
    const bf16_t* d_o;                   // cotangent of the forward output
    const bf16_t *aq, *ak, *av;          // incoming gradients w.r.t. dq, dk, dv ([B][n][h*64])
    const bf16_t *ak0, *av0;             // incoming gradients w.r.t. dk0, dv0 ([h][64])
    const float *lse, *dvec;             // [B*h][n]: log-sum-exp (forward) and D_i = dO_i.O_i (first backward)
    float *mu, *gi;                      // [B*h][n] row statistics (stats kernel -> the other two)
    bf16_t *gq, *gk, *gv, *gdo;          // outputs
    float* null_part;                    // [q-kernel blocks][3][64]: [0] sum_i (R0 q_i + dS0 A_q,i), [1] sum_i Pd0 dO_i,
                                         //                           [2][0] sum_i R0, [2][1] sum_i dS0
    int B, n, h;
    float alpha, beta;
    int xcd;                             // XCD-aware block order (gg_attn_block)
};

// one 32-row x 64-d tile: 256 threads, one 16-byte vector each (row t>>3, chunk t&7)
GG_DEVICE u16x8 gga2_tile_load(const bf16_t* base, long long row_stride, int t0) {
    const int t = threadIdx.x;
    return *(const u16x8*)(base + (long long)(t0 + (t >> 3)) * row_stride + (t & 7) * 8);
}
GG_DEVICE void gga2_tile_store(u16x8 x, bf16_t (*rowk)[GGA_KP], char* tr) {
    const int t = threadIdx.x;
    const int row = t >> 3, c8 = t & 7;
    if (rowk) *(u16x8*)&rowk[row][c8 * 8] = x;
    if (tr) *(u16x8*)(tr + row * GGA_TP + c8 * 16) = x;
}
// sum over the 8 lanes that hold one tile row
GG_DEVICE float gga2_row_sum(float s) {
    s += gg_shfl_xor(s, 1);
    s += gg_shfl_xor(s, 2);
    s += gg_shfl_xor(s, 4);
    return s;
}
GG_DEVICE float gga2_dot8(u16x8 a, u16x8 b) {
    float s = 0.f;
    for (int e = 0; e < 8; ++e) s += gg_bf2f(a[e]) * gg_bf2f(b[e]);
    return s;
}
// dot product of two B-operand fragment sets of one token (lane halves hold complementary d's)
GG_DEVICE float gga2_frag_dot_vec(const u16x8* f, const bf16_t* vec, int hi) {
    float s = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) s += gga2_dot8(f[kk], *(const u16x8*)(vec + kk * 16 + 8 * hi));
    return s + gg_shfl_xor(s, 32);
}

// ---- per query tile ------------------------------------------------------------------------------------------
// grid: (n / 128, B*h); wave w owns queries q0 + 32w .. +31
template <bool STATS>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_attn_bwd2_q_kernel(GgAttn2Params p) {
    GG_SHARED __attribute__((aligned(16))) bf16_t sK[32][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) bf16_t sV[32][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) bf16_t sAK[32][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) bf16_t sAV[32][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) char sKt[STATS ? 16 : 32 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) char sVt[STATS ? 16 : 32 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) char sAKt[STATS ? 16 : 32 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) char sAVt[STATS ? 16 : 32 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) float sKsq[32];
    GG_SHARED __attribute__((aligned(16))) float sKak[32];
    GG_SHARED float sRed[4][3][64];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    int bx, bh;
    gg_attn_block(p.xcd, bx, bh);
    const int b = bh / p.h, hd = bh % p.h;
    const long long rs = (long long)p.h * GGA_D;
    const long long boff = (long long)b * p.n * rs + hd * GGA_D;
    const int qi0 = bx * 128 + wave * 32;
    const int qi = qi0 + (lane & 31);

    u16x8 qf[4], aqf[4], dof[4];
    gga_load_frags(qf, p.q + boff, rs, qi0, lane);
    gga_load_frags(aqf, p.aq + boff, rs, qi0, lane);
    gga_load_frags(dof, p.d_o + boff, rs, qi0, lane);
    const float lse = p.lse[(long long)bh * p.n + qi];
    const float Di = p.dvec[(long long)bh * p.n + qi];
    float mu = 0.f, gi = 0.f;
    if (!STATS) {
        mu = p.mu[(long long)bh * p.n + qi];
        gi = p.gi[(long long)bh * p.n + qi];
    }
    float mu_acc = 0.f, w_acc = 0.f;
    f32x16 gqt[2], gdot[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gqt[db][r] = 0.f; gdot[db][r] = 0.f; }

    // ---- the null token (j = 0), lane-local --------------------------------------------------------------------
    {
        const bf16_t* k0 = p.k0 + hd * GGA_D;
        const bf16_t* v0 = p.v0 + hd * GGA_D;
        const bf16_t* ak0 = p.ak0 + hd * GGA_D;
        const bf16_t* av0 = p.av0 + hd * GGA_D;
        const float qk = gga2_frag_dot_vec(qf, k0, hi), aqk = gga2_frag_dot_vec(aqf, k0, hi);
        const float qak = gga2_frag_dot_vec(qf, ak0, hi);
        const float e0 = gga2_frag_dot_vec(dof, v0, hi), u0 = gga2_frag_dot_vec(dof, av0, hi);
        float sq = 0.f, kak = 0.f;
        for (int d = 0; d < GGA_D; ++d) {
            const float kf = gg_bf2f(k0[d]);
            sq += kf * kf;
            kak += kf * gg_bf2f(ak0[d]);
        }
        const float P0 = gg_expf(p.alpha * qk + p.beta * sq - lse);
        const float T0 = p.alpha * (aqk + qak) + 2.f * p.beta * kak;
        if (STATS) {
            if (hi == 0) {      // both half-wave lanes of a query hold the same scalars: count the null token once
                mu_acc = P0 * T0;
                w_acc = P0 * (T0 * e0 + u0);
            }
        } else {
            const float dS0 = P0 * (e0 - Di);
            const float Pd0 = P0 * (T0 - mu);
            const float R0 = P0 * (T0 * e0 - T0 * Di - mu * e0 + u0 - gi);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    gqt[db][r] = R0 * gg_bf2f(k0[d]) + dS0 * gg_bf2f(ak0[d]);
                    gdot[db][r] = Pd0 * gg_bf2f(v0[d]) + P0 * gg_bf2f(av0[d]);
                }
            // sums over this wave's 32 queries for the null token's own gradients
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                for (int e = 0; e < 8; ++e) {
                    float a = R0 * gg_bf2f(qf[kk][e]) + dS0 * gg_bf2f(aqf[kk][e]);
                    float c = Pd0 * gg_bf2f(dof[kk][e]);
                    for (int o = 1; o < 32; o <<= 1) { a += gg_shfl_xor(a, o); c += gg_shfl_xor(c, o); }
                    if ((lane & 31) == 0) {
                        sRed[wave][0][kk * 16 + 8 * hi + e] = a;
                        sRed[wave][1][kk * 16 + 8 * hi + e] = c;
                    }
                }
            float sR = R0, sS = dS0;
            for (int o = 1; o < 32; o <<= 1) { sR += gg_shfl_xor(sR, o); sS += gg_shfl_xor(sS, o); }
            if (lane == 0) { sRed[wave][2][0] = sR; sRed[wave][2][1] = sS; }
        }
    }

    // ---- the n tokens, 32 keys per step ----------------------------------------------------------------------
    u16x8 rk = gga2_tile_load(p.k + boff, rs, 0), rv = gga2_tile_load(p.v + boff, rs, 0);
    u16x8 rak = gga2_tile_load(p.ak + boff, rs, 0), rav = gga2_tile_load(p.av + boff, rs, 0);
    for (int j0 = 0; j0 < p.n; j0 += 32) {
        gg_sync();
        gga2_tile_store(rk, sK, STATS ? nullptr : sKt);
        gga2_tile_store(rv, sV, STATS ? nullptr : sVt);
        gga2_tile_store(rak, sAK, STATS ? nullptr : sAKt);
        gga2_tile_store(rav, sAV, STATS ? nullptr : sAVt);
        {
            const float s1 = gga2_row_sum(gga2_dot8(rk, rk)), s2 = gga2_row_sum(gga2_dot8(rk, rak));
            if ((threadIdx.x & 7) == 0) { sKsq[threadIdx.x >> 3] = s1; sKak[threadIdx.x >> 3] = s2; }
        }
        gg_sync();
        if (j0 + 32 < p.n) {
            rk = gga2_tile_load(p.k + boff, rs, j0 + 32);
            rv = gga2_tile_load(p.v + boff, rs, j0 + 32);
            rak = gga2_tile_load(p.ak + boff, rs, j0 + 32);
            rav = gga2_tile_load(p.av + boff, rs, j0 + 32);
        }
        f32x16 S, T1, T2, E, U;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; T1[r] = 0.f; T2[r] = 0.f; E[r] = 0.f; U[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u16x8 fk = gga_frag_rowk(sK, 0, kk, lane);
            S = gg_mfma_32x32x16_bf16(fk, qf[kk], S);
            T1 = gg_mfma_32x32x16_bf16(fk, aqf[kk], T1);
            T2 = gg_mfma_32x32x16_bf16(gga_frag_rowk(sAK, 0, kk, lane), qf[kk], T2);
            E = gg_mfma_32x32x16_bf16(gga_frag_rowk(sV, 0, kk, lane), dof[kk], E);
            U = gg_mfma_32x32x16_bf16(gga_frag_rowk(sAV, 0, kk, lane), dof[kk], U);
        }
        f32x16 Rr, dSr, Pdr;     // R, dS, Pd (and P kept in S)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ks = gga_rows4(sKsq, 0, g, lane), ka = gga_rows4(sKak, 0, g, lane);
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const float P = gg_expf(p.alpha * S[r] + p.beta * ks[e] - lse);
                const float T = p.alpha * (T1[r] + T2[r]) + 2.f * p.beta * ka[e];
                if (STATS) {
                    mu_acc += P * T;
                    w_acc += P * (T * E[r] + U[r]);
                } else {
                    S[r] = P;
                    dSr[r] = P * (E[r] - Di);
                    Pdr[r] = P * (T - mu);
                    Rr[r] = P * (T * E[r] - T * Di - mu * E[r] + U[r] - gi);
                }
            }
        }
        if (!STATS) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u16x8 fR = gga_pack8(Rr, c), fdS = gga_pack8(dSr, c), fPd = gga_pack8(Pdr, c), fP = gga_pack8(S, c);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    gqt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sKt, db * 32, 16 * c, lane), fR, gqt[db]);
                    gqt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sAKt, db * 32, 16 * c, lane), fdS, gqt[db]);
                    gdot[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sVt, db * 32, 16 * c, lane), fPd, gdot[db]);
                    gdot[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sAVt, db * 32, 16 * c, lane), fP, gdot[db]);
                }
            }
        }
    }

    if (STATS) {
        mu_acc += gg_shfl_xor(mu_acc, 32);
        w_acc += gg_shfl_xor(w_acc, 32);
        if (hi == 0) {
            p.mu[(long long)bh * p.n + qi] = mu_acc;
            p.gi[(long long)bh * p.n + qi] = w_acc - 2.f * mu_acc * Di;
        }
        return;
    }

    bf16_t* gqrow = p.gq + boff + (long long)qi * rs;
    bf16_t* gdorow = p.gdo + boff + (long long)qi * rs;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 a4, b4;
            for (int e = 0; e < 4; ++e) {
                a4[e] = gg_f2bf(gqt[db][4 * g + e] * p.alpha);
                b4[e] = gg_f2bf(gdot[db][4 * g + e]);
            }
            *(u16x4*)(gqrow + db * 32 + 8 * g + 4 * hi) = a4;
            *(u16x4*)(gdorow + db * 32 + 8 * g + 4 * hi) = b4;
        }
    gg_sync();
    if (threadIdx.x < 192) {
        const int which = threadIdx.x >> 6, d = threadIdx.x & 63;
        float s = 0.f;
        if (which < 2 || d < 2)
            for (int w = 0; w < 4; ++w) s += sRed[w][which][d];
        p.null_part[(((long long)bh * gridDim.x + bx) * 3 + which) * 64 + d] = s;
    }
}

// ---- per key tile -----------------------------------------------------------------------------------------------
// grid: (n / 128, B*h); wave w owns keys j0 + 32w .. +31 and loops over all queries, 32 per step
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_attn_bwd2_kv_kernel(GgAttn2Params p) {
    GG_SHARED __attribute__((aligned(16))) bf16_t sQ[32][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) bf16_t sAQ[32][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) bf16_t sDO[32][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) char sQt[32 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) char sAQt[32 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) char sDOt[32 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) float sStat[4][32];   // lse, D, mu, g of the staged queries

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    int bx, bh;
    gg_attn_block(p.xcd, bx, bh);
    const int b = bh / p.h, hd = bh % p.h;
    const long long rs = (long long)p.h * GGA_D;
    const long long boff = (long long)b * p.n * rs + hd * GGA_D;
    const int kj0 = bx * 128 + wave * 32;
    const int kj = kj0 + (lane & 31);

    u16x8 kf[4], vf[4], akf[4], avf[4];
    gga_load_frags(kf, p.k + boff, rs, kj0, lane);
    gga_load_frags(vf, p.v + boff, rs, kj0, lane);
    gga_load_frags(akf, p.ak + boff, rs, kj0, lane);
    gga_load_frags(avf, p.av + boff, rs, kj0, lane);
    float ksq = 0.f, kak = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { ksq += gga2_dot8(kf[kk], kf[kk]); kak += gga2_dot8(kf[kk], akf[kk]); }
    ksq += gg_shfl_xor(ksq, 32);
    kak += gg_shfl_xor(kak, 32);
    const float xb = p.beta * ksq, tb = 2.f * p.beta * kak;

    f32x16 gkt[2], gvt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gkt[db][r] = 0.f; gvt[db][r] = 0.f; }
    float cR = 0.f, cdS = 0.f;

    const float* stat_src[4] = {p.lse, p.dvec, p.mu, p.gi};
    u16x8 rq = gga2_tile_load(p.q + boff, rs, 0), raq = gga2_tile_load(p.aq + boff, rs, 0);
    u16x8 rdo = gga2_tile_load(p.d_o + boff, rs, 0);
    float rstat = 0.f;
    if (threadIdx.x < 128) rstat = stat_src[threadIdx.x >> 5][(long long)bh * p.n + (threadIdx.x & 31)];
    for (int i0 = 0; i0 < p.n; i0 += 32) {
        gg_sync();
        gga2_tile_store(rq, sQ, sQt);
        gga2_tile_store(raq, sAQ, sAQt);
        gga2_tile_store(rdo, sDO, sDOt);
        if (threadIdx.x < 128) sStat[threadIdx.x >> 5][threadIdx.x & 31] = rstat;
        gg_sync();
        if (i0 + 32 < p.n) {
            rq = gga2_tile_load(p.q + boff, rs, i0 + 32);
            raq = gga2_tile_load(p.aq + boff, rs, i0 + 32);
            rdo = gga2_tile_load(p.d_o + boff, rs, i0 + 32);
            if (threadIdx.x < 128) rstat = stat_src[threadIdx.x >> 5][(long long)bh * p.n + i0 + 32 + (threadIdx.x & 31)];
        }
        f32x16 S, T1, T2, E, U;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; T1[r] = 0.f; T2[r] = 0.f; E[r] = 0.f; U[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u16x8 fq = gga_frag_rowk(sQ, 0, kk, lane), fdo = gga_frag_rowk(sDO, 0, kk, lane);
            S = gg_mfma_32x32x16_bf16(fq, kf[kk], S);
            T1 = gg_mfma_32x32x16_bf16(gga_frag_rowk(sAQ, 0, kk, lane), kf[kk], T1);
            T2 = gg_mfma_32x32x16_bf16(fq, akf[kk], T2);
            E = gg_mfma_32x32x16_bf16(fdo, vf[kk], E);
            U = gg_mfma_32x32x16_bf16(fdo, avf[kk], U);
        }
        f32x16 Rr, dSr, Pdr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ls = gga_rows4(sStat[0], 0, g, lane), dd = gga_rows4(sStat[1], 0, g, lane);
            const f32x4 mm = gga_rows4(sStat[2], 0, g, lane), gg_ = gga_rows4(sStat[3], 0, g, lane);
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const float P = gg_expf(p.alpha * S[r] + xb - ls[e]);
                const float T = p.alpha * (T1[r] + T2[r]) + tb;
                const float ds = P * (E[r] - dd[e]);
                const float rr = P * (T * E[r] - T * dd[e] - mm[e] * E[r] + U[r] - gg_[e]);
                dSr[r] = ds;
                Pdr[r] = P * (T - mm[e]);
                Rr[r] = rr;
                cR += rr;
                cdS += ds;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u16x8 fR = gga_pack8(Rr, c), fdS = gga_pack8(dSr, c), fPd = gga_pack8(Pdr, c);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                gkt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sQt, db * 32, 16 * c, lane), fR, gkt[db]);
                gkt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sAQt, db * 32, 16 * c, lane), fdS, gkt[db]);
                gvt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sDOt, db * 32, 16 * c, lane), fPd, gvt[db]);
            }
        }
    }
    cR += gg_shfl_xor(cR, 32);
    cdS += gg_shfl_xor(cdS, 32);

    const bf16_t* krow = p.k + boff + (long long)kj * rs;
    const bf16_t* akrow = p.ak + boff + (long long)kj * rs;
    bf16_t* gkrow = p.gk + boff + (long long)kj * rs;
    bf16_t* gvrow = p.gv + boff + (long long)kj * rs;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + 8 * g + 4 * hi;
            const u16x4 k4 = *(const u16x4*)(krow + d), ak4 = *(const u16x4*)(akrow + d);
            u16x4 a4, b4;
            for (int e = 0; e < 4; ++e) {
                a4[e] = gg_f2bf(p.alpha * gkt[db][4 * g + e] + 2.f * p.beta * (cR * gg_bf2f(k4[e]) + cdS * gg_bf2f(ak4[e])));
                b4[e] = gg_f2bf(gvt[db][4 * g + e]);
            }
            *(u16x4*)(gkrow + d) = a4;
            *(u16x4*)(gvrow + d) = b4;
        }
}
