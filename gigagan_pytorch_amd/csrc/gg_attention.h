// gg_attention.h — fused (flash-style) self-attention of the GigaGAN generator / discriminator blocks
// (reference SelfAttention.forward gp.py:538-594): heads of dimension 64 over n = H*W image tokens plus ONE learned
// null key/value (gp.py:534, :568), dot-product similarity (generator, gp.py:574) or negative squared L2 distance
// (discriminator, gp.py:577-580), softmax, A·V — without materialising the (b*h, n, n+1) similarity tensor
// (4.3 GB in fp32 at the discriminator's 32x32 stage, batch 32).
//
// Tensors: q, k, v, o, do, dq, dk, dv are [B][n][h*64] bf16 — exactly what the 1x1 projections read and write, so no
// head split / merge copies exist. lse, dvec are [B*h][n] fp32. k0, v0: the null key / value, [h][64] bf16.
//
// logits:  x_ij = alpha * q_i.k_j + beta * |k_j|^2      (dot: alpha = scale, beta = 0;
//                                                         L2:  alpha = 2*scale, beta = -scale; the -scale*|q_i|^2 term is
//                                                         constant along j and cancels in the softmax)
// The null key is folded in as the initial state of the online softmax (m = x_i0, l = 1, O = v0), which keeps the
// key loop free of ragged tiles.
//
// Mapping (64-lane wavefronts, v_mfma_f32_32x32x16_bf16): every contraction is issued "transposed", e.g.
// S^T = K Q^T, so that the MFMA result leaves lane l with 16 values of ONE query (or key) — the softmax row
// statistics are lane-local plus one exchange with lane l^32 — and the same registers are, after bf16 packing, the
// k-contiguous B operand of the next MFMA (P^T for O^T = V^T P^T): the reduction index is simply re-labelled
// (slot (c, hi, e) <-> row 16c + 4hi + (e&3) + 8(e>>2) of the 32-row block), which only changes which LDS rows the
// other operand's transpose reads (ds_read_b64_tr_b16) start from. No cross-lane movement of P.
//
// Work per launch: forward 4*n*(n+1)*64 flops per head; backward 2.5x that, recomputing P in both backward kernels
// (dq: 3 contractions, dk/dv: 4). HBM traffic: q, k, v, o (+ gradients) once per head.
#pragma once
#include "gg_device.h"

#define GGA_D 64
#define GGA_KP 72                  // bf16 pitch of row-major [row][d] tiles (144 B: conflict-free ds_read_b128)
#define GGA_TP 192                 // byte pitch of transpose-read tiles [row][d] (4 consecutive rows -> 4 bank quarters)

struct GgAttnParams {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    const bf16_t* k0;   // [h][64]
    const bf16_t* v0;   // [h][64]
    bf16_t* o;          // fwd out / bwd in
    float* lse;         // [B*h][n]  fwd out / bwd in
    const bf16_t* d_o;  // bwd in
    float* dvec;        // [B*h][n]  written by the dq kernel (rowsum(dO * O)), read by the dk/dv kernel
    bf16_t* dq;
    bf16_t* dk;
    bf16_t* dv;
    float* null_part;   // [blocks of the dq kernel][3][64]: partial sums of dk0 (q part), dv0, and [2][0] = dbias0
    int B, n, h;
    float alpha, beta;
    int xcd;            // 1: XCD-aware block order (gg_attn_block); 0: dispatch order (GG_ATTN_XCD=0, A/B runs)
    // general form (GEN instantiations: cross attention gp.py:617-655, the text transformer's attention gp.py:659-722, the unet's
    // Attend attend.py:64-110): m keys / values per (batch, head) that need not equal the n queries, any n and m (tails are
    // clamped on load and masked), strided q / k / v rows (channel slices of a fused projection), an optional additive per-key
    // bias [B][m] (natural-log domain; key-padding masks are -1e30), and the null key / value optional (has_null)
    int m;
    long long ldq, ldk, ldv;    // row pitch (elements) of q, k, v; o / dO / dq / dk / dv are dense [B][len][h*64]
    const float* kbias;
    int has_null;
};

// ---- fragment helpers ----------------------------------------------------------------------------------------

// B-operand fragments straight from global memory: lane l -> token row0 + (l & 31), d = kk*16 + 8*(l >> 5) + 0..7
GG_DEVICE void gga_load_frags(u16x8* f, const bf16_t* base, long long row_stride, int row0, int lane) {
    const bf16_t* p = base + (long long)(row0 + (lane & 31)) * row_stride + 8 * (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = *(const u16x8*)(p + kk * 16);
}

// the same with the token row clamped to nrows - 1 (ragged tails of the general form)
GG_DEVICE void gga_load_frags_c(u16x8* f, const bf16_t* base, long long row_stride, int row0, int nrows, int lane) {
    const int r = row0 + (lane & 31);
    const bf16_t* p = base + (long long)(r < nrows ? r : nrows - 1) * row_stride + 8 * (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = *(const u16x8*)(p + kk * 16);
}

GG_DEVICE u16x8 gga_frag_rowk(const bf16_t (*tile)[GGA_KP], int row0, int kk, int lane) {
    return *(const u16x8*)&tile[row0 + (lane & 31)][kk * 16 + 8 * (lane >> 5)];
}

// A-operand fragment of the TRANSPOSE of a [row][d] tile: lane l -> d = d0 + (l & 31), reduction slots (hi = l >> 5, e)
// <-> rows rb + 4*hi + (e & 3) + 8*(e >> 2)   (rb = first row of the 16-row group (block, c))
GG_DEVICE u16x8 gga_frag_tr(const char* tile, int d0, int rb, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const char* p = tile + (rb + 4 * (g >> 1) + (i >> 2)) * GGA_TP + (d0 + (g & 1) * 16 + 4 * (i & 3)) * 2;
    u16x4 a = gg_lds_read_tr16((const bf16_t*)p);
    u16x4 b = gg_lds_read_tr16((const bf16_t*)(p + 8 * GGA_TP));
    u16x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
}

// registers 8c .. 8c+7 of an MFMA result, packed to bf16: the lane-local B operand of the follow-up contraction
GG_DEVICE u16x8 gga_pack8(const f32x16& v, int c) {
    u16x8 f;
    for (int e = 0; e < 8; ++e) f[e] = gg_f2bf(v[8 * c + e]);
    return f;
}

// A 64-token x 64-d tile (tokens t0 .. t0+63 of one (batch, head)) is staged in two halves: global -> registers
// (issued one tile ahead, so the HBM/L2 latency overlaps the MFMAs of the current tile) and registers -> LDS in
// row-major and/or transpose-read form (256 threads: 8 threads per token, 16 bytes each, two passes); optionally the
// squared norms of the rows.
struct GgaTileRegs {
    u16x8 v[2];
};

GG_DEVICE GgaTileRegs gga_tile_load(const bf16_t* base, long long row_stride, int t0) {
    GgaTileRegs r;
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = t + 256 * it;
        r.v[it] = *(const u16x8*)(base + (long long)(t0 + (v >> 3)) * row_stride + (v & 7) * 8);
    }
    return r;
}

GG_DEVICE GgaTileRegs gga_tile_load_c(const bf16_t* base, long long row_stride, int t0, int nrows) {
    GgaTileRegs r;
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = t + 256 * it;
        const int row = t0 + (v >> 3);
        r.v[it] = *(const u16x8*)(base + (long long)(row < nrows ? row : nrows - 1) * row_stride + (v & 7) * 8);
    }
    return r;
}

// general form: the per-key bias of a staged tile (log2 domain) on top of the |k|^2 term gga_tile_store left in kb[]; rows beyond the
// last key are switched off. Written by the threads that wrote kb[] (c8 == 0), so no barrier is needed in between.
GG_DEVICE void gga_tile_bias(float* kb, const float* kbias_row, int t0, int nkeys) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = t + 256 * it;
        const int row = v >> 3;
        if ((v & 7) == 0) {
            const int j = t0 + row;
            float x = kb[row];
            if (kbias_row) x += kbias_row[j < nkeys ? j : nkeys - 1] * 1.4426950408889634f;
            kb[row] = j < nkeys ? x : -1.0e30f;
        }
    }
}

GG_DEVICE void gga_tile_store(const GgaTileRegs& r, bf16_t (*rowk)[GGA_KP], char* tr, float* sq, float sq_scale = 1.f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = t + 256 * it;
        const int row = v >> 3, c8 = v & 7;
        const u16x8 x = r.v[it];
        if (rowk) *(u16x8*)&rowk[row][c8 * 8] = x;
        if (tr) *(u16x8*)(tr + row * GGA_TP + c8 * 16) = x;
        if (sq) {
            float s = 0.f;
            for (int e = 0; e < 8; ++e) { float f = gg_bf2f(x[e]); s += f * f; }
            s += gg_shfl_xor(s, 1);
            s += gg_shfl_xor(s, 2);
            s += gg_shfl_xor(s, 4);
            if (c8 == 0) sq[row] = s * sq_scale;
        }
    }
}

// the value of per-row array `arr` (LDS) at the 4 consecutive rows owned by register quad g of an MFMA result
GG_DEVICE f32x4 gga_rows4(const float* arr, int blk, int g, int lane) {
    return *(const f32x4*)(arr + blk * 32 + 8 * g + 4 * (lane >> 5));
}

// ---- forward ---------------------------------------------------------------------------------------------------
// grid: (n / 128, B*h); 256 threads; wave w owns queries q0 + 32w .. +31.
//
// The kernel is bound by the vector ALU (the softmax), not by the matrix pipe: per 64-key tile a wave issues 16 MFMAs (512
// cycles) but the exponentials of its 32 x 64 scores cost more than that. So the per-score work is kept to
//     x = fma(s, alpha', bias'_j)   max3   x - m   v_exp_f32   (+ row sum, bf16 pack)
// by (1) working in the log2 domain (alpha' = alpha log2 e, bias'_j = beta log2 e |k_j|^2, written to LDS with the tile),
// (2) rescaling the running output only when some row's maximum grew by more than 2^GGA_TAU since the last rescale (a
// wave-uniform branch; until then the stale maximum is the reference of BOTH the probabilities and the row sums, so the
// final O / l is unchanged; probabilities stay below 2^GGA_TAU), which removes the 32 accumulator read-modify-writes per
// tile, and (3) keeping the row sums per lane (the two half-wave partners own different keys of the same query) until the end.
// K / V tiles are double buffered: one barrier per tile.
#define GGA_LOG2E 1.4426950408889634f
#define GGA_LN2 0.6931471805599453f
#define GGA_TAU 8.0f

// XCD-aware block order of the (key/query block, batch x head) grids. The dispatcher places consecutive workgroups (x fastest) on
// consecutive XCDs, so the n / 128 blocks of ONE (batch, head) - which all stream that head's whole K and V (or Q and dO) - land on
// eight different XCDs and every L2 fetches the same tiles: rocprofv3 FETCH_SIZE x 2 = 38 / 49 / 51 GB per four steps for forward / dq /
// dk-dv against ~9 GB of q, k, v, o (profiles/r05_pmc_step.log) - the kernels ran at 3.8-5.5 TB/s of HBM-side traffic. Remapped: XCD x
// works through a contiguous range of (batch x head, block) items, so the blocks of a head share one L2.
GG_DEVICE void gg_attn_block(int xcd_order, int& bx, int& bh) {
    if (!xcd_order) { bx = blockIdx.x; bh = blockIdx.y; return; }
    const int gx = gridDim.x, total = gx * gridDim.y;
    const int lin = blockIdx.y * gx + blockIdx.x;
    const int xq = total >> 3, xr = total & 7, xcd = lin & 7, pos = lin >> 3;
    const int item = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    bh = item / gx;
    bx = item - bh * gx;
}

template <bool GEN>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_attn_fwd_kernel(GgAttnParams p) {
    GG_SHARED __attribute__((aligned(16))) bf16_t sK[2][64][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) char sV[2][64 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) float sKb[2][64];          // beta' |k_j|^2 of the tile's keys

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    int bx, bh;
    gg_attn_block(p.xcd, bx, bh);
    const int b = bh / p.h, hd = bh % p.h;
    const long long rs = (long long)p.h * GGA_D;
    const long long rsq = GEN ? p.ldq : rs, rsk = GEN ? p.ldk : rs, rsv = GEN ? p.ldv : rs;
    const int nk = GEN ? p.m : p.n;                                   // keys per (batch, head)
    const bf16_t* qb = p.q + (long long)b * p.n * rsq + hd * GGA_D;
    const bf16_t* kb = p.k + (long long)b * nk * rsk + hd * GGA_D;
    const bf16_t* vb = p.v + (long long)b * nk * rsv + hd * GGA_D;
    const float* kbias = (GEN && p.kbias) ? p.kbias + (long long)b * nk : nullptr;
    const int qi0 = bx * 128 + wave * 32;
    const float a2 = p.alpha * GGA_LOG2E, b2 = p.beta * GGA_LOG2E, inv_a2 = 1.f / a2;

    u16x8 qf[4];
    if (GEN) gga_load_frags_c(qf, qb, rsq, qi0, p.n, lane);
    else gga_load_frags(qf, qb, rs, qi0, lane);

    // null key / value: initial state of the online softmax (its probability 2^0 = 1 is counted by the hi = 0 lane)
    float m, l = hi ? 0.f : 1.f;
    f32x16 ot[2];
    if (GEN && !p.has_null) {
        // no null token: the reference maximum starts at the (unbiased) score of key 0 - any finite value of the scores' magnitude
        // will do, the first tiles move it - with nothing counted yet (l = 0, O = 0). (A -inf start would be folded into the score
        // MFMA's accumulate input below and absorb the scores.)
        float dot = 0.f, sq = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u16x8 kv = *(const u16x8*)(kb + kk * 16 + 8 * hi);
            for (int e = 0; e < 8; ++e) {
                float kf = gg_bf2f(kv[e]);
                dot += gg_bf2f(qf[kk][e]) * kf;
                sq += kf * kf;
            }
        }
        dot += gg_shfl_xor(dot, 32);
        sq += gg_shfl_xor(sq, 32);
        m = a2 * dot + b2 * sq;
        l = 0.f;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[db][r] = 0.f;
    } else {
        const bf16_t* k0 = p.k0 + hd * GGA_D;
        float dot = 0.f, sq = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u16x8 kv = *(const u16x8*)(k0 + kk * 16 + 8 * hi);
            for (int e = 0; e < 8; ++e) {
                float kf = gg_bf2f(kv[e]);
                dot += gg_bf2f(qf[kk][e]) * kf;
                sq += kf * kf;
            }
        }
        dot += gg_shfl_xor(dot, 32);
        sq += gg_shfl_xor(sq, 32);
        m = a2 * dot + b2 * sq;
        const bf16_t* v0 = p.v0 + hd * GGA_D;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[db][r] = gg_bf2f(v0[db * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]);
    }

    auto tload = [&](const bf16_t* base, long long ld, int t0) {
        return GEN ? gga_tile_load_c(base, ld, t0, nk) : gga_tile_load(base, ld, t0);
    };
    GgaTileRegs rk = tload(kb, rsk, 0), rv = tload(vb, rsv, 0);
    gga_tile_store(rk, sK[0], nullptr, sKb[0], b2);
    if (GEN) gga_tile_bias(sKb[0], kbias, 0, nk);
    gga_tile_store(rv, nullptr, sV[0], nullptr);
    if (64 < nk) {
        rk = tload(kb, rsk, 64);
        rv = tload(vb, rsv, 64);
    }
    gg_sync();
    for (int j0 = 0, buf = 0; j0 < nk; j0 += 64, buf ^= 1) {
        // tile j0 + 64 sits in the staging registers: park it in the other buffer (free since the barrier that ended the
        // previous iteration) and put tile j0 + 128 in flight
        if (j0 + 64 < nk) {
            gga_tile_store(rk, sK[buf ^ 1], nullptr, sKb[buf ^ 1], b2);
            if (GEN) gga_tile_bias(sKb[buf ^ 1], kbias, j0 + 64, nk);
            gga_tile_store(rv, nullptr, sV[buf ^ 1], nullptr);
            if (j0 + 128 < nk) {
                rk = tload(kb, rsk, j0 + 128);
                rv = tload(vb, rsv, j0 + 128);
            }
        }

        // the accumulators start at -m / alpha', so that alpha' * S + bias' comes out of the fma already relative to the reference
        // maximum m (the subtraction rides on the MFMA's accumulate input)
        const float minit = -m * inv_a2;
        f32x16 st[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[jb][r] = minit;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) st[jb] = gg_mfma_32x32x16_bf16(gga_frag_rowk(sK[buf], jb * 32, kk, lane), qf[kk], st[jb]);
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kb4 = gga_rows4(sKb[buf], jb, g, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = a2 * st[jb][4 * g + e] + kb4[e];        // log2 score - m
                    st[jb][4 * g + e] = x;
                    mx = fmaxf(mx, x);
                }
            }
        mx = fmaxf(mx, gg_shfl_xor(mx, 32));
        if (gg_wave_any(mx > GGA_TAU)) {        // rare after the first tiles: move the reference to the new maxima
            const float up = fmaxf(mx, 0.f);
            const float corr = gg_exp2f(-up);
            l *= corr;
            m += up;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[db][r] *= corr;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[jb][r] -= up;
        }
        f32x2 rowsum = {0.f, 0.f};          // pairs: v_pk_add_f32
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 pv = {gg_exp2f(st[jb][r]), gg_exp2f(st[jb][r + 1])};
                st[jb][r] = pv[0];
                st[jb][r + 1] = pv[1];
                rowsum += pv;
            }
        l += rowsum[0] + rowsum[1];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u16x8 pf = gga_pack8(st[jb], c);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    ot[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sV[buf], db * 32, jb * 32 + 16 * c, lane), pf, ot[db]);
            }
        gg_sync();
    }

    l += gg_shfl_xor(l, 32);
    if (GEN) l = fmaxf(l, 1.0e-30f);                     // (every key masked: zeros instead of NaN)
    const float inv = 1.f / l;
    const int qi = qi0 + (lane & 31);
    if (GEN && qi >= p.n) return;
    bf16_t* orow = p.o + ((long long)b * p.n + qi) * rs + hd * GGA_D;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 o4 = {gg_f2bf(ot[db][4 * g + 0] * inv), gg_f2bf(ot[db][4 * g + 1] * inv), gg_f2bf(ot[db][4 * g + 2] * inv),
                        gg_f2bf(ot[db][4 * g + 3] * inv)};
            *(u16x4*)(orow + db * 32 + 8 * g + 4 * hi) = o4;
        }
    if (hi == 0) p.lse[(long long)bh * p.n + qi] = m * GGA_LN2 + logf(l);
}

// ---- backward, part 1: dq (+ rowsum(dO*O), + the null key/value partial gradients) ---------------------------------
// grid: (n / 128, B*h); wave w owns queries q0 + 32w .. +31
template <bool GEN>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_attn_bwd_dq_kernel(GgAttnParams p) {
    GG_SHARED __attribute__((aligned(16))) bf16_t sK[2][64][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) bf16_t sV[2][64][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) char sKt[2][64 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) float sKb[2][64];          // beta' |k_j|^2 (log2 domain)
    GG_SHARED float sRed[4][3][64];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    int bx, bh;
    gg_attn_block(p.xcd, bx, bh);
    const int b = bh / p.h, hd = bh % p.h;
    const long long rs = (long long)p.h * GGA_D;
    const long long boff = (long long)b * p.n * rs + hd * GGA_D;          // dense query-side tensors: o, dO, dq
    const long long rsq = GEN ? p.ldq : rs, rsk = GEN ? p.ldk : rs, rsv = GEN ? p.ldv : rs;
    const int nk = GEN ? p.m : p.n;
    const bf16_t* qb = p.q + (long long)b * p.n * rsq + hd * GGA_D;
    const bf16_t* kb = p.k + (long long)b * nk * rsk + hd * GGA_D;
    const bf16_t* vb = p.v + (long long)b * nk * rsv + hd * GGA_D;
    const float* kbias = (GEN && p.kbias) ? p.kbias + (long long)b * nk : nullptr;
    const int qi0 = bx * 128 + wave * 32;
    const int qi = qi0 + (lane & 31);
    const int qic = (GEN && qi >= p.n) ? p.n - 1 : qi;
    const float a2 = p.alpha * GGA_LOG2E, b2 = p.beta * GGA_LOG2E, inv_a2 = 1.f / a2;

    u16x8 qf[4], dof[4];
    if (GEN) {
        gga_load_frags_c(qf, qb, rsq, qi0, p.n, lane);
        gga_load_frags_c(dof, p.d_o + boff, rs, qi0, p.n, lane);
    } else {
        gga_load_frags(qf, p.q + boff, rs, qi0, lane);
        gga_load_frags(dof, p.d_o + boff, rs, qi0, lane);
    }
    float dsum = 0.f;   // D_i = sum_d dO_i[d] * O_i[d]
    {
        u16x8 of[4];
        if (GEN) gga_load_frags_c(of, p.o + boff, rs, qi0, p.n, lane);
        else gga_load_frags(of, p.o + boff, rs, qi0, lane);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            for (int e = 0; e < 8; ++e) dsum += gg_bf2f(dof[kk][e]) * gg_bf2f(of[kk][e]);
        dsum += gg_shfl_xor(dsum, 32);
    }
    const float lse = p.lse[(long long)bh * p.n + qic];
    if (hi == 0 && qi == qic) p.dvec[(long long)bh * p.n + qi] = dsum;

    f32x16 dqt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[db][r] = 0.f;

    // null key: ds0 = p0 * (dO_i . v0 - D_i); dq_i += ds0 * k0 (alpha applied at the end); partial sums for dk0 / dv0
    if (!GEN || p.has_null) {
        const bf16_t* k0 = p.k0 + hd * GGA_D;
        const bf16_t* v0 = p.v0 + hd * GGA_D;
        float dot = 0.f, sq = 0.f, dp0 = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u16x8 kv = *(const u16x8*)(k0 + kk * 16 + 8 * hi);
            u16x8 vv = *(const u16x8*)(v0 + kk * 16 + 8 * hi);
            for (int e = 0; e < 8; ++e) {
                float kf = gg_bf2f(kv[e]);
                dot += gg_bf2f(qf[kk][e]) * kf;
                sq += kf * kf;
                dp0 += gg_bf2f(dof[kk][e]) * gg_bf2f(vv[e]);
            }
        }
        dot += gg_shfl_xor(dot, 32);
        sq += gg_shfl_xor(sq, 32);
        dp0 += gg_shfl_xor(dp0, 32);
        const float p0 = (GEN && qi != qic) ? 0.f : gg_expf(p.alpha * dot + p.beta * sq - lse);      // (rows past the last query: nothing)
        const float ds0 = p0 * (dp0 - dsum);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqt[db][r] = ds0 * gg_bf2f(k0[db * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]);
        // sums over this wave's 32 queries: dk0[d] += ds0 * q_i[d], dv0[d] += p0 * dO_i[d], dbias0 += ds0
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            for (int e = 0; e < 8; ++e) {
                float a = ds0 * gg_bf2f(qf[kk][e]);
                float c = p0 * gg_bf2f(dof[kk][e]);
                for (int o = 1; o < 32; o <<= 1) { a += gg_shfl_xor(a, o); c += gg_shfl_xor(c, o); }
                if ((lane & 31) == 0) {
                    sRed[wave][0][kk * 16 + 8 * hi + e] = a;
                    sRed[wave][1][kk * 16 + 8 * hi + e] = c;
                }
            }
        float s0 = ds0;
        for (int o = 1; o < 32; o <<= 1) s0 += gg_shfl_xor(s0, o);
        if (lane == 0) sRed[wave][2][0] = s0;
    }

    // per score: p = 2^(alpha' s + bias'_j - lse'), dS = p (dP - D): the two subtractions ride on the MFMAs' accumulate
    // inputs (S starts at -lse' / alpha', dP at -D), leaving fma, v_exp_f32, mul and the bf16 pack on the vector ALU
    const float sinit = -lse * GGA_LOG2E * inv_a2, dinit = -dsum;
    auto tload = [&](const bf16_t* base, long long ld, int t0) {
        return GEN ? gga_tile_load_c(base, ld, t0, nk) : gga_tile_load(base, ld, t0);
    };
    GgaTileRegs rk = tload(kb, rsk, 0), rv = tload(vb, rsv, 0);
    gga_tile_store(rk, sK[0], sKt[0], sKb[0], b2);
    if (GEN) gga_tile_bias(sKb[0], kbias, 0, nk);
    gga_tile_store(rv, sV[0], nullptr, nullptr);
    if (64 < nk) {
        rk = tload(kb, rsk, 64);
        rv = tload(vb, rsv, 64);
    }
    gg_sync();
    for (int j0 = 0, buf = 0; j0 < nk; j0 += 64, buf ^= 1) {
        if (j0 + 64 < nk) {
            gga_tile_store(rk, sK[buf ^ 1], sKt[buf ^ 1], sKb[buf ^ 1], b2);
            if (GEN) gga_tile_bias(sKb[buf ^ 1], kbias, j0 + 64, nk);
            gga_tile_store(rv, sV[buf ^ 1], nullptr, nullptr);
            if (j0 + 128 < nk) {
                rk = tload(kb, rsk, j0 + 128);
                rv = tload(vb, rsv, j0 + 128);
            }
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = sinit; dp[r] = dinit; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                st = gg_mfma_32x32x16_bf16(gga_frag_rowk(sK[buf], jb * 32, kk, lane), qf[kk], st);
                dp = gg_mfma_32x32x16_bf16(gga_frag_rowk(sV[buf], jb * 32, kk, lane), dof[kk], dp);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kb4 = gga_rows4(sKb[buf], jb, g, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = gg_exp2f(a2 * st[4 * g + e] + kb4[e]);
                    st[4 * g + e] = pv * dp[4 * g + e];      // dS^T
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u16x8 dsf = gga_pack8(st, c);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dqt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sKt[buf], db * 32, jb * 32 + 16 * c, lane), dsf, dqt[db]);
            }
        }
        gg_sync();
    }

    bf16_t* dqrow = p.dq + ((long long)b * p.n + qic) * rs + hd * GGA_D;
    if (qi == qic) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u16x4 o4 = {gg_f2bf(dqt[db][4 * g + 0] * p.alpha), gg_f2bf(dqt[db][4 * g + 1] * p.alpha),
                            gg_f2bf(dqt[db][4 * g + 2] * p.alpha), gg_f2bf(dqt[db][4 * g + 3] * p.alpha)};
                *(u16x4*)(dqrow + db * 32 + 8 * g + 4 * hi) = o4;
            }
    }
    gg_sync();
    if (GEN && !p.has_null) return;
    if (threadIdx.x < 192) {
        const int which = threadIdx.x >> 6, d = threadIdx.x & 63;
        float s = 0.f;
        if (which < 2 || d == 0)
            for (int w = 0; w < 4; ++w) s += sRed[w][which][d];
        p.null_part[(((long long)bh * gridDim.x + bx) * 3 + which) * 64 + d] = s;
    }
}

// ---- backward, part 2: dk, dv ----------------------------------------------------------------------------------
// grid: (n / 128, B*h); wave w owns keys j0 + 32w .. +31 and loops over all queries
template <bool GEN>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_attn_bwd_dkv_kernel(GgAttnParams p) {
    GG_SHARED __attribute__((aligned(16))) bf16_t sQ[64][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) bf16_t sDO[64][GGA_KP];
    GG_SHARED __attribute__((aligned(16))) char sQt[64 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) char sDOt[64 * GGA_TP];
    GG_SHARED __attribute__((aligned(16))) float sLse[64];
    GG_SHARED __attribute__((aligned(16))) float sD[64];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    int bx, bh;
    gg_attn_block(p.xcd, bx, bh);
    const int b = bh / p.h, hd = bh % p.h;
    const long long rs = (long long)p.h * GGA_D;
    const long long boff = (long long)b * p.n * rs + hd * GGA_D;           // dense query-side tensors (dO) and, when !GEN, all
    const long long rsq = GEN ? p.ldq : rs, rsk = GEN ? p.ldk : rs, rsv = GEN ? p.ldv : rs;
    const int nk = GEN ? p.m : p.n;
    const bf16_t* qb = p.q + (long long)b * p.n * rsq + hd * GGA_D;
    const bf16_t* kb = p.k + (long long)b * nk * rsk + hd * GGA_D;
    const bf16_t* vb = p.v + (long long)b * nk * rsv + hd * GGA_D;
    const long long koff = (long long)b * nk * rs + hd * GGA_D;            // dense key-side outputs: dk, dv
    const int kj0 = bx * 128 + wave * 32;
    const int kj = kj0 + (lane & 31);
    const int kjc = (GEN && kj >= nk) ? nk - 1 : kj;

    u16x8 kf[4], vf[4];
    if (GEN) {
        gga_load_frags_c(kf, kb, rsk, kj0, nk, lane);
        gga_load_frags_c(vf, vb, rsv, kj0, nk, lane);
    } else {
        gga_load_frags(kf, p.k + boff, rs, kj0, lane);
        gga_load_frags(vf, p.v + boff, rs, kj0, lane);
    }
    float ksq = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        for (int e = 0; e < 8; ++e) { float f = gg_bf2f(kf[kk][e]); ksq += f * f; }
    ksq += gg_shfl_xor(ksq, 32);
    const float a2 = p.alpha * GGA_LOG2E;
    float sinit = p.beta * ksq / p.alpha;                  // (beta' |k_j|^2) / alpha'
    if (GEN) {
        if (p.kbias) sinit += p.kbias[(long long)b * nk + kjc] / p.alpha;
        if (kj != kjc) sinit = -1.0e30f;                    // past the last key: probability 0
    }

    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[db][r] = 0.f; dvt[db][r] = 0.f; }
    float dbias = 0.f;

    auto qload = [&](const bf16_t* base, long long ld, int t0) {
        return GEN ? gga_tile_load_c(base, ld, t0, p.n) : gga_tile_load(base, ld, t0);
    };
    auto stat_load = [&](int i, float& lse_o, float& d_o_) {          // rows past the last query get lse = +big: probability 0
        const int ic = (GEN && i >= p.n) ? p.n - 1 : i;
        lse_o = p.lse[(long long)bh * p.n + ic];
        d_o_ = p.dvec[(long long)bh * p.n + ic];
        if (GEN && i >= p.n) lse_o = 1.0e30f;
    };
    GgaTileRegs rq = qload(qb, rsq, 0), rdo = qload(p.d_o + boff, rs, 0);
    float r_lse = 0.f, r_d = 0.f;
    if (threadIdx.x < 64) stat_load(threadIdx.x, r_lse, r_d);
    for (int i0 = 0; i0 < p.n; i0 += 64) {
        gg_sync();
        gga_tile_store(rq, sQ, sQt, nullptr);
        gga_tile_store(rdo, sDO, sDOt, nullptr);
        if (threadIdx.x < 64) {
            sLse[threadIdx.x] = r_lse * GGA_LOG2E;
            sD[threadIdx.x] = r_d;
        }
        gg_sync();
        if (i0 + 64 < p.n) {
            rq = qload(qb, rsq, i0 + 64);
            rdo = qload(p.d_o + boff, rs, i0 + 64);
            if (threadIdx.x < 64) stat_load(i0 + 64 + threadIdx.x, r_lse, r_d);
        }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            // S starts at bias'_j / alpha' (this lane's key), so p = 2^(alpha' S - lse'_i) is one fma (negated LDS operand) + v_exp_f32
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = sinit; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = gg_mfma_32x32x16_bf16(gga_frag_rowk(sQ, ib * 32, kk, lane), kf[kk], s);
                dp = gg_mfma_32x32x16_bf16(gga_frag_rowk(sDO, ib * 32, kk, lane), vf[kk], dp);
            }
            f32x16 pr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ls = gga_rows4(sLse, ib, g, lane);       // lse' = lse log2 e
                const f32x4 dd = gga_rows4(sD, ib, g, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = gg_exp2f(a2 * s[4 * g + e] - ls[e]);
                    pr[4 * g + e] = pv;
                    const float ds = pv * (dp[4 * g + e] - dd[e]);
                    s[4 * g + e] = ds;
                    dbias += ds;
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u16x8 pf = gga_pack8(pr, c), dsf = gga_pack8(s, c);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dvt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sDOt, db * 32, ib * 32 + 16 * c, lane), pf, dvt[db]);
                    dkt[db] = gg_mfma_32x32x16_bf16(gga_frag_tr(sQt, db * 32, ib * 32 + 16 * c, lane), dsf, dkt[db]);
                }
            }
        }
    }
    dbias += gg_shfl_xor(dbias, 32);

    // dk_j = alpha * sum_i dS_ij q_i + dbias_j * 2*beta*k_j   (d/dk of beta*|k|^2)
    if (GEN && kj != kjc) return;
    const bf16_t* krow = GEN ? kb + (long long)kj * rsk : p.k + boff + (long long)kj * rs;
    const bool tied = p.dk == p.dq;
    bf16_t* dkrow = p.dk + (GEN ? koff : boff) + (long long)kj * rs;
    bf16_t* dvrow = p.dv + (GEN ? koff : boff) + (long long)kj * rs;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + 8 * g + 4 * hi;
            u16x4 k4 = *(const u16x4*)(krow + d);
            u16x4 q4 = {0, 0, 0, 0};
            if (tied) q4 = *(const u16x4*)(dkrow + d);       // tied projections (k is q): dk aliases dq, the sum is stored
            u16x4 dk4, dv4;
            for (int e = 0; e < 4; ++e) {
                dk4[e] = gg_f2bf(p.alpha * dkt[db][4 * g + e] + 2.f * p.beta * dbias * gg_bf2f(k4[e]) + gg_bf2f(q4[e]));
                dv4[e] = gg_f2bf(dvt[db][4 * g + e]);
            }
            *(u16x4*)(dkrow + d) = dk4;
            *(u16x4*)(dvrow + d) = dv4;
        }
}
