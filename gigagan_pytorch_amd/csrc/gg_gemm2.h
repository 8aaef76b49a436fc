// gg_gemm2.h — the large-tile variant of the contraction kernel (gg_gemm.h) for the MFMA-bound layers of the
// GigaGAN step: the discriminator's 128..512-channel 3x3 convolutions, their data/weight gradients, and the
// generator's low-resolution adaptive convolutions (reference gp.py:402-409, :1608-1621, :1454-1470).
//
// Same operand model, conv gather, epilogue and split-K contract as gg_gemm_kernel; what changes is the tiling:
//   * 512 threads = 8 wavefronts (two per SIMD), block tile BM x BN with BM = 256, BN in {256, 128}, k-tile 64;
//     each wave owns (BM/WM) x (BN/WN) = 128 x 64 (or 128 x 32) outputs = 4 x 2 (4 x 1) MFMA 32x32 tiles, so one
//     k-tile costs a wave 24 ds_read_b128 for 32 v_mfma_f32_32x32x16_bf16 (v1: 8 reads per 8 MFMAs).
//   * LDS rows hold 64 k-elements at a 144-byte pitch (9 sixteen-byte slots: 9*r mod 16 is injective on every
//     16-lane ds_read_b128 service group, MI355X_MICROARCH §LDS), double buffered: 2*(BM+BN)*144 B <= 144 KiB,
//     one workgroup per CU.
//   * register staging ordered as "write the staged tile after the barrier, re-issue the next global loads at
//     once" (cdna_hip_programming.md T14/G15): the global loads of tile t+2 are in flight during the MFMAs of
//     tile t, one barrier per k-tile.
//   * conv gather with CV % 64 == 0: a k-tile lies inside one filter tap, so the tap offset is a scalar and each
//     staged 16-byte vector needs one add and one mask test (per-row corner offset + tap-validity bitmask are
//     computed once per workgroup).
//   * reduction-major ("KROW") operands — both operands of a weight gradient — are staged UNTRANSPOSED, [k][column]
//     rows of 16-byte vectors exactly as they sit in HBM (pitch = 2*COLS + 64 bytes, so the four k-rows a
//     transpose read touches fall on different 64-byte bank quarters), and the k-contiguous MFMA fragments are
//     produced by ds_read_b64_tr_b16 (two per fragment). No per-element transposing ds_write_b32 pass as in v1.
//   * XCD-aware tile order over a flattened 1-D grid: the dispatcher places block b on XCD b % 8; consecutive logical
//     work items (output tile fastest, then k-slice / batch) are remapped onto the same XCD so that the N-tiles of
//     one M-tile, vertically adjacent pixel rows and — for weight gradients — all output tiles of one pixel slice
//     share that XCD's L2.
#pragma once
#include <type_traits>
#include "gg_gemm.h"

#define GG2_BK 64
#define GG2_PITCH 72   // bf16 elements per LDS row: 64 + 8 pad = 144 bytes
#define GG2_NT 512

// ---- staging loaders (512 threads) -------------------------------------------------------------------------

template <int ROWS>
struct Gg2RowK {   // ROWS x 64 tile as ROWS*8 vectors of 8 k-elements: vector v = t + 512*i -> row v>>3, chunk v&7
    static constexpr int NV = ROWS * 8 / GG2_NT;
};
template <int COLS>
struct Gg2KRow {   // 64 k-rows x COLS columns kept reduction-major: vector v = t + 512*i -> k-row v / (COLS/8), group v % (COLS/8)
    static constexpr int CG = COLS / 8;                 // 16-byte column groups per k-row (divides 512)
    static constexpr int NV = 64 * CG / GG2_NT;
    static constexpr int KSTEP = GG2_NT / CG;           // k-row distance between a thread's successive vectors
    static constexpr int PITCH = COLS * 2 + 64;         // bytes
    static constexpr int BYTES = 64 * PITCH;
};
template <int ROWS>
struct Gg2RowKBytes {
    static constexpr int BYTES = ROWS * GG2_PITCH * 2;
};

template <int ROWS>
GG_DEVICE void gg2_load_rowk_dense(u16x8* regs, const bf16_t* base, int ld, int nrows, int r0, int kend, int k0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Gg2RowK<ROWS>::NV; ++i) {
        int v = t + GG2_NT * i;
        int row = v >> 3, kc = v & 7;
        u16x8 x = gg_zero8();
        int r = r0 + row, k = k0 + kc * 8;
        if (r < nrows && k < kend) {
            x = *(const u16x8*)(base + (long long)r * ld + k);
            if (k + 8 > kend) {
                for (int e = 0; e < 8; ++e)
                    if (k + e >= kend) x[e] = 0;
            }
        }
        regs[i] = x;
    }
}

template <int ROWS>
GG_DEVICE void gg2_store_rowk(bf16_t (*s)[GG2_PITCH], const u16x8* regs) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Gg2RowK<ROWS>::NV; ++i) {
        int v = t + GG2_NT * i;
        *(u16x8*)&s[v >> 3][(v & 7) * 8] = regs[i];
    }
}

// ---- buffer-addressed ROWK loaders -------------------------------------------------------------------------------------
// per-lane byte offsets are loop invariant (set up once per workgroup), the k-tile contributes one scalar offset, rows beyond the
// matrix and padding taps are zero-filled by the descriptor's bounds check: the k-loop issues its 16-byte loads back to back with
// a handful of scalar / vector instructions in front of them (the pointer-arithmetic form spent ~30 per load: as many issue cycles
// as the tile's MFMAs).
template <int ROWS>
GG_DEVICE void gg2_brows_init(unsigned* voff, int ld, int nrows, int r0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Gg2RowK<ROWS>::NV; ++i) {
        const int v = t + GG2_NT * i;
        const int row = v >> 3, kc = v & 7;
        voff[i] = (r0 + row < nrows) ? (unsigned)(((long long)row * ld + kc * 8) * 2) : 0xFFFFFFFFu;
    }
}

// full 64-wide k-tiles only (the caller sends ragged tails through gg2_load_rowk_dense)
template <int ROWS>
GG_DEVICE void gg2_bload_rowk_dense(u16x8* regs, GgBuf buf, const unsigned* voff, long long r0_ld, int k0) {
    const unsigned soff = (unsigned)((r0_ld + k0) * 2);
#pragma unroll
    for (int i = 0; i < Gg2RowK<ROWS>::NV; ++i) regs[i] = gg_buf_load16(buf, voff[i], soff);
}

struct Gg2ConvRowB {
    unsigned voff;      // byte offset of (window corner + lane's 8-channel chunk) from the biased base; 0xFFFFFFFF for rows beyond M
    unsigned mask;      // bit (kh*S + kw) set when that tap reads inside the image
    int img;
};

// `bias` (elements): the descriptor base sits that far BEFORE the tensor so that corners of padded windows stay non-negative
GG_DEVICE Gg2ConvRowB gg2_conv_row_b(const GgGemmParams& p, int m, int kc, long long bias) {
    Gg2ConvRowB r;
    r.voff = 0xFFFFFFFFu; r.mask = 0; r.img = 0;
    if (m < p.M) {
        const int hw = p.OH * p.OW;
        const int img = m / hw, rem = m - img * hw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
        r.img = img;
        const long long corner = (((long long)img * p.H + ih0) * p.W + iw0) * p.C;
        r.voff = (unsigned)((corner + bias + kc * 8) * 2);
        unsigned int mask = 0;
        for (int kh = 0; kh < p.R; ++kh)
            for (int kw = 0; kw < p.S; ++kw) {
                const int ih = ih0 + kh, iw = iw0 + kw;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) mask |= 1u << (kh * p.S + kw);
            }
        r.mask = mask;
    }
    return r;
}

template <int ROWS>
GG_DEVICE void gg2_conv_rows_init_b(Gg2ConvRowB* rows, const GgGemmParams& p, int m0, long long bias) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Gg2RowK<ROWS>::NV; ++i) {
        const int v = t + GG2_NT * i;
        rows[i] = gg2_conv_row_b(p, m0 + (v >> 3), v & 7, bias);
    }
}

// position of the NEXT k-tile to load inside the (tap, channel) reduction index: k-tiles are visited in order, 64 at a time, so
// the tap decode (two integer divisions per tile in scalar code, ~70 instructions) is done once and then stepped
struct Gg2ConvCursor {
    int tap, cv0, ci0, kh, kw;      // ci0 = cv0 % C: the physical channel of the tile's first element (CV = n * C for stacked inputs)
};

GG_DEVICE Gg2ConvCursor gg2_conv_cursor(const GgGemmParams& p, int k0) {
    Gg2ConvCursor c;
    c.tap = k0 / p.CV;
    c.cv0 = k0 - c.tap * p.CV;
    c.ci0 = (p.CV == p.C) ? c.cv0 : c.cv0 % p.C;
    c.kh = c.tap / p.S;
    c.kw = c.tap - c.kh * p.S;
    return c;
}

template <int ROWS>
GG_DEVICE void gg2_bload_rowk_conv(u16x8* regs, const Gg2ConvRowB* rows, GgBuf buf, const GgGemmParams& p, int kend, int k0,
                                   Gg2ConvCursor& cur) {
    // workgroup-uniform: the tap of this k-tile and its byte offset from a window corner
    const unsigned soff = (unsigned)((((long long)cur.kh * p.W + cur.kw) * p.C + cur.ci0) * 2);
    const int rem = kend - k0;                            // <= 0: the tile is past the slice; < 64: a 32-wide split-K tail
    const unsigned lane_live = ((int)((threadIdx.x & 7) * 8) < rem) ? 1u : 0u;
    const int tap = cur.tap;
#pragma unroll
    for (int i = 0; i < Gg2RowK<ROWS>::NV; ++i) {
        const unsigned ok = (rows[i].mask >> tap) & lane_live;             // 1: inside the image and the slice
        regs[i] = gg_buf_load16(buf, rows[i].voff | (ok - 1u), soff);      // not ok -> offset 0xFFFFFFFF -> hardware zero fill
    }
    cur.cv0 += GG2_BK;
    cur.ci0 += GG2_BK;
    if (cur.ci0 >= p.C) cur.ci0 -= p.C;            // C % 64 == 0 whenever CV != C (planner), so the step lands exactly
    if (cur.cv0 >= p.CV) {
        cur.cv0 = 0;
        cur.ci0 = 0;
        ++cur.tap;
        if (++cur.kw == p.S) { cur.kw = 0; ++cur.kh; }
    }
}

// the per-sample input scale of a modulated conv, applied when the staged tile is parked in LDS (one k-tile after its loads
// were issued: the data has arrived, nothing stalls); k0 is the tile's first reduction index
template <int ROWS>
GG_DEVICE void gg2_scale_rowk_conv(u16x8* regs, const Gg2ConvRowB* rows, const GgGemmParams& p, int k0) {
    const int tap = k0 / p.CV;
    const int cv = k0 - tap * p.CV + (threadIdx.x & 7) * 8;
#pragma unroll
    for (int i = 0; i < Gg2RowK<ROWS>::NV; ++i) regs[i] = gg_scale8(regs[i], p.in_scale + (long long)rows[i].img * p.CV + cv);
}

template <int COLS>
GG_DEVICE void gg2_load_krow_dense(u16x8* regs, const bf16_t* base, int ld, int ncols, int c0, int kend, int k0) {
    const int t = threadIdx.x;
    const int cg = t % Gg2KRow<COLS>::CG, kr = t / Gg2KRow<COLS>::CG;
    const int c = c0 + cg * 8;
#pragma unroll
    for (int i = 0; i < Gg2KRow<COLS>::NV; ++i) {
        int k = k0 + kr + Gg2KRow<COLS>::KSTEP * i;
        u16x8 x = gg_zero8();
        if (c < ncols && k < kend) {
            x = *(const u16x8*)(base + (long long)k * ld + c);
            if (c + 8 > ncols) {
                for (int e = 0; e < 8; ++e)
                    if (c + e >= ncols) x[e] = 0;
            }
        }
        regs[i] = x;
    }
}

template <int COLS>
GG_DEVICE void gg2_store_krow(char* tile, const u16x8* regs) {
    const int t = threadIdx.x;
    const int cg = t % Gg2KRow<COLS>::CG, kr = t / Gg2KRow<COLS>::CG;
#pragma unroll
    for (int i = 0; i < Gg2KRow<COLS>::NV; ++i)
        *(u16x8*)(tile + (kr + Gg2KRow<COLS>::KSTEP * i) * Gg2KRow<COLS>::PITCH + cg * 16) = regs[i];
}

// k-contiguous MFMA fragment (lane l: tile column col0 + (l & 31), k = kk*16 + 8*(l >> 5) + 0..7) out of a
// reduction-major tile: two transpose reads, each fed by the 16 lanes of a group pointing at a [4 k][16 col] block
template <int COLS>
GG_DEVICE u16x8 gg2_frag_krow(const char* tile, int col0, int kk, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int col = col0 + (g & 1) * 16 + 4 * (i & 3);
    const int row = kk * 16 + (g >> 1) * 8 + (i >> 2);
    const char* p0 = tile + row * Gg2KRow<COLS>::PITCH + col * 2;
    u16x4 a = gg_lds_read_tr16((const bf16_t*)p0);
    u16x4 b = gg_lds_read_tr16((const bf16_t*)(p0 + 4 * Gg2KRow<COLS>::PITCH));
    u16x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
}

// conv gather, KROW (weight gradient): k = output pixel, column = (tap, cv); a thread stages ONE column group
template <int COLS>
GG_DEVICE GgConvCol gg2_conv_col_init(const GgGemmParams& p, int c0) {
    const int c = c0 + (threadIdx.x % Gg2KRow<COLS>::CG) * 8;
    GgConvCol cc;
    cc.valid = c < p.M;
    int tap = cc.valid ? c / p.CV : 0;
    cc.cv = cc.valid ? c - tap * p.CV : 0;
    cc.kh = tap / p.S;
    cc.kw = tap - cc.kh * p.S;
    cc.ci = (p.CV == p.C) ? cc.cv : cc.cv % p.C;
    return cc;
}

template <int COLS>
GG_DEVICE void gg2_load_krow_conv(u16x8* regs, const GgConvCol& cc, const GgGemmParams& p, int kend, int k0) {
    const int kr = threadIdx.x / Gg2KRow<COLS>::CG;
    const int hw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < Gg2KRow<COLS>::NV; ++i) {
        int pix = k0 + kr + Gg2KRow<COLS>::KSTEP * i;
        u16x8 x = gg_zero8();
        if (cc.valid && pix < kend) {
            int img, oh, ow;
            if (p.hw_shift >= 0) {
                img = pix >> p.hw_shift;
                int rem = pix & (hw - 1);
                oh = rem >> p.w_shift;
                ow = rem & (p.OW - 1);
            } else {
                img = pix / hw;
                int rem = pix - img * hw;
                oh = rem / p.OW;
                ow = rem - oh * p.OW;
            }
            int ih = oh * p.stride - p.pad + cc.kh, iw = ow * p.stride - p.pad + cc.kw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
                x = *(const u16x8*)(p.A + (((long long)img * p.H + ih) * p.W + iw) * p.C + cc.ci);
                if (p.in_scale) x = gg_scale8(x, p.in_scale + (long long)img * p.CV + cc.cv);
            }
        }
        regs[i] = x;
    }
}

// ---- buffer-addressed KROW loaders (weight gradients: both operands are indexed by the output pixel) -------------------------
// dense (dy [pixel][co]): lane offset (k-row within the tile, column group) is loop invariant, the tile adds one scalar.
template <int COLS>
GG_DEVICE void gg2_bkrow_init(unsigned* voff, int ld, int ncols, int c0) {
    const int cg = threadIdx.x % Gg2KRow<COLS>::CG, kr = threadIdx.x / Gg2KRow<COLS>::CG;
#pragma unroll
    for (int i = 0; i < Gg2KRow<COLS>::NV; ++i)
        voff[i] = (c0 + cg * 8 + 8 <= ncols) ? (unsigned)(((long long)(kr + Gg2KRow<COLS>::KSTEP * i) * ld + cg * 8) * 2) : 0xFFFFFFFFu;
}

template <int COLS>
GG_DEVICE void gg2_bload_krow_dense(u16x8* regs, GgBuf buf, const unsigned* voff, int ld, int c0, int kend, int k0) {
    const unsigned soff = (unsigned)(((long long)k0 * ld + c0) * 2);
    const int rem = kend - k0, kr = threadIdx.x / Gg2KRow<COLS>::CG;
#pragma unroll
    for (int i = 0; i < Gg2KRow<COLS>::NV; ++i) {
        const unsigned ok = (kr + Gg2KRow<COLS>::KSTEP * i < rem) ? 1u : 0u;
        regs[i] = gg_buf_load16(buf, voff[i] | (ok - 1u), soff);
    }
}

// conv gather for stride-1 'same' windows (OH == H, OW == W: every 3x3 / 1x1 layer) with power-of-two image sides: the input
// pixel of output pixel `pix` and tap (kh, kw) is pix + (kh - pad) * W + (kw - pad), so a lane's byte offset is
// (k-row * C + tap shift + channel group) -- loop invariant -- plus the scalar k0 * C; only the inside-the-image test needs the
// pixel's (oh, ow), two shifts and two unsigned compares per vector (the generic form spent ~110 vector instructions per load)
struct Gg2ConvColB {
    int dkh, dkw, valid;
};

template <int COLS>
GG_DEVICE Gg2ConvColB gg2_conv_col_init_b(unsigned* voff, const GgGemmParams& p, int c0, long long bias) {
    const int cg = threadIdx.x % Gg2KRow<COLS>::CG, kr = threadIdx.x / Gg2KRow<COLS>::CG;
    const int c = c0 + cg * 8;
    Gg2ConvColB cc;
    cc.valid = c < p.M;
    const int tap = cc.valid ? c / p.CV : 0;
    const int cv = cc.valid ? c - tap * p.CV : 0;
    const int kh = tap / p.S, kw = tap - kh * p.S;
    const int ci = (p.CV == p.C) ? cv : cv % p.C;
    cc.dkh = kh - p.pad;
    cc.dkw = kw - p.pad;
    const long long shift = ((long long)cc.dkh * p.W + cc.dkw) * p.C + ci + bias;
#pragma unroll
    for (int i = 0; i < Gg2KRow<COLS>::NV; ++i)
        voff[i] = cc.valid ? (unsigned)(((long long)(kr + Gg2KRow<COLS>::KSTEP * i) * p.C + shift) * 2) : 0xFFFFFFFFu;
    return cc;
}

template <int COLS>
GG_DEVICE void gg2_bload_krow_conv(u16x8* regs, GgBuf buf, const unsigned* voff, const Gg2ConvColB& cc, const GgGemmParams& p,
                                   int kend, int k0) {
    const unsigned soff = (unsigned)((long long)k0 * p.C * 2);
    const int kr = threadIdx.x / Gg2KRow<COLS>::CG;
#pragma unroll
    for (int i = 0; i < Gg2KRow<COLS>::NV; ++i) {
        const int pix = k0 + kr + Gg2KRow<COLS>::KSTEP * i;
        const int ow = pix & (p.OW - 1), oh = (pix >> p.w_shift) & (p.OH - 1);
        const bool in = (unsigned)(oh + cc.dkh) < (unsigned)p.H && (unsigned)(ow + cc.dkw) < (unsigned)p.W && pix < kend;
        const unsigned ok = in ? 1u : 0u;
        regs[i] = gg_buf_load16(buf, voff[i] | (ok - 1u), soff);
    }
}

// ---- epilogue -----------------------------------------------------------------------------------------------
// Every accumulator index below is a compile-time constant by construction (template recursion, one (i, j, g)
// register quad per step). A `#pragma unroll` nest over the full epilogue body exceeds clang's pragma-unroll budget
// for the 256x256 tile; the accumulators were then indexed dynamically, i.e. demoted to scratch memory for the WHOLE
// kernel — measured 4x slower.
//
// Two store paths:
//  * direct: each lane stores its 4-column quads itself (fp32 outputs, split-K partials, the depth-to-space scatter,
//    ragged N). 8 or 16 bytes per lane and 32 different rows per instruction.
//  * staged (bf16 [m][ldc] outputs, N % 4 == 0): the wave parks its 128 x WTN sub-tile in LDS (pitch WTN*2 + 8 bytes:
//    conflict-free ds_write_b64 / ds_read_b64) and writes it back row-contiguous — 16 lanes cover 128 contiguous bytes
//    of one output row — with the residual read the same way. Short-K launches (1x1 convs, attention projections) are
//    bound by this store pass, not by the MFMAs.

GG_DEVICE f32x4 gg_ld4(const float* p) { return *(const f32x4*)p; }

// activation( acc*alpha * out_scale + bias*bias_scale + noise*noise_w ) for one quad of 4 consecutive columns
template <bool FULL_EPI>
GG_DEVICE void gg2_quad_math(const GgGemmParams& e, float* v, int m, int n, f32x4 cb, f32x4 cnw, float nz) {
    for (int q = 0; q < 4; ++q) v[q] *= e.alpha;
    if (FULL_EPI) {
        if (e.out_scale) {
            const float* os = e.out_scale + (long long)(m / e.rows_per_group) * e.N + n;
            for (int q = 0; q < 4; ++q)
                if (n + q < e.N) v[q] *= os[q];
        }
        for (int q = 0; q < 4; ++q) v[q] += cb[q] + nz * cnw[q];
        if (e.act != GG_ACT_NONE)
            for (int q = 0; q < 4; ++q) v[q] = gg_apply_act(v[q], e.act, e.act_slope);
    }
}

// per-column epilogue operands of the quad starting at column n (zeros where absent / out of range)
GG_DEVICE void gg2_quad_columns(const GgGemmParams& e, int n, f32x4& cb, f32x4& cnw) {
    cb = (f32x4){0.f, 0.f, 0.f, 0.f};
    cnw = cb;
    if (n >= e.N) return;
    const bool vec = (n + 3 < e.N) && ((e.N & 3) == 0);
    if (e.bias) {
        if (vec && (((unsigned long long)e.bias) & 15) == 0) cb = gg_ld4(e.bias + n);
        else
            for (int q = 0; q < 4; ++q)
                if (n + q < e.N) cb[q] = e.bias[n + q];
        for (int q = 0; q < 4; ++q) cb[q] *= e.bias_scale;
    }
    if (e.noise) {
        if (vec && (((unsigned long long)e.noise_w) & 15) == 0) cnw = gg_ld4(e.noise_w + n);
        else
            for (int q = 0; q < 4; ++q)
                if (n + q < e.N) cnw[q] = e.noise_w[n + q];
    }
}

// IDX = jg * TM + i: the column quad (j, g) is the slow index so that its operands are fetched once for the TM rows
template <int IDX, int TM, int TN, bool FULL_EPI, bool STAGED>
GG_DEVICE void gg2_epilogue_step(const f32x16 (&acc)[TM][TN], const GgGemmParams& e, int b, int bz, int m_lane, int n_lane,
                                 char* stage, int stage_pitch, int lane, f32x4 cb, f32x4 cnw) {
    if constexpr (IDX < TM * TN * 4) {
        constexpr int jg = IDX / TM, i = IDX % TM, j = jg / 4, g = jg % 4;
        const int m = m_lane + i * 32, n = n_lane + j * 32 + 8 * g;
        if (FULL_EPI && i == 0) gg2_quad_columns(e, n, cb, cnw);
        if (STAGED || (m < e.M && n < e.N)) {
            float v[4] = {acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
            if (!STAGED && e.splitk > 1) {
                float* dst = e.partial + ((long long)bz * e.M + m) * e.N + n;
                if (n + 3 < e.N && (e.N & 3) == 0) {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    *(f32x4*)dst = o;
                } else {
                    for (int q = 0; q < 4; ++q)
                        if (n + q < e.N) dst[q] = v[q];
                }
            } else {
                const float nz = (FULL_EPI && e.noise && m < e.M) ? e.noise[m] : 0.f;
                gg2_quad_math<FULL_EPI>(e, v, m < e.M ? m : 0, n, cb, cnw, nz);
                if (STAGED) {
                    u16x4 o = {gg_f2bf(v[0]), gg_f2bf(v[1]), gg_f2bf(v[2]), gg_f2bf(v[3])};
                    *(u16x4*)(stage + (i * 32 + (lane & 31)) * stage_pitch + (j * 32 + 8 * g + 4 * (lane >> 5)) * 2) = o;
                } else {
                    if (FULL_EPI && e.residual) {
                        for (int q = 0; q < 4; ++q)
                            if (n + q < e.N) v[q] += gg_bf2f(e.residual[(long long)m * e.ldr + n + q]) * e.res_scale;
                    }
                    gg_store4(e, b, m, n, v);
                }
            }
        }
        gg2_epilogue_step<IDX + 1, TM, TN, FULL_EPI, STAGED>(acc, e, b, bz, m_lane, n_lane, stage, stage_pitch, lane, cb, cnw);
    }
}

// write-back of a staged WTM x WTN sub-tile, V = 8 or 4 columns per lane: lane -> (row lane / (WTN/V) + it * rows_per_pass, V columns).
// V = 8 (N, every row pitch and base 16-byte aligned: the host-visible case of every model layer): 16 bytes per lane and instruction -
// short-K launches (1x1 convolutions, attention / FeedForward projections) are bound by the ISSUE of this pass's loads and stores, not by
// bandwidth (MI355X_MICROARCH.md: 8x dwordx4 halves a row-per-lane store tail against 16x dwordx2); the staged tile keeps its pitch
// (WTN*2 + 8 bytes: conflict-free ds_write_b64), so a lane reads its 16 bytes as two 8-byte halves.
template <int WTM, int WTN, int V>
GG_DEVICE void gg2_stage_writeback_v(const GgGemmParams& e, int b, const char* stage, int stage_pitch, int m_wave, int n_wave, int lane) {
    constexpr int CPR = WTN / V;          // lane slots per row
    constexpr int RPP = 64 / CPR;         // rows per pass
    typedef typename std::conditional<V == 8, u16x8, u16x4>::type vec_t;
    const int qc = lane % CPR, rr = lane / CPR;
    const int n = n_wave + qc * V;
    bf16_t* cbase = (bf16_t*)e.Cout + (long long)b * e.c_bs;
#pragma unroll 4
    for (int it = 0; it < WTM / RPP; ++it) {
        const int row = it * RPP + rr;
        const int m = m_wave + row;
        if (m < e.M && n < e.N) {
            vec_t o;
            if constexpr (V == 8) {
                const u16x4 lo = *(const u16x4*)(stage + row * stage_pitch + qc * 16), hi = *(const u16x4*)(stage + row * stage_pitch + qc * 16 + 8);
                o = u16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
                o = *(const u16x4*)(stage + row * stage_pitch + qc * 8);
            }
            if (e.aux_mode == 1) {            // FeedForward up-projection: keep the pre-activation, emit gelu of its bf16 value
                *(vec_t*)(e.aux + (long long)m * e.ld_aux + n) = o;
                for (int q = 0; q < V; ++q) o[q] = gg_f2bf(gg_gelu_f(gg_bf2f(o[q])));
            } else if (e.aux_mode == 2) {     // data gradient of the down-projection: times gelu'(h) = Phi(h) + h phi(h)
                const vec_t h = *(const vec_t*)(e.aux + (long long)m * e.ld_aux + n);
                for (int q = 0; q < V; ++q) {
                    float c, d;
                    const float x = gg_bf2f(h[q]);
                    gg_normal_cdf_pdf(x, c, d);
                    o[q] = gg_f2bf(gg_bf2f(o[q]) * (c + x * d));
                }
            }
            if (e.residual) {
                const vec_t r = *(const vec_t*)(e.residual + (long long)m * e.ldr + n);
                for (int q = 0; q < V; ++q) o[q] = gg_f2bf(gg_bf2f(o[q]) + gg_bf2f(r[q]) * e.res_scale);
            }
            *(vec_t*)(cbase + (long long)m * e.ldc + n) = o;
        }
    }
}

template <int WTM, int WTN>
GG_DEVICE void gg2_stage_writeback(const GgGemmParams& e, int b, const char* stage, int stage_pitch, int m_wave, int n_wave, int lane) {
    const bool wide = !(e.N & 7) && !(e.ldc & 7) && !(e.c_bs & 7) && !((unsigned long long)e.Cout & 15) &&
                      (!e.residual || (!(e.ldr & 7) && !((unsigned long long)e.residual & 15))) &&
                      (!e.aux_mode || (!(e.ld_aux & 7) && !((unsigned long long)e.aux & 15)));
    if (wide) gg2_stage_writeback_v<WTM, WTN, 8>(e, b, stage, stage_pitch, m_wave, n_wave, lane);
    else gg2_stage_writeback_v<WTM, WTN, 4>(e, b, stage, stage_pitch, m_wave, n_wave, lane);
}

// ---- the kernel ---------------------------------------------------------------------------------------------

template <int BM, int BN, int WM, int WN, bool A_KROW, bool B_KROW, bool A_CONV, bool FULL_EPI>
GG_KERNEL GG_LAUNCH_BOUNDS(GG2_NT) void gg_gemm2_kernel(GgGemmParams p) {
    static_assert(WM * WN == 8, "8 wavefronts per workgroup");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    constexpr int ANV = A_KROW ? Gg2KRow<BM>::NV : Gg2RowK<BM>::NV;
    constexpr int BNV = B_KROW ? Gg2KRow<BN>::NV : Gg2RowK<BN>::NV;
    constexpr int ABYTES = A_KROW ? Gg2KRow<BM>::BYTES : Gg2RowKBytes<BM>::BYTES;
    constexpr int BBYTES = B_KROW ? Gg2KRow<BN>::BYTES : Gg2RowKBytes<BN>::BYTES;

    // one LDS object: [buffer 0: A tile | B tile][buffer 1: A tile | B tile]
    GG_SHARED __attribute__((aligned(16))) char smem[2 * (ABYTES + BBYTES)];
    auto tileA = [&](int buf) { return smem + buf * (ABYTES + BBYTES); };
    auto tileB = [&](int buf) { return smem + buf * (ABYTES + BBYTES) + ABYTES; };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order (block b runs on XCD b % 8): XCD x works through a contiguous range of logical tiles
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    // flattened grid, output tile fastest: the tiles of one (batch, k-slice) follow each other on one XCD, so a weight
    // gradient's pixel slice (both operands stream over the same pixels) is fetched into that XCD's L2 once
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_mn = tiles_n * ((p.M + BM - 1) / BM);
    const int bz = wg / tiles_mn, tile = wg - bz * tiles_mn;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int b = bz / p.splitk, ks = bz % p.splitk;
    const int kbeg = ks * p.k_per_split;
    int kend = kbeg + p.k_per_split;
    if (kend > p.K) kend = p.K;

    const bf16_t* Ab = p.A + (long long)b * p.a_bs;
    // per-image weight operand (per-sample weights of the adaptive conv): a tile never straddles two images (planner)
    const bf16_t* Bb = p.B + (long long)b * p.b_bs + ((A_CONV && !A_KROW && p.b_img_stride) ? (long long)(m0 / (p.OH * p.OW)) * p.b_img_stride : 0);

    // ROWK operands go through buffer descriptors (gg2_bload_*); the conv window corners are made non-negative by a base that
    // sits (pad * W + pad) * C elements before the tensor
    const long long abias = A_CONV ? ((long long)p.pad * p.W + p.pad) * p.C : 0;
    Gg2ConvRowB crow[A_CONV && !A_KROW ? ANV : 1];
    if (A_CONV && !A_KROW) gg2_conv_rows_init_b<BM>(crow, p, m0, abias);
    unsigned avoff[(!A_CONV || A_KROW) ? ANV : 1], bvoff[BNV];
    // reduction-major (weight-gradient) operands: buffer path for 8-aligned column counts; the conv gather additionally needs
    // stride-1 'same' windows on power-of-two images without an input scale (p.krow_fast, set by the host)
    const bool a_kfast = A_KROW && (A_CONV ? p.krow_fast != 0 : (p.M & 7) == 0);
    const bool b_kfast = B_KROW && (p.N & 7) == 0;
    Gg2ConvColB ccb;
    ccb.dkh = ccb.dkw = ccb.valid = 0;
    if (!A_CONV && !A_KROW) gg2_brows_init<BM>(avoff, p.lda, p.M, m0);
    if (!A_CONV && A_KROW) gg2_bkrow_init<BM>(avoff, p.lda, p.M, m0);
    if (A_CONV && A_KROW && a_kfast) ccb = gg2_conv_col_init_b<BM>(avoff, p, m0, abias);
    if (!B_KROW) gg2_brows_init<BN>(bvoff, p.ldb, p.N, n0);
    else gg2_bkrow_init<BN>(bvoff, p.ldb, p.N, n0);
    GgBuf bufA = gg_make_buf((const void*)(Ab - abias), (unsigned long long)(p.a_bytes - (long long)b * p.a_bs * 2 + abias * 2));
    GgBuf bufB = gg_make_buf((const void*)Bb, (unsigned long long)(p.b_bytes - (Bb - p.B) * 2));
    GgConvCol ccol;
    ccol.kh = ccol.kw = ccol.ci = ccol.cv = ccol.valid = 0;
    if (A_CONV && A_KROW) ccol = gg2_conv_col_init<BM>(p, m0);

    u16x8 ra[ANV], rb[BNV];

    Gg2ConvCursor ccur = {0, 0, 0, 0, 0};              // load_tiles is called for kbeg, kbeg + 64, kbeg + 128, ... in this order
    if (A_CONV && !A_KROW) ccur = gg2_conv_cursor(p, kbeg);
    auto load_tiles = [&](int k0) {
        const bool full = kend - k0 >= GG2_BK;          // workgroup-uniform; ragged tails keep the pointer-arithmetic loaders
        if (A_CONV) {
            if (A_KROW) {
                if (a_kfast) gg2_bload_krow_conv<BM>(ra, bufA, avoff, ccb, p, kend, k0);
                else gg2_load_krow_conv<BM>(ra, ccol, p, kend, k0);
            } else gg2_bload_rowk_conv<BM>(ra, crow, bufA, p, kend, k0, ccur);
        } else {
            if (A_KROW) {
                if (a_kfast) gg2_bload_krow_dense<BM>(ra, bufA, avoff, p.lda, m0, kend, k0);
                else gg2_load_krow_dense<BM>(ra, Ab, p.lda, p.M, m0, kend, k0);
            } else if (full) gg2_bload_rowk_dense<BM>(ra, bufA, avoff, (long long)m0 * p.lda, k0);
            else gg2_load_rowk_dense<BM>(ra, Ab, p.lda, p.M, m0, kend, k0);
        }
        if (B_KROW) {
            if (b_kfast) gg2_bload_krow_dense<BN>(rb, bufB, bvoff, p.ldb, n0, kend, k0);
            else gg2_load_krow_dense<BN>(rb, Bb, p.ldb, p.N, n0, kend, k0);
        } else if (full) gg2_bload_rowk_dense<BN>(rb, bufB, bvoff, (long long)n0 * p.ldb, k0);
        else gg2_load_rowk_dense<BN>(rb, Bb, p.ldb, p.N, n0, kend, k0);
    };
    auto store_tiles = [&](int buf, int k0) {
        if (A_CONV && !A_KROW && p.in_scale && k0 < kend) gg2_scale_rowk_conv<BM>(ra, crow, p, k0);
        if (A_KROW) gg2_store_krow<BM>(tileA(buf), ra);
        else gg2_store_rowk<BM>((bf16_t(*)[GG2_PITCH])tileA(buf), ra);
        if (B_KROW) gg2_store_krow<BN>(tileB(buf), rb);
        else gg2_store_rowk<BN>((bf16_t(*)[GG2_PITCH])tileB(buf), rb);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend > kbeg) ? (kend - kbeg + GG2_BK - 1) / GG2_BK : 0;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0, kbeg);
        if (nk > 1) load_tiles(kbeg + GG2_BK);
    }
    gg_sync();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
#ifdef GG2_PROBE
    // probe builds only (tests/probes/build_gemm2_probe.sh): phases of the k-loop can be switched off to time the others.
    // bit 0: no LDS stores, bit 1: no global loads, bit 2: no LDS fragment reads, bit 3: no MFMAs. Results are garbage.
    const int dbg = p.xcd_slices;
    u16x8 pfa[TM], pfb[TN];
    float psink = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) pfa[i] = (u16x8){0x3f80, 0x3f00, 0x3e80, 0x3f80, 0x3f00, 0x3e80, 0x3f80, 0x3f00};
#pragma unroll
    for (int j = 0; j < TN; ++j) pfb[j] = (u16x8){0x3f00, 0x3f80, 0x3f00, 0x3e80, 0x3f80, 0x3f00, 0x3e80, 0x3f80};
#else
    constexpr int dbg = 0;
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        // tile kt+1 sits in the staging registers (its loads were issued one MFMA phase ago): park it in the
        // other LDS buffer (free since the barrier that ended iteration kt-1), then put tile kt+2 in flight
        if (kt + 1 < nk && !(dbg & 1)) store_tiles(buf ^ 1, kbeg + (kt + 1) * GG2_BK);
        if (kt + 2 < nk && !(dbg & 2)) load_tiles(kbeg + (kt + 2) * GG2_BK);
#pragma unroll
        for (int kk = 0; kk < GG2_BK / 16; ++kk) {
            u16x8 fa[TM], fb[TN];
#ifdef GG2_PROBE
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = pfa[i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = pfb[j];
            if (!(dbg & 4))
#endif
            {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (A_KROW) fa[i] = gg2_frag_krow<BM>(tileA(buf), wm * WTM + i * 32, kk, lane);
                else fa[i] = *(const u16x8*)&((const bf16_t(*)[GG2_PITCH])tileA(buf))[wm * WTM + i * 32 + frow][kk * 16 + fk];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (B_KROW) fb[j] = gg2_frag_krow<BN>(tileB(buf), wn * WTN + j * 32, kk, lane);
                else fb[j] = *(const u16x8*)&((const bf16_t(*)[GG2_PITCH])tileB(buf))[wn * WTN + j * 32 + frow][kk * 16 + fk];
            }
            }
#ifdef GG2_PROBE
            if (dbg & 8) {      // keep the fragment reads alive without the matrix pipe
#pragma unroll
                for (int i = 0; i < TM; ++i) psink += gg_bf2f(fa[i][0]);
#pragma unroll
                for (int j = 0; j < TN; ++j) psink += gg_bf2f(fb[j][1]);
            } else
#endif
            {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = gg_mfma_32x32x16_bf16(fb[j], fa[i], acc[i][j]);   // swapped: lane registers run along n
            }
        }
        gg_sync();
    }
#ifdef GG2_PROBE
    acc[0][0][0] += psink;
#endif

    // epilogue (same fragment ownership as gg_gemm_kernel): lane owns row m = ... + (lane & 31); register r holds
    // column n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5). The argument struct is copied from the kernarg segment HERE
    // (gg_late_params) so that epilogue-only fields do not occupy registers during the main loop.
    const GgGemmParams e = *gg_late_params(p);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int m_wave = m0 + wm * WTM, n_wave = n0 + wn * WTN;
    const bool staged = e.splitk == 1 && !e.c_f32 && !e.d2s && (e.N & 3) == 0 && (e.ldc & 3) == 0 &&
                        (!e.residual || (e.ldr & 3) == 0);
    if (staged) {
        constexpr int SP = WTN * 2 + 8;                      // stage pitch in bytes
        static_assert(8 * WTM * SP <= 2 * (ABYTES + BBYTES), "staging area must fit the operand tiles' LDS");
        char* stage = smem + wave * (WTM * SP);              // (the barrier that ended the k-loop freed the tiles)
        gg2_epilogue_step<0, TM, TN, FULL_EPI, true>(acc, e, b, bz, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), stage, SP,
                                                     lane, z4, z4);
        gg_sync();
        gg2_stage_writeback<WTM, WTN>(e, b, stage, SP, m_wave, n_wave, lane);
    } else {
        gg2_epilogue_step<0, TM, TN, FULL_EPI, false>(acc, e, b, bz, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), nullptr, 0,
                                                      lane, z4, z4);
    }
}

// ---- LDS-DMA staged row-major tiles (used by gg_conv3.h for its weight tiles) ------------------------------------------------
// A wave's DMA instruction (buffer_load_dwordx4 ... lds) deposits 64 x 16 contiguous bytes, so rows cannot be padded; the tile is
// stored as 128-byte rows with its eight 16-byte chunks XOR-swizzled: chunk c of row r sits in slot c ^ ((r >> 1) & 7). The
// fragment reads (32 rows x one chunk per half wave) are conflict-free under the ds_read_b128 lane grouping: rows two apart share
// banks and get different slots.
// (Non-temporal stores in gg2_stage_writeback_v - every staged epilogue of gg_gemm2 / gg_conv3 - measured 402.3 / 401.3 -> 401.7 img/s on the
// step, profiles/r05_nt_wb_ab.log: neutral, not kept; gg_pgemm.h keeps them, -4 % on its own launches.)
// (A whole-kernel variant of gg_gemm2_kernel staged this way — no staging registers, no ds_write phase — measured within +-2 % of
// the register-staged kernel on every config-2 layer and 0.0 % on the step, profiles/r02_dma_ab_{off,on}.log: the 256x256 tile is
// fed at the L2's delivery rate either way. It was deleted; gg_conv3.h attacks the bytes instead.)
template <int ROWS>
struct Gg2Dma {
    static constexpr int NV = ROWS / 64;          // DMA instructions per thread and k-tile (8 rows x 128 bytes per wave each)
    static constexpr int BYTES = ROWS * 128;
    static_assert(NV >= 1, "at least eight rows per wave");
};

// the thread's i-th DMA vector: row 8 * (wave * NV + i) + lane / 8 of the tile, LDS slot lane % 8
template <int ROWS>
GG_DEVICE int gg2d_row(int i) { return 8 * ((threadIdx.x >> 6) * Gg2Dma<ROWS>::NV + i) + ((threadIdx.x & 63) >> 3); }
// ... which holds chunk slot ^ ((row >> 1) & 7) = (lane % 8) ^ (lane / 16) ^ (4 if the 8-row group wave * NV + i is odd)
template <int ROWS>
GG_DEVICE int gg2d_chunk(int i) {
    const int grp = (int)(threadIdx.x >> 6) * Gg2Dma<ROWS>::NV + i;
    return (int)((threadIdx.x & 7) ^ ((threadIdx.x & 63) >> 4) ^ ((grp & 1) << 2));
}

