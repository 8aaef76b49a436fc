// gg_gemm.h — the one contraction kernel of the GigaGAN hot path: a batched, LDS-tiled bf16 MFMA GEMM
// whose A operand can be an implicit im2col gather over an NHWC activation (stride-1 "same" convolution)
// and whose operands can each be stored reduction-major ("KROW": the matrix is stored [k][m]) or
// reduction-minor ("ROWK": stored [m][k]).  Every dense contraction of the reference's G+D step maps
// onto it:
//
//   reference op (gigagan_pytorch.py)                     A                      B
//   F.conv2d fwd / dgrad  (:407, :1608-1621, :1454-1470)  conv gather, ROWK      weights [co][tap][ci], ROWK
//   conv weight gradient (autograd of the above)          conv gather, KROW      dy [pixel][co], KROW
//   nn.Linear / 1x1 conv / EqualLinear (:530-536,:871)    dense ROWK             dense ROWK
//   einsum('b i d, b j d -> b i j') (:574, :579)          q ROWK                 k ROWK
//   einsum('b i j, b j d -> b i d') (:590)                attn ROWK              v KROW
//   and their gradients (all four transposes).
//
// Tiling (CDNA4): 256 threads = 4 wavefronts of 64 lanes; block tile BM x BN x 32; each wave owns a
// (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Operand tiles are staged global -> registers -> LDS with 16-byte loads, double-buffered in LDS (one
// barrier per k-tile, next tile's global loads in flight during the MFMAs).  LDS rows are k-contiguous
// with an 80-byte pitch, which makes the fragment `ds_read_b128`s conflict-free (MI355X_MICROARCH §LDS).
// KROW operands are transposed while staging: each thread loads two k-rows x 8 columns and writes 8
// packed (k, k+1) dwords (kp-fastest lane order => at most the free 2-way ds_write_b32 conflict).
// The MFMA is issued with swapped operands (B-tile as the A operand) so that each lane's accumulator
// registers run along N: a lane owns 4 consecutive output channels per register quad and stores them
// with one 8-byte (bf16) or 16-byte (fp32) store.
//
// Algorithmic work per launch: 2*M*N*K*batch flops; algorithmic HBM bytes: (M*K + N*K + M*N)*2*batch
// for dense operands, and for the conv gather the activation is counted once (H*W*C per image).
#pragma once
#include "gg_device.h"

enum { GG_ACT_NONE = 0, GG_ACT_LRELU = 1, GG_ACT_GELU = 2, GG_ACT_SILU = 3 };

struct GgGemmParams {
    // problem: C[b][m][n] = epilogue( sum_k A[b][m][k] * B[b][n][k] )
    int M, N, K;
    int batch;
    int splitk;       // >= 1; > 1: fp32 partials go to `partial`, finished by gg_splitk_reduce_kernel
    int k_per_split;  // multiple of 32
    // A operand. dense: element (m,k) at A[b*a_bs + m*lda + k] (ROWK) or A[b*a_bs + k*lda + m] (KROW)
    const bf16_t* A;
    long long a_bs;
    int lda;
    // B operand. element (n,k) at B[b*b_bs + n*ldb + k] (ROWK) or B[b*b_bs + k*ldb + n] (KROW)
    const bf16_t* B;
    long long b_bs;
    int ldb;
    // conv gather on A: activation [img][H][W][C] bf16, output grid OH x OW, input pixel of output (oh, ow) and
    // tap (kh, kw) is (oh*stride - pad + kh, ow*stride - pad + kw); reduction/gather index = (tap = kh*S + kw, cv)
    // with cv in [0, CV); physical channel = cv % C.
    int H, W, C, CV, R, S, pad, stride, OH, OW;
    int w_shift, hw_shift;  // log2(OW), log2(OH*OW) when both are powers of two, else -1 (pixel decode by shifts)
    const float* in_scale;  // optional [img][CV] multiplier applied to the gathered activation
    // output / epilogue
    void* Cout;
    long long c_bs;
    int ldc;
    int c_f32;  // 0: bf16 output, 1: fp32 output
    float alpha;
    const float* bias;       // [N]
    float bias_scale;        // multiplies the bias (lets alpha scale conv + bias together)
    const float* out_scale;  // [m / rows_per_group][N]
    int rows_per_group;
    const float* noise;    // [M]
    const float* noise_w;  // [N]
    int act;
    float act_slope;
    const bf16_t* residual;  // optional bf16 [M][ldr] added after the activation (times res_scale)
    int ldr;
    float res_scale;
    // depth-to-space scatter store (adjoint of a stride-`d2s` gather): column n = (tap, c) with c in [0, d2s_c),
    // row m = (img, oh, ow) on the d2s_oh x d2s_ow grid; element goes to pixel (oh*d2s + tap/d2s_t, ow*d2s + tap%d2s_t)
    // of an [img][d2s_oh*d2s][d2s_ow*d2s][d2s_c] tensor. d2s == 0: plain [m][ldc] store.
    int d2s, d2s_t, d2s_c, d2s_oh, d2s_ow;
    float* partial;  // [batch][splitk][M][N] fp32
    // 4-wave kernel, split-K launches with few output tiles (narrow weight gradients): 1-D grid in which all tiles of
    // one k-slice land on the same XCD (block b -> XCD b % 8), so the slice of x / dy they all stream is fetched from
    // HBM once and shared through that XCD's L2. 0: (tiles, 1, batch*splitk) grid.
    int xcd_slices;
    long long b_img_stride;    // conv forward only: > 0: image i's weights start at B + i * b_img_stride (per-sample weights)
    const float* bank_mix;     // gg_lrconv MIX: [img][CV / C] per-image weights of the stacked banks (null: the banks stay stacked along k)
    // byte extents of the A / B operands as seen from their base pointers (buffer descriptors of the 8-wave kernel's ROWK loaders)
    long long a_bytes, b_bytes;
    int krow_fast;     // weight-gradient conv gather: stride-1 'same' windows, power-of-two image sides, no input scale
    int ws_spx, ws_depth;   // gg_wgrads_kernel: pixels per step (256 / 128) and prefetch depth in steps
    // gg_gemm2's staged bf16 epilogue, GELU fused around a 1x1 convolution pair (FeedForward, gp.py:726-740): mode 1 = the staged value h
    // is ALSO stored to aux and the output is gelu(h); mode 2 = aux holds h and the output is staged * gelu'(h)
    bf16_t* aux; int aux_mode, ld_aux;
    int ws_cs, ws_cstore, ws_gmul;   // ... channels per x slot, rows stored per tap, memory rows per ring row
    int pg_order;      // gg_pgemm: 1 = tiles dealt round-robin over the workgroups, 0 = a contiguous run per workgroup
    int buf_ok;        // 31 when both operands' byte extents fit the 32-bit offsets of a buffer descriptor, else 0 (bits: A conv rows,
                       // A dense rows, A reduction-major, B dense rows, B reduction-major)
};

// Phi(x) and phi(x) of the standard normal for the exact (erf) GELU of gp.py:731 and its derivatives: Abramowitz-Stegun 7.1.26,
// |error| < 1.5e-7 in erf - far below the bf16 rounding of everything it feeds - with ONE exponential shared by both (erf's
// e^{-z^2} at z = x / sqrt 2 IS sqrt(2 pi) phi(x)), one reciprocal and five fmas: cheap enough for a GEMM epilogue (erff + expf cost
// ~50 instructions per element)
GG_DEVICE void gg_normal_cdf_pdf(float x, float& cdf, float& pdf) {
    const float ax = x < 0.f ? -x : x;
    const float e = gg_expf(-0.5f * x * x);
    const float t = gg_rcpf(1.f + 0.3275911f * 0.70710678118654752f * ax);      // (1 ulp: two orders below the approximation's own error)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.f - poly * e;
    cdf = 0.5f * (1.f + (x < 0.f ? -erf_abs : erf_abs));
    pdf = 0.3989422804014327f * e;
}
GG_DEVICE float gg_gelu_f(float x) {
    float c, d;
    gg_normal_cdf_pdf(x, c, d);
    return x * c;
}

GG_DEVICE float gg_apply_act(float v, int act, float slope) {
    if (act == GG_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == GG_ACT_GELU) return gg_gelu_f(v);
    if (act == GG_ACT_SILU) return v / (1.f + gg_expf(-v));
    return v;
}

GG_DEVICE float gg_epilogue(const GgGemmParams& p, float acc, int m, int n) {
    float v = acc * p.alpha;
    if (p.out_scale) v *= p.out_scale[(long long)(m / p.rows_per_group) * p.N + n];
    if (p.bias) v += p.bias[n] * p.bias_scale;
    if (p.noise) v += p.noise[m] * p.noise_w[n];
    v = gg_apply_act(v, p.act, p.act_slope);
    if (p.residual) v += gg_bf2f(p.residual[(long long)m * p.ldr + n]) * p.res_scale;
    return v;
}

// element offset of output (m, n) for the depth-to-space scatter store (n .. n+3 stay inside one tap: d2s_c % 4 == 0)
GG_DEVICE long long gg_d2s_offset(const GgGemmParams& p, int m, int n) {
    const int tap = n / p.d2s_c, c = n - tap * p.d2s_c;
    const int ty = tap / p.d2s_t, tx = tap - ty * p.d2s_t;
    const int hw = p.d2s_oh * p.d2s_ow;
    const int img = m / hw, rem = m - img * hw;
    const int oh = rem / p.d2s_ow, ow = rem - oh * p.d2s_ow;
    const long long W2 = (long long)p.d2s_ow * p.d2s;
    return (((long long)img * p.d2s_oh * p.d2s + oh * p.d2s + ty) * W2 + (ow * p.d2s + tx)) * p.d2s_c + c;
}

// store 4 consecutive-n results of row m
GG_DEVICE void gg_store4(const GgGemmParams& p, int b, int m, int n, const float* v) {
    const long long off = p.d2s ? gg_d2s_offset(p, m, n) : (long long)b * p.c_bs + (long long)m * p.ldc + n;
    if (p.c_f32) {
        float* c = (float*)p.Cout + off;
        if (n + 3 < p.N && (p.ldc & 3) == 0) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)c = o;
        } else {
            for (int e = 0; e < 4; ++e)
                if (n + e < p.N) c[e] = v[e];
        }
    } else {
        bf16_t* c = (bf16_t*)p.Cout + off;
        if (n + 3 < p.N && (p.ldc & 3) == 0) {
            u16x4 o = {gg_f2bf(v[0]), gg_f2bf(v[1]), gg_f2bf(v[2]), gg_f2bf(v[3])};
            *(u16x4*)c = o;
        } else {
            for (int e = 0; e < 4; ++e)
                if (n + e < p.N) c[e] = gg_f2bf(v[e]);
        }
    }
}

#define GG_BK 32
#define GG_LDS_PITCH 40  // bf16 elements per LDS row: 32 + 8 pad = 80 bytes

// ---- operand tile loaders -------------------------------------------------------------------------
// A loader fills `NV` 16-byte registers per thread for one k-tile; a matching store writes them to LDS.

template <int ROWS>
struct GgRowKLayout {  // ROWK: ROWS*4 vectors of 8 k-elements
    static constexpr int NV = (ROWS * 4 + 255) / 256;
};
template <int ROWS>
struct GgKRowLayout {  // KROW: 16 k-pairs x ROWS/8 column groups, 2 vectors each
    static constexpr int ITEMS = 16 * (ROWS / 8);
    static constexpr int NI = (ITEMS + 255) / 256;
    static constexpr int NV = 2 * NI;
};

GG_DEVICE u16x8 gg_zero8() {
    u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return z;
}

GG_DEVICE u16x8 gg_scale8(u16x8 v, const float* s) {
    u16x8 o;
    for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(v[e]) * s[e]);
    return o;
}

// dense ROWK: rows [r0, r0+ROWS) of a [nrows][K] matrix with pitch ld, k-tile at k0
template <int ROWS>
GG_DEVICE void gg_load_rowk_dense(u16x8* regs, const bf16_t* base, int ld, int nrows, int r0, int K,
                                  int kend, int k0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) {
        int v = t + 256 * i;
        int row = v >> 2, kv = v & 3;
        u16x8 x = gg_zero8();
        int r = r0 + row, k = k0 + kv * 8;
        if (row < ROWS && r < nrows && k < kend) {
            x = *(const u16x8*)(base + (long long)r * ld + k);
            if (k + 8 > kend) {
                for (int e = 0; e < 8; ++e)
                    if (k + e >= kend) x[e] = 0;
            }
        }
        regs[i] = x;
    }
    (void)K;
}

template <int ROWS>
GG_DEVICE void gg_store_rowk(bf16_t (*s)[GG_LDS_PITCH], const u16x8* regs) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) {
        int v = t + 256 * i;
        int row = v >> 2, kv = v & 3;
        if (row < ROWS) *(u16x8*)&s[row][kv * 8] = regs[i];
    }
}

// dense KROW: matrix stored [k][ncols] with pitch ld; tile columns [c0, c0+ROWS), k-tile at k0.
template <int ROWS>
GG_DEVICE void gg_load_krow_dense(u16x8* regs, const bf16_t* base, int ld, int ncols, int c0, int kend,
                                  int k0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        int item = t + 256 * i;
        int kp = item & 15, cg = item >> 4;
        u16x8 x0 = gg_zero8(), x1 = gg_zero8();
        int c = c0 + cg * 8, k = k0 + 2 * kp;
        if (item < GgKRowLayout<ROWS>::ITEMS && c < ncols) {
            if (k < kend) x0 = *(const u16x8*)(base + (long long)k * ld + c);
            if (k + 1 < kend) x1 = *(const u16x8*)(base + (long long)(k + 1) * ld + c);
            if (c + 8 > ncols) {
                for (int e = 0; e < 8; ++e)
                    if (c + e >= ncols) { x0[e] = 0; x1[e] = 0; }
            }
        }
        regs[2 * i] = x0;
        regs[2 * i + 1] = x1;
    }
}

template <int ROWS>
GG_DEVICE void gg_store_krow(bf16_t (*s)[GG_LDS_PITCH], const u16x8* regs) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        int item = t + 256 * i;
        int kp = item & 15, cg = item >> 4;
        if (item < GgKRowLayout<ROWS>::ITEMS) {
            u16x8 x0 = regs[2 * i], x1 = regs[2 * i + 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned int d = (unsigned int)x0[e] | ((unsigned int)x1[e] << 16);
                *(unsigned int*)&s[cg * 8 + e][2 * kp] = d;
            }
        }
    }
}

// conv gather, ROWK: row m = output pixel (img, oh, ow); k = (tap, cv)
struct GgConvRow {
    long long img_off;  // img * H*W*C
    int img, ih0, iw0, valid;
};

template <int ROWS>
GG_DEVICE void gg_conv_rows_init(GgConvRow* rows, const GgGemmParams& p, int m0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) {
        int v = t + 256 * i;
        int row = v >> 2;
        int m = m0 + row;
        GgConvRow r;
        r.valid = (row < ROWS) && (m < p.M);
        int hw = p.OH * p.OW;
        int img = r.valid ? m / hw : 0;
        int rem = r.valid ? m - img * hw : 0;
        int oh = rem / p.OW, ow = rem - oh * p.OW;
        r.img = img;
        r.img_off = (long long)img * p.H * p.W * p.C;
        r.ih0 = oh * p.stride - p.pad;
        r.iw0 = ow * p.stride - p.pad;
        rows[i] = r;
    }
}

template <int ROWS>
GG_DEVICE void gg_load_rowk_conv(u16x8* regs, const GgConvRow* rows, const GgGemmParams& p, int kend,
                                 int k0) {
    const int t = threadIdx.x;
    // CV % 32 == 0: the whole 32-wide k-tile lies inside one filter tap, so (kh, kw) are workgroup-uniform
    // scalars computed once per k-tile instead of two integer divisions per 16-byte vector.
    const bool uniform_tap = (p.CV & 31) == 0;
    int u_kh = 0, u_kw = 0, u_cv0 = 0;
    if (uniform_tap) {
        int tap = k0 / p.CV;
        u_cv0 = k0 - tap * p.CV;
        u_kh = tap / p.S;
        u_kw = tap - u_kh * p.S;
    }
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) {
        int v = t + 256 * i;
        int kv = v & 3;
        int k = k0 + kv * 8;
        u16x8 x = gg_zero8();
        const GgConvRow& r = rows[i];
        if (r.valid && k < kend) {
            int kh, kw, cv;
            if (uniform_tap) {
                kh = u_kh; kw = u_kw; cv = u_cv0 + kv * 8;
            } else {
                int tap = k / p.CV;
                cv = k - tap * p.CV;
                kh = tap / p.S;
                kw = tap - kh * p.S;
            }
            int ih = r.ih0 + kh, iw = r.iw0 + kw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
                int ci = (p.CV == p.C) ? cv : cv % p.C;
                x = *(const u16x8*)(p.A + r.img_off + ((long long)ih * p.W + iw) * p.C + ci);
                if (p.in_scale) x = gg_scale8(x, p.in_scale + (long long)r.img * p.CV + cv);
            }
        }
        regs[i] = x;
    }
}

// conv gather, KROW (weight gradient): k = output pixel, column = (tap, cv)
struct GgConvCol {
    int kh, kw, ci, cv, valid;
};

template <int ROWS>
GG_DEVICE void gg_conv_cols_init(GgConvCol* cols, const GgGemmParams& p, int c0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        int item = t + 256 * i;
        int cg = item >> 4;
        int c = c0 + cg * 8;
        GgConvCol cc;
        cc.valid = (item < GgKRowLayout<ROWS>::ITEMS) && (c < p.M);
        int tap = cc.valid ? c / p.CV : 0;
        cc.cv = cc.valid ? c - tap * p.CV : 0;
        cc.kh = tap / p.S;
        cc.kw = tap - cc.kh * p.S;
        cc.ci = (p.CV == p.C) ? cc.cv : cc.cv % p.C;
        cols[i] = cc;
    }
}

template <int ROWS>
GG_DEVICE void gg_load_krow_conv(u16x8* regs, const GgConvCol* cols, const GgGemmParams& p, int kend, int k0) {
    const int t = threadIdx.x;
    const int hw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        int item = t + 256 * i;
        int kp = item & 15;
        u16x8 x[2] = {gg_zero8(), gg_zero8()};
        const GgConvCol& cc = cols[i];
        if (cc.valid) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int pix = k0 + 2 * kp + h;
                if (pix < kend) {
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
                        u16x8 y = *(const u16x8*)(p.A + (long long)img * p.H * p.W * p.C +
                                                  ((long long)ih * p.W + iw) * p.C + cc.ci);
                        if (p.in_scale) y = gg_scale8(y, p.in_scale + (long long)img * p.CV + cc.cv);
                        x[h] = y;
                    }
                }
            }
        }
        regs[2 * i] = x[0];
        regs[2 * i + 1] = x[1];
    }
}


// ---- buffer-addressed loaders (see gg_device.h GgBuf and gg_gemm2.h): loop-invariant per-lane byte offsets, one scalar offset per
// k-tile, zero fill by the descriptor's bounds check. Used when the operand's byte extent fits 32-bit offsets (p.buf_ok); the
// pointer-arithmetic loaders above stay for larger tensors, ragged tails and taps that change inside a k-tile (CV % 32 != 0).
template <int ROWS>
GG_DEVICE void gg_brows_init(unsigned* voff, int ld, int nrows, int r0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) {
        const int v = t + 256 * i;
        const int row = v >> 2, kv = v & 3;
        voff[i] = (row < ROWS && r0 + row < nrows) ? (unsigned)(((long long)row * ld + kv * 8) * 2) : 0xFFFFFFFFu;
    }
}

template <int ROWS>
GG_DEVICE void gg_bload_rowk_dense(u16x8* regs, GgBuf buf, const unsigned* voff, long long r0_ld, int k0) {
    const unsigned soff = (unsigned)((r0_ld + k0) * 2);
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) regs[i] = gg_buf_load16(buf, voff[i], soff);
}

struct GgConvRowB {
    unsigned voff, mask;
    int img;
};

template <int ROWS>
GG_DEVICE void gg_conv_rows_init_b(GgConvRowB* rows, const GgGemmParams& p, int m0, long long bias) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) {
        const int v = t + 256 * i;
        const int row = v >> 2, kv = v & 3, m = m0 + row;
        GgConvRowB r;
        r.voff = 0xFFFFFFFFu; r.mask = 0; r.img = 0;
        if (row < ROWS && m < p.M) {
            const int hw = p.OH * p.OW;
            const int img = m / hw, rem = m - img * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            r.img = img;
            r.voff = (unsigned)(((((long long)img * p.H + ih0) * p.W + iw0) * p.C + bias + kv * 8) * 2);
            unsigned int mask = 0;
            for (int kh = 0; kh < p.R; ++kh)
                for (int kw = 0; kw < p.S; ++kw) {
                    const int ih = ih0 + kh, iw = iw0 + kw;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) mask |= 1u << (kh * p.S + kw);
                }
            r.mask = mask;
        }
        rows[i] = r;
    }
}

struct GgConvCursor {
    int tap, cv0, ci0, kh, kw;
};

GG_DEVICE GgConvCursor gg_conv_cursor(const GgGemmParams& p, int k0) {
    GgConvCursor c;
    c.tap = k0 / p.CV;
    c.cv0 = k0 - c.tap * p.CV;
    c.ci0 = (p.CV == p.C) ? c.cv0 : c.cv0 % p.C;
    c.kh = c.tap / p.S;
    c.kw = c.tap - c.kh * p.S;
    return c;
}

// k-tiles of GG_BK (32) visited in order; needs CV % 32 == 0 (one tap per tile) and, for stacked inputs, C % 32 == 0
template <int ROWS>
GG_DEVICE void gg_bload_rowk_conv(u16x8* regs, const GgConvRowB* rows, GgBuf buf, const GgGemmParams& p, int kend, int k0,
                                  GgConvCursor& cur) {
    const unsigned soff = (unsigned)((((long long)cur.kh * p.W + cur.kw) * p.C + cur.ci0) * 2);
    const int rem = kend - k0;
    const unsigned lane_live = ((int)((threadIdx.x & 3) * 8) < rem) ? 1u : 0u;
    const int tap = cur.tap;
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) {
        const unsigned ok = (rows[i].mask >> tap) & lane_live;
        regs[i] = gg_buf_load16(buf, rows[i].voff | (ok - 1u), soff);
    }
    cur.cv0 += GG_BK;
    cur.ci0 += GG_BK;
    if (cur.ci0 >= p.C) cur.ci0 -= p.C;
    if (cur.cv0 >= p.CV) {
        cur.cv0 = 0;
        cur.ci0 = 0;
        ++cur.tap;
        if (++cur.kw == p.S) { cur.kw = 0; ++cur.kh; }
    }
}

template <int ROWS>
GG_DEVICE void gg_scale_rowk_conv(u16x8* regs, const GgConvRowB* rows, const GgGemmParams& p, int k0) {
    const int tap = k0 / p.CV;
    const int cv = k0 - tap * p.CV + (threadIdx.x & 3) * 8;
#pragma unroll
    for (int i = 0; i < GgRowKLayout<ROWS>::NV; ++i) regs[i] = gg_scale8(regs[i], p.in_scale + (long long)rows[i].img * p.CV + cv);
}

// reduction-major operands (weight gradients): item = (k-pair kp, column group cg), two vectors (pixels k0 + 2 kp, + 1) each
template <int ROWS>
GG_DEVICE void gg_bkrow_init(unsigned* voff, int ld, int ncols, int c0) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        const int item = t + 256 * i;
        const int kp = item & 15, cg = item >> 4;
        const bool ok = item < GgKRowLayout<ROWS>::ITEMS && c0 + cg * 8 + 8 <= ncols;
        voff[2 * i] = ok ? (unsigned)(((long long)(2 * kp) * ld + cg * 8) * 2) : 0xFFFFFFFFu;
        voff[2 * i + 1] = ok ? (unsigned)(((long long)(2 * kp + 1) * ld + cg * 8) * 2) : 0xFFFFFFFFu;
    }
}

template <int ROWS>
GG_DEVICE void gg_bload_krow_dense(u16x8* regs, GgBuf buf, const unsigned* voff, int ld, int c0, int kend, int k0) {
    const unsigned soff = (unsigned)(((long long)k0 * ld + c0) * 2);
    const int rem = kend - k0, t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        const int kp = (t + 256 * i) & 15;
        const unsigned ok0 = (2 * kp < rem) ? 1u : 0u, ok1 = (2 * kp + 1 < rem) ? 1u : 0u;
        regs[2 * i] = gg_buf_load16(buf, voff[2 * i] | (ok0 - 1u), soff);
        regs[2 * i + 1] = gg_buf_load16(buf, voff[2 * i + 1] | (ok1 - 1u), soff);
    }
}

// stride-1 'same' windows on power-of-two images, no input scale (p.krow_fast): see gg2_bload_krow_conv
struct GgConvColB {
    int dkh, dkw;
};

template <int ROWS>
GG_DEVICE void gg_conv_cols_init_b(GgConvColB* cols, unsigned* voff, const GgGemmParams& p, int c0, long long bias) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        const int item = t + 256 * i;
        const int kp = item & 15, cg = item >> 4;
        const int c = c0 + cg * 8;
        const bool valid = (item < GgKRowLayout<ROWS>::ITEMS) && (c < p.M);
        const int tap = valid ? c / p.CV : 0;
        const int cv = valid ? c - tap * p.CV : 0;
        const int kh = tap / p.S, kw = tap - kh * p.S;
        const int ci = (p.CV == p.C) ? cv : cv % p.C;
        cols[i].dkh = kh - p.pad;
        cols[i].dkw = kw - p.pad;
        const long long shift = ((long long)cols[i].dkh * p.W + cols[i].dkw) * p.C + ci + bias;
        voff[2 * i] = valid ? (unsigned)(((long long)(2 * kp) * p.C + shift) * 2) : 0xFFFFFFFFu;
        voff[2 * i + 1] = valid ? (unsigned)(((long long)(2 * kp + 1) * p.C + shift) * 2) : 0xFFFFFFFFu;
    }
}

template <int ROWS>
GG_DEVICE void gg_bload_krow_conv(u16x8* regs, GgBuf buf, const unsigned* voff, const GgConvColB* cols, const GgGemmParams& p,
                                  int kend, int k0) {
    const unsigned soff = (unsigned)((long long)k0 * p.C * 2);
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < GgKRowLayout<ROWS>::NI; ++i) {
        const int kp = (t + 256 * i) & 15;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pix = k0 + 2 * kp + h;
            const int ow = pix & (p.OW - 1), oh = (pix >> p.w_shift) & (p.OH - 1);
            const bool in = (unsigned)(oh + cols[i].dkh) < (unsigned)p.H && (unsigned)(ow + cols[i].dkw) < (unsigned)p.W && pix < kend;
            const unsigned ok = in ? 1u : 0u;
            regs[2 * i + h] = gg_buf_load16(buf, voff[2 * i + h] | (ok - 1u), soff);
        }
    }
}

// ---- the kernel ---------------------------------------------------------------------------------

template <int BM, int BN, int WM, int WN, bool A_KROW, bool B_KROW, bool A_CONV, bool FULL_EPI>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_gemm_kernel(GgGemmParams p) {
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    constexpr int WTM = BM / WM, WTN = BN / WN;  // per-wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    constexpr int ANV = A_KROW ? GgKRowLayout<BM>::NV : GgRowKLayout<BM>::NV;
    constexpr int BNV = B_KROW ? GgKRowLayout<BN>::NV : GgRowKLayout<BN>::NV;

    GG_SHARED __attribute__((aligned(16))) bf16_t sA[2][BM][GG_LDS_PITCH];
    GG_SHARED __attribute__((aligned(16))) bf16_t sB[2][BN][GG_LDS_PITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN;
    int bx = blockIdx.x, bz = blockIdx.z;
    if (p.xcd_slices) {
        const int blocks_mn = ((p.M + BM - 1) / BM) * tiles_n;
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        bx = q % blocks_mn;
        bz = (q / blocks_mn) * 8 + xcd;
        if (bz >= p.batch * p.splitk) return;
    }
    const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int b = bz / p.splitk, ks = bz % p.splitk;
    const int kbeg = ks * p.k_per_split;
    int kend = kbeg + p.k_per_split;
    if (kend > p.K) kend = p.K;

    const bf16_t* Ab = p.A + (long long)b * p.a_bs;
    // per-image weight operand (per-sample weights of the adaptive conv): a tile never straddles two images (planner)
    const bf16_t* Bb = p.B + (long long)b * p.b_bs + ((A_CONV && !A_KROW && p.b_img_stride) ? (long long)(m0 / (p.OH * p.OW)) * p.b_img_stride : 0);

    // which operands go through buffer descriptors (workgroup-uniform; see the loaders above)
    const bool a_rconv = A_CONV && !A_KROW && (p.buf_ok & 1) && (p.CV & 31) == 0 && (p.CV == p.C || (p.C & 31) == 0) &&
                         p.R * p.S <= 32;                        // the tap-validity mask is one 32-bit word
    const bool a_rdense = !A_CONV && !A_KROW && (p.buf_ok & 2);
    const bool a_kfast = A_KROW && (p.buf_ok & 4) && (A_CONV ? p.krow_fast != 0 : (p.M & 7) == 0);
    const bool b_rdense = !B_KROW && (p.buf_ok & 8);
    const bool b_kfast = B_KROW && (p.buf_ok & 16) && (p.N & 7) == 0;
    const long long abias = A_CONV ? ((long long)p.pad * p.W + p.pad) * p.C : 0;

    GgConvRow crow[A_CONV && !A_KROW ? ANV : 1];
    GgConvRowB crowb[A_CONV && !A_KROW ? ANV : 1];
    if (A_CONV && !A_KROW) {
        if (a_rconv) gg_conv_rows_init_b<BM>(crowb, p, m0, abias);
        else gg_conv_rows_init<BM>(crow, p, m0);
    }
    GgConvCol ccol[A_CONV && A_KROW ? GgKRowLayout<BM>::NI : 1];
    GgConvColB ccolb[A_CONV && A_KROW ? GgKRowLayout<BM>::NI : 1];
    unsigned avoff[ANV], bvoff[BNV];
    if (A_CONV && A_KROW) {
        if (a_kfast) gg_conv_cols_init_b<BM>(ccolb, avoff, p, m0, abias);
        else gg_conv_cols_init<BM>(ccol, p, m0);
    }
    if (!A_CONV && A_KROW && a_kfast) gg_bkrow_init<BM>(avoff, p.lda, p.M, m0);
    if (a_rdense) gg_brows_init<BM>(avoff, p.lda, p.M, m0);
    if (b_rdense) gg_brows_init<BN>(bvoff, p.ldb, p.N, n0);
    if (b_kfast) gg_bkrow_init<BN>(bvoff, p.ldb, p.N, n0);
    GgBuf bufA = gg_make_buf((const void*)(Ab - abias), (unsigned long long)(p.a_bytes - (long long)b * p.a_bs * 2 + abias * 2));
    GgBuf bufB = gg_make_buf((const void*)Bb, (unsigned long long)(p.b_bytes - (Bb - p.B) * 2));
    GgConvCursor ccur = {0, 0, 0, 0, 0};
    if (a_rconv) ccur = gg_conv_cursor(p, kbeg);          // load_tiles visits kbeg, kbeg + 32, ... in order

    u16x8 ra[ANV], rb[BNV];

    auto load_tiles = [&](int k0) {
        const bool full = kend - k0 >= GG_BK;
        if (A_CONV) {
            if (A_KROW) {
                if (a_kfast) gg_bload_krow_conv<BM>(ra, bufA, avoff, ccolb, p, kend, k0);
                else gg_load_krow_conv<BM>(ra, ccol, p, kend, k0);
            } else {
                if (a_rconv) gg_bload_rowk_conv<BM>(ra, crowb, bufA, p, kend, k0, ccur);
                else gg_load_rowk_conv<BM>(ra, crow, p, kend, k0);
            }
        } else {
            if (A_KROW) {
                if (a_kfast) gg_bload_krow_dense<BM>(ra, bufA, avoff, p.lda, m0, kend, k0);
                else gg_load_krow_dense<BM>(ra, Ab, p.lda, p.M, m0, kend, k0);
            } else {
                if (a_rdense && full) gg_bload_rowk_dense<BM>(ra, bufA, avoff, (long long)m0 * p.lda, k0);
                else gg_load_rowk_dense<BM>(ra, Ab, p.lda, p.M, m0, p.K, kend, k0);
            }
        }
        if (B_KROW) {
            if (b_kfast) gg_bload_krow_dense<BN>(rb, bufB, bvoff, p.ldb, n0, kend, k0);
            else gg_load_krow_dense<BN>(rb, Bb, p.ldb, p.N, n0, kend, k0);
        } else {
            if (b_rdense && full) gg_bload_rowk_dense<BN>(rb, bufB, bvoff, (long long)n0 * p.ldb, k0);
            else gg_load_rowk_dense<BN>(rb, Bb, p.ldb, p.N, n0, p.K, kend, k0);
        }
    };
    auto store_tiles = [&](int buf, int k0) {
        if (a_rconv && p.in_scale && k0 < kend) gg_scale_rowk_conv<BM>(ra, crowb, p, k0);
        if (A_KROW) gg_store_krow<BM>(sA[buf], ra);
        else gg_store_rowk<BM>(sA[buf], ra);
        if (B_KROW) gg_store_krow<BN>(sB[buf], rb);
        else gg_store_rowk<BN>(sB[buf], rb);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend > kbeg) ? (kend - kbeg + GG_BK - 1) / GG_BK : 0;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0, kbeg);
    }
    gg_sync();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool has_next = (kt + 1 < nk);
        if (has_next) load_tiles(kbeg + (kt + 1) * GG_BK);
#pragma unroll
        for (int kk = 0; kk < GG_BK / 16; ++kk) {
            u16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *(const u16x8*)&sA[buf][wm * WTM + i * 32 + frow][kk * 16 + fk];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *(const u16x8*)&sB[buf][wn * WTN + j * 32 + frow][kk * 16 + fk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // swapped operands: D[n_local][m_local], so a lane's registers run along n
                    acc[i][j] = gg_mfma_32x32x16_bf16(fb[j], fa[i], acc[i][j]);
        }
        if (has_next) store_tiles(buf ^ 1, kbeg + (kt + 1) * GG_BK);
        gg_sync();
    }

    // epilogue: lane owns output row m = ... + (lane & 31); register r is column
    //   n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WTN + j * 32 + 8 * g + 4 * hi;
                if (n >= p.N) continue;
                float v[4];
                if (p.splitk > 1) {
                    float* dst = p.partial + ((long long)bz * p.M + m) * p.N + n;
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) dst[e] = acc[i][j][g * 4 + e];
                } else {
                    if (FULL_EPI) {
                        for (int e = 0; e < 4; ++e)
                            v[e] = (n + e < p.N) ? gg_epilogue(p, acc[i][j][g * 4 + e], m, n + e) : 0.f;
                    } else {
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e] * p.alpha;
                    }
                    gg_store4(p, b, m, n, v);
                }
            }
        }
    }
}

// split-K finish for 2..8 slices of a [M][N] result with N % 4 == 0: a thread owns four consecutive columns and issues the 16-byte
// loads of ALL slices before the first add (the generic kernel below walks the slices in a run-time loop: one exposed round trip
// per slice). Slices are added in index order, as there.
template <int SK>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_splitk_reduce4_kernel(GgGemmParams p) {
    const long long per = (long long)p.M * p.N;
    const long long total4 = per * p.batch / 4;
    for (long long v4 = (long long)blockIdx.x * 256 + threadIdx.x; v4 < total4; v4 += (long long)gridDim.x * 256) {
        const long long idx = v4 * 4;
        int b = 0;
        long long rem = idx;
        if (p.batch > 1) {
            b = (int)(idx / per);
            rem = idx - (long long)b * per;
        }
        const float* src = p.partial + (long long)b * SK * per + rem;
        f32x4 v[SK];
#pragma unroll
        for (int ks = 0; ks < SK; ++ks) v[ks] = *(const f32x4*)(src + (long long)ks * per);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < SK; ++ks) s += v[ks];
        const int m = (int)(rem / p.N), n = (int)(rem - (long long)m * p.N);
        float o[4];
        for (int e = 0; e < 4; ++e) o[e] = gg_epilogue(p, s[e], m, n + e);
        const long long off = (long long)b * p.c_bs + (long long)m * p.ldc + n;
        if (p.c_f32) {
            const f32x4 w = {o[0], o[1], o[2], o[3]};
            *(f32x4*)((float*)p.Cout + off) = w;
        } else {
            const u16x4 w = {gg_f2bf(o[0]), gg_f2bf(o[1]), gg_f2bf(o[2]), gg_f2bf(o[3])};
            *(u16x4*)((bf16_t*)p.Cout + off) = w;
        }
    }
}

// finishes a split-K launch: sums the fp32 partials and applies the epilogue. Few splits: one thread per output
// element. Many splits (narrow weight gradients: a few thousand outputs, hundreds to thousands of slices): a wave per 64
// consecutive outputs (256-byte coalesced rows of the partial buffer), the 4 waves of the workgroup interleave over the
// slices and combine through LDS - the summation order is fixed, so the result is run-to-run deterministic.
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_splitk_reduce_kernel(GgGemmParams p) {
    GG_SHARED float part[3][64];
    const long long per = (long long)p.M * p.N;
    const long long total = per * p.batch;
    if (p.splitk <= 8) {
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
             idx += (long long)gridDim.x * 256) {
            int b = (int)(idx / per);
            long long rem = idx - (long long)b * per;
            int m = (int)(rem / p.N), n = (int)(rem - (long long)m * p.N);
            float s = 0.f;
            for (int ks = 0; ks < p.splitk; ++ks)
                s += p.partial[((long long)(b * p.splitk + ks) * p.M + m) * p.N + n];
            float v = gg_epilogue(p, s, m, n);
            const long long off = p.d2s ? gg_d2s_offset(p, m, n) : (long long)b * p.c_bs + (long long)m * p.ldc + n;
            if (p.c_f32) ((float*)p.Cout)[off] = v;
            else ((bf16_t*)p.Cout)[off] = gg_f2bf(v);
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        const long long idx = base + lane;
        const bool valid = idx < total;
        int b = 0;
        long long rem = 0;
        float s = 0.f;
        if (valid) {
            b = (int)(idx / per);
            rem = idx - (long long)b * per;
            const float* src = p.partial + (long long)b * p.splitk * per + rem;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int ks = wave;
            for (; ks + 12 < p.splitk; ks += 16) {
                s0 += src[(long long)ks * per];
                s1 += src[(long long)(ks + 4) * per];
                s2 += src[(long long)(ks + 8) * per];
                s3 += src[(long long)(ks + 12) * per];
            }
            for (; ks < p.splitk; ks += 4) s0 += src[(long long)ks * per];
            s = (s0 + s1) + (s2 + s3);
        }
        if (wave) part[wave - 1][lane] = s;
        gg_sync();
        if (wave == 0 && valid) {
            s += part[0][lane] + part[1][lane] + part[2][lane];
            int m = (int)(rem / p.N), n = (int)(rem - (long long)m * p.N);
            float v = gg_epilogue(p, s, m, n);
            const long long off = p.d2s ? gg_d2s_offset(p, m, n) : (long long)b * p.c_bs + (long long)m * p.ldc + n;
            if (p.c_f32) ((float*)p.Cout)[off] = v;
            else ((bf16_t*)p.Cout)[off] = gg_f2bf(v);
        }
        gg_sync();
    }
}
