// gg_api.hip — the C ABI of libgigagan_amd.so (declared in include/gigagan_amd.h).
// Argument validation, tile / split-K selection and kernel launches; no allocation, no synchronisation.
#include "gg_device.h"
#include "gg_gemm.h"
#include "gg_elementwise.h"
#include "../../include/gigagan_amd.h"

#include <stdio.h>
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

static int gg_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int gg_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "hip launch error %d: %s", (int)e, hipGetErrorString(e));
        return (int)e > 0 ? (int)e : 1;
    }
    return 0;
}

extern "C" int gg_version(void) { return GG_ABI_VERSION; }
extern "C" const char* gg_last_error(void) { return g_err; }
extern "C" int gg_is_emulator(void) {
#if defined(GG_HOST_EMULATION)
    return 1;
#else
    return 0;
#endif
}

// ---- GEMM / conv ------------------------------------------------------------------------------------

namespace {

struct GemmPlan {
    int tile;  // 1: 128x128, 2: 128x64, 3: 128x32
    int bm, bn;
    int splitk, k_per_split;
    long long blocks_mn;
};

int gg_validate_gemm(const gg_gemm_desc* d) {
    if (!d) return gg_fail(-1, "gg_gemm: null descriptor");
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0)
        return gg_fail(-2, "gg_gemm: M,N,K,batch must be positive (got %d %d %d %d)", d->M, d->N, d->K, d->batch);
    if (!d->A || !d->B || !d->C_out) return gg_fail(-3, "gg_gemm: null operand pointer");
    if (((uintptr_t)d->A & 15) || ((uintptr_t)d->B & 15) || ((uintptr_t)d->C_out & 7))
        return gg_fail(-4, "gg_gemm: operands must be 16-byte aligned (C 8-byte)");
    if (d->a_conv) {
        if (d->H <= 0 || d->W <= 0 || d->C <= 0 || d->CV <= 0 || d->R <= 0 || d->S <= 0)
            return gg_fail(-5, "gg_gemm: bad conv geometry");
        if ((d->R & 1) == 0 || d->R != d->S) return gg_fail(-5, "gg_gemm: conv kernel must be odd and square");
        if (d->C % 8 || d->CV % d->C) return gg_fail(-6, "gg_gemm: conv needs C %% 8 == 0 and CV %% C == 0 (C=%d CV=%d)", d->C, d->CV);
        if (d->batch != 1) return gg_fail(-7, "gg_gemm: conv gather takes batch == 1 (images are folded into M or K)");
        long long red = (long long)d->R * d->S * d->CV;
        if (d->a_layout == GG_ROWK) {
            if (d->K != red) return gg_fail(-8, "gg_gemm: conv fwd needs K == R*S*CV (%d vs %lld)", d->K, red);
            if (d->M % (d->H * d->W)) return gg_fail(-8, "gg_gemm: conv fwd needs M %% (H*W) == 0");
        } else {
            if (d->M != red) return gg_fail(-8, "gg_gemm: conv wgrad needs M == R*S*CV (%d vs %lld)", d->M, red);
            if (d->K % (d->H * d->W)) return gg_fail(-8, "gg_gemm: conv wgrad needs K %% (H*W) == 0");
        }
    } else {
        if (d->lda % 8) return gg_fail(-9, "gg_gemm: lda must be a multiple of 8 (got %d)", d->lda);
        int need = d->a_layout == GG_ROWK ? d->K : d->M;
        if (d->lda < need) return gg_fail(-9, "gg_gemm: lda %d < extent %d", d->lda, need);
        if (d->a_batch_stride % 8) return gg_fail(-9, "gg_gemm: A batch stride must be a multiple of 8");
        if (d->in_scale) return gg_fail(-9, "gg_gemm: in_scale only applies to the conv gather");
    }
    if (d->ldb % 8) return gg_fail(-10, "gg_gemm: ldb must be a multiple of 8 (got %d)", d->ldb);
    {
        int need = d->b_layout == GG_ROWK ? d->K : d->N;
        if (d->ldb < need) return gg_fail(-10, "gg_gemm: ldb %d < extent %d", d->ldb, need);
    }
    if (d->b_batch_stride % 8) return gg_fail(-10, "gg_gemm: B batch stride must be a multiple of 8");
    if (d->ldc < d->N) return gg_fail(-11, "gg_gemm: ldc %d < N %d", d->ldc, d->N);
    if (d->out_scale && d->rows_per_group <= 0) return gg_fail(-12, "gg_gemm: out_scale needs rows_per_group > 0");
    if ((d->noise != nullptr) != (d->noise_w != nullptr)) return gg_fail(-12, "gg_gemm: noise and noise_w go together");
    if (d->act < 0 || d->act > 3) return gg_fail(-13, "gg_gemm: unknown activation %d", d->act);
    return 0;
}

GemmPlan gg_plan_gemm(const gg_gemm_desc* d) {
    GemmPlan pl;
    pl.tile = d->force_tile;
    if (pl.tile < 1 || pl.tile > 3) pl.tile = d->N <= 32 ? 3 : (d->N <= 64 ? 2 : 1);
    pl.bm = 128;
    pl.bn = pl.tile == 1 ? 128 : (pl.tile == 2 ? 64 : 32);
    long long tm = (d->M + pl.bm - 1) / pl.bm, tn = (d->N + pl.bn - 1) / pl.bn;
    pl.blocks_mn = tm * tn;
    long long blocks = pl.blocks_mn * d->batch;
    int ktiles = (d->K + GG_BK - 1) / GG_BK;
    int sk = d->force_splitk;
    if (sk <= 0) {
        sk = 1;
        // fill 256 CUs x 2 workgroups when the M x N grid is small and the reduction is long
        if (blocks < 384 && ktiles >= 16) {
            long long want = (512 + blocks - 1) / blocks;
            long long cap = ktiles / 8;  // keep >= 8 k-tiles (256 reduction elements) per split
            if (want > cap) want = cap;
            if (want > 256) want = 256;
            if (want > 1) sk = (int)want;
        }
    }
    if (sk > ktiles) sk = ktiles;
    int tiles_per = (ktiles + sk - 1) / sk;
    sk = (ktiles + tiles_per - 1) / tiles_per;
    pl.splitk = sk;
    pl.k_per_split = tiles_per * GG_BK;
    return pl;
}

template <int BM, int BN, int WM, int WN>
void gg_launch_gemm_tile(const GgGemmParams& p, bool akrow, bool bkrow, bool aconv, dim3 grid, hipStream_t s) {
    dim3 block(256);
    // the plain epilogue (alpha only) is a separate instantiation: short-K launches (attention, K = 64) would
    // otherwise spend most of their time in the bias / scale / noise / activation branches
    const bool full = p.bias || p.out_scale || p.noise || p.act != GG_ACT_NONE;
#define GG_CASE(AK, BK_, AC)                                                                     \
    if (akrow == AK && bkrow == BK_ && aconv == AC) {                                            \
        if (full) GG_LAUNCH((gg_gemm_kernel<BM, BN, WM, WN, AK, BK_, AC, true>), grid, block, s, p);   \
        else GG_LAUNCH((gg_gemm_kernel<BM, BN, WM, WN, AK, BK_, AC, false>), grid, block, s, p);       \
        return;                                                                                  \
    }
    GG_CASE(false, false, false)
    GG_CASE(false, true, false)
    GG_CASE(true, false, false)
    GG_CASE(true, true, false)
    GG_CASE(false, false, true)
    GG_CASE(false, true, true)
    GG_CASE(true, false, true)
    GG_CASE(true, true, true)
#undef GG_CASE
}

}  // namespace

extern "C" size_t gg_gemm_workspace_bytes(const gg_gemm_desc* d) {
    if (gg_validate_gemm(d) != 0) return 0;
    GemmPlan pl = gg_plan_gemm(d);
    if (pl.splitk <= 1) return 0;
    return (size_t)d->batch * pl.splitk * d->M * d->N * sizeof(float);
}

extern "C" int gg_gemm_plan(const gg_gemm_desc* d, int32_t* tile, int32_t* splitk) {
    int rc = gg_validate_gemm(d);
    if (rc) return rc;
    GemmPlan pl = gg_plan_gemm(d);
    if (tile) *tile = pl.tile;
    if (splitk) *splitk = pl.splitk;
    return 0;
}

extern "C" int gg_gemm_bf16(const gg_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = gg_validate_gemm(d);
    if (rc) return rc;
    GemmPlan pl = gg_plan_gemm(d);
    size_t need = pl.splitk > 1 ? (size_t)d->batch * pl.splitk * d->M * d->N * sizeof(float) : 0;
    if (need > workspace_bytes || (need && !workspace))
        return gg_fail(-20, "gg_gemm: workspace too small (%zu < %zu)", workspace_bytes, need);
    if ((long long)d->batch * pl.splitk > 65535) return gg_fail(-21, "gg_gemm: batch*splitk exceeds grid.z");

    GgGemmParams p;
    memset(&p, 0, sizeof(p));
    p.M = d->M; p.N = d->N; p.K = d->K; p.batch = d->batch;
    p.splitk = pl.splitk; p.k_per_split = pl.k_per_split;
    p.A = (const bf16_t*)d->A; p.a_bs = d->a_batch_stride; p.lda = d->lda;
    p.B = (const bf16_t*)d->B; p.b_bs = d->b_batch_stride; p.ldb = d->ldb;
    p.H = d->H; p.W = d->W; p.C = d->C; p.CV = d->CV; p.R = d->R; p.S = d->S; p.pad = (d->R - 1) / 2;
    p.in_scale = d->in_scale;
    p.w_shift = p.hw_shift = -1;
    if (d->a_conv && (d->W & (d->W - 1)) == 0 && ((d->H * d->W) & (d->H * d->W - 1)) == 0) {
        int ws = 0, hs = 0;
        while ((1 << ws) < d->W) ++ws;
        while ((1 << hs) < d->H * d->W) ++hs;
        p.w_shift = ws; p.hw_shift = hs;
    }
    p.Cout = d->C_out; p.c_bs = d->c_batch_stride; p.ldc = d->ldc; p.c_f32 = d->c_is_f32;
    p.alpha = d->alpha;
    p.bias = d->bias; p.out_scale = d->out_scale; p.rows_per_group = d->rows_per_group;
    p.noise = d->noise; p.noise_w = d->noise_w;
    p.act = d->act; p.act_slope = d->act_slope;
    p.partial = (float*)workspace;

    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)pl.blocks_mn, 1, (unsigned)(d->batch * pl.splitk));
    bool akrow = d->a_layout == GG_KROW, bkrow = d->b_layout == GG_KROW, aconv = d->a_conv != 0;
    if (pl.tile == 1) gg_launch_gemm_tile<128, 128, 2, 2>(p, akrow, bkrow, aconv, grid, s);
    else if (pl.tile == 2) gg_launch_gemm_tile<128, 64, 2, 2>(p, akrow, bkrow, aconv, grid, s);
    else gg_launch_gemm_tile<128, 32, 4, 1>(p, akrow, bkrow, aconv, grid, s);
    rc = gg_check_launch();
    if (rc) return rc;
    if (pl.splitk > 1) {
        long long total = (long long)d->M * d->N * d->batch;
        long long nb = (total + 255) / 256;
        if (nb > 4096) nb = 4096;
        GG_LAUNCH(gg_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), s, p);
        rc = gg_check_launch();
    }
    return rc;
}

// ---- element-wise / stencil kernels -------------------------------------------------------------------

static unsigned gg_grid_for(long long work_items) {
    long long nb = (work_items + 255) / 256;
    if (nb > 8192) nb = 8192;  // 256 CUs x 8 workgroups x 4: grid-stride the rest
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

extern "C" int gg_resample_nhwc_bf16(const void* in, void* out, int32_t n, int32_t ih, int32_t iw, int32_t oh,
                                     int32_t ow, int32_t c, int32_t ty, int32_t tx, const int32_t* iy0,
                                     const int32_t* ix0, const float* wy, const float* wx, void* stream) {
    if (!in || !out || !iy0 || !ix0 || !wy || !wx) return gg_fail(-1, "gg_resample: null pointer");
    if (n <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0 || c <= 0 || ty <= 0 || tx <= 0)
        return gg_fail(-2, "gg_resample: non-positive extent");
    if ((c % 8) == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15))
        return gg_fail(-3, "gg_resample: 16-byte alignment required when C %% 8 == 0");
    GgResampleParams p;
    p.in = (const bf16_t*)in; p.out = (bf16_t*)out;
    p.n = n; p.IH = ih; p.IW = iw; p.OH = oh; p.OW = ow; p.C = c; p.TY = ty; p.TX = tx;
    p.iy0 = iy0; p.ix0 = ix0; p.wy = wy; p.wx = wx;
    long long total = (long long)n * oh * ow * ((c + 7) / 8);
    GG_LAUNCH(gg_resample_kernel, dim3(gg_grid_for(total)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_adamw_flat_f32(float* p, const float* g, float* m, float* v, const uint8_t* flags, int64_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay,
                                 float bias_corr1, float bias_corr2_sqrt, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || !flags) return gg_fail(-1, "gg_adamw: null pointer");
    if (n <= 0 || (n % 256)) return gg_fail(-2, "gg_adamw: n must be a positive multiple of 256 (got %lld)", (long long)n);
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15)
        return gg_fail(-3, "gg_adamw: buffers must be 16-byte aligned");
    if (bias_corr1 <= 0.f || bias_corr2_sqrt <= 0.f) return gg_fail(-4, "gg_adamw: bias corrections must be positive");
    GgAdamWParams a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.flags = flags; a.n = n;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = bias_corr1; a.bc2_sqrt = bias_corr2_sqrt; a.grad_scale = grad_scale;
    GG_LAUNCH(gg_adamw_kernel, dim3(gg_grid_for(n / 4)), dim3(256), (hipStream_t)stream, a);
    return gg_check_launch();
}

extern "C" int gg_ema_flat_f32(float* ema, const float* p, int64_t n, float one_minus_beta, void* stream) {
    if (!ema || !p) return gg_fail(-1, "gg_ema: null pointer");
    if (n <= 0 || (n % 4)) return gg_fail(-2, "gg_ema: n must be a positive multiple of 4");
    if ((((uintptr_t)ema) | ((uintptr_t)p)) & 15) return gg_fail(-3, "gg_ema: buffers must be 16-byte aligned");
    GG_LAUNCH(gg_ema_kernel, dim3(gg_grid_for(n / 4)), dim3(256), (hipStream_t)stream, ema, p, (long long)n, one_minus_beta);
    return gg_check_launch();
}

static int gg_softmax_common(GgSoftmaxParams& p, int64_t rows, int32_t rows_per_batch, int32_t n_valid, int32_t ld,
                             float alpha) {
    if (rows <= 0 || rows_per_batch <= 0 || n_valid <= 0 || ld < n_valid) return gg_fail(-2, "gg_softmax: bad extents");
    if (ld % 4 || ld > 256 * GG_SM_MAXV) return gg_fail(-3, "gg_softmax: ld must be a multiple of 4 and <= %d", 256 * GG_SM_MAXV);
    if (rows % rows_per_batch) return gg_fail(-4, "gg_softmax: rows must be a multiple of rows_per_batch");
    p.rows = rows; p.rows_per_batch = rows_per_batch; p.n_valid = n_valid; p.ld = ld; p.alpha = alpha;
    return 0;
}

extern "C" int gg_softmax_fwd(const float* x, void* S, const float* bias, int64_t rows, int32_t rows_per_batch,
                              int32_t n_valid, int32_t ld, float alpha, void* stream) {
    if (!x || !S) return gg_fail(-1, "gg_softmax_fwd: null pointer");
    GgSoftmaxParams p;
    memset(&p, 0, sizeof(p));
    int rc = gg_softmax_common(p, rows, rows_per_batch, n_valid, ld, alpha);
    if (rc) return rc;
    p.x = x; p.out = (bf16_t*)S; p.bias = bias;
    long long waves = (rows + GG_SM_ROWS - 1) / GG_SM_ROWS;
    GG_LAUNCH(gg_softmax_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_softmax_bwd(const void* S, const void* dS, void* dx, float* dbias, int64_t rows,
                              int32_t rows_per_batch, int32_t n_valid, int32_t ld, float alpha, void* stream) {
    if (!S || !dS || !dx) return gg_fail(-1, "gg_softmax_bwd: null pointer");
    GgSoftmaxParams p;
    memset(&p, 0, sizeof(p));
    int rc = gg_softmax_common(p, rows, rows_per_batch, n_valid, ld, alpha);
    if (rc) return rc;
    if (dbias && (rows_per_batch % GG_SM_ROWS)) return gg_fail(-5, "gg_softmax_bwd: rows_per_batch must be a multiple of %d when dbias is requested", GG_SM_ROWS);
    p.S = (const bf16_t*)S; p.dS = (const bf16_t*)dS; p.out = (bf16_t*)dx; p.dbias = dbias;
    long long waves = (rows + GG_SM_ROWS - 1) / GG_SM_ROWS;
    GG_LAUNCH(gg_softmax_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

static long long gg_bias_act_blocks(int64_t rows, int32_t C) {
    long long nb = (rows * (long long)(C / 8) + 4095) / 4096;   // ~16 vectors per thread
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return nb;
}

extern "C" int32_t gg_bias_act_bwd_partials(int64_t rows, int32_t C) { return (int32_t)gg_bias_act_blocks(rows, C); }

extern "C" int gg_bias_act_bwd(const void* dy, const void* y, void* dz, float* db, int64_t rows, int32_t C,
                               float slope, void* stream) {
    if (!dy) return gg_fail(-1, "gg_bias_act_bwd: null dy");
    if ((y != nullptr) != (dz != nullptr)) return gg_fail(-1, "gg_bias_act_bwd: y and dz go together");
    if (!y && !db) return gg_fail(-1, "gg_bias_act_bwd: nothing to do");
    if (rows <= 0 || C <= 0 || (C % 8)) return gg_fail(-2, "gg_bias_act_bwd: need rows > 0 and C %% 8 == 0 (C=%d)", C);
    GgBiasActBwdParams p;
    p.dy = (const bf16_t*)dy; p.y = (const bf16_t*)y; p.dz = (bf16_t*)dz; p.db = db; p.rows = rows; p.C = C; p.slope = slope;
    long long nb = gg_bias_act_blocks(rows, C);
    GG_LAUNCH(gg_bias_act_bwd_kernel, dim3((unsigned)nb), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}
