// gg_api.hip — the C ABI of libgigagan_amd.so (declared in include/gigagan_amd.h).
// Argument validation, tile / split-K selection and kernel launches; no allocation, no synchronisation.
#include "gg_device.h"
#include "gg_gemm.h"
#include "gg_gemm2.h"
#include "gg_conv3.h"
#include "gg_lrconv.h"
#include "gg_wgrad9.h"
#include "gg_wgrads.h"
#include "gg_sfwd.h"
#include "gg_pgemm.h"
#include "gg_elementwise.h"
#include "gg_modconv.h"
#include "gg_attention.h"
#include "gg_attention2.h"
#include "gg_weights.h"
#include "gg_dconv.h"
#include "gg_modcoef.h"
#include "gg_modfwd.h"
#include "gg_aconv.h"
#include "gg_spair.h"
#include "gg_semlp.h"
#include "gg_comm.h"
#include "../../include/gigagan_amd.h"

#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <unordered_map>
#include <vector>
#include <string>

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

// ---- hipGraph repair ---------------------------------------------------------------------------------
// This ROCm runtime (HIP 7.0.51831, the one PyTorch 2.10+rocm7.0 carries) re-executes a captured hipMemsetAsync with a corrupted VALUE
// from the second replay on (tests/gpu_graph_memset_probe.py: the cleared bytes read 16 / 57 / 64 / 128 instead of 0). PyTorch clears
// the semaphores of its split reductions with exactly such a memset (ATen/native/cuda/Reduce.cuh:1301), so a captured `t.sum(0)` over
// >= 1024 columns returns garbage on every replay but the first. The captured graph is repaired before it is instantiated: behind every
// memset node goes a kernel node that writes the intended value, and every node that depended on the memset now also depends on it.
#if !defined(GG_HOST_EMULATION)
struct GgGraphFillParams { void* dst; unsigned value; unsigned elem; unsigned long long width, height, pitch; };

__global__ void gg_graph_fill_kernel(GgGraphFillParams p) {
    const unsigned long long n = p.width * p.height;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long r = i / p.width, c = i - r * p.width;
        char* row = (char*)p.dst + r * p.pitch;
        if (p.elem == 1) ((unsigned char*)row)[c] = (unsigned char)p.value;
        else if (p.elem == 2) ((unsigned short*)row)[c] = (unsigned short)p.value;
        else ((unsigned*)row)[c] = p.value;
    }
}
#endif

extern "C" int gg_graph_patch_memsets(void* hip_graph, int32_t* n_patched) {
    if (n_patched) *n_patched = 0;
#if defined(GG_HOST_EMULATION)
    (void)hip_graph;
    return 0;
#else
    if (!hip_graph) return gg_fail(-1, "gg_graph_patch_memsets: null graph");
    hipGraph_t g = (hipGraph_t)hip_graph;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(g, nullptr, &n);
    if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphGetNodes: %s", hipGetErrorString(e));
    if (!n) return 0;
    std::vector<hipGraphNode_t> nodes(n);
    e = hipGraphGetNodes(g, nodes.data(), &n);
    if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphGetNodes: %s", hipGetErrorString(e));
    int patched = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType ty;
        e = hipGraphNodeGetType(nodes[i], &ty);
        if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphNodeGetType: %s", hipGetErrorString(e));
        if (ty != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        e = hipGraphMemsetNodeGetParams(nodes[i], &mp);
        if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphMemsetNodeGetParams: %s", hipGetErrorString(e));
        if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4)
            return gg_fail(-3, "gg_graph_patch_memsets: memset node with element size %u", mp.elementSize);
        size_t nd = 0;
        e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nd);
        if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphNodeGetDependentNodes: %s", hipGetErrorString(e));
        std::vector<hipGraphNode_t> after(nd);
        if (nd) {
            e = hipGraphNodeGetDependentNodes(nodes[i], after.data(), &nd);
            if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphNodeGetDependentNodes: %s", hipGetErrorString(e));
        }
        GgGraphFillParams fp;
        fp.dst = mp.dst; fp.value = mp.value; fp.elem = mp.elementSize;
        fp.width = mp.width; fp.height = mp.height ? mp.height : 1; fp.pitch = mp.pitch;
        const unsigned long long total = fp.width * fp.height;
        if (!total) continue;
        void* args[] = {&fp};
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = (void*)gg_graph_fill_kernel;
        kp.blockDim = dim3(256);
        unsigned long long blocks = (total + 255) / 256;
        kp.gridDim = dim3((unsigned)(blocks > 1024 ? 1024 : blocks));
        kp.kernelParams = args;
        hipGraphNode_t fill;
        e = hipGraphAddKernelNode(&fill, g, &nodes[i], 1, &kp);
        if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphAddKernelNode: %s", hipGetErrorString(e));
        for (size_t j = 0; j < nd; ++j) {
            e = hipGraphAddDependencies(g, &fill, &after[j], 1);
            if (e != hipSuccess) return gg_fail(-2, "gg_graph_patch_memsets: hipGraphAddDependencies: %s", hipGetErrorString(e));
        }
        ++patched;
    }
    if (n_patched) *n_patched = patched;
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
        if (d->R != d->S) return gg_fail(-5, "gg_gemm: conv kernel must be square");
        if (d->conv_stride < 1 || d->conv_pad < 0) return gg_fail(-5, "gg_gemm: conv_stride must be >= 1 and conv_pad >= 0 (got %d, %d)", d->conv_stride, d->conv_pad);
        if (d->H + 2 * d->conv_pad < d->R || d->W + 2 * d->conv_pad < d->S) return gg_fail(-5, "gg_gemm: conv window larger than the padded image");
        if (d->C % 8 || d->CV % d->C) return gg_fail(-6, "gg_gemm: conv needs C %% 8 == 0 and CV %% C == 0 (C=%d CV=%d)", d->C, d->CV);
        if (d->batch != 1) return gg_fail(-7, "gg_gemm: conv gather takes batch == 1 (images are folded into M or K)");
        long long red = (long long)d->R * d->S * d->CV;
        const int oh = (d->H + 2 * d->conv_pad - d->R) / d->conv_stride + 1, ow = (d->W + 2 * d->conv_pad - d->S) / d->conv_stride + 1;
        if (d->a_layout == GG_ROWK) {
            if (d->K != red) return gg_fail(-8, "gg_gemm: conv fwd needs K == R*S*CV (%d vs %lld)", d->K, red);
            if (d->M % (oh * ow)) return gg_fail(-8, "gg_gemm: conv fwd needs M %% (OH*OW) == 0");
        } else {
            if (d->M != red) return gg_fail(-8, "gg_gemm: conv wgrad needs M == R*S*CV (%d vs %lld)", d->M, red);
            if (d->K % (oh * ow)) return gg_fail(-8, "gg_gemm: conv wgrad needs K %% (OH*OW) == 0");
        }
    } else {
        if (d->lda % 8) return gg_fail(-9, "gg_gemm: lda must be a multiple of 8 (got %d)", d->lda);
        int need = d->a_layout == GG_ROWK ? d->K : d->M;
        if (d->lda < need) return gg_fail(-9, "gg_gemm: lda %d < extent %d", d->lda, need);
        if (d->a_batch_stride % 8) return gg_fail(-9, "gg_gemm: A batch stride must be a multiple of 8");
        if (d->in_scale) return gg_fail(-9, "gg_gemm: in_scale only applies to the conv gather");
    }
    if (d->b_image_stride) {
        if (!d->a_conv || d->a_layout != GG_ROWK || d->b_layout != GG_ROWK || d->b_image_stride < 0 || (d->b_image_stride & 7))
            return gg_fail(-10, "gg_gemm: b_image_stride applies to the conv forward (row-major operands), multiple of 8");
        const int oh = (d->H + 2 * d->conv_pad - d->R) / d->conv_stride + 1, ow = (d->W + 2 * d->conv_pad - d->S) / d->conv_stride + 1;
        if ((oh * ow) % 128) return gg_fail(-10, "gg_gemm: b_image_stride needs OH*OW %% 128 == 0 (got %d)", oh * ow);
    }
    if (d->bank_mix) {      // per-image mix of a stacked bank: only the low-resolution kernel does it (one image per 256-pixel tile)
        const bool ok = d->a_conv && d->a_layout == GG_ROWK && d->b_layout == GG_ROWK && d->R == 3 && d->S == 3 && d->conv_stride == 1 &&
                        d->conv_pad == 1 && d->H == 16 && d->W == 16 && !(d->C & 31) && d->CV == 2 * d->C && d->K == 9 * d->CV && d->in_scale &&
                        d->batch == 1 && !d->d2s && !d->b_image_stride && d->M % 256 == 0 && d->force_tile == 0;
        if (!ok) return gg_fail(-16, "gg_gemm: bank_mix needs a 3x3 / stride 1 / pad 1 convolution of 16x16 images over two stacked banks with in_scale [img][C]");
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
    if (d->residual) {
        if (d->ldr < d->N) return gg_fail(-14, "gg_gemm: ldr %d < N %d", d->ldr, d->N);
        if (d->batch != 1) return gg_fail(-14, "gg_gemm: residual needs batch == 1");
    }
    if (d->gelu_mode) {
        if ((d->gelu_mode != 1 && d->gelu_mode != 2) || !d->gelu_aux || (d->ld_aux & 3) || d->ld_aux < d->N || d->c_is_f32 || d->d2s || (d->N & 3) ||
            (d->ldc & 3) || d->batch != 1 || (d->residual && (d->ldr & 3)))
            return gg_fail(-18, "gg_gemm: gelu_mode needs a bf16 [M][N] output with N, ldc, ld_aux multiples of 4 and gelu_aux");
    }
    if (d->keep_partials && (!d->c_is_f32 || d->bias || d->out_scale || d->noise || d->residual || d->act != GG_ACT_NONE || d->d2s || d->batch != 1))
        return gg_fail(-17, "gg_gemm: keep_partials needs a plain fp32 [M][N] output (alpha-only epilogue, batch 1)");
    if (d->d2s) {
        if (d->d2s < 1 || d->d2s_taps < 1 || d->d2s_taps > d->d2s || d->d2s_c <= 0 || (d->d2s_c & 3) || d->d2s_oh <= 0 || d->d2s_ow <= 0)
            return gg_fail(-15, "gg_gemm: bad depth-to-space geometry");
        if (d->N != d->d2s_taps * d->d2s_taps * d->d2s_c) return gg_fail(-15, "gg_gemm: d2s needs N == taps^2 * d2s_c");
        if (d->M % (d->d2s_oh * d->d2s_ow) || d->batch != 1) return gg_fail(-15, "gg_gemm: d2s needs M %% (oh*ow) == 0 and batch == 1");
    }
    return 0;
}

// byte extent of an operand as seen from its base pointer (whole batch)
static long long gg_a_bytes(const gg_gemm_desc* d) {
    if (d->a_conv) {
        const int oh = (d->H + 2 * d->conv_pad - d->R) / d->conv_stride + 1, ow = (d->W + 2 * d->conv_pad - d->S) / d->conv_stride + 1;
        const long long rows = d->a_layout == GG_ROWK ? d->M : d->K;          // output pixels
        const long long imgs = (rows + (long long)oh * ow - 1) / ((long long)oh * ow);
        return imgs * d->H * d->W * d->C * 2;
    }
    const long long rows = d->a_layout == GG_ROWK ? d->M : d->K;
    return ((long long)(d->batch - 1) * d->a_batch_stride + rows * d->lda) * 2;
}
static long long gg_b_bytes(const gg_gemm_desc* d) {
    const long long rows = d->b_layout == GG_ROWK ? d->N : d->K;
    long long e = (long long)(d->batch - 1) * d->b_batch_stride + rows * d->ldb;
    if (d->b_image_stride) {
        const int oh = (d->H + 2 * d->conv_pad - d->R) / d->conv_stride + 1, ow = (d->W + 2 * d->conv_pad - d->S) / d->conv_stride + 1;
        e += ((long long)d->M / ((long long)oh * ow) - 1) * d->b_image_stride;
    }
    return e * 2;
}

bool gg_v2_has_variant(const gg_gemm_desc* d);

bool gg_v2_eligible(const gg_gemm_desc* d) {
    if (d->a_conv && d->a_layout == GG_ROWK) {
        if ((d->CV & 63) || d->R * d->S > 32) return false;
        if (d->CV != d->C && (d->C & 63)) return false;     // a 64-wide k-tile must not wrap around the physical channels
    }
    // ROWK operands are read through 32-bit buffer offsets (gg_gemm2.h): keep a margin below 4 GiB
    if (gg_a_bytes(d) + (1ll << 24) >= (1ll << 32) || gg_b_bytes(d) >= (1ll << 32)) return false;
    return gg_v2_has_variant(d);
}

// Launch planning by a small cost model (all times in microseconds, constants fitted to gpu_gemm_bench.py runs on
// MI355X): a launch runs in ceil(workgroups / resident) rounds; a round of a tile costs its k-tiles times the
// measured per-k-tile time plus a fixed prologue/epilogue; split-K adds the fp32 partial round trip and a launch.
// What it encodes: a split or tile choice that leaves 1.1 rounds of workgroups costs as much as 2.0 rounds
// (the weight gradients' 288-block launches measured 2x slower than their 252-block siblings).
struct GgTileModel {
    int tile, bm, bn, bk;
    double us_per_ktile;   // one workgroup, one k-tile, chip fully occupied by such workgroups
    int resident;          // workgroups resident on the chip at once (256 CUs x blocks per CU)
    double fixed_us;
};

static const GgTileModel kTileModels[] = {
    {1, 128, 128, 32, 1.95, 1024, 1.5},   // 40 KB LDS: 4 workgroups per CU
    {2, 128, 64, 32, 1.75, 1280, 1.5},
    {3, 128, 32, 32, 2.00, 1536, 1.5},
    {4, 256, 256, 64, 2.25, 256, 3.0},
    {5, 256, 128, 64, 1.45, 256, 3.0},
    {6, 128, 128, 64, 1.50, 512, 2.5},    // 8 waves, 72 KB LDS: 2 workgroups per CU (small-M layers: no split-K needed)
};

static double gg_plan_cost(const gg_gemm_desc* d, const GgTileModel& tm, int sk, int* k_per_split) {
    const long long blocks = (long long)((d->M + tm.bm - 1) / tm.bm) * ((d->N + tm.bn - 1) / tm.bn) * d->batch;
    const int ktiles = (d->K + tm.bk - 1) / tm.bk;
    const int per = (ktiles + sk - 1) / sk;
    if (k_per_split) *k_per_split = per * tm.bk;
    const long long rounds = (blocks * sk + tm.resident - 1) / tm.resident;
    double t = (double)rounds * (per * tm.us_per_ktile + tm.fixed_us);
    if (sk > 1) t += 3.0 + (double)d->M * d->N * d->batch * 4.0 * (sk + 1) / 4.0e6;   // partials out + back at ~4 TB/s
    return t;
}

// the direct convolution (gg_dconv.h, plan tile 9): narrow 3x3 / stride 1 / pad 1 layers on large feature maps.
// GG_DCONV=0 disables it (A/B runs); force_tile 9 selects it wherever eligible, any other force_tile bypasses it.
static int gg_dconv_policy() { return 1; }     // (round 6: the GG_DCONV A/B switch is gone - settled since its round; tests reach the other kernels with force_tile)

static bool gg_dconv_eligible(const gg_gemm_desc* d) {
    if (!d->a_conv || d->a_layout != GG_ROWK || d->b_layout != GG_ROWK) return false;
    if (d->R != 3 || d->S != 3 || d->conv_stride != 1 || d->conv_pad != 1) return false;
    if (d->C != d->CV || !(d->C == 16 || d->C == 32 || d->C == 64)) return false;
    if (d->N > 64 || (d->N & 7) || d->batch != 1 || d->d2s || d->c_is_f32) return false;
    if ((d->W % GG_DC_TW) || (d->H % GG_DC_TH) || (d->ldc & 7) || d->K != 9 * d->C) return false;
    if (d->M % (d->H * d->W)) return false;
    if ((long long)d->M * d->C * 2 + (1ll << 24) >= (1ll << 32)) return false;      // 32-bit buffer offsets into the activation
    return true;
}

static int gg_img_pixels(const gg_gemm_desc* d) {
    const int oh = (d->H + 2 * d->conv_pad - d->R) / d->conv_stride + 1, ow = (d->W + 2 * d->conv_pad - d->S) / d->conv_stride + 1;
    return oh * ow;
}

static bool gg_use_dconv(const gg_gemm_desc* d) {
    if (d->b_image_stride) return false;
    if (!gg_dconv_eligible(d)) return false;
    if (d->force_tile == 9) return true;
    if (d->force_tile != 0 || d->force_splitk > 1) return false;
    // 64 -> 64 channels is MFMA/LDS-bound in this form (one 124 KB workgroup per CU): the implicit GEMM measured faster
    return gg_dconv_policy() != 0 && d->M >= 65536 && (d->C <= 32 || d->N <= 32);
}

template <int C, int TN>
static void gg_launch_dconv(const GgGemmParams& p, hipStream_t s) {
    using L = GgDconvLds<C, TN>;
    const int lds = (L::XO_ELEMS + L::W_ELEMS) * 2;
    int per_cu = (160 * 1024) / lds;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    const long long total = (long long)(p.M / (p.H * p.W)) * (p.W / GG_DC_TW) * (p.H / GG_DC_TH);
    long long blocks = 256LL * per_cu;
    if (blocks > total) blocks = total;
    const bool full = p.bias || p.out_scale || p.noise || p.residual || p.act != GG_ACT_NONE;
    if (full) GG_LAUNCH((gg_dconv_kernel<C, TN, true>), dim3((unsigned)blocks), dim3(256), s, p);
    else GG_LAUNCH((gg_dconv_kernel<C, TN, false>), dim3((unsigned)blocks), dim3(256), s, p);
}

static bool gg_conv3_eligible(const gg_gemm_desc* d);
static GemmPlan gg_conv3_plan(const gg_gemm_desc* d, int tile, int splitk);
static bool gg_wgrad9_eligible(const gg_gemm_desc* d);
static GemmPlan gg_wgrad9_plan(const gg_gemm_desc* d, int splitk);
static bool gg_lrconv_eligible(const gg_gemm_desc* d);
static GemmPlan gg_lrconv_plan(const gg_gemm_desc* d, int splitk);
static bool gg_wgrads_eligible(const gg_gemm_desc* d);
static GemmPlan gg_wgrads_plan(const gg_gemm_desc* d, int splitk);
static bool gg_sfwd_eligible(const gg_gemm_desc* d);
static GemmPlan gg_sfwd_plan(const gg_gemm_desc* d);

// ---- tuning cache: measured-best (tile, split-K) per exact geometry (gg_gemm_plan_table) ----------------------------------
struct GgPlanChoice { int tile, splitk; };
static std::unordered_map<std::string, GgPlanChoice> g_plan_table;

static std::string gg_plan_key(const int32_t* f) { return std::string((const char*)f, 18 * sizeof(int32_t)); }

static std::string gg_plan_key_of(const gg_gemm_desc* d) {
    const bool full = d->bias || d->out_scale || d->noise || d->residual || d->act != GG_ACT_NONE;
    int32_t f[18] = {d->M, d->N, d->K, d->batch, d->a_layout, d->b_layout, d->a_conv ? 1 : 0,
                     d->a_conv ? d->H : 0, d->a_conv ? d->W : 0, d->a_conv ? d->C : 0, d->a_conv ? d->CV : 0,
                     d->a_conv ? d->R : 0, d->a_conv ? d->conv_stride : 0, d->a_conv ? d->conv_pad : 0,
                     d->c_is_f32 ? 1 : 0, d->d2s, full ? 1 : 0, d->in_scale ? 1 : 0};
    return gg_plan_key(f);
}

static int gg_pgemm_policy();
static bool gg_pgemm_eligible(const gg_gemm_desc* d);
static GemmPlan gg_pgemm_plan(const gg_gemm_desc* d);
static bool gg_table_plan(const gg_gemm_desc* d, GemmPlan& pl) {
    if (g_plan_table.empty() || d->force_tile != 0 || d->force_splitk != 0 || d->b_image_stride || d->bank_mix) return false;
    auto it = g_plan_table.find(gg_plan_key_of(d));
    if (it == g_plan_table.end()) return false;
    const int tile = it->second.tile;
    if (tile == 9) {
        if (!gg_dconv_eligible(d)) return false;
        pl.tile = 9; pl.bm = GG_DC_TH * GG_DC_TW; pl.bn = d->N <= 32 ? 32 : 64;
        pl.splitk = 1; pl.k_per_split = d->K; pl.blocks_mn = d->M / pl.bm;
        return true;
    }
    if (tile == 7 || tile == 8 || tile == 12) {
        if (!gg_conv3_eligible(d)) return false;
        pl = gg_conv3_plan(d, tile, it->second.splitk);
        return true;
    }
    if (tile == 14) {
        if (!gg_sfwd_eligible(d)) return false;
        pl = gg_sfwd_plan(d);
        return true;
    }
    if (tile == 15) {
        if (!gg_pgemm_policy() || !gg_pgemm_eligible(d)) return false;
        pl = gg_pgemm_plan(d);
        return true;
    }
    if (tile == 13) {
        if (!gg_wgrads_eligible(d)) return false;
        pl = gg_wgrads_plan(d, it->second.splitk);
        return true;
    }
    if (tile == 10) {
        if (!gg_wgrad9_eligible(d)) return false;
        pl = gg_wgrad9_plan(d, it->second.splitk);
        return true;
    }
    if (tile == 11) {
        if (!gg_lrconv_eligible(d)) return false;
        pl = gg_lrconv_plan(d, it->second.splitk);
        return true;
    }
    if (tile < 1 || tile > 6) return false;
    if (tile >= 4 && !gg_v2_eligible(d)) return false;
    const GgTileModel& tm = kTileModels[tile - 1];
    const int ktiles = (d->K + tm.bk - 1) / tm.bk;
    int sk = it->second.splitk < 1 ? 1 : it->second.splitk;
    if (sk > ktiles) sk = ktiles;
    if ((long long)d->batch * sk > 65535) return false;
    const int per = (ktiles + sk - 1) / sk;
    pl.tile = tile; pl.bm = tm.bm; pl.bn = tm.bn;
    pl.splitk = (ktiles + per - 1) / per;
    pl.k_per_split = per * tm.bk;
    pl.blocks_mn = (long long)((d->M + tm.bm - 1) / tm.bm) * ((d->N + tm.bn - 1) / tm.bn);
    return true;
}

// the halo-staged 3x3 convolution (gg_conv3.h, plan tile 7): stride 1 / pad 1 on 64-channel-multiple inputs whose 256-pixel tiles are
// whole image rows or whole images; 7: 256 output channels per workgroup, 8: 128). It takes over every unsplit 256-row implicit-GEMM
// choice of the planner on eligible layers (measured +17-36 % per layer, profiles/r02_conv3_ab.log); GG_CONV3=0 disables that
// (A/B runs); force_tile 7 / 8 selects it wherever eligible.
static int gg_conv3_policy() { return 1; }     // (round 6: the GG_CONV3 A/B switch is gone - settled since its round; tests reach the other kernels with force_tile)

static bool gg_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static bool gg_conv3_eligible(const gg_gemm_desc* d) {
    if (!d->a_conv || d->a_layout != GG_ROWK || d->b_layout != GG_ROWK) return false;
    if (d->R != 3 || d->S != 3 || d->conv_stride != 1 || d->conv_pad != 1) return false;
    // CV = C, or N * C: a kernel bank stacked along the reduction (weights [co][tap][n][ci]); chunks of 64 never wrap around C
    if ((d->C & 63) || d->CV % d->C || d->K != 9 * d->CV || (d->ldb & 7)) return false;
    if (d->CV != d->C && !d->in_scale) return false;      // (an unscaled stacked bank would just double the work)
    if (d->batch != 1 || d->d2s) return false;
    if (!gg_pow2(d->H) || !gg_pow2(d->W) || d->W < 8 || d->W > 64 || d->H * d->W < 64) return false;
    if (d->M % (d->H * d->W)) return false;
    if (d->b_image_stride && ((d->H * d->W) & 255)) return false;        // per-image weights: a 256-pixel tile inside one image
    if (gg_a_bytes(d) >= (1ll << 32) || gg_b_bytes(d) >= (1ll << 32)) return false;
    // the tile's halo (PH + 2 rows of W + 2 slots for each of its TI images) must fit the LDS area reserved for it
    const int hw = d->H * d->W, ph = hw >= 256 ? 256 / d->W : d->H, ti = hw >= 256 ? 1 : 256 / hw;
    if (ti * (ph + 2) * (d->W + 2) > GG_C3_MAX_SLOTS || ti * (ph + 2) > GG_C3_MAX_ROWS) return false;
    return true;
}

// tile 7: 256 x 256, tile 8: 256 x 128 (one 64-column tile when N <= 64), tile 12: 256 x 64 (layers whose 128-column grid would leave
// CUs idle: the generator's per-image-weight convolutions at 32x32). splitk <= 0: chosen by the cost form of gg_plan_cost (rounds of 256 resident workgroups x
// taps of the slice + the fp32 partial round trip); per-tap times from the round-2 layer measurements (1200 / 1100 TFLOP/s).
static double gg_conv3_cost(const gg_gemm_desc* d, int tile, int sk, int* per_out) {
    const int bn = tile == 7 ? 256 : (tile == 12 ? 64 : 128);
    const long long blocks = (long long)((d->M + 255) / 256) * ((d->N + bn - 1) / bn);
    const int nchunks = d->CV / 64;
    const int per = (nchunks + sk - 1) / sk;
    if (per_out) *per_out = per;
    const long long rounds = (blocks * sk + 255) / 256;
    double t = (double)rounds * (per * 9 * (tile == 7 ? 1.8 : (tile == 12 ? 0.65 : 1.0)) + 4.0);
    if (sk > 1) t += 3.0 + (double)d->M * d->N * 4.0 * (sk + 1) / 4.0e6;
    return t;
}

static GemmPlan gg_conv3_plan(const gg_gemm_desc* d, int tile, int splitk) {
    GemmPlan pl;
    pl.tile = tile; pl.bm = 256; pl.bn = tile == 7 ? 256 : (tile == 12 ? 64 : 128);
    pl.blocks_mn = (long long)((d->M + 255) / 256) * ((d->N + pl.bn - 1) / pl.bn);
    const int nchunks = d->CV / 64;
    // SCALED: the scale values of (images of a tile) x (channels of a k-slice) live in LDS (GG_C3_SC_FLOATS)
    const int hw = d->H * d->W, ti = hw >= 256 ? 1 : 256 / hw;
    int min_sk = 1;
    if (d->in_scale) {
        const int max_per = GG_C3_SC_FLOATS / (ti * 64);
        min_sk = (nchunks + max_per - 1) / max_per;
    }
    int sk = splitk;
    if (sk <= 0) {
        double best = 1e30;
        sk = min_sk;
        const int max_sk = nchunks < 64 ? nchunks : 64;
        for (int c = min_sk; c <= max_sk; ++c) {
            int per;
            const double t = gg_conv3_cost(d, tile, c, &per);
            if ((nchunks + per - 1) / per != c) continue;          // split counts that leave empty slices
            if (t < best) { best = t; sk = c; }
        }
    }
    if (sk < min_sk) sk = min_sk;
    if (sk > nchunks) sk = nchunks;
    const int per = (nchunks + sk - 1) / sk;
    pl.splitk = (nchunks + per - 1) / per;
    pl.k_per_split = per * 64;            // CHANNELS per slice (each with its nine taps): what gg_conv3_kernel reads
    return pl;
}

// the nine-tap weight gradient (gg_wgrad9.h, plan tile 10): 3x3 / stride 1 / pad 1 on 32-channel-multiple inputs, 8..64-wide power-of-
// two images. It takes over the 8-wave implicit-GEMM weight gradient wherever the planner chose a 256-row tile on an eligible layer
// (measured +16-35 % per layer on the 256/512-channel layers, +1.9 % on the step: profiles/r02_wgrad9_ab.log); GG_WGRAD9=0 disables
// that (A/B runs); force_tile 10 selects it wherever eligible.
static int gg_wgrad9_policy() { return 1; }     // (round 6: the GG_WGRAD9 A/B switch is gone - settled since its round; tests reach the other kernels with force_tile)

static bool gg_wgrad9_eligible(const gg_gemm_desc* d) {
    if (!d->a_conv || d->a_layout != GG_KROW || d->b_layout != GG_KROW) return false;
    if (d->R != 3 || d->S != 3 || d->conv_stride != 1 || d->conv_pad != 1) return false;
    if (d->C != d->CV || (d->C & 31) || d->M != 9 * d->C || (d->N & 7) || (d->ldb & 7) || (d->ldc & 3)) return false;
    if (d->in_scale || d->b_image_stride || d->batch != 1 || d->d2s || !d->c_is_f32) return false;
    if (d->bias || d->out_scale || d->noise || d->residual || d->act != GG_ACT_NONE) return false;
    if (!gg_pow2(d->H) || !gg_pow2(d->W) || d->W < 8 || d->W > 64 || d->H * d->W < 64 || (d->K & 63)) return false;
    if (d->K % (d->H * d->W)) return false;
    if (gg_a_bytes(d) + (1ll << 24) >= (1ll << 32) || gg_b_bytes(d) >= (1ll << 32)) return false;
    return true;
}

// split-K for the nine-tap kernel: one 99 KB workgroup per CU; the same cost form as gg_plan_cost (rounds x k-tiles + partial traffic)
static GemmPlan gg_wgrad9_plan(const gg_gemm_desc* d, int splitk) {
    GemmPlan pl;
    pl.tile = 10; pl.bm = 288; pl.bn = 256;
    pl.blocks_mn = (long long)(d->C / 32) * ((d->N + 255) / 256);
    const int ktiles = d->K / 64;
    int sk = splitk;
    if (sk <= 0) {
        double best = 1e30;
        sk = 1;
        int max_sk = ktiles / 4;
        if (max_sk < 1) max_sk = 1;
        if (max_sk > 256) max_sk = 256;
        for (int c = 1; c <= max_sk; ++c) {
            const int per = (ktiles + c - 1) / c;
            if ((ktiles + per - 1) / per != c) continue;
            const long long rounds = (pl.blocks_mn * c + 255) / 256;
            double t = (double)rounds * (per * 1.9 + 3.0);
            if (c > 1) t += 3.0 + (double)d->M * d->N * 4.0 * (c + 1) / 4.0e6;
            if (t < best) { best = t; sk = c; }
        }
    }
    if (sk > ktiles) sk = ktiles;
    const int per = (ktiles + sk - 1) / sk;
    pl.splitk = (ktiles + per - 1) / per;
    pl.k_per_split = per * 64;
    return pl;
}

// the low-resolution 3x3 convolution (gg_lrconv.h, plan tile 11): 4x4 / 8x8 / 16x16 images, 32-channel-multiple inputs, a plain or
// stacked (CV = N * C, scaled) weight bank; workgroups of 256 pixels (whole images) x 64 output channels x a channel slice, every
// chunk of 32 channels carrying all nine taps. Chosen by force_tile 11 or a plan-table entry (tests/gpu_modconv_layers.py measures
// it against the halo-staged kernel and the implicit GEMM on the generator's low-resolution adaptive convolutions).
static int gg_lrconv_scale_floats(const gg_gemm_desc* d) { return d->W == 4 ? 4096 : 2048; }

static bool gg_lrconv_eligible(const gg_gemm_desc* d) {
    if (!d->a_conv || d->a_layout != GG_ROWK || d->b_layout != GG_ROWK) return false;
    if (d->R != 3 || d->S != 3 || d->conv_stride != 1 || d->conv_pad != 1) return false;
    if ((d->C & 31) || d->CV % d->C || d->K != 9 * d->CV || (d->ldb & 7)) return false;
    if (d->CV != d->C && !d->in_scale) return false;
    if (d->batch != 1 || d->d2s || d->b_image_stride) return false;
    if (d->H != d->W || !(d->W == 4 || d->W == 8 || d->W == 16)) return false;
    if (d->bank_mix && (d->W != 16 || d->CV != 2 * d->C || !d->in_scale)) return false;      // per-image mix: one image per tile, two banks
    if (d->M % (d->H * d->W)) return false;
    if (gg_a_bytes(d) >= (1ll << 32) || gg_b_bytes(d) >= (1ll << 32)) return false;
    return true;
}

static GemmPlan gg_lrconv_plan(const gg_gemm_desc* d, int splitk) {
    GemmPlan pl;
    pl.tile = 11; pl.bm = GG_LR_BM; pl.bn = GG_LR_BN;
    pl.blocks_mn = (long long)((d->M + GG_LR_BM - 1) / GG_LR_BM) * ((d->N + GG_LR_BN - 1) / GG_LR_BN);
    const int nchunks = (d->bank_mix ? d->C : d->CV) / GG_LR_KC;
    // the scale values of (images of a tile) x (channels of a slice) live in LDS
    const int hw = d->H * d->W, ti = GG_LR_BM / hw;
    int min_sk = 1;
    if (d->in_scale) {
        const int max_per = gg_lrconv_scale_floats(d) / (ti * GG_LR_KC);
        min_sk = (nchunks + max_per - 1) / max_per;
    }
    int sk = splitk;
    if (sk <= 0) {      // fill the chip once (256 workgroups), at most 16 slices (the vectorised finish)
        sk = (int)((256 + pl.blocks_mn - 1) / pl.blocks_mn);
        if (sk > 16) sk = 16;
        if (sk > 8 && sk < 16) sk = 8;
    }
    if (sk < min_sk) sk = min_sk;
    if (sk > nchunks) sk = nchunks;
    const int per = (nchunks + sk - 1) / sk;
    pl.splitk = (nchunks + per - 1) / per;
    pl.k_per_split = per * GG_LR_KC;       // CHANNELS per slice
    return pl;
}

// the streaming weight gradient of the narrow high-resolution layers (gg_wgrads.h, plan tile 13): takes over wherever the planner (table
// or cost model) would run a 3x3 / stride 1 / pad 1, 1x1, 2x2 / stride 2 or 1x1 / stride 2 weight gradient with <= 64 input and output
// channels over >= 64K pixels
// on the 4-wave kernel. GG_WGRADS=0 disables that (A/B runs); force_tile 13 selects it wherever eligible.
static int gg_wgrads_policy() { return 1; }     // (round 6: the GG_WGRADS A/B switch is gone - settled since its round; tests reach the other kernels with force_tile)

struct GgWsMode { int kh, kw, halo, rpr, cs, cstore, gmul, spx, depth; };

static bool gg_wgrads_shape(const gg_gemm_desc* d, GgWsMode* out) {
    if (!d->a_conv || d->a_layout != GG_KROW || d->b_layout != GG_KROW || d->R != d->S) return false;
    GgWsMode m = {0, 0, 0, 1, d->C, d->C, 1, 0, 0};
    if (d->R == 3 && d->conv_stride == 1 && d->conv_pad == 1) { m.kh = 3; m.kw = 3; m.halo = 1; }
    else if (d->R == 1 && d->conv_stride == 1 && d->conv_pad == 0) { m.kh = 1; m.kw = 1; }
    else if (d->R == 2 && d->conv_stride == 2 && d->conv_pad == 0) { m.kh = 2; m.kw = 1; m.rpr = 2; m.cs = m.cstore = 2 * d->C; }   // space-to-depth
    else if (d->R == 1 && d->conv_stride == 2 && d->conv_pad == 0) { m.kh = 1; m.kw = 1; m.cs = 2 * d->C; m.gmul = 2; }              // even rows, first C of 2C
    else return false;
    const int N = d->N, cs = m.cs;
    if (d->C != d->CV || d->M != d->R * d->S * d->C || (d->ldb & 7) || (d->ldc & 3)) return false;
    if (!((cs == 8 || cs == 16 || cs == 32 || cs == 64) && (N == 8 || N == 16 || N == 32 || N == 64))) return false;
    if (d->in_scale || d->b_image_stride || d->batch != 1 || d->d2s || !d->c_is_f32) return false;
    if (d->bias || d->out_scale || d->noise || d->residual || d->act != GG_ACT_NONE) return false;
    if (!gg_pow2(d->H) || !gg_pow2(d->W)) return false;
    const int OW = d->W / d->conv_stride, OH = d->H / d->conv_stride;       // ('same' 3x3, 1x1, or non-overlapping stride-2 windows)
    if (OW < 64 || OW > 256 || d->K % (OH * OW)) return false;
    if (gg_a_bytes(d) >= (1ll << 32) || gg_b_bytes(d) >= (1ll << 32)) return false;
    for (int spx : {256, 128}) {
        // a step is whole output rows of one image; every wave issues the same number of 1 KB transfers per step and operand (<= 8)
        const int xch = cs * m.rpr;              // x channels moved per output pixel
        if (OW > spx || OH * OW < spx || ((spx * xch) & 2047) || ((spx * N) & 2047) || spx * xch > 16384 || spx * N > 16384) continue;
        const int npw = spx * (xch + N) / 2048;         // DMA instructions per wave and step
        for (int depth : {3, 2}) {
            if (depth * npw > 48) continue;             // vmcnt is a 6-bit counter; gg_wait_vm_le carries literals up to 48
            if (gg_ws_geom(m.halo, m.rpr, OW, cs, N, spx, depth).bytes > GG_WS_LDS) continue;
            m.spx = spx; m.depth = depth;
            if (out) *out = m;
            return true;
        }
    }
    return false;
}

static bool gg_wgrads_eligible(const gg_gemm_desc* d) { return gg_wgrads_shape(d, nullptr); }

static GemmPlan gg_wgrads_plan(const gg_gemm_desc* d, int splitk) {
    GemmPlan pl;
    GgWsMode m;
    gg_wgrads_shape(d, &m);
    const int spx = m.spx;
    pl.tile = 13; pl.bm = d->M; pl.bn = d->N; pl.blocks_mn = 1;
    const int steps = d->K / spx;
    int sk = splitk > 0 ? splitk : 256;                 // one 152 KB workgroup per CU
    if (sk > steps) sk = steps;
    const int per = (steps + sk - 1) / sk;
    pl.splitk = (steps + per - 1) / per;
    pl.k_per_split = per * spx;
    return pl;
}

// the streaming forward / data-gradient convolution (gg_sfwd.h, plan tile 14): 3x3 / stride 1 / pad 1, <= 64 channels either side,
// 64..256-wide images, shared or per-image weights. Takes over from the direct convolution (tile 9) and from the 4-wave kernel
// (tiles 1-3) on eligible launches of >= 64K pixels. GG_SFWD=0 disables that (A/B runs); force_tile 14 selects it wherever eligible.
static int gg_sfwd_policy() { return 1; }     // (round 6: the GG_SFWD A/B switch is gone - settled since its round; tests reach the other kernels with force_tile)

static bool gg_sfwd_shape(const gg_gemm_desc* d, int* spx_out, int* depth_out) {
    if (!d->a_conv || d->a_layout != GG_ROWK || d->b_layout != GG_ROWK) return false;
    if (d->R != 3 || d->S != 3 || d->conv_stride != 1 || d->conv_pad != 1) return false;
    const int C = d->C, N = d->N;
    if (C != d->CV || d->K != 9 * C || !(C == 8 || C == 16 || C == 32 || C == 64) || N > 64 || (N & 7) || N < 8) return false;
    if (d->batch != 1 || d->d2s || d->c_is_f32 || d->bank_mix || (d->ldb & 7) || (d->ldc & 7) || (d->residual && d->ldr != d->ldc)) return false;
    if (d->out_scale || !(d->act == GG_ACT_NONE || d->act == GG_ACT_LRELU)) return false;      // (the epilogue is branch-free: alpha, bias, noise, leaky-relu, residual)
    if (!gg_pow2(d->H) || !gg_pow2(d->W) || d->W < 64 || d->W > 256 || d->W * C > 8192) return false;
    if (d->M % (d->H * d->W)) return false;
    if (gg_a_bytes(d) >= (1ll << 32)) return false;
    for (int spx : {C == 64 ? 128 : 256}) {             // (the kernel's pixel blocks per wave are tied to this choice)
        if (d->W > spx || d->H * d->W < spx) continue;
        const int per_step = spx * C * 2 / 1024;        // 1 KB transfers of the loader wave per step
        if (per_step < 1) continue;
        for (int depth : {3, 2}) {
            if (depth * per_step > 48) continue;
            if (gg_sf_geom(d->W, C, N, spx, depth).bytes > GG_WS_LDS) continue;
            if (spx_out) *spx_out = spx;
            if (depth_out) *depth_out = depth;
            return true;
        }
    }
    return false;
}

static bool gg_sfwd_eligible(const gg_gemm_desc* d) { return gg_sfwd_shape(d, nullptr, nullptr); }

static GemmPlan gg_sfwd_plan(const gg_gemm_desc* d) {
    GemmPlan pl;
    int spx = 256, depth = 2;
    gg_sfwd_shape(d, &spx, &depth);
    pl.tile = 14; pl.bm = spx; pl.bn = d->N; pl.splitk = 1;
    const int steps = d->M / spx;
    int wgs = steps < 256 ? steps : 256;                // one 152 KB workgroup per CU, a contiguous run of steps each
    const int per = (steps + wgs - 1) / wgs;
    pl.blocks_mn = (steps + per - 1) / per;
    pl.k_per_split = per * spx;                         // (pixels per workgroup)
    return pl;
}

static GemmPlan gg_sfwd_substitute(const gg_gemm_desc* d, const GemmPlan& pl) {
    if (!((pl.tile >= 1 && pl.tile <= 3) || pl.tile == 9) || d->force_tile != 0 || d->force_splitk != 0 || !gg_sfwd_policy()) return pl;
    if (pl.splitk != 1 || d->M < 65536 || !gg_sfwd_eligible(d)) return pl;
    // measured (profiles/r04_sfwd_ab.log): against the 4-wave kernel it wins everywhere (stem 8 -> 32: 285 -> 110 us, 64 -> 64: 239 -> 143 us);
    // against the direct convolution only when the launch carries a bias / noise / activation / residual epilogue (32 -> 32 @256x256
    // b=64: 343 -> 159 us; with the plain alpha epilogue gg_dconv runs 3.2-4.3 TB/s and stays)
    const bool full = d->bias || d->noise || d->residual || d->act != GG_ACT_NONE;
    if (pl.tile == 9 && !full) return pl;
    return gg_sfwd_plan(d);
}

// the persistent short-K contraction (gg_pgemm.h, plan tile 15): row-major A (dense, or a 1x1 / stride 1 convolution gather, which is
// the same addressing) times row-major B into a bf16 [M][N] output through the staged epilogue (alpha, bias, activation, residual,
// GELU aux modes). GG_PGEMM=0 disables the substitution (A/B runs); force_tile 15 selects it wherever eligible.
static int gg_pgemm_policy() { return 1; }     // (round 6: the GG_PGEMM A/B switch is gone - settled since its round; tests reach the other kernels with force_tile)

static bool gg_pgemm_eligible(const gg_gemm_desc* d) {
    if (d->a_layout != GG_ROWK || d->b_layout != GG_ROWK || d->batch != 1) return false;
    if (d->a_conv) {
        if (d->R != 1 || d->S != 1 || d->conv_stride != 1 || d->conv_pad != 0 || d->CV != d->C || d->K != d->C) return false;
        if (d->in_scale || d->b_image_stride) return false;
    }
    if (d->bank_mix || d->d2s || d->c_is_f32 || d->keep_partials || d->out_scale || d->noise) return false;
    if ((d->K & 63) || d->K < 64 || (d->N & 7) || (d->ldc & 7) || (d->ldb & 7)) return false;
    const int pitch = d->a_conv ? d->C : d->lda;
    if (pitch & 7) return false;
    if ((((uintptr_t)d->A) | ((uintptr_t)d->B) | ((uintptr_t)d->C_out)) & 15) return false;
    if (d->residual && d->gelu_mode == 2) return false;     // (one pre-loaded epilogue operand per tile)
    if (d->residual && ((d->ldr & 7) || (((uintptr_t)d->residual) & 15))) return false;
    if (d->gelu_mode && ((d->ld_aux & 7) || (((uintptr_t)d->gelu_aux) & 15))) return false;
    if (d->bias && (((uintptr_t)d->bias) & 15)) return false;
    if (gg_a_bytes(d) >= (1ll << 32) || gg_b_bytes(d) >= (1ll << 32)) return false;
    return true;
}

static GemmPlan gg_pgemm_plan(const gg_gemm_desc* d) {
    GemmPlan pl;
    pl.tile = 15; pl.bm = 128; pl.bn = 128; pl.splitk = 1; pl.k_per_split = d->K;
    const long long tiles = (long long)((d->M + 127) / 128) * ((d->N + 127) / 128);
    long long wgs = 256;                                 // one 150 KB workgroup per CU, every 256th tile each
    if (const char* e = getenv("GG_PGEMM_WGS")) wgs = atoi(e) > 0 ? atoi(e) : wgs;     // (tests: runs of several tiles on small problems)
    pl.blocks_mn = tiles < wgs ? tiles : wgs;
    return pl;
}

static GemmPlan gg_pgemm_substitute(const gg_gemm_desc* d, const GemmPlan& pl) {
    if (pl.tile < 4 || pl.tile > 6 || pl.splitk != 1 || d->force_tile != 0 || d->force_splitk != 0 || !gg_pgemm_policy()) return pl;
    if (d->M < 8192 || d->K > 1024 || !gg_pgemm_eligible(d)) return pl;
    // measured (profiles/r05_pgemm_probe_v2.log, r05_plan_sweep_pgemm.log): 1.07-1.69x on launches that carry a bias / residual / GELU
    // epilogue and on K <= 256, down to 8192 rows; the plain alpha-only launches of K >= 512 stay on the 256 x 256 tile (0.86-0.99x)
    const bool full = d->bias || d->residual || d->act != GG_ACT_NONE || d->gelu_mode;
    if (!full && d->K > 256) return pl;
    return gg_pgemm_plan(d);
}

static GemmPlan gg_wgrads_substitute(const gg_gemm_desc* d, const GemmPlan& pl) {
    if (pl.tile < 1 || pl.tile > 3 || d->force_tile != 0 || d->force_splitk != 0 || !gg_wgrads_policy()) return pl;
    if (d->K < 65536 || !gg_wgrads_eligible(d)) return pl;
    return gg_wgrads_plan(d, 0);
}

static GemmPlan gg_wgrad9_substitute(const gg_gemm_desc* d, const GemmPlan& pl) {
    if ((pl.tile != 4 && pl.tile != 5) || d->force_tile != 0 || d->force_splitk != 0 || !gg_wgrad9_policy() || !gg_wgrad9_eligible(d)) return pl;
    return gg_wgrad9_plan(d, 0);
}

// the planner (table or cost model) thinks in implicit-GEMM tiles; an unsplit 256-row choice on an eligible layer runs halo-staged.
// `modelled`: the plan came from the cost model (not from the measured table): the halo-staged kernel with its own split-K is then
// compared with it by modelled cost (small-M layers: the generator's 8x8 .. 32x32 adaptive convs at batch 32)
static GemmPlan gg_conv3_substitute(const gg_gemm_desc* d, const GemmPlan& pl, bool modelled) {
    if (d->force_tile != 0 || !gg_conv3_policy() || !gg_conv3_eligible(d)) return pl;
    if (!modelled) {        // a measured choice: only the unsplit 256-row tiles are known to lose against the halo-staged kernel
        if ((pl.tile == 4 || pl.tile == 5) && pl.splitk == 1 && !d->in_scale) return gg_conv3_plan(d, pl.tile == 4 ? 7 : 8, 1);
        return pl;
    }
    if (pl.tile < 1 || pl.tile > 6 || d->force_splitk != 0) return pl;
    GemmPlan best = pl;
    double best_t = gg_plan_cost(d, kTileModels[pl.tile - 1], pl.splitk, nullptr);
    for (int tile : {7, 8, 12}) {
        if (tile == 7 && d->N < 192) continue;
        if (tile == 12 && (d->N <= 64 || d->N > 256)) continue;      // (N <= 64: tile 8 already runs the 64-column kernel)
        const GemmPlan c = gg_conv3_plan(d, tile, 0);
        const double t = gg_conv3_cost(d, tile, c.splitk, nullptr);
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}

// a stacked, scaled bank on 4x4 images (the generator's first adaptive convolutions): the low-resolution kernel with the modulation on
// its halo store measured 30.7 us against 38.4 us for modulation pass + implicit GEMM + finish (profiles/r03_lowres_sweep.log); the
// halo-staged kernel does not take 4-wide images, the implicit GEMM would apply the scale once per tap
static bool gg_use_lrconv(const gg_gemm_desc* d) {
    return d->force_tile == 0 && d->a_conv && d->in_scale && d->W == 4 && d->CV != d->C && gg_lrconv_eligible(d);
}

GemmPlan gg_plan_gemm(const gg_gemm_desc* d) {
    GemmPlan pl;
    if (d->bank_mix) return gg_lrconv_plan(d, d->force_splitk);        // (validated: only the low-resolution kernel mixes banks)
    if ((d->force_tile == 7 || d->force_tile == 8 || d->force_tile == 12) && gg_conv3_eligible(d))
        return gg_conv3_plan(d, d->force_tile, d->force_splitk);
    if (d->force_tile == 10 && gg_wgrad9_eligible(d)) return gg_wgrad9_plan(d, d->force_splitk);
    if (d->force_tile == 11 && gg_lrconv_eligible(d)) return gg_lrconv_plan(d, d->force_splitk);
    if (d->force_tile == 13 && gg_wgrads_eligible(d)) return gg_wgrads_plan(d, d->force_splitk);
    if (d->force_tile == 14 && gg_sfwd_eligible(d)) return gg_sfwd_plan(d);
    if (d->force_tile == 15 && gg_pgemm_eligible(d)) return gg_pgemm_plan(d);
    // (a measured plan is final as far as tile 15 goes: the sweep timed it against the table's choice, tests/gpu_plan_sweep.py --tiles 15)
    if (gg_table_plan(d, pl)) return gg_sfwd_substitute(d, gg_wgrads_substitute(d, gg_wgrad9_substitute(d, gg_conv3_substitute(d, pl, false))));
    if (gg_use_lrconv(d)) return gg_lrconv_plan(d, d->force_splitk);
    if (gg_use_dconv(d)) {
        pl.tile = 9; pl.bm = GG_DC_TH * GG_DC_TW; pl.bn = d->N <= 32 ? 32 : 64;
        pl.splitk = 1; pl.k_per_split = d->K; pl.blocks_mn = d->M / pl.bm;
        return gg_sfwd_substitute(d, pl);
    }
    const bool v2ok = gg_v2_eligible(d) && d->N >= 96 && d->M >= 192;
    const int v1_tile = d->N <= 32 ? 3 : (d->N <= 64 ? 2 : 1);
    int forced = d->force_tile;
    if (forced < 0 || forced > 6) forced = 0;      // (9 = direct convolution: handled above when eligible)
    if (forced >= 4 && !gg_v2_eligible(d)) forced = 0;
    if (d->b_image_stride && forced && gg_img_pixels(d) % kTileModels[forced - 1].bm) forced = 0;
    double best = 1e30;
    pl.tile = v1_tile; pl.splitk = 1;
    for (const GgTileModel& tm : kTileModels) {
        if (forced) {
            if (tm.tile != forced) continue;
        } else {
            if (d->b_image_stride && gg_img_pixels(d) % tm.bm) continue;     // a row tile must stay inside one image
            if (tm.tile <= 3 && tm.tile != v1_tile) continue;
            if (tm.tile >= 4 && !v2ok) continue;
            if (tm.tile == 4 && d->N < 192) continue;
            // the small 8-wave tile is for row-major launches whose grid the 256-row tiles cannot fill; weight
            // gradients measured better on 256x256 + split-K at every size except the 4x4-resolution layers
            if (tm.tile == 6 && (d->M > 32768 || d->a_layout == GG_KROW)) continue;
        }
        const int ktiles = (d->K + tm.bk - 1) / tm.bk;
        int max_sk = ktiles / (tm.bk == 64 ? 4 : 8);   // keep >= 256 reduction elements per split
        if (max_sk < 1) max_sk = 1;
        // the 4-wave tiles may split much further: a narrow weight gradient (M*N of a few thousand, K = millions of
        // pixels) needs thousands of workgroups in flight to pull HBM bandwidth; its partials stay small
        const int sk_cap = tm.tile <= 3 ? 4096 : 256;
        if (max_sk > sk_cap) max_sk = sk_cap;
        if ((long long)d->batch * max_sk > 65535) max_sk = (int)(65535 / d->batch);
        int lo = 1, hi = max_sk;
        if (d->force_splitk > 0) { lo = hi = d->force_splitk < ktiles ? d->force_splitk : ktiles; }
        for (int sk = lo; sk <= hi; sk += (sk < 256 ? 1 : (sk < 1024 ? 64 : 256))) {
            const int per = (ktiles + sk - 1) / sk;
            if ((ktiles + per - 1) / per != sk && d->force_splitk <= 0) continue;   // split counts that leave empty slices
            double c = gg_plan_cost(d, tm, sk, nullptr);
            if (c < best) { best = c; pl.tile = tm.tile; pl.splitk = sk; }
        }
    }
    const GgTileModel& tm = kTileModels[pl.tile - 1];
    pl.bm = tm.bm; pl.bn = tm.bn;
    pl.blocks_mn = (long long)((d->M + tm.bm - 1) / tm.bm) * ((d->N + tm.bn - 1) / tm.bn);
    const int ktiles = (d->K + tm.bk - 1) / tm.bk;
    const int per = (ktiles + pl.splitk - 1) / pl.splitk;
    pl.splitk = (ktiles + per - 1) / per;
    pl.k_per_split = per * tm.bk;
    return gg_pgemm_substitute(d, gg_sfwd_substitute(d, gg_wgrads_substitute(d, gg_wgrad9_substitute(d, gg_conv3_substitute(d, pl, true)))));
}

template <int BM, int BN, int WM, int WN>
void gg_launch_gemm_tile(const GgGemmParams& p, bool akrow, bool bkrow, bool aconv, dim3 grid, hipStream_t s) {
    dim3 block(256);
    // the plain epilogue (alpha only) is a separate instantiation: short-K launches (attention, K = 64) would
    // otherwise spend most of their time in the bias / scale / noise / activation branches
    const bool full = p.bias || p.out_scale || p.noise || p.residual || p.act != GG_ACT_NONE;
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

// the 8-wave kernel is instantiated for the layout / epilogue combinations the step uses: row-major x row-major
// (conv forward / data gradient, linear) with either epilogue; reduction-major x reduction-major (weight gradients)
// and the two mixed dense layouts (attention transposes, depth-to-space data gradient) with the plain epilogue
bool gg_v2_has_variant(const gg_gemm_desc* d) {
    const bool akrow = d->a_layout == GG_KROW, bkrow = d->b_layout == GG_KROW, aconv = d->a_conv != 0;
    const bool full = d->bias || d->out_scale || d->noise || d->residual || d->act != GG_ACT_NONE;
    if (!akrow && !bkrow) return true;
    if (full) return false;
    if (akrow && bkrow) return true;
    return !aconv;
}

template <int BM, int BN, int WM, int WN>
void gg_launch_gemm2_tile(const GgGemmParams& p, bool akrow, bool bkrow, bool aconv, dim3 grid, hipStream_t s) {
    dim3 block(GG2_NT);
    const bool full = p.bias || p.out_scale || p.noise || p.residual || p.act != GG_ACT_NONE;
#define GG_CASE(AK, BK_, AC, FE)                                                                  \
    if (akrow == AK && bkrow == BK_ && aconv == AC && full == FE) {                               \
        GG_LAUNCH((gg_gemm2_kernel<BM, BN, WM, WN, AK, BK_, AC, FE>), grid, block, s, p);          \
        return;                                                                                   \
    }
    GG_CASE(false, false, false, false)
    GG_CASE(false, false, false, true)
    GG_CASE(false, false, true, false)
    GG_CASE(false, false, true, true)
    GG_CASE(true, true, false, false)
    GG_CASE(true, true, true, false)
    GG_CASE(false, true, false, false)
    GG_CASE(true, false, false, false)
#undef GG_CASE
}

}  // namespace

extern "C" int gg_gemm_plan_table(const gg_plan_entry* entries, int32_t n) {
    if (n < 0 || (n > 0 && !entries)) return gg_fail(-1, "gg_gemm_plan_table: bad arguments");
    g_plan_table.clear();
    for (int i = 0; i < n; ++i) {
        const gg_plan_entry& e = entries[i];
        if (e.tile < 1 || e.tile > 15 || e.splitk < 1) return gg_fail(-2, "gg_gemm_plan_table: entry %d has tile %d split-K %d", i, e.tile, e.splitk);
        g_plan_table[gg_plan_key(&e.M)] = GgPlanChoice{e.tile, e.splitk};
    }
    return 0;
}

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
    p.H = d->H; p.W = d->W; p.C = d->C; p.CV = d->CV; p.R = d->R; p.S = d->S;
    p.pad = d->conv_pad; p.stride = d->conv_stride;
    p.in_scale = d->in_scale;
    p.w_shift = p.hw_shift = -1;
    if (d->a_conv) {
        p.OH = (d->H + 2 * d->conv_pad - d->R) / d->conv_stride + 1;
        p.OW = (d->W + 2 * d->conv_pad - d->S) / d->conv_stride + 1;
        if ((p.OW & (p.OW - 1)) == 0 && ((p.OH * p.OW) & (p.OH * p.OW - 1)) == 0) {
            int ws = 0, hs = 0;
            while ((1 << ws) < p.OW) ++ws;
            while ((1 << hs) < p.OH * p.OW) ++hs;
            p.w_shift = ws; p.hw_shift = hs;
        }
    }
    p.Cout = d->C_out; p.c_bs = d->c_batch_stride; p.ldc = d->ldc; p.c_f32 = d->c_is_f32;
    p.alpha = d->alpha;
    p.bias = d->bias; p.bias_scale = d->bias_scale; p.out_scale = d->out_scale; p.rows_per_group = d->rows_per_group;
    p.residual = (const bf16_t*)d->residual; p.ldr = d->ldr; p.res_scale = d->res_scale;
    p.d2s = d->d2s; p.d2s_t = d->d2s_taps; p.d2s_c = d->d2s_c; p.d2s_oh = d->d2s_oh; p.d2s_ow = d->d2s_ow;
    p.noise = d->noise; p.noise_w = d->noise_w;
    p.act = d->act; p.act_slope = d->act_slope;
    if (d->gelu_mode && !(((pl.tile >= 4 && pl.tile <= 6) || pl.tile == 15) && pl.splitk == 1))
        return gg_fail(-18, "gg_gemm: gelu_mode runs on the 8-wave tiles' staged epilogue only (planned tile %d, split-K %d)", pl.tile, pl.splitk);
    p.aux = (bf16_t*)d->gelu_aux; p.aux_mode = d->gelu_mode; p.ld_aux = d->ld_aux;
    p.partial = (float*)workspace;
    p.b_img_stride = d->b_image_stride;
    p.bank_mix = d->bank_mix;
    p.a_bytes = gg_a_bytes(d); p.b_bytes = gg_b_bytes(d);
    p.buf_ok = (p.a_bytes + (1ll << 24) < (1ll << 32) && p.b_bytes < (1ll << 32)) ? 31 : 0;
    p.krow_fast = d->a_conv && d->a_layout == GG_KROW && d->conv_stride == 1 && p.OH == d->H && p.OW == d->W && p.w_shift >= 0 &&
                  p.hw_shift >= 0 && !d->in_scale && (d->CV == d->C || !(d->C & 7));
#ifdef GG2_PROBE
    if (pl.tile > 3) p.xcd_slices = getenv("GG2_DBG") ? atoi(getenv("GG2_DBG")) : 0;   // probe builds: k-loop phase mask
#endif

    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)pl.blocks_mn, 1, (unsigned)(d->batch * pl.splitk));
    if (pl.tile <= 3 && pl.splitk > 1 && pl.blocks_mn <= 64) {
        p.xcd_slices = 1;
        const long long groups = ((long long)d->batch * pl.splitk + 7) / 8;
        grid = dim3((unsigned)(8 * pl.blocks_mn * groups), 1, 1);
    }
    bool akrow = d->a_layout == GG_KROW, bkrow = d->b_layout == GG_KROW, aconv = d->a_conv != 0;
    dim3 grid2((unsigned)(pl.blocks_mn * d->batch * pl.splitk), 1, 1);
    if (pl.tile == 9) {
        const bool wide = d->N > 32;
        if (d->C == 16) { if (wide) gg_launch_dconv<16, 2>(p, s); else gg_launch_dconv<16, 1>(p, s); }
        else if (d->C == 32) { if (wide) gg_launch_dconv<32, 2>(p, s); else gg_launch_dconv<32, 1>(p, s); }
        else { if (wide) gg_launch_dconv<64, 2>(p, s); else gg_launch_dconv<64, 1>(p, s); }
    }
    else if (pl.tile == 10) GG_LAUNCH(gg_wgrad9_kernel, grid2, dim3(GG2_NT), s, p);
    else if (pl.tile == 14) {
        gg_sfwd_shape(d, &p.ws_spx, &p.ws_depth);
        const int ck = d->C <= 16 ? 1 : d->C / 16;
        const bool nsplit = d->N > 32;
#define GG_SF(CK)                                                                                          \
        do {                                                                                               \
            if (nsplit) GG_LAUNCH((gg_sfwd_kernel<CK, true>), grid2, dim3(GG_SF_NT), s, p);                \
            else GG_LAUNCH((gg_sfwd_kernel<CK, false>), grid2, dim3(GG_SF_NT), s, p);                      \
        } while (0)
        if (ck == 1) GG_SF(1);
        else if (ck == 2) GG_SF(2);
        else GG_SF(4);
#undef GG_SF
    }
    else if (pl.tile == 15) {
        if (d->a_conv) p.lda = d->C;                // (a 1x1 / stride 1 gather reads pixel rows of C channels)
        {   // tile order, measured per shape on one box (profiles/r05_pgemm_probe_v8_orders.log): round-robin wins on narrow outputs (one
            // or two column tiles; four from K = 512 on) and on launches that read a second M x N operand (GELU data gradient); in the
            // step: this rule 411.5 / 411.6 img/s, round-robin everywhere 411.4, contiguous everywhere 409.7 (r05_pgemm_order_ab.log)
            const int tiles_n = (d->N + 127) / 128;
            p.pg_order = tiles_n <= 2 || (tiles_n <= 4 && d->K >= 512) || d->gelu_mode == 2;
        }
        // GG_PGEMM_ORDER=flip: the other tile order (same results bit for bit: tests/test_emulator_kernels.py runs both; the order probe)
        if (const char* e = getenv("GG_PGEMM_ORDER")) if (e[0] == 'f') p.pg_order = !p.pg_order;
#if defined(GG_PROBE)
        if (const char* e = getenv("GG_PGEMM_DBG")) p.xcd_slices = atoi(e);      // (probe builds: phases switched off, results are garbage)
#endif
        const bool full = p.bias || p.act != GG_ACT_NONE;
        if (full) GG_LAUNCH((gg_pgemm_kernel<true>), grid2, dim3(GG_PG_NT), s, p);
        else GG_LAUNCH((gg_pgemm_kernel<false>), grid2, dim3(GG_PG_NT), s, p);
    }
    else if (pl.tile == 13) {
        GgWsMode m;
        gg_wgrads_shape(d, &m);
        p.ws_spx = m.spx; p.ws_depth = m.depth; p.ws_cs = m.cs; p.ws_cstore = m.cstore; p.ws_gmul = m.gmul;
        if (m.kh == 3) GG_LAUNCH((gg_wgrads_kernel<3, 3>), grid2, dim3(GG_WS_NT), s, p);
        else if (m.kh == 2) GG_LAUNCH((gg_wgrads_kernel<2, 1>), grid2, dim3(GG_WS_NT), s, p);
        else GG_LAUNCH((gg_wgrads_kernel<1, 1>), grid2, dim3(GG_WS_NT), s, p);
    }
    else if (pl.tile == 11) {
        const bool full = pl.splitk == 1 && (p.bias || p.out_scale || p.noise || p.residual || p.act != GG_ACT_NONE);
        if (d->bank_mix) {
            if (full) GG_LAUNCH((gg_lrconv_kernel<35840, 2048, true, 2>), grid2, dim3(GG_LR_NT), s, p);
            else GG_LAUNCH((gg_lrconv_kernel<35840, 2048, false, 2>), grid2, dim3(GG_LR_NT), s, p);
        } else if (d->W == 4) {
            if (full) GG_LAUNCH((gg_lrconv_kernel<57344, 4096, true>), grid2, dim3(GG_LR_NT), s, p);
            else GG_LAUNCH((gg_lrconv_kernel<57344, 4096, false>), grid2, dim3(GG_LR_NT), s, p);
        } else {
            if (full) GG_LAUNCH((gg_lrconv_kernel<35840, 2048, true>), grid2, dim3(GG_LR_NT), s, p);
            else GG_LAUNCH((gg_lrconv_kernel<35840, 2048, false>), grid2, dim3(GG_LR_NT), s, p);
        }
    }
    else if (pl.tile == 7 || pl.tile == 8 || pl.tile == 12) {
        // (a split launch writes fp32 partials: its epilogue runs in the reduce pass, so it takes the plain instantiation)
        const bool full = pl.splitk == 1 && (p.bias || p.out_scale || p.noise || p.residual || p.act != GG_ACT_NONE);
        const bool scaled = p.in_scale != nullptr;
#define GG_C3(BN, WM, WN)                                                                                       \
        do {                                                                                                    \
            if (scaled) {                                                                                       \
                if (full) GG_LAUNCH((gg_conv3_kernel<BN, WM, WN, true, true>), grid2, dim3(GG2_NT), s, p);      \
                else GG_LAUNCH((gg_conv3_kernel<BN, WM, WN, false, true>), grid2, dim3(GG2_NT), s, p);          \
            } else {                                                                                            \
                if (full) GG_LAUNCH((gg_conv3_kernel<BN, WM, WN, true, false>), grid2, dim3(GG2_NT), s, p);     \
                else GG_LAUNCH((gg_conv3_kernel<BN, WM, WN, false, false>), grid2, dim3(GG2_NT), s, p);         \
            }                                                                                                   \
        } while (0)
        if (pl.tile == 7) GG_C3(256, 2, 4);
        else if ((pl.tile == 12 || d->N <= 64) && d->W >= 32 && scaled && pl.k_per_split <= GG_C3_SC_FLOATS_PAIR) {
            if (full) GG_LAUNCH((gg_conv3_kernel<GG_C3_BN64_PAIR, 8, 1, true, true>), grid2, dim3(GG2_NT), s, p);
            else GG_LAUNCH((gg_conv3_kernel<GG_C3_BN64_PAIR, 8, 1, false, true>), grid2, dim3(GG2_NT), s, p);
        }
        else if ((pl.tile == 12 || d->N <= 64) && !scaled && d->W >= 32) {
            // images of 32 / 64 pixels a side: two workgroups per CU (gg_conv3.h PAIR; the adaptive 64x64 layers on per-sample weights
            // 50.7 -> 35.2 us and 31.8 -> 26.4 us same-box, profiles/r06_conv3_pair_ab.log)
            if (full) GG_LAUNCH((gg_conv3_kernel<GG_C3_BN64_PAIR, 8, 1, true, false>), grid2, dim3(GG2_NT), s, p);
            else GG_LAUNCH((gg_conv3_kernel<GG_C3_BN64_PAIR, 8, 1, false, false>), grid2, dim3(GG2_NT), s, p);
        }
        else if (pl.tile == 12 || d->N <= 64) GG_C3(64, 8, 1);        // 64-column tiles: all eight waves along the pixels
        else GG_C3(128, 4, 2);
#undef GG_C3
    }
    else if (pl.tile == 4) gg_launch_gemm2_tile<256, 256, 2, 4>(p, akrow, bkrow, aconv, grid2, s);
    else if (pl.tile == 5) gg_launch_gemm2_tile<256, 128, 2, 4>(p, akrow, bkrow, aconv, grid2, s);
    else if (pl.tile == 6) gg_launch_gemm2_tile<128, 128, 2, 4>(p, akrow, bkrow, aconv, grid2, s);
    else if (pl.tile == 1) gg_launch_gemm_tile<128, 128, 2, 2>(p, akrow, bkrow, aconv, grid, s);
    else if (pl.tile == 2) gg_launch_gemm_tile<128, 64, 2, 2>(p, akrow, bkrow, aconv, grid, s);
    else gg_launch_gemm_tile<128, 32, 4, 1>(p, akrow, bkrow, aconv, grid, s);
    rc = gg_check_launch();
    if (rc) return rc;
    if (pl.splitk > 1 && !d->keep_partials) {
        long long total = (long long)d->M * d->N * d->batch;
        long long nb = pl.splitk <= 8 ? (total + 255) / 256 : (total + 63) / 64;
        if (nb > 8192) nb = 8192;
        // 2..8 slices of a 4-column-aligned plain [m][n] result: the vectorised finish (all slice loads in flight)
        const bool vec = (pl.splitk <= 8 || pl.splitk == 16) && !(d->N & 3) && !d->d2s && !(d->ldc & 3) && !(d->c_batch_stride & 3) &&
                         !(((uintptr_t)d->C_out) & 15) && !(((uintptr_t)workspace) & 15);
        if (vec) {
            long long nb4 = (total / 4 + 255) / 256;
            if (nb4 > 8192) nb4 = 8192;
            const dim3 g4((unsigned)nb4), blk(256);
            switch (pl.splitk) {
                case 2: GG_LAUNCH((gg_splitk_reduce4_kernel<2>), g4, blk, s, p); break;
                case 3: GG_LAUNCH((gg_splitk_reduce4_kernel<3>), g4, blk, s, p); break;
                case 4: GG_LAUNCH((gg_splitk_reduce4_kernel<4>), g4, blk, s, p); break;
                case 5: GG_LAUNCH((gg_splitk_reduce4_kernel<5>), g4, blk, s, p); break;
                case 6: GG_LAUNCH((gg_splitk_reduce4_kernel<6>), g4, blk, s, p); break;
                case 7: GG_LAUNCH((gg_splitk_reduce4_kernel<7>), g4, blk, s, p); break;
                case 8: GG_LAUNCH((gg_splitk_reduce4_kernel<8>), g4, blk, s, p); break;
                default: GG_LAUNCH((gg_splitk_reduce4_kernel<16>), g4, blk, s, p); break;
            }
        } else {
            GG_LAUNCH(gg_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), s, p);
        }
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
    const dim3 grid(gg_grid_for(total));
    hipStream_t s = (hipStream_t)stream;
    const char* blk_env = getenv("GG_RESAMPLE_2X2");
    if ((c % 8) == 0 && ty == tx && (ty == 2 || ty == 3) && oh >= ih && ow >= iw && !(oh & 1) && !(ow & 1) && (!blk_env || atoi(blk_env))) {
        // up-sampling / same-size filters: 2 x 2 output pixels per thread from one shared window
        const dim3 grid4(gg_grid_for(total / 4));
        if (ty == 2) GG_LAUNCH((gg_resample_taps2x2_kernel<2, 2>), grid4, dim3(256), s, p);
        else GG_LAUNCH((gg_resample_taps2x2_kernel<3, 3>), grid4, dim3(256), s, p);
    } else if (c == 3 && ty == tx && (ty == 1 || ty == 2 || ty == 3 || ty == 6) && (!blk_env || atoi(blk_env))) {
        // the rgb maps: a thread per output pixel, all taps loaded unconditionally
        const dim3 grid1(gg_grid_for((long long)n * oh * ow));
        if (ty == 1) GG_LAUNCH((gg_resample_taps_small_kernel<1, 1, 3>), grid1, dim3(256), s, p);
        else if (ty == 2) GG_LAUNCH((gg_resample_taps_small_kernel<2, 2, 3>), grid1, dim3(256), s, p);
        else if (ty == 3) GG_LAUNCH((gg_resample_taps_small_kernel<3, 3, 3>), grid1, dim3(256), s, p);
        else GG_LAUNCH((gg_resample_taps_small_kernel<6, 6, 3>), grid1, dim3(256), s, p);
    } else if ((c % 8) == 0 && ty == tx && (ty == 1 || ty == 2 || ty == 3 || ty == 6)) {
        if (ty == 1) GG_LAUNCH((gg_resample_taps_kernel<1, 1>), grid, dim3(256), s, p);
        else if (ty == 2) GG_LAUNCH((gg_resample_taps_kernel<2, 2>), grid, dim3(256), s, p);
        else if (ty == 3) GG_LAUNCH((gg_resample_taps_kernel<3, 3>), grid, dim3(256), s, p);
        else GG_LAUNCH((gg_resample_taps_kernel<6, 6>), grid, dim3(256), s, p);
    } else {
        GG_LAUNCH(gg_resample_kernel, grid, dim3(256), s, p);
    }
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

static_assert(sizeof(gg_pack_entry) == sizeof(GgPackEntry), "gg_pack_entry must mirror GgPackEntry");

extern "C" int gg_pack_weights(const gg_pack_entry* table, const int64_t* header, int32_t max_blocks, void* stream) {
    if (!table || !header) return gg_fail(-1, "gg_pack_weights: null pointer");
    if ((((uintptr_t)table) | ((uintptr_t)header)) & 7) return gg_fail(-3, "gg_pack_weights: table/header must be 8-byte aligned");
    if (max_blocks <= 0) max_blocks = 2048;
    GG_LAUNCH(gg_pack_weights_kernel, dim3((unsigned)max_blocks), dim3(256), (hipStream_t)stream,
              (const GgPackEntry*)table, (const long long*)header);
    return gg_check_launch();
}

extern "C" int gg_wgrad_finish(const float* g, float* dst, int32_t O, int32_t I, int32_t T, int32_t C8, int32_t O8,
                               float alpha, int32_t accumulate, void* stream) {
    if (!g || !dst) return gg_fail(-1, "gg_wgrad_finish: null pointer");
    if (O <= 0 || I <= 0 || T <= 0 || C8 < I || O8 < O) return gg_fail(-2, "gg_wgrad_finish: bad extents");
    GgWgradFinishParams p;
    p.g = g; p.dst = dst; p.O = O; p.I = I; p.T = T; p.C8 = C8; p.O8 = O8; p.accumulate = accumulate; p.alpha = alpha; p.nsplit = 1;
    GG_LAUNCH(gg_wgrad_finish_kernel, dim3((unsigned)((O + 31) / 32), (unsigned)((I + GG_WF_IB - 1) / GG_WF_IB)), dim3(256),
              (hipStream_t)stream, p);
    return gg_check_launch();
}

// workgroups that share one column block of a column-sum finish. Each issues ONE fp32 atomic add per channel, so with more than one the
// sum's last bit depends on their arrival order: that was the run-to-run noise of every bias gradient. ONE workgroup
// per column block folds all partial rows in a fixed order (eight rows in flight per wavefront): bit-reproducible bias gradients at
// +0.0 ... +0.2 ms per step (profiles/r05_colsum_det_ab.log; +0.8 ms before the loads were batched). GG_COLSUM_GROUPS=32 restores the
// spread form (16 partial rows per workgroup, at most 32 workgroups).
static int gg_colsum_groups(int P) {
    static int cap = -1;
    if (cap < 0) { const char* e = getenv("GG_COLSUM_GROUPS"); cap = e ? atoi(e) : 1; if (cap < 1) cap = 1; if (cap > 32) cap = 32; }
    int groups = (P + 15) / 16;
    if (groups > cap) groups = cap;
    return groups < 1 ? 1 : groups;
}

extern "C" int gg_colsum_finish(const float* part, float* dst, int32_t P, int32_t C, int32_t n, float alpha, void* stream) {
    if (!part || !dst) return gg_fail(-1, "gg_colsum_finish: null pointer");
    if (P <= 0 || C <= 0 || n <= 0 || n > C) return gg_fail(-2, "gg_colsum_finish: bad extents");
    int groups = gg_colsum_groups(P);
    GG_LAUNCH(gg_colsum_finish_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)groups), dim3(256), (hipStream_t)stream, part,
              dst, (int)P, (int)C, (int)n, alpha);
    return gg_check_launch();
}

extern "C" int gg_reduce_multi(const gg_reduce_item* items, int32_t n, void* stream) {
    if (!items || n <= 0) return gg_fail(-1, "gg_reduce_multi: no items");
    for (int base = 0; base < n; base += GG_FM_MAX) {
        GgReduceBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - base < GG_FM_MAX ? n - base : GG_FM_MAX;
        long long wgs = 0;
        for (int i = 0; i < b.n; ++i) {
            const gg_reduce_item& it = items[base + i];
            if (!it.src || it.n <= 0 || it.nsplit < 2) return gg_fail(-2, "gg_reduce_multi: item %d: need a slice stack of >= 2 slices", base + i);
            if (((uintptr_t)it.src) & 3) return gg_fail(-3, "gg_reduce_multi: item %d: unaligned", base + i);
            b.item[i].src = it.src; b.item[i].n = it.n; b.item[i].nsplit = it.nsplit;
            b.first_wg[i] = (int)wgs;
            wgs += (it.n + 63) / 64;
            if (wgs > 0x7fffffffll) return gg_fail(-2, "gg_reduce_multi: too many outputs in one batch");
        }
        b.first_wg[b.n] = (int)wgs;
        GG_LAUNCH(gg_reduce_multi_kernel, dim3((unsigned)wgs), dim3(256), (hipStream_t)stream, b);
        int rc = gg_check_launch();
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gg_finish_multi(const gg_finish_item* items, int32_t n, void* stream) {
    if (n < 0 || (n > 0 && !items)) return gg_fail(-1, "gg_finish_multi: bad arguments");
    for (int i0 = 0; i0 < n; i0 += GG_FM_MAX) {
        GgFinishBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - i0 < GG_FM_MAX ? n - i0 : GG_FM_MAX;
        int wgs = 0;
        for (int j = 0; j < b.n; ++j) {
            const gg_finish_item& it = items[i0 + j];
            if (!it.src || !it.dst) return gg_fail(-1, "gg_finish_multi: item %d: null pointer", i0 + j);
            GgFinishItem& o = b.item[j];
            o.src = it.src; o.dst = it.dst; o.kind = it.kind; o.alpha = it.alpha; o.accumulate = it.accumulate;
            o.nsplit = it.nsplit > 1 ? it.nsplit : 1;
            b.first_wg[j] = wgs;
            if (it.kind == 0) {
                if (it.O <= 0 || it.I <= 0 || it.T <= 0 || it.C8 < it.I || it.O8 < it.O) return gg_fail(-2, "gg_finish_multi: item %d: bad extents", i0 + j);
                o.O = it.O; o.I = it.I; o.T = it.T; o.C8 = it.C8; o.O8 = it.O8;
                wgs += ((it.O + 31) / 32) * ((it.I + GG_WF_IB - 1) / GG_WF_IB);
            } else if (it.kind == 1) {        // column sums: O = partial rows P, I = row pitch C, T = columns n
                if (it.O <= 0 || it.I <= 0 || it.T <= 0 || it.T > it.I) return gg_fail(-2, "gg_finish_multi: item %d: bad extents", i0 + j);
                const int groups = gg_colsum_groups(it.O);
                o.O = it.O; o.I = it.I; o.T = it.T; o.C8 = groups; o.O8 = 0;
                wgs += ((it.T + 63) / 64) * groups;
            } else if (it.kind == 2) {        // dst += alpha * src over O elements
                if (it.O <= 0) return gg_fail(-2, "gg_finish_multi: item %d: bad extents", i0 + j);
                o.O = it.O;
                wgs += (it.O + 1023) / 1024;
            } else return gg_fail(-3, "gg_finish_multi: item %d: unknown kind %d", i0 + j, it.kind);
        }
        b.first_wg[b.n] = wgs;
        GG_LAUNCH(gg_finish_multi_kernel, dim3((unsigned)wgs), dim3(256), (hipStream_t)stream, b);
        int rc = gg_check_launch();
        if (rc) return rc;
    }
    return 0;
}

static int gg_modcoef_common(GgModCoefParams& p, const float* w, const float* kmod, int32_t b, int32_t N, int32_t O, int32_t I,
                             int32_t T, int32_t Ip, int32_t Op, float eps) {
    if (!w) return gg_fail(-1, "gg_modcoef: null weights");
    if (b <= 0 || N <= 0 || O <= 0 || I <= 0 || T <= 0 || Ip < I || Op < O) return gg_fail(-2, "gg_modcoef: bad extents");
    if (N > GG_MC_NMAX || I > GG_MC_IMAX || O > GG_MC_IMAX)
        return gg_fail(-3, "gg_modcoef: supports N <= %d kernels and I, O <= %d channels", GG_MC_NMAX, GG_MC_IMAX);
    if (N > 1 && !kmod) return gg_fail(-1, "gg_modcoef: kernel_mod is required for N > 1");
    memset(&p, 0, sizeof(p));
    p.w = w; p.kmod = kmod; p.b = b; p.N = N; p.O = O; p.I = I; p.T = T; p.Ip = Ip; p.Op = Op; p.eps = eps;
    return 0;
}

extern "C" int gg_modcoef_fwd(const float* w, const float* mod, const float* kmod, float* s, float* a, float* d, int32_t b,
                              int32_t N, int32_t O, int32_t I, int32_t T, int32_t Ip, int32_t Op, float eps, void* stream) {
    GgModCoefParams p;
    int rc = gg_modcoef_common(p, w, kmod, b, N, O, I, T, Ip, Op, eps);
    if (rc) return rc;
    if (!mod || !s || !a) return gg_fail(-1, "gg_modcoef_fwd: null pointer");
    p.mod = mod; p.s = s; p.a = a; p.d = d;
    const unsigned grid = d ? (unsigned)O : (unsigned)(b < 256 ? b : 256);
    GG_LAUNCH(gg_modcoef_fwd_kernel, dim3(grid), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_modcoef_bwd(const float* w, const float* kmod, const float* s, const float* d, const float* gs,
                              const float* ga, const float* gd, float* gmod, float* gkmod, float* da_acc, float* gw,
                              int32_t b, int32_t N, int32_t O, int32_t I, int32_t T, int32_t Ip, int32_t Op, float eps,
                              void* stream) {
    GgModCoefParams p;
    int rc = gg_modcoef_common(p, w, kmod, b, N, O, I, T, Ip, Op, eps);
    if (rc) return rc;
    if (!s || !d || !gd || !gmod || !da_acc) return gg_fail(-1, "gg_modcoef_bwd: null pointer");
    if (N > 1 && !gkmod) return gg_fail(-1, "gg_modcoef_bwd: gkmod is required for N > 1");
    p.s = (float*)s; p.d = (float*)d; p.gs = gs; p.ga = ga; p.gd = gd; p.gmod = gmod; p.gkmod = gkmod; p.da_acc = da_acc; p.gw = gw;
    if (gw || N > 1) {
        GG_LAUNCH(gg_modcoef_bwd_w_kernel, dim3((unsigned)O), dim3(256), (hipStream_t)stream, p);
        rc = gg_check_launch();
        if (rc) return rc;
    }
    GG_LAUNCH(gg_modcoef_bwd_s_kernel, dim3((unsigned)I), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

// ---- the coefficients through the bank's Gram rows (gg_modcoef.h, second half) --------------------------------------------------------
static int gg_modgram_common(GgModGramParams& p, const float* w, const float* gram, const float* kmod, int32_t b, int32_t N, int32_t O,
                             int32_t I, int32_t T, int32_t Ip, int32_t Op, float eps) {
    if (!gram) return gg_fail(-1, "gg_modcoef_gram: null Gram rows");
    if (b <= 0 || N <= 0 || O <= 0 || I <= 0 || T <= 0 || Ip < I || Op < O) return gg_fail(-2, "gg_modcoef_gram: bad extents");
    if (N > GG_MC_NMAX || I > GG_MC_IMAX || O > GG_MC_IMAX || b > GG_MG_BMAX)
        return gg_fail(-3, "gg_modcoef_gram: supports N <= %d kernels, I, O <= %d channels, b <= %d samples", GG_MC_NMAX, GG_MC_IMAX, GG_MG_BMAX);
    if (N > 1 && !kmod) return gg_fail(-1, "gg_modcoef_gram: kernel_mod is required for N > 1");
    memset(&p, 0, sizeof(p));
    p.w = w; p.gram = (float*)gram; p.kmod = kmod; p.b = b; p.N = N; p.P = N * (N + 1) / 2; p.O = O; p.I = I; p.T = T; p.Ip = Ip; p.Op = Op;
    p.eps = eps;
    return 0;
}

extern "C" int gg_modgram(const float* w, float* gram, int32_t N, int32_t O, int32_t I, int32_t T, void* stream) {
    if (!w) return gg_fail(-1, "gg_modgram: null weights");
    GgModGramParams p;
    int rc = gg_modgram_common(p, w, gram, (const float*)w, 1, N, O, I, T, I, O, 0.f);
    if (rc) return rc;
    GG_LAUNCH(gg_modgram_kernel, dim3((unsigned)O), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_modcoef_gram_fwd(const float* gram, const float* mod, const float* kmod, float* s, float* a, float* d, float* tsum,
                                   int32_t b, int32_t N, int32_t O, int32_t I, int32_t Ip, int32_t Op, float eps, void* stream) {
    GgModGramParams p;
    int rc = gg_modgram_common(p, nullptr, gram, kmod, b, N, O, I, 1, Ip, Op, eps);
    if (rc) return rc;
    if (!mod || !s || !a || !d || !tsum) return gg_fail(-1, "gg_modcoef_gram_fwd: null pointer");
    p.mod = mod; p.s = s; p.a = a; p.d = d; p.tsum = tsum;
#define GG_MG_BY_N(KERNEL, GRID) \
    do { if (N == 1) GG_LAUNCH((KERNEL<1>), GRID, dim3(256), (hipStream_t)stream, p); else if (N == 2) GG_LAUNCH((KERNEL<2>), GRID, dim3(256), (hipStream_t)stream, p); \
         else if (N == 3) GG_LAUNCH((KERNEL<3>), GRID, dim3(256), (hipStream_t)stream, p); else GG_LAUNCH((KERNEL<4>), GRID, dim3(256), (hipStream_t)stream, p); } while (0)
    GG_MG_BY_N(gg_modcoef_gram_fwd_kernel, dim3((unsigned)O));
    return gg_check_launch();
}

extern "C" int gg_modcoef_gram_bwd(const float* w, const float* gram, const float* kmod, const float* s, const float* d,
                                   const float* tsum, const float* gs, const float* ga, const float* gd, float* gmod, float* gkmod,
                                   float* da_slots, float* gw, int32_t b, int32_t N, int32_t O, int32_t I, int32_t T, int32_t Ip,
                                   int32_t Op, float eps, void* stream) {
    GgModGramParams p;
    int rc = gg_modgram_common(p, w, gram, kmod, b, N, O, I, T, Ip, Op, eps);
    if (rc) return rc;
    if (!s || !d || !tsum || !gd || !gmod) return gg_fail(-1, "gg_modcoef_gram_bwd: null pointer");
    if (N > 1 && (!gkmod || !da_slots)) return gg_fail(-1, "gg_modcoef_gram_bwd: gkmod and da_slots are required for N > 1");
    if (gw && !w) return gg_fail(-1, "gg_modcoef_gram_bwd: the weights are required with gw");
    p.s = (float*)s; p.d = (float*)d; p.tsum = (float*)tsum; p.gs = gs; p.ga = ga; p.gd = gd; p.gmod = gmod; p.gkmod = gkmod;
    p.da_slots = da_slots; p.gw = gw;
    if (gw || N > 1) {
        GG_MG_BY_N(gg_modcoef_gram_bwd_o_kernel, dim3((unsigned)O));
        rc = gg_check_launch();
        if (rc) return rc;
    }
    GG_MG_BY_N(gg_modcoef_gram_bwd_i_kernel, dim3((unsigned)((I + 63) / 64), (unsigned)b));
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

extern "C" int gg_softmax_bwd2(const void* S, const void* dS, const void* g_dx, const float* g_dbias, void* g_S, void* g_dS,
                               int64_t rows, int32_t rows_per_batch, int32_t n_valid, int32_t ld, float alpha, void* stream) {
    if (!S || !dS || !g_S || !g_dS) return gg_fail(-1, "gg_softmax_bwd2: null pointer");
    if (!g_dx && !g_dbias) return gg_fail(-1, "gg_softmax_bwd2: no incoming gradient");
    if (rows <= 0 || rows_per_batch <= 0 || n_valid <= 0 || ld < n_valid) return gg_fail(-2, "gg_softmax_bwd2: bad extents");
    if (ld % 4 || ld > 256 * GG_SM_MAXV) return gg_fail(-3, "gg_softmax_bwd2: ld must be a multiple of 4 and <= %d", 256 * GG_SM_MAXV);
    if (rows % rows_per_batch) return gg_fail(-4, "gg_softmax_bwd2: rows must be a multiple of rows_per_batch");
    GgSoftmaxBwd2Params p;
    p.S = (const bf16_t*)S; p.dS = (const bf16_t*)dS; p.g_dx = (const bf16_t*)g_dx; p.g_dbias = g_dbias;
    p.g_S = (bf16_t*)g_S; p.g_dS = (bf16_t*)g_dS;
    p.rows = rows; p.rows_per_batch = rows_per_batch; p.n_valid = n_valid; p.ld = ld; p.alpha = alpha;
    long long waves = (rows + GG_SM_ROWS - 1) / GG_SM_ROWS;
    GG_LAUNCH(gg_softmax_bwd2_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_gelu(const void* x, const void* dy, const void* g, void* out0, void* out1, int64_t n, int32_t mode,
                       void* stream) {
    if (!x || !out0) return gg_fail(-1, "gg_gelu: null pointer");
    if (mode < 0 || mode > 2) return gg_fail(-2, "gg_gelu: mode must be 0 (forward), 1 (backward) or 2 (second order)");
    if ((mode >= 1 && !dy) || (mode == 2 && (!g || !out1))) return gg_fail(-1, "gg_gelu: missing operand for mode %d", mode);
    if (n <= 0 || n % 8) return gg_fail(-3, "gg_gelu: n must be a positive multiple of 8");
    GgGeluParams p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.g = (const bf16_t*)g;
    p.out0 = (bf16_t*)out0; p.out1 = (bf16_t*)out1; p.n8 = n / 8; p.mode = mode;
    long long nb = (p.n8 + 256 * 4 - 1) / (256 * 4);     // ~4 vectors per thread
    if (nb > 4096) nb = 4096;
    GG_LAUNCH(gg_gelu_kernel, dim3((unsigned)nb), dim3(256), (hipStream_t)stream, p);
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

// ---- adaptive-conv passes (gg_modconv.h) ------------------------------------------------------------------------

extern "C" int gg_modulate_fwd(const void* x, const float* s, void* out, int32_t b, int32_t P, int32_t C, void* stream) {
    if (!x || !s || !out) return gg_fail(-1, "gg_modulate_fwd: null pointer");
    if (b <= 0 || P <= 0 || C <= 0 || (C % 8)) return gg_fail(-2, "gg_modulate_fwd: need positive extents and C %% 8 == 0 (C=%d)", C);
    GgModulateParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.s = s; p.out = (bf16_t*)out; p.b = b; p.P = P; p.C = C; p.chunks = 1;
    GG_LAUNCH(gg_modulate_kernel, dim3(gg_grid_for((long long)b * P * (C / 8))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_modulate_bwd(const void* g, const void* x, const float* s, void* dx, float* ds_part, int32_t b, int32_t P,
                               int32_t C, int32_t chunks, void* stream) {
    if (!g || !x || !s || !dx || !ds_part) return gg_fail(-1, "gg_modulate_bwd: null pointer");
    if (b <= 0 || P <= 0 || C <= 0 || (C % 8) || chunks <= 0) return gg_fail(-2, "gg_modulate_bwd: bad extents (C=%d chunks=%d)", C, chunks);
    GgModulateParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.g = (const bf16_t*)g; p.s = s; p.out = (bf16_t*)dx; p.ds_part = ds_part;
    p.b = b; p.P = P; p.C = C; p.chunks = chunks;
    GG_LAUNCH(gg_modulate_bwd_kernel, dim3((unsigned)(b * chunks)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_modulate_bank_fwd(const void* x, const float* s, const float* a, void* out, int32_t b, int32_t P, int32_t Cin,
                                    int32_t Cout, void* stream) {
    if (!x || !s || !a || !out) return gg_fail(-1, "gg_modulate_bank_fwd: null pointer");
    if (b <= 0 || P <= 0 || Cin <= 0 || (Cin % 8) || Cout < Cin || (Cout % Cin))
        return gg_fail(-2, "gg_modulate_bank_fwd: need Cin %% 8 == 0 and Cout a multiple of Cin (Cin=%d Cout=%d)", Cin, Cout);
    GgModulateParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.s = s; p.a = a; p.out = (bf16_t*)out; p.b = b; p.P = P; p.C = Cout; p.Cin = Cin; p.chunks = 1;
    GG_LAUNCH(gg_modulate_kernel, dim3(gg_grid_for((long long)b * P * (Cout / 8))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

// ---- adaptive convolution on a shared, fragment-ordered bank (gg_aconv.h) --------------------------------------------------------
struct GgAconvPlan { int tm, nwn, nwk, lds, grid, mt; };

static int gg_log2i(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

// tile shape (TM x 32 pixels, NWN x 32 output channels, NWK K-slices): modelled time = rounds of 256 resident workgroups x
// max(matrix pipe, L2 -> register weight stream at ~45 B/clk/CU) + a fixed prologue; forced values (probes) override the choice
static int gg_aconv_plan_of(const gg_aconv_desc* d, GgAconvPlan* out) {
    if (!d) return gg_fail(-1, "gg_aconv: null descriptor");
    if (!d->x || !d->wf || !d->y || !d->s) return gg_fail(-1, "gg_aconv: null operand pointer");
    if (d->b <= 0 || d->H <= 0 || d->W <= 0 || !gg_pow2(d->H) || !gg_pow2(d->W) || d->H != d->W || d->W < 4 || d->W > 64)
        return gg_fail(-2, "gg_aconv: square power-of-two images of 4..64 pixels a side (got %d x %d)", d->H, d->W);
    if (d->C < 16 || d->C > 512 || !gg_pow2(d->C) || d->O < 32 || (d->O & 31))
        return gg_fail(-2, "gg_aconv: C a power of two in 16..512, O %% 32 == 0 (C=%d O=%d)", d->C, d->O);
    if (d->NB < 1 || d->NB > 2) return gg_fail(-2, "gg_aconv: banks of 1 or 2 kernels (got %d)", d->NB);
    if (d->NB > 1 && !d->a) return gg_fail(-2, "gg_aconv: a bank of %d kernels needs their weights a[b][N]", d->NB);
    if ((d->noise != nullptr) != (d->noise_w != nullptr)) return gg_fail(-2, "gg_aconv: noise and noise_w go together");
    if (d->act < 0 || d->act > 1) return gg_fail(-2, "gg_aconv: activation 0 (none) or 1 (leaky relu)");
    const int HW = d->H * d->W, KS = 9 * (d->C / 16);
    const long long M = (long long)d->b * HW;
    double best = 1e30;
    GgAconvPlan bp = {0, 0, 0, 0, 0, 0};
    for (int tm : {4, 2, 1}) {
        if (d->force_tm && tm != d->force_tm) continue;
        const int bmt = 32 * tm;
        int ti = 1, rt = d->H;
        if (HW >= bmt) { if (bmt % d->W) continue; rt = bmt / d->W; }
        else { if (bmt % HW) continue; ti = bmt / HW; }
        if (ti > 2 || ti * (rt + 2) * (d->W + 2) >= 900) continue;      // (the kernel keeps two images' scales; 16.16 slot reciprocals)
        const long long halo = (long long)ti * (rt + 2) * (d->W + 2) * (d->C * 2 + 16);
        const int mt = (int)((M + bmt - 1) / bmt);
        for (int nwn : {4, 2, 1}) {
            if (d->force_nwn && nwn != d->force_nwn) continue;
            if (d->O % (32 * nwn)) continue;
            const int nwk = 8 / nwn;
            if (d->NB * tm * 16 > 128) continue;                       // accumulator registers of a wavefront
            const long long red = nwk > 1 ? (long long)nwk * nwn * tm * 4096 : 0;      // every slice's quads, [wk][wn][4 tm][64 lanes] x 16 bytes
            const long long lds = halo > red ? halo : red;
            if (lds > 152 * 1024) continue;
            const long long grid = (long long)mt * (d->O / (32 * nwn));
            if (grid > 0x7fffffff) continue;
            const double mfma = 2.0 * d->NB * tm * ((KS + nwk - 1) / nwk) * 32.0;      // cycles: two wavefronts per SIMD
            const double l2 = (double)nwn * d->NB * KS * 1024.0 / 45.0;
            const double t = (double)((grid + 255) / 256) * ((mfma > l2 ? mfma : l2) + 3000.0 + (double)halo / 64.0);
            if (t < best) { best = t; bp.tm = tm; bp.nwn = nwn; bp.nwk = nwk; bp.lds = (int)((lds + 1023) & ~1023ll); bp.grid = (int)grid; bp.mt = mt; }
        }
    }
    if (!bp.tm) return gg_fail(-3, "gg_aconv: no tile shape fits %dx%d images with C=%d O=%d (forced TM %d NWN %d)", d->H, d->W, d->C, d->O,
                               d->force_tm, d->force_nwn);
    if (out) *out = bp;
    return 0;
}

extern "C" int gg_aconv_plan(const gg_aconv_desc* d, int32_t* tm, int32_t* nwn, int32_t* nwk, int32_t* lds_bytes, int32_t* grid) {
    GgAconvPlan pl;
    int rc = gg_aconv_plan_of(d, &pl);
    if (rc) return rc;
    if (tm) *tm = pl.tm;
    if (nwn) *nwn = pl.nwn;
    if (nwk) *nwk = pl.nwk;
    if (lds_bytes) *lds_bytes = pl.lds;
    if (grid) *grid = pl.grid;
    return 0;
}

extern "C" int gg_aconv_fwd(const gg_aconv_desc* d, void* stream) {
    GgAconvPlan pl;
    int rc = gg_aconv_plan_of(d, &pl);
    if (rc) return rc;
    GgAconvParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)d->x; p.wf = (const bf16_t*)d->wf; p.y = (bf16_t*)d->y;
    p.s = d->s; p.xs = d->xs; p.a = d->a; p.d = d->d; p.noise = d->noise; p.noise_w = d->noise_w;
    p.b = d->b; p.H = d->H; p.W = d->W; p.C = d->C; p.O = d->O;
    p.w_shift = gg_log2i(d->W); p.hw_shift = gg_log2i(d->H * d->W); p.c8_shift = gg_log2i(d->C / 8);
    p.act = d->act; p.slope = d->slope; p.mt = pl.mt;
#if defined(GG_PROBE)
    p.dbg = d->reserved;        // (probe builds: phases switched off)
#endif
    {
        const int bmt = 32 * pl.tm, hw = d->H * d->W;
        const int rt = hw >= bmt ? bmt / d->W : d->H;
        p.inv_spi = 65536 / ((rt + 2) * (d->W + 2)) + 1;
        p.inv_hwp = 65536 / (d->W + 2) + 1;
    }
    p.x_bytes = (long long)d->b * d->H * d->W * d->C * 2;
    p.wf_bytes = (long long)d->O * d->NB * 9 * d->C * 2;
    if (p.x_bytes >= (1ll << 32) || p.wf_bytes >= (1ll << 32)) return gg_fail(-4, "gg_aconv: operands beyond 4 GiB");
    if (d->next_wf) {       // the next launch's bank: which bytes each of ITS workgroups will stream (its own plan, asked of the same planner)
        gg_aconv_desc nd;
        memset(&nd, 0, sizeof(nd));
        nd.x = d->x; nd.wf = d->next_wf; nd.y = d->y; nd.s = d->s; nd.a = d->s;
        nd.b = d->next_b; nd.H = nd.W = d->next_H; nd.C = d->next_C; nd.O = d->next_O; nd.NB = d->next_NB;
        GgAconvPlan np;
        if (gg_aconv_plan_of(&nd, &np) == 0) {
            p.pf_wf = (const bf16_t*)d->next_wf;
            p.pf_bytes = (long long)nd.O * nd.NB * 9 * nd.C * 2;
            p.pf_tn_bytes = np.nwn * nd.NB * 9 * (nd.C / 16) * 1024;
            p.pf_mt = np.mt; p.pf_grid = np.grid;
            if (p.pf_bytes >= (1ll << 32)) p.pf_wf = nullptr;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)pl.grid), block(64 * pl.nwn * pl.nwk);
#define GG_AC(NB_, TM_, NWN_, NWK_) GG_LAUNCH_DYN((gg_aconv_kernel<NB_, TM_, NWN_, NWK_>), grid, block, pl.lds, s, p)
#define GG_AC_NB(NB_)                                                               \
    do {                                                                            \
        if (pl.tm == 1 && pl.nwn == 1) GG_AC(NB_, 1, 1, 8);                         \
        else if (pl.tm == 1 && pl.nwn == 2) GG_AC(NB_, 1, 2, 4);                    \
        else if (pl.tm == 1 && pl.nwn == 4) GG_AC(NB_, 1, 4, 2);                    \
        else if (pl.tm == 2 && pl.nwn == 1) GG_AC(NB_, 2, 1, 8);                    \
        else if (pl.tm == 2 && pl.nwn == 2) GG_AC(NB_, 2, 2, 4);                    \
        else if (pl.tm == 2 && pl.nwn == 4) GG_AC(NB_, 2, 4, 2);                    \
        else if (pl.tm == 4 && pl.nwn == 1) GG_AC(NB_, 4, 1, 8);                    \
        else if (pl.tm == 4 && pl.nwn == 2) GG_AC(NB_, 4, 2, 4);                    \
        else if (pl.tm == 4 && pl.nwn == 4) GG_AC(NB_, 4, 4, 2);                    \
        else return gg_fail(-3, "gg_aconv: no instantiation for TM %d NWN %d", pl.tm, pl.nwn);   \
    } while (0)
    if (d->NB == 1) GG_AC_NB(1);
    else GG_AC_NB(2);
#undef GG_AC_NB
#undef GG_AC
    return gg_check_launch();
}

static int gg_linattn_check(const char* who, int32_t C, int32_t ld_x, int32_t ld_y) {
    if (C <= 0 || (C & 63) || C > 2048 || ld_x < C || ld_y < C || (ld_x & 7) || (ld_y & 7))
        return gg_fail(-2, "%s: need C %% 64 == 0, C <= 2048, row pitches >= C and multiples of 8 (C=%d ld_x=%d ld_y=%d)", who, C, ld_x, ld_y);
    return 0;
}

extern "C" int gg_linattn_q_fwd(const void* q, int32_t ld_q, void* qs, int32_t ld_qs, int64_t rows, int32_t C, float scale, void* stream) {
    if (!q || !qs) return gg_fail(-1, "gg_linattn_q_fwd: null pointer");
    int rc = gg_linattn_check("gg_linattn_q_fwd", C, ld_q, ld_qs);
    if (rc) return rc;
    GgLinAttnParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)q; p.y = (bf16_t*)qs; p.ld_x = ld_q; p.ld_y = ld_qs; p.rows = rows; p.C = C; p.scale = scale;
    GG_LAUNCH(gg_linattn_q_kernel<0>, dim3(gg_grid_for(rows * (C >> 6) * 8)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_linattn_q_bwd(const void* qs, int32_t ld_qs, const void* dqs, int32_t ld_dqs, void* dq, int32_t ld_dq, int64_t rows,
                                int32_t C, float scale, void* stream) {
    if (!qs || !dqs || !dq) return gg_fail(-1, "gg_linattn_q_bwd: null pointer");
    int rc = gg_linattn_check("gg_linattn_q_bwd", C, ld_qs, ld_dq);
    if (rc) return rc;
    if (ld_dqs < C || (ld_dqs & 7) || scale == 0.f) return gg_fail(-2, "gg_linattn_q_bwd: bad gradient pitch / zero scale");
    GgLinAttnParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)qs; p.g = (const bf16_t*)dqs; p.y = (bf16_t*)dq; p.ld_x = ld_qs; p.ld_g = ld_dqs; p.ld_y = ld_dq;
    p.rows = rows; p.C = C; p.scale = scale;
    GG_LAUNCH(gg_linattn_q_kernel<1>, dim3(gg_grid_for(rows * (C >> 6) * 8)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int32_t gg_linattn_chunks(int32_t b, int32_t n) { return gg_pool_chunks(b, n); }

// x [b][n][ld_x] (+ g for the backward); part / stat are caller-owned fp32 scratch: part b * chunks * C * (2 | 1), stat b * C * (2 | 1)
static int gg_linattn_k(int mode, const void* x, int32_t ld_x, const void* g, int32_t ld_g, void* y, int32_t ld_y, float* part,
                        float* stat, int32_t b, int32_t n, int32_t C, void* stream) {
    if (!x || !y || !part || !stat || (mode == 1 && !g)) return gg_fail(-1, "gg_linattn_k: null pointer");
    int rc = gg_linattn_check("gg_linattn_k", C, ld_x, ld_y);
    if (rc) return rc;
    if (C > 512 || b <= 0 || n <= 0 || b > 65535) return gg_fail(-2, "gg_linattn_k: need C <= 512 and 0 < b <= 65535 (C=%d b=%d)", C, b);
    if (mode == 1 && (ld_g < C || (ld_g & 7))) return gg_fail(-2, "gg_linattn_k: bad gradient pitch");
    GgLinAttnParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.g = (const bf16_t*)g; p.y = (bf16_t*)y; p.part = part; p.stat = stat;
    p.ld_x = ld_x; p.ld_g = ld_g; p.ld_y = ld_y; p.C = C; p.b = b; p.n = n; p.chunks = gg_pool_chunks(b, n);
    hipStream_t s = (hipStream_t)stream;
    const dim3 g1((unsigned)p.chunks, (unsigned)b);
    const dim3 g2(gg_grid_for((long long)b * C)), g3(gg_grid_for((long long)b * n * (C >> 3)));
    if (mode == 0) {
        GG_LAUNCH(gg_linattn_kpart_kernel<0>, g1, dim3(256), s, p);
        GG_LAUNCH(gg_linattn_kfinish_kernel<0>, g2, dim3(256), s, p);
        GG_LAUNCH(gg_linattn_kapply_kernel<0>, g3, dim3(256), s, p);
    } else {
        GG_LAUNCH(gg_linattn_kpart_kernel<1>, g1, dim3(256), s, p);
        GG_LAUNCH(gg_linattn_kfinish_kernel<1>, g2, dim3(256), s, p);
        GG_LAUNCH(gg_linattn_kapply_kernel<1>, g3, dim3(256), s, p);
    }
    return gg_check_launch();
}

extern "C" int gg_linattn_k_fwd(const void* k, int32_t ld_k, void* eks, int32_t ld_eks, float* part, float* stat, int32_t b, int32_t n,
                                int32_t C, void* stream) {
    return gg_linattn_k(0, k, ld_k, nullptr, 0, eks, ld_eks, part, stat, b, n, C, stream);
}

extern "C" int gg_linattn_k_bwd(const void* eks, int32_t ld_eks, const void* deks, int32_t ld_deks, void* dk, int32_t ld_dk, float* part,
                                float* stat, int32_t b, int32_t n, int32_t C, void* stream) {
    return gg_linattn_k(1, eks, ld_eks, deks, ld_deks, dk, ld_dk, part, stat, b, n, C, stream);
}

extern "C" int gg_hinge(const void* x, void* dx, const float* gscale, float* loss, void* scratch, int64_t n, int64_t inner, int32_t nb,
                        int32_t split, int32_t mode, int32_t x_is_f32, void* stream) {
    if (!x || (!dx && !loss) || (dx && !gscale)) return gg_fail(-1, "gg_hinge: null pointer");
    if (n <= 0 || n >= (1ll << 31) || inner <= 0 || nb <= 0 || (mode != 0 && mode != 1) ||
        (mode == 1 && (split <= 0 || split >= nb || n % (inner * nb))))
        return gg_fail(-2, "gg_hinge: bad extents");
    GgHingeParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.dx = dx; p.gscale = gscale; p.loss = loss; p.n = n; p.inner = inner; p.nb = nb; p.split = split; p.mode = mode; p.x_f32 = x_is_f32;
    p.scratch = (unsigned*)scratch;
    long long blocks = (n + 2047) / 2048;                  // >= 8 elements per lane
    if (blocks > GG_HINGE_MAXB) blocks = GG_HINGE_MAXB;
    if (!dx && !scratch) blocks = 1;                       // forward without scratch: one workgroup
    GG_LAUNCH(gg_hinge_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_scaled_add(const void* a, const void* b, const void* d, void* y, int64_t n, float c, void* stream) {
    if (!a || !y) return gg_fail(-1, "gg_scaled_add: null pointer");
    if (n <= 0 || (n & 7)) return gg_fail(-2, "gg_scaled_add: n must be a positive multiple of 8");
    GgScaledAddParams p;
    memset(&p, 0, sizeof(p));
    p.a = (const bf16_t*)a; p.b = (const bf16_t*)b; p.d = (const bf16_t*)d; p.y = (bf16_t*)y; p.n = n; p.c = c;
    GG_LAUNCH(gg_scaled_add_kernel, dim3(gg_grid_for(n >> 3)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_addcat_fwd(const void* x, const void* feats, void* out, int32_t B, int32_t f, int64_t n, void* stream) {
    if (!x || !feats || !out) return gg_fail(-1, "gg_addcat_fwd: null pointer");
    if (B <= 0 || f <= 0 || (B % f) || n <= 0 || (n & 7)) return gg_fail(-2, "gg_addcat_fwd: need B %% f == 0 and n %% 8 == 0");
    GgAddCatParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.feats = (const bf16_t*)feats; p.out = (bf16_t*)out; p.n = n; p.B = B; p.f = f;
    GG_LAUNCH(gg_addcat_fwd_kernel, dim3(gg_grid_for((long long)B * (n >> 3))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_addcat_bwd(const void* g, void* gfeats, int32_t B, int32_t f, int64_t n, void* stream) {
    if (!g || !gfeats) return gg_fail(-1, "gg_addcat_bwd: null pointer");
    if (B <= 0 || f <= 0 || (B % f) || n <= 0 || (n & 7)) return gg_fail(-2, "gg_addcat_bwd: need B %% f == 0 and n %% 8 == 0");
    GgAddCatParams p;
    memset(&p, 0, sizeof(p));
    p.g = (const bf16_t*)g; p.out = (bf16_t*)gfeats; p.n = n; p.B = B; p.f = f;
    GG_LAUNCH(gg_addcat_bwd_kernel, dim3(gg_grid_for((long long)f * (n >> 3))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int32_t gg_pool_chunks(int32_t b, int32_t P) {
    // enough workgroups to fill the chip (>= 1024) without chunks shorter than 64 pixels
    int c = (1024 + b - 1) / (b > 0 ? b : 1);
    if (c > (P + 63) / 64) c = (P + 63) / 64;
    if (c < 1) c = 1;
    return c;
}

extern "C" int gg_pool_mean_fwd(const void* x, float* part, float* out, int32_t b, int32_t P, int32_t C, void* stream) {
    if (!x || !part || !out) return gg_fail(-1, "gg_pool_mean_fwd: null pointer");
    if (b <= 0 || P <= 0 || C <= 0 || (C % 8) || C > 512 || b > 65535)
        return gg_fail(-2, "gg_pool_mean_fwd: need C %% 8 == 0, C <= 512, b <= 65535 (b=%d C=%d)", b, C);
    GgPoolParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.part = part; p.out = out; p.b = b; p.P = P; p.C = C; p.chunks = gg_pool_chunks(b, P);
    p.scale = 1.f / (float)P;
    GG_LAUNCH(gg_pool_partial_kernel, dim3((unsigned)p.chunks, (unsigned)b), dim3(256), (hipStream_t)stream, p);
    int rc = gg_check_launch();
    if (rc) return rc;
    GG_LAUNCH(gg_pool_finish_kernel, dim3(gg_grid_for((long long)b * C)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_pool_mean_bwd(const void* g, const float* gs, void* y, int32_t b, int32_t P, int32_t C, void* stream) {
    if (!gs || !y) return gg_fail(-1, "gg_pool_mean_bwd: null pointer");
    if (b <= 0 || P <= 0 || C <= 0 || (C % 8)) return gg_fail(-2, "gg_pool_mean_bwd: need C %% 8 == 0 (C=%d)", C);
    GgPoolParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)g; p.gs = gs; p.y = (bf16_t*)y; p.b = b; p.P = P; p.C = C;
    GG_LAUNCH(gg_pool_bwd_kernel, dim3(gg_grid_for((long long)b * P * (C / 8))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

static int gg_se_mlp_check(const char* who, int32_t b, int32_t C, int32_t H, int32_t O) {
    if (b <= 0 || b > 65535 || C <= 0 || H <= 0 || O <= 0 || C > GG_SEMLP_MAX_C || O > GG_SEMLP_MAX_C || H > GG_SEMLP_MAX_H)
        return gg_fail(-2, "%s: need 0 < b <= 65535, C, O <= %d, H <= %d (b=%d C=%d H=%d O=%d)", who, GG_SEMLP_MAX_C, GG_SEMLP_MAX_H, b, C, H, O);
    return 0;
}

extern "C" int gg_se_mlp_fwd(const float* m, const float* w1, const float* b1, const float* w2, const float* b2, float* h, float* hs,
                             float* e, int32_t b, int32_t C, int32_t H, int32_t O, void* stream) {
    if (!m || !w1 || !w2 || !h || !hs || !e) return gg_fail(-1, "gg_se_mlp_fwd: null pointer");
    int rc = gg_se_mlp_check("gg_se_mlp_fwd", b, C, H, O);
    if (rc) return rc;
    GgSeMlpParams p;
    memset(&p, 0, sizeof(p));
    p.m = m; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.h = h; p.hs = hs; p.e = e; p.b = b; p.C = C; p.H = H; p.O = O;
    GG_LAUNCH(gg_se_mlp_fwd_kernel, dim3((unsigned)b), dim3(GG_SEMLP_THREADS), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_se_mlp_bwd(const float* de, const float* e, const float* h, const float* hs, const float* m, const float* w1,
                             const float* w2, float* dz2, float* dz1, float* dm, float* gw, int32_t b, int32_t C, int32_t H, int32_t O,
                             void* stream) {
    if (!de || !e || !h || !hs || !m || !w1 || !w2 || !dz2 || !dz1) return gg_fail(-1, "gg_se_mlp_bwd: null pointer");
    int rc = gg_se_mlp_check("gg_se_mlp_bwd", b, C, H, O);
    if (rc) return rc;
    GgSeMlpParams p;
    memset(&p, 0, sizeof(p));
    p.de = de; p.e = (float*)e; p.h = (float*)h; p.hs = (float*)hs; p.m = m; p.w1 = w1; p.w2 = w2; p.dz2 = dz2; p.dz1 = dz1; p.dm = dm;
    p.gw = gw; p.b = b; p.C = C; p.H = H; p.O = O;
    GG_LAUNCH(gg_se_mlp_bwd_rows_kernel, dim3((unsigned)b), dim3(GG_SEMLP_THREADS), (hipStream_t)stream, p);
    rc = gg_check_launch();
    if (rc || !gw) return rc;
    const long long n = (long long)H * C + H + (long long)O * H + O;
    GG_LAUNCH(gg_se_mlp_bwd_weights_kernel, dim3(gg_grid_for(n)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

static int gg_poolhf_check(const char* who, int32_t b, int32_t H, int32_t W, int32_t C) {
    if (b <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0 || (C & 7))
        return gg_fail(-2, "%s: needs even H, W >= 2 and C %% 8 == 0 (b=%d H=%d W=%d C=%d)", who, b, H, W, C);
    return 0;
}

extern "C" int gg_poolhf_fwd(const void* x, void* pool, void* hf, int32_t b, int32_t H, int32_t W, int32_t C, void* stream) {
    if (!x || !pool || !hf) return gg_fail(-1, "gg_poolhf_fwd: null pointer");
    int rc = gg_poolhf_check("gg_poolhf_fwd", b, H, W, C);
    if (rc) return rc;
    GgPoolHfParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.pool = (bf16_t*)pool; p.hf = (bf16_t*)hf; p.b = b; p.H = H; p.W = W; p.C = C;
    GG_LAUNCH(gg_poolhf_fwd_kernel, dim3(gg_grid_for((long long)b * (H / 2) * (W / 2) * (C / 8))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_poolhf_bwd(const void* x, const void* g_pool, const void* g_hf, void* dx, int32_t b, int32_t H, int32_t W, int32_t C,
                             void* stream) {
    if (!x || !dx || (!g_pool && !g_hf)) return gg_fail(-1, "gg_poolhf_bwd: null pointer");
    int rc = gg_poolhf_check("gg_poolhf_bwd", b, H, W, C);
    if (rc) return rc;
    GgPoolHfParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.g_pool = (const bf16_t*)g_pool; p.g_hf = (const bf16_t*)g_hf; p.dx = (bf16_t*)dx;
    p.b = b; p.H = H; p.W = W; p.C = C;
    GG_LAUNCH(gg_poolhf_bwd_kernel, dim3(gg_grid_for((long long)b * (H / 2) * (W / 2) * (C / 8))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

static int gg_modw_fill(GgModWParams& p, const float* w, const float* gram, const float* mod, int32_t mod_ld, const float* kmod,
                        int32_t kmod_ld, const float* xs, int32_t xs_ld, float* s, float* a, float* d, float* insc, void* wmix,
                        int32_t layout, int32_t b,
                        int32_t N, int32_t O, int32_t I, int32_t T, int32_t Ip, int32_t Op, int32_t demod, float eps) {
    if (!w || !mod) return gg_fail(-1, "gg_modw_fwd: null pointer");
    if (b <= 0 || b > GG_MW_BMAX || N <= 0 || N > GG_MW_NMAX || O <= 0 || I <= 0 || (I & 3) || T <= 0 || Ip < I || Op < O)
        return gg_fail(-2, "gg_modw_fwd: bad extents (b=%d N=%d O=%d I=%d T=%d)", b, N, O, I, T);
    if ((long long)N * I * T > GG_MW_WMAX || (long long)(N * (N + 1) / 2) * I > GG_MW_GMAX)
        return gg_fail(-3, "gg_modw_fwd: bank too large for one workgroup (N*I*T=%lld)", (long long)N * I * T);
    if (N > 1 && !kmod) return gg_fail(-1, "gg_modw_fwd: kernel_mod is required for N > 1");
    if (mod_ld < I || (kmod && kmod_ld < N) || (xs && xs_ld < I)) return gg_fail(-2, "gg_modw_fwd: row pitches smaller than the rows");
    if (wmix) {
        if (layout != 1 && layout != 2) return gg_fail(-4, "gg_modw_fwd: layout must be 1 or 2");
        if (layout == 2 && ((I & 15) || O > 32)) return gg_fail(-4, "gg_modw_fwd: layout 2 needs I %% 16 == 0 and O <= 32");
        if (((uintptr_t)wmix) & 15) return gg_fail(-4, "gg_modw_fwd: wmix must be 16-byte aligned");
    }
    memset(&p, 0, sizeof(p));
    p.w = w; p.gram = gram; p.mod = mod; p.kmod = kmod; p.s = s; p.a = a; p.d = d; p.insc = insc; p.wmix = (bf16_t*)wmix;
    p.layout = layout;
    p.b = b; p.N = N; p.O = O; p.I = I; p.T = T; p.Ip = Ip; p.Op = Op; p.demod = demod; p.eps = eps;
    p.mod_ld = mod_ld; p.kmod_ld = kmod_ld; p.xs = xs; p.xs_ld = xs_ld;
    // coefficient-only launches: one workgroup per channel handles every sample; with per-sample weights the samples are spread
    // over enough workgroups to fill the chip (a 16-channel layer would otherwise run on 16 CUs)
    int bc = b;
    if (wmix) {
        bc = (int)(((long long)b * O + 767) / 768);
        if (bc < 1) bc = 1;
        if (bc > b) bc = b;
    }
    p.bc = bc;
    return 0;
}

extern "C" int gg_modw_fwd(const float* w, const float* mod, int32_t mod_ld, const float* kmod, int32_t kmod_ld, const float* xs,
                           int32_t xs_ld, float* s, float* a, float* d, void* wmix, int32_t layout, int32_t b, int32_t N, int32_t O,
                           int32_t I, int32_t T, int32_t Ip, int32_t Op, int32_t demod, float eps, void* stream) {
    GgModWParams p;
    int rc = gg_modw_fill(p, w, nullptr, mod, mod_ld, kmod, kmod_ld, xs, xs_ld, s, a, d, nullptr, wmix, layout, b, N, O, I, T, Ip, Op,
                          demod, eps);
    if (rc) return rc;
    const dim3 grid((unsigned)O, (unsigned)((b + p.bc - 1) / p.bc));
    if (N == 1) GG_LAUNCH((gg_modw_kernel<1>), grid, dim3(256), (hipStream_t)stream, p);
    else if (N == 2) GG_LAUNCH((gg_modw_kernel<2>), grid, dim3(256), (hipStream_t)stream, p);
    else if (N == 3) GG_LAUNCH((gg_modw_kernel<3>), grid, dim3(256), (hipStream_t)stream, p);
    else GG_LAUNCH((gg_modw_kernel<4>), grid, dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_modw_multi_fwd(const gg_modw_item* items, int32_t n_items, void* stream) {
    if (!items || n_items <= 0) return gg_fail(-1, "gg_modw_multi_fwd: no items");
    GgModWMulti tab;
    int blocks = 0;
    auto flush = [&]() -> int {
        if (tab.n == 0) return 0;
        for (int j = tab.n; j <= GG_MW_MAX_ITEMS; ++j) tab.first_block[j] = blocks;
        GG_LAUNCH(gg_modw_multi_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, tab);
        tab.n = 0;
        blocks = 0;
        return gg_check_launch();
    };
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < n_items; ++i) {
        const gg_modw_item& it = items[i];
        GgModWParams q;
        int rc = gg_modw_fill(q, it.w, it.gram, it.mod, it.mod_ld, it.kmod, it.kmod_ld, it.xs, it.xs_ld, it.s, it.a, it.d, it.insc,
                              it.wmix, it.layout, it.b, it.N, it.O, it.I, it.T, it.Ip, it.Op, it.demod, it.eps);
        if (rc) return rc;
        if (it.N > 2) {         // rare: its own launch (the single-layer kernel)
            const dim3 grid((unsigned)it.O, (unsigned)((it.b + q.bc - 1) / q.bc));
            if (it.N == 3) GG_LAUNCH((gg_modw_kernel<3>), grid, dim3(256), (hipStream_t)stream, q);
            else GG_LAUNCH((gg_modw_kernel<4>), grid, dim3(256), (hipStream_t)stream, q);
            if ((rc = gg_check_launch())) return rc;
            continue;
        }
        // one launch carries hundreds of workgroups already: every workgroup takes all samples of its channel (the per-workgroup
        // prologue - bank rows, Gram rows, barriers - is the cost, not the samples); coefficient-only items with cached Gram rows
        // run a workgroup per channel pair without LDS (gg_modw_coef_body)
        q.bc = it.b;
        q.fast = (it.gram && !it.wmix && it.I <= 512) ? 1 : 0;
        if (tab.n == GG_MW_MAX_ITEMS && (rc = flush())) return rc;
        tab.item[tab.n] = q;
        tab.first_block[tab.n] = blocks;
        blocks += q.fast ? (it.O + 1) / 2 : it.O * ((it.b + q.bc - 1) / q.bc);
        ++tab.n;
    }
    return flush();
}

extern "C" int gg_sconv_fwd(const void* x, const void* w, int64_t w_bs, void* y, const float* noise, const float* noise_w,
                            const float* xs, int32_t b, int32_t H, int32_t W, int32_t C, int32_t O, int32_t act, float slope,
                            void* stream) {
    if (!x || !w || !y) return gg_fail(-1, "gg_sconv_fwd: null pointer");
    if (b <= 0 || H <= 0 || W <= 0 || (W & 31) || !(C == 16 || C == 32 || C == 64) || O <= 0 || O > 32 || (O & 7))
        return gg_fail(-2, "gg_sconv_fwd: needs W %% 32 == 0, C in {16, 32, 64}, O <= 32 and O %% 8 == 0 (W=%d C=%d O=%d)", W, C, O);
    if ((noise != nullptr) != (noise_w != nullptr)) return gg_fail(-1, "gg_sconv_fwd: noise and noise_w go together");
    if (act < 0 || act > 1) return gg_fail(-3, "gg_sconv_fwd: activation must be none (0) or leaky-relu (1)");
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)xs)) & 15) return gg_fail(-4, "gg_sconv_fwd: 16-byte alignment required");
    GgSconvParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.w_bs = w_bs; p.y = (bf16_t*)y; p.noise = noise; p.noise_w = noise_w; p.xs = xs;
    p.b = b; p.H = H; p.W = W; p.O = O; p.act = act; p.slope = slope;
    // work item of a wavefront: a 32-pixel-wide strip of `rows` rows (+ one halo row above and below: 2 / rows extra reads);
    // a workgroup = 4 wavefronts inside one image sharing that image's filter bank in LDS
    // strip height by measurement at batch 32 (profiles/r03_sconv_rows.log; 8 / 16 / 32 rows): 64 -> 32 @128x128 49.5 / 46.5 / 68.1 us,
    // 32 -> 32 @128x128 27.3 / 30.9 / 46.4, 32 -> 16 @256x256 58.4 / 56.6 / 51.9, 16 -> 16 @256x256 35.8 / 34.4 / 37.0
    const bool big = (long long)b * H * W >= (2 << 20);
    const int rows = C == 64 ? 16 : (C == 32 ? (big ? 32 : 8) : (big ? 16 : 8));
    const int items = (W >> 5) * ((H + rows - 1) / rows);
    int ipw = 1;
    while ((long long)b * ((items + 4 * ipw - 1) / (4 * ipw)) > 2048) ++ipw;
    p.rows_per_item = rows;
    p.items_per_wave = ipw;
    const long long blocks = (long long)b * ((items + 4 * ipw - 1) / (4 * ipw));
    if (C == 16) GG_LAUNCH((gg_sconv_kernel<16>), dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, p);
    else if (C == 32) GG_LAUNCH((gg_sconv_kernel<32>), dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, p);
    else GG_LAUNCH((gg_sconv_kernel<64>), dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

// ---- gg_spair_fwd: two adaptive 3x3 convolutions of one generator block in one launch (csrc/gg_spair.h) ----------------------------
template <int C0, int C1, int PT, int NCW, bool M16 = false>
static int gg_spair_launch(GgSpairParams& p, int32_t b, void* stream) {
    typedef GgSpGeom<C0, C1, PT, NCW, M16> G;
    const int lds = G::bytes(p.C2);
    if (lds > 160 * 1024) return gg_fail(-2, "gg_spair_fwd: C2 = %d needs %d bytes of LDS", p.C2, lds);
    // output rows per workgroup: one workgroup per CU in ONE round where the batch allows it (the halo - two conv1 rows and four x rows
    // per strip - is the price of a shorter strip)
    int rows = p.H;
    while (rows > 8 && (long long)b * ((p.H + rows - 1) / rows) < 256) rows >>= 1;
    p.rows = rows;
    p.strips = (p.H + rows - 1) / rows;
    GG_LAUNCH_DYN((gg_spair_kernel<C0, C1, PT, NCW, M16>), dim3((unsigned)(b * p.strips)), dim3(G::NT), lds, (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_spair_supported(int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t C2) {
    if (H <= 0 || C2 <= 0 || C2 > 32 || (C2 & 7)) return 0;
    if (W == 256 && C0 == 32 && C1 == 16) return GgSpGeom<32, 16, 1, 8>::bytes(C2) <= 160 * 1024;
    if (W == 128 && C0 == 64 && C1 == 32) return GgSpGeom<64, 32, 1, 4>::bytes(C2) <= 160 * 1024;
    return 0;
}

extern "C" int gg_spair_fwd(const void* x, const void* w1, int64_t w1_bs, const void* w2, int64_t w2_bs, void* y,
                            const float* noise1, const float* nw1, const float* noise2, const float* nw2, const float* xs,
                            int32_t b, int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t C2, int32_t act1, int32_t act2,
                            float slope, void* stream) {
    if (!x || !w1 || !w2 || !y) return gg_fail(-1, "gg_spair_fwd: null pointer");
    if (b <= 0 || !gg_spair_supported(H, W, C0, C1, C2))
        return gg_fail(-2, "gg_spair_fwd: supported geometries are (W 256, C0 32, C1 16) and (W 128, C0 64, C1 32) with C2 <= 32, C2 %% 8 == 0 "
                           "(b=%d H=%d W=%d C0=%d C1=%d C2=%d)", b, H, W, C0, C1, C2);
    if ((noise1 != nullptr) != (nw1 != nullptr) || (noise2 != nullptr) != (nw2 != nullptr))
        return gg_fail(-1, "gg_spair_fwd: a noise map and its weights go together");
    if (act1 < 0 || act1 > 1 || act2 < 0 || act2 > 1) return gg_fail(-3, "gg_spair_fwd: activation must be none (0) or leaky-relu (1)");
    if ((((uintptr_t)x) | ((uintptr_t)w1) | ((uintptr_t)w2) | ((uintptr_t)y) | ((uintptr_t)xs) | ((uintptr_t)noise1) | ((uintptr_t)noise2)) & 15)
        return gg_fail(-4, "gg_spair_fwd: 16-byte alignment required");
    GgSpairParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.y = (bf16_t*)y; p.w1_bs = w1_bs; p.w2_bs = w2_bs;
    p.noise1 = noise1; p.nw1 = nw1; p.noise2 = noise2; p.nw2 = nw2; p.xs = xs;
    p.b = b; p.H = H; p.C2 = C2; p.act1 = act1; p.act2 = act2; p.slope = slope;
    // 32 -> 16 -> <= 16 channels (the 256x256 block): 16 channels are a whole row block of v_mfma_f32_16x16x32_bf16 - half the matrix-pipe
    // time of the 32x32x16 form, which multiplies 16 zero weight rows (profiles/r06_spair_m16_ab.txt)
    if (C0 == 32 && C2 <= 16) return gg_spair_launch<32, 16, 1, 8, true>(p, b, stream);
    if (C0 == 32) return gg_spair_launch<32, 16, 1, 8>(p, b, stream);
    return gg_spair_launch<64, 32, 1, 4>(p, b, stream);
}

static int gg_modmix_common(GgModMixParams& p, int32_t b, int32_t P, int32_t O, int32_t Os, int32_t N, int32_t act, float slope) {
    if (b <= 0 || P <= 0 || O <= 0 || (O % 8) || Os < O || (Os % 8) || N < 1 || N > GG_MIX_MAXN)
        return gg_fail(-2, "gg_modmix: bad extents (O=%d Os=%d N=%d)", O, Os, N);
    if (act < 0 || act > 1) return gg_fail(-3, "gg_modmix: activation must be none (0) or leaky-relu (1)");
    p.b = b; p.P = P; p.O = O; p.Os = Os; p.N = N; p.act = act; p.slope = slope;
    return 0;
}

extern "C" int gg_modmix_fwd(const void* Y, const float* a, const float* d, const float* noise, const float* noise_w, void* y,
                             int32_t b, int32_t P, int32_t O, int32_t Os, int32_t N, int32_t act, float slope, void* stream) {
    if (!Y || !a || !y) return gg_fail(-1, "gg_modmix_fwd: null pointer");
    if ((noise != nullptr) != (noise_w != nullptr)) return gg_fail(-1, "gg_modmix_fwd: noise and noise_w go together");
    GgModMixParams p;
    memset(&p, 0, sizeof(p));
    int rc = gg_modmix_common(p, b, P, O, Os, N, act, slope);
    if (rc) return rc;
    p.Y = (const bf16_t*)Y; p.a = a; p.d = d; p.noise = noise; p.noise_w = noise_w; p.y = (bf16_t*)y; p.chunks = 1;
    GG_LAUNCH(gg_modmix_fwd_kernel, dim3(gg_grid_for((long long)b * P * (O / 8))), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_modmix_bwd(const void* dy, const void* y, const void* Y, const float* a, const float* d, const float* noise,
                             void* dY, float* da_part, float* dd_part, float* dnw_part, int32_t b, int32_t P, int32_t O,
                             int32_t Os, int32_t N, int32_t chunks, int32_t act, float slope, void* stream) {
    if (!dy || !Y || !a || !dY) return gg_fail(-1, "gg_modmix_bwd: null pointer");
    if (act == 1 && !y) return gg_fail(-1, "gg_modmix_bwd: the leaky-relu mask needs the forward output");
    if ((d != nullptr) != (dd_part != nullptr)) return gg_fail(-1, "gg_modmix_bwd: d and dd_part go together");
    if ((noise != nullptr) != (dnw_part != nullptr)) return gg_fail(-1, "gg_modmix_bwd: noise and dnw_part go together");
    if (chunks <= 0) return gg_fail(-2, "gg_modmix_bwd: chunks must be positive");
    GgModMixParams p;
    memset(&p, 0, sizeof(p));
    int rc = gg_modmix_common(p, b, P, O, Os, N, act, slope);
    if (rc) return rc;
    p.dy = (const bf16_t*)dy; p.y = (bf16_t*)y; p.Y = (const bf16_t*)Y; p.a = a; p.d = d; p.noise = noise;
    p.dY = (bf16_t*)dY; p.da_part = da_part; p.dd_part = dd_part; p.dnw_part = dnw_part; p.chunks = chunks;
    GG_LAUNCH(gg_modmix_bwd_kernel, dim3((unsigned)(b * chunks)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

// ---- fused self-attention (gg_attention.h) ------------------------------------------------------------------------

static int gg_attn_xcd() { return 1; }     // (round 6: the GG_ATTN_XCD A/B switch is gone: XCD-aware block order, profiles/r05_attn_xcd_ab.log)

static int gg_attn_common(GgAttnParams& p, const void* q, const void* k, const void* v, const void* k0, const void* v0,
                          int32_t B, int32_t n, int32_t h, float alpha, float beta) {
    if (!q || !k || !v || !k0 || !v0) return gg_fail(-1, "gg_attn: null pointer");
    if (B <= 0 || h <= 0 || n <= 0 || (n % 128)) return gg_fail(-2, "gg_attn: need n %% 128 == 0 tokens (got %d)", n);
    if ((long long)B * h > 65535) return gg_fail(-2, "gg_attn: B*h exceeds grid.y");
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)k0) | ((uintptr_t)v0)) & 15)
        return gg_fail(-3, "gg_attn: operands must be 16-byte aligned");
    memset(&p, 0, sizeof(p));
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.k0 = (const bf16_t*)k0; p.v0 = (const bf16_t*)v0;
    p.B = B; p.n = n; p.h = h; p.alpha = alpha; p.beta = beta; p.xcd = gg_attn_xcd();
    return 0;
}

extern "C" int gg_attn_fwd(const void* q, const void* k, const void* v, const void* k0, const void* v0, void* o, float* lse,
                           int32_t B, int32_t n, int32_t h, float alpha, float beta, void* stream) {
    GgAttnParams p;
    int rc = gg_attn_common(p, q, k, v, k0, v0, B, n, h, alpha, beta);
    if (rc) return rc;
    if (!o || !lse) return gg_fail(-1, "gg_attn_fwd: null output");
    p.o = (bf16_t*)o; p.lse = lse;
    GG_LAUNCH(gg_attn_fwd_kernel<false>, dim3((unsigned)(n / 128), (unsigned)(B * h)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_attn_bwd(const void* q, const void* k, const void* v, const void* k0, const void* v0, const void* o,
                           const float* lse, const void* d_o, float* dvec, void* dq, void* dk, void* dv, float* null_part,
                           int32_t B, int32_t n, int32_t h, float alpha, float beta, void* stream) {
    GgAttnParams p;
    int rc = gg_attn_common(p, q, k, v, k0, v0, B, n, h, alpha, beta);
    if (rc) return rc;
    if (!o || !lse || !d_o || !dvec || !dq || !dk || !dv || !null_part) return gg_fail(-1, "gg_attn_bwd: null pointer");
    p.o = (bf16_t*)o; p.lse = (float*)lse; p.d_o = (const bf16_t*)d_o; p.dvec = dvec;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.null_part = null_part;
    dim3 grid((unsigned)(n / 128), (unsigned)(B * h));
    GG_LAUNCH(gg_attn_bwd_dq_kernel<false>, grid, dim3(256), (hipStream_t)stream, p);
    rc = gg_check_launch();
    if (rc) return rc;
    GG_LAUNCH(gg_attn_bwd_dkv_kernel<false>, grid, dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

// general form: n queries against m keys, strided q / k / v rows, optional null token, optional per-key bias (gg_attention.h GEN)
static int gg_attn_gen_common(GgAttnParams& p, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                              const void* k0, const void* v0, const float* kbias, int32_t B, int32_t n, int32_t m, int32_t h,
                              float alpha, float beta) {
    if (!q || !k || !v || ((k0 == nullptr) != (v0 == nullptr))) return gg_fail(-1, "gg_attn_gen: null pointer");
    if (B <= 0 || h <= 0 || n <= 0 || m <= 0) return gg_fail(-2, "gg_attn_gen: B, h, n, m must be positive");
    if ((long long)B * h > 65535) return gg_fail(-2, "gg_attn_gen: B*h exceeds grid.y");
    if (ldq < h * 64 || ldk < h * 64 || ldv < h * 64 || ((ldq | ldk | ldv) & 7))
        return gg_fail(-2, "gg_attn_gen: row pitches must be >= h*64 and multiples of 8");
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)k0) | ((uintptr_t)v0)) & 15)
        return gg_fail(-3, "gg_attn_gen: operands must be 16-byte aligned");
    memset(&p, 0, sizeof(p));
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.k0 = (const bf16_t*)k0; p.v0 = (const bf16_t*)v0;
    p.B = B; p.n = n; p.m = m; p.h = h; p.alpha = alpha; p.beta = beta; p.xcd = gg_attn_xcd();
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.kbias = kbias; p.has_null = k0 != nullptr;
    return 0;
}

extern "C" int gg_attn_gen_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* k0,
                               const void* v0, const float* kbias, void* o, float* lse, int32_t B, int32_t n, int32_t m, int32_t h,
                               float alpha, float beta, void* stream) {
    GgAttnParams p;
    int rc = gg_attn_gen_common(p, q, ldq, k, ldk, v, ldv, k0, v0, kbias, B, n, m, h, alpha, beta);
    if (rc) return rc;
    if (!o || !lse) return gg_fail(-1, "gg_attn_gen_fwd: null output");
    p.o = (bf16_t*)o; p.lse = lse;
    GG_LAUNCH(gg_attn_fwd_kernel<true>, dim3((unsigned)((n + 127) / 128), (unsigned)(B * h)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

extern "C" int gg_attn_gen_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* k0,
                               const void* v0, const float* kbias, const void* o, const float* lse, const void* d_o, float* dvec,
                               void* dq, void* dk, void* dv, float* null_part, int32_t B, int32_t n, int32_t m, int32_t h,
                               float alpha, float beta, void* stream) {
    GgAttnParams p;
    int rc = gg_attn_gen_common(p, q, ldq, k, ldk, v, ldv, k0, v0, kbias, B, n, m, h, alpha, beta);
    if (rc) return rc;
    if (!o || !lse || !d_o || !dvec || !dq || !dk || !dv || (k0 && !null_part)) return gg_fail(-1, "gg_attn_gen_bwd: null pointer");
    if (dk == dq) return gg_fail(-4, "gg_attn_gen_bwd: tied projections are the self-attention form (gg_attn_bwd)");
    p.o = (bf16_t*)o; p.lse = (float*)lse; p.d_o = (const bf16_t*)d_o; p.dvec = dvec;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.null_part = null_part;
    GG_LAUNCH(gg_attn_bwd_dq_kernel<true>, dim3((unsigned)((n + 127) / 128), (unsigned)(B * h)), dim3(256), (hipStream_t)stream, p);
    rc = gg_check_launch();
    if (rc) return rc;
    GG_LAUNCH(gg_attn_bwd_dkv_kernel<true>, dim3((unsigned)((m + 127) / 128), (unsigned)(B * h)), dim3(256), (hipStream_t)stream, p);
    return gg_check_launch();
}

// ---- ChannelRMSNorm passes ------------------------------------------------------------------------------------------

static int gg_rms_launch(int mode, const void* x, const void* g, const void* v, const float* gamma, void* out0, void* out1,
                         float* dgamma_part, int64_t rows, int32_t C, int32_t blocks, float eps, void* stream, int act = 0) {
    if (!x || !gamma || !out0) return gg_fail(-1, "gg_rmsnorm: null pointer");
    if (rows <= 0 || C <= 0 || (C % 8) || C > 512 * GG_RMS_MAXV) return gg_fail(-2, "gg_rmsnorm: need C %% 8 == 0 and C <= %d (C=%d)", 512 * GG_RMS_MAXV, C);
    if (blocks <= 0) return gg_fail(-2, "gg_rmsnorm: blocks must be positive");
    GgRmsParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.g = (const bf16_t*)g; p.v = (const bf16_t*)v; p.gamma = gamma;
    p.out0 = (bf16_t*)out0; p.out1 = (bf16_t*)out1; p.dgamma_part = dgamma_part; p.rows = rows; p.C = C; p.eps = eps;
    p.act = act;
    if (act < 0 || act > 1 || (act && mode == 2)) return gg_fail(-3, "gg_rmsnorm: activation must be none (0) or silu (1), first order only");
    hipStream_t s = (hipStream_t)stream;
    // C = 8 * LPR <= 512: LPR lanes per row, several rows per wavefront (same summation order, 2-4x the bandwidth)
    const char* rows_env = getenv("GG_RMS_ROWS");             // (read per call: the tests compare both kernels in one process)
    const int rows_kernel = rows_env ? atoi(rows_env) : 1;
    if (rows_kernel && mode <= 2 && (C == 32 || C == 64 || C == 128 || C == 256 || C == 512)) {
#define GG_RMS_ROWS_LAUNCH(M, A, L) GG_LAUNCH((gg_rmsnorm_rows_kernel<M, A, L>), dim3((unsigned)blocks), dim3(256), s, p)
#define GG_RMS_ROWS_C(M, A) \
        do { if (C == 32) GG_RMS_ROWS_LAUNCH(M, A, 4); else if (C == 64) GG_RMS_ROWS_LAUNCH(M, A, 8); else if (C == 128) GG_RMS_ROWS_LAUNCH(M, A, 16); \
             else if (C == 256) GG_RMS_ROWS_LAUNCH(M, A, 32); else GG_RMS_ROWS_LAUNCH(M, A, 64); } while (0)
        if (mode == 0 && act) GG_RMS_ROWS_C(0, true);
        else if (mode == 0) GG_RMS_ROWS_C(0, false);
        else if (mode == 2) GG_RMS_ROWS_C(2, false);
        else if (act) GG_RMS_ROWS_C(1, true);
        else GG_RMS_ROWS_C(1, false);
#undef GG_RMS_ROWS_C
#undef GG_RMS_ROWS_LAUNCH
        return gg_check_launch();
    }
    if (mode == 0 && act) GG_LAUNCH((gg_rmsnorm_kernel<0, true>), dim3((unsigned)blocks), dim3(256), s, p);
    else if (mode == 1 && act) GG_LAUNCH((gg_rmsnorm_kernel<1, true>), dim3((unsigned)blocks), dim3(256), s, p);
    else if (mode == 0) GG_LAUNCH(gg_rmsnorm_kernel<0>, dim3((unsigned)blocks), dim3(256), s, p);
    else if (mode == 1) GG_LAUNCH(gg_rmsnorm_kernel<1>, dim3((unsigned)blocks), dim3(256), s, p);
    else GG_LAUNCH(gg_rmsnorm_kernel<2>, dim3((unsigned)blocks), dim3(256), s, p);
    return gg_check_launch();
}

extern "C" int32_t gg_rmsnorm_blocks(int64_t rows) {
    long long nb = (rows + 15) / 16;     // >= 4 rows per wavefront
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int32_t)nb;
}

extern "C" int gg_rmsnorm_fwd(const void* x, const float* gamma, void* y, int64_t rows, int32_t C, float eps, int32_t act, void* stream) {
    return gg_rms_launch(0, x, nullptr, nullptr, gamma, y, nullptr, nullptr, rows, C, gg_rmsnorm_blocks(rows), eps, stream, act);
}

extern "C" int gg_rmsnorm_bwd(const void* x, const void* g, const float* gamma, const void* carry, void* dx, float* dgamma_part,
                              int64_t rows, int32_t C, float eps, int32_t act, void* stream) {
    if (!g) return gg_fail(-1, "gg_rmsnorm_bwd: null gradient");
    return gg_rms_launch(1, x, g, carry, gamma, dx, nullptr, dgamma_part, rows, C, gg_rmsnorm_blocks(rows), eps, stream, act);
}

extern "C" int gg_rmsnorm_bwd2(const void* x, const void* g, const void* v, const float* gamma, void* gx, void* gg,
                               float* dgamma_part, int64_t rows, int32_t C, float eps, void* stream) {
    if (!g || !v || !gg) return gg_fail(-1, "gg_rmsnorm_bwd2: null pointer");
    return gg_rms_launch(2, x, g, v, gamma, gx, gg, dgamma_part, rows, C, gg_rmsnorm_blocks(rows), eps, stream);
}

extern "C" int gg_attn_bwd2(const void* q, const void* k, const void* v, const void* k0, const void* v0, const void* d_o,
                            const void* aq, const void* ak, const void* av, const void* ak0, const void* av0,
                            const float* lse, const float* dvec, float* mu, float* gi, void* gq, void* gk, void* gv,
                            void* gdo, float* null_part, int32_t B, int32_t n, int32_t h, float alpha, float beta,
                            void* stream) {
    if (!q || !k || !v || !k0 || !v0 || !d_o || !aq || !ak || !av || !ak0 || !av0 || !lse || !dvec || !mu || !gi || !gq ||
        !gk || !gv || !gdo || !null_part)
        return gg_fail(-1, "gg_attn_bwd2: null pointer");
    if (B <= 0 || h <= 0 || n <= 0 || (n % 128)) return gg_fail(-2, "gg_attn_bwd2: need n %% 128 == 0 tokens (got %d)", n);
    if ((long long)B * h > 65535) return gg_fail(-2, "gg_attn_bwd2: B*h exceeds grid.y");
    GgAttn2Params p;
    memset(&p, 0, sizeof(p));
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.k0 = (const bf16_t*)k0; p.v0 = (const bf16_t*)v0;
    p.d_o = (const bf16_t*)d_o; p.aq = (const bf16_t*)aq; p.ak = (const bf16_t*)ak; p.av = (const bf16_t*)av;
    p.ak0 = (const bf16_t*)ak0; p.av0 = (const bf16_t*)av0; p.lse = lse; p.dvec = dvec; p.mu = mu; p.gi = gi;
    p.gq = (bf16_t*)gq; p.gk = (bf16_t*)gk; p.gv = (bf16_t*)gv; p.gdo = (bf16_t*)gdo; p.null_part = null_part;
    p.B = B; p.n = n; p.h = h; p.alpha = alpha; p.beta = beta; p.xcd = gg_attn_xcd();
    dim3 grid((unsigned)(n / 128), (unsigned)(B * h));
    hipStream_t s = (hipStream_t)stream;
    GG_LAUNCH(gg_attn_bwd2_q_kernel<true>, grid, dim3(256), s, p);
    int rc = gg_check_launch();
    if (rc) return rc;
    GG_LAUNCH(gg_attn_bwd2_q_kernel<false>, grid, dim3(256), s, p);
    rc = gg_check_launch();
    if (rc) return rc;
    GG_LAUNCH(gg_attn_bwd2_kv_kernel, grid, dim3(256), s, p);
    return gg_check_launch();
}

// ---- data-parallel exchange (gg_comm.h) -----------------------------------------------------------------------------

static int gg_comm_rc(int rc, const char* what) {
    if (rc == 0) return 0;
    const char* msg = gg_comm::g_api.GetErrorString ? gg_comm::g_api.GetErrorString(rc) : "?";
    snprintf(g_err, sizeof(g_err), "%s: ncclResult %d (%s)", what, rc, msg);
    return 1000 + rc;
}

extern "C" int gg_comm_load(const char* librccl_path) {
    return gg_comm::load(librccl_path, g_err, sizeof(g_err)) ? 0 : -30;
}

extern "C" int gg_comm_unique_id(void* id128) {
    if (!id128) return gg_fail(-1, "gg_comm_unique_id: null pointer");
    if (!gg_comm::load(nullptr, g_err, sizeof(g_err))) return -30;
    return gg_comm_rc(gg_comm::g_api.GetUniqueId((gg_comm::UniqueId*)id128), "ncclGetUniqueId");
}

extern "C" int gg_comm_init(int32_t rank, int32_t world, const void* id128) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return gg_fail(-1, "gg_comm_init: bad arguments (rank %d of %d)", rank, world);
    if (gg_comm::g_comm) return gg_fail(-31, "gg_comm_init: a communicator is already live (one per process)");
    if (!gg_comm::load(nullptr, g_err, sizeof(g_err))) return -30;
    gg_comm::UniqueId id;
    memcpy(&id, id128, sizeof(id));
    int rc = gg_comm_rc(gg_comm::g_api.CommInitRank(&gg_comm::g_comm, world, id, rank), "ncclCommInitRank");
    if (rc) { gg_comm::g_comm = nullptr; return rc; }
    gg_comm::g_world = world;
    gg_comm::g_rank = rank;
    return 0;
}

extern "C" int gg_comm_world(void) {
    if (!gg_comm::g_comm) return 0;
    int n = 0;
    if (gg_comm::g_api.CommCount(gg_comm::g_comm, &n) != 0) return 0;
    return n;
}

extern "C" int gg_comm_allreduce(void* buf, size_t n, int32_t dtype, void* stream) {
    if (!gg_comm::g_comm) return gg_fail(-32, "gg_comm_allreduce: no communicator (call gg_comm_init)");
    size_t elem;
    const int dt = gg_comm::dtype_of(dtype, &elem);
    if (!buf || n == 0 || dt < 0) return gg_fail(-1, "gg_comm_allreduce: bad arguments");
    return gg_comm_rc(gg_comm::g_api.AllReduce(buf, buf, n, dt, gg_comm::kSum, gg_comm::g_comm, stream), "ncclAllReduce");
}

extern "C" int gg_comm_allgather(const void* send, void* recv, size_t n_per_rank, int32_t dtype, void* stream) {
    if (!gg_comm::g_comm) return gg_fail(-32, "gg_comm_allgather: no communicator (call gg_comm_init)");
    size_t elem;
    const int dt = gg_comm::dtype_of(dtype, &elem);
    if (!send || !recv || n_per_rank == 0 || dt < 0) return gg_fail(-1, "gg_comm_allgather: bad arguments");
    return gg_comm_rc(gg_comm::g_api.AllGather(send, recv, n_per_rank, dt, gg_comm::g_comm, stream), "ncclAllGather");
}

extern "C" int gg_comm_destroy(void) {
    if (!gg_comm::g_comm) return 0;
    int rc = gg_comm_rc(gg_comm::g_api.CommDestroy(gg_comm::g_comm), "ncclCommDestroy");
    gg_comm::g_comm = nullptr;
    gg_comm::g_world = 0;
    gg_comm::g_rank = -1;
    return rc;
}
