/* gigagan_amd.h — C ABI of libgigagan_amd.so (hand-written gfx950 HIP kernels for the GigaGAN G+D step).
 *
 * The reference (lucidrains/gigagan-pytorch) has no native layer: every op below replaces a stock
 * PyTorch call made from gigagan_pytorch/gigagan_pytorch.py ("gp.py"), cited per entry point. The Python
 * host code in gigagan_pytorch_amd/ binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions:
 *  - plain pointers and sizes only; every buffer (inputs, outputs, workspaces) is owned by the caller;
 *    nothing is allocated, freed or retained by the library; all pointers are device pointers.
 *  - every call only ENQUEUES work on the caller's `stream` (a hipStream_t passed as void*); there is no
 *    host synchronisation and no default-stream use, so calls are hipGraph-capture safe.
 *  - return 0 = enqueued; < 0 = argument error (text via gg_last_error(), thread-local);
 *    > 0 = hipError_t passthrough.
 *  - bf16 tensors are raw 16-bit words (round-to-nearest-even), activations are NHWC.
 */
#ifndef GIGAGAN_AMD_H
#define GIGAGAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GG_ABI_VERSION 12

int gg_version(void);
const char* gg_last_error(void);
/* 1 when this library was built for the host-side kernel emulator (tests only), 0 for the gfx950 build */
int gg_is_emulator(void);

enum { GG_ACT_NONE_ = 0, GG_ACT_LRELU_ = 1, GG_ACT_GELU_ = 2, GG_ACT_SILU_ = 3 };
enum { GG_ROWK = 0, GG_KROW = 1 };

/* One batched contraction C[b][m][n] = act(alpha * out_scale * sum_k A[b][m][k] B[b][n][k] + bias + noise).
 *
 * Replaces: F.conv2d forward/backward-data (gp.py:407 grouped modulated conv; nn.Conv2d at gp.py:1608-1621,
 * :1454-1470, :1275-1281, :1656), conv weight gradients, nn.Linear / 1x1 nn.Conv2d / EqualLinear
 * (gp.py:530-536, :726-740, :871-887, :1121, :1658), and the attention einsums (gp.py:574, :579, :590) with
 * all their autograd transposes.
 *
 * a_layout / b_layout: GG_ROWK = operand stored [row][k] (k contiguous); GG_KROW = stored [k][row].
 * Leading dimensions must be multiples of 8 elements and base pointers 16-byte aligned.
 * a_conv != 0: A is an NHWC activation [n_img][H][W][C] gathered as an R x S convolution window with stride
 *   `conv_stride` and zero padding `conv_pad` (output grid OH = (H + 2*pad - R)/stride + 1, same for OW): 3x3 / 7x7 /
 *   1x1 "same" convs (stride 1, pad (R-1)/2), the stride-2 1x1 residual conv (gp.py:1612: R = 1, stride 2, pad 0) and
 *   space-to-depth + 1x1 (gp.py:289-293: R = 2, stride 2, pad 0, weights ordered [co][s1][s2][c]).
 *   With GG_ROWK the rows are output pixels (M = n_img*OH*OW) and k = (tap, cv); with GG_KROW (weight gradient) k
 *   runs over output pixels (K = n_img*OH*OW) and rows are (tap, cv) (M = R*S*CV). CV is the virtual channel count
 *   (CV % C == 0, physical channel = cv % C); in_scale, if given, is an fp32 [n_img][CV] multiplier on the gathered
 *   activation (style modulation, gp.py:396).
 * Epilogue order: acc*alpha -> *out_scale[m / rows_per_group][n] -> +bias[n]*bias_scale -> +noise[m]*noise_w[n]
 *   -> activation (leaky-relu slope `act_slope`, exact-erf GELU, SiLU) -> +residual[m][n]*res_scale.
 * d2s != 0: depth-to-space scatter store, the adjoint of a stride-`d2s` gather (data gradient of the two stride-2
 *   convs above): column n = (tap, c), c < d2s_c, tap = ty*d2s_taps + tx; row m = (img, oh, ow) on the d2s_oh x d2s_ow
 *   grid; the element is written to pixel (oh*d2s + ty, ow*d2s + tx) of an [img][d2s_oh*d2s][d2s_ow*d2s][d2s_c] tensor
 *   (pixels no tap maps to are not touched: pre-zero them). d2s_c %% 4 == 0.
 */
typedef struct gg_gemm_desc {
    int32_t M, N, K, batch;
    const void* A; int64_t a_batch_stride; int32_t lda; int32_t a_layout; int32_t a_conv;
    const void* B; int64_t b_batch_stride; int32_t ldb; int32_t b_layout;
    int32_t H, W, C, CV, R, S;
    const float* in_scale;
    void* C_out; int64_t c_batch_stride; int32_t ldc; int32_t c_is_f32;
    float alpha;
    const float* bias;
    const float* out_scale; int32_t rows_per_group;
    const float* noise; const float* noise_w;
    int32_t act; float act_slope;
    int32_t force_splitk; /* 0 = heuristic */
    int32_t force_tile;   /* 0 = heuristic; 1: 128x128, 2: 128x64, 3: 128x32 (4 waves); 4: 256x256, 5: 256x128, 6: 128x128 (8 waves);
                           * 7: halo-staged 3x3 convolution (stride 1, pad 1, C %% 64 == 0, H and W powers of two, 8 <= W <= 64);
                           * 10: nine-tap 3x3 weight gradient (reduction-major operands, stride 1, pad 1, C %% 32 == 0, 8 <= W <= 64);
                           * 9: direct 3x3 convolution (C in {16,32,64}, N <= 64, W %% 32 == 0, H %% 8 == 0);
                           * 11: low-resolution 3x3 convolution (stride 1, pad 1, C %% 32 == 0, 4x4 / 8x8 / 16x16 images: the first
                           *     adaptive convolutions of Generator.forward, gp.py:1184-1245);
                           * 8 / 12: the halo-staged convolution with 128 / 64 output channels per workgroup (same eligibility as 7);
                           * 13 / 14: the streaming 3x3 weight gradient / forward convolution of the narrow high-resolution layers;
                           * 15: the persistent short-K contraction (row-major operands or a 1x1 / stride 1 gather, K %% 64 == 0,
                           *     N, ldc, row pitches multiples of 8, bf16 output, batch 1, no out_scale / noise / in_scale / d2s):
                           *     the 1x1 convolutions around attention and FeedForward, gp.py:620-740;
                           * else heuristic */
    int32_t conv_stride;  /* >= 1 */
    int32_t conv_pad;     /* >= 0 */
    float bias_scale;     /* multiplies bias (set 1.0f) */
    const void* residual; int32_t ldr; float res_scale;   /* bf16 [M][ldr], optional */
    int32_t d2s, d2s_taps, d2s_c, d2s_oh, d2s_ow;
    int64_t b_image_stride;  /* a_conv forward (GG_ROWK x GG_ROWK) only: > 0 = per-image weight operands, image i reads B + i * b_image_stride
                              * elements (the reference's per-sample weights of AdaptiveConv2DMod, gp.py:388-407); OH*OW must be a
                              * multiple of the row tile, which the planner guarantees (128 or 256 rows) or rejects */
    const float* bank_mix;   /* optional fp32 [n_img][CV / C]: a kernel bank stacked along the reduction (weights [co][tap][n][ci], CV = n_banks * C)
                              * is MIXED per image, w_img = sum_n bank_mix[img][n] * W_n, while its tiles are staged (AdaptiveConv2DMod's
                              * softmax-weighted kernel sum, gp.py:378-386) and the contraction runs over the C physical channels:
                              * algorithmic flops instead of n_banks times that. in_scale is then fp32 [n_img][C]. 16x16 images,
                              * 2 banks, 3x3 / stride 1 / pad 1 (gg_lrconv, one image per 256-pixel tile); other shapes are rejected */
    int32_t keep_partials;   /* split-K launches only (fp32 output, alpha-only epilogue): 1 = leave the slices [splitk][M][N] in the workspace
                              * and skip the reduction launch - the caller folds it into its own consumer (gg_finish_multi's nsplit) */
    int32_t gelu_mode;       /* GELU fused around the 1x1 pair of a FeedForward (gp.py:726-740), bf16 outputs on a staged epilogue
                              * only (gg_gemm_plan tile 4-6 or 15, split-K 1; anything else is rejected): 1 = the result h is ALSO stored to
                              * gelu_aux and C_out receives gelu(h); 2 = gelu_aux holds h and C_out receives result * gelu'(h) */
    void* gelu_aux;          /* bf16 [M][ld_aux] */
    int32_t ld_aux, reserved1;
} gg_gemm_desc;

/* Per-device tuning cache (the only persistent native state besides the communicator): measured-best launch
 * plans for exact problem geometries, consulted before the cost model whenever the caller forces neither tile nor split-K.
 * `epi` = 1 when any of bias / out_scale / noise / residual / activation is present, `scaled` = 1 with in_scale. An entry whose
 * tile is not eligible for the geometry is ignored. gg_gemm_plan_table copies `n` entries (n = 0 clears); not thread-safe
 * against concurrent launches: set it once after loading the library (tests/gpu_plan_sweep.py produces the table). */
typedef struct gg_plan_entry {
    int32_t M, N, K, batch, a_layout, b_layout, a_conv, H, W, C, CV, R, conv_stride, conv_pad, c_is_f32, d2s, epi, scaled;
    int32_t tile, splitk;
} gg_plan_entry;
int gg_gemm_plan_table(const gg_plan_entry* entries, int32_t n);

size_t gg_gemm_workspace_bytes(const gg_gemm_desc* d);
/* reports the launch plan the library will use for `d`: tile (1: 128x128, 2: 128x64, 3: 128x32, 4: 256x256,
 * 5: 256x128, 6: 128x128 with 8 waves, 7 - 15: the specialised kernels listed at force_tile) and the split-K factor; used by bench.py to attribute measured time to kernel instantiations. */
int gg_gemm_plan(const gg_gemm_desc* d, int32_t* tile, int32_t* splitk);
int gg_gemm_bf16(const gg_gemm_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* Separable banded linear resampling of an NHWC bf16 tensor:
 *   out[n][oy][ox][c] = sum_{a<ty} sum_{b<tx} wy[oy*ty+a] * wx[ox*tx+b] * in[n][iy0[oy]+a][ix0[ox]+b][c]
 * (taps that fall outside the input contribute zero). Replaces nn.Upsample(x2, bilinear) + kornia
 * filter2d blur (gp.py:246-261) as one pass, F.interpolate bilinear / nearest (gp.py:1683-1687, :2210,
 * unet_upsampler.py:33-61) and, with transposed tables, their backward passes. */
int gg_resample_nhwc_bf16(const void* in, void* out, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                          int32_t c, int32_t ty, int32_t tx, const int32_t* iy0, const int32_t* ix0,
                          const float* wy, const float* wx, void* stream);

/* Fused multi-tensor AdamW over flat fp32 buffers (replaces torch.optim.AdamW from optimizer.py:10-34,
 * stepped at gp.py:2477 and :2596). n %% 256 == 0; flags[n/256]: bit0 = chunk is stepped, bit1 = decoupled
 * weight decay applies. bias_corr1 = 1 - beta1^t, bias_corr2_sqrt = sqrt(1 - beta2^t); grad_scale
 * multiplies the gradient first (e.g. 1/world_size after a summing all-reduce). */
int gg_adamw_flat_f32(float* p, const float* g, float* m, float* v, const uint8_t* flags, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                      float bias_corr2_sqrt, float grad_scale, void* stream);

/* ema += (1 - beta) * (p - ema) over flat fp32 buffers (ema_pytorch update, gp.py:2603); n %% 4 == 0. */
int gg_ema_flat_f32(float* ema, const float* p, int64_t n, float one_minus_beta, void* stream);

/* One work-table entry of gg_pack_weights: a conv weight (O, I, T = kh*kw) in the reference's fp32 parameter layout
 * (gp.py:352 `weights`, nn.Conv2d.weight) and the bf16 GEMM operand it is packed into. */
typedef struct gg_pack_entry {
    const float* src;     /* (O, I, T) fp32 contiguous (device) */
    uint16_t* dst;        /* kind 0: (O8, T, I8) [co][tap][ci] ; kind 1: (I8, T, O8) [ci][T-1-tap][co] ; bf16 (device) */
    int64_t first_item;   /* prefix sum over the table of work items per entry. T <= 16: kind 0: O8*ceil(I8/256),
                           * kind 1: ceil(O8/64)*ceil(I8/16); T > 16: ceil((O8*I8/8)/256) */
    int32_t O, I, T, O8, I8, kind;   /* O8, I8 = O, I rounded up to multiples of 8 (zero filled) */
    int32_t dst_row, dst_tap;        /* kind 0 only: element pitch of a dst row / of a tap inside a row; 0 = dense (T*I8, I8).
                                      * Lets the N kernels of an AdaptiveConv2DMod bank (gp.py:352) interleave along the
                                      * reduction as [co][tap][n][ci]: entry n has dst + n*I8, dst_row = T*N*I8, dst_tap = N*I8 */
} gg_pack_entry;

/* Re-pack every registered weight of a model in one launch (replaces the filter transforms behind the reference's
 * F.conv2d / nn.Conv2d calls, gp.py:402-409, :1608-1621, and their transposes in autograd's conv backward).
 * `table` (device) holds header[0] entries, header[1] (device) is the total item count: both are read on the device,
 * so a captured graph follows later registrations. max_blocks <= 0 picks a default grid. */
int gg_pack_weights(const gg_pack_entry* table, const int64_t* header, int32_t max_blocks, void* stream);

/* dst[o][i][t] (+)= alpha * g[(t*C8 + i)*O8 + o]: the weight-gradient GEMM output ([tap][ci][co] fp32) moved into the
 * parameter layout (O, I, T), optionally accumulated in place - `dst` may be the parameter's .grad inside the flat
 * gradient buffer (what autograd's AccumulateGrad does after conv2d_weight backward in the reference). */
int gg_wgrad_finish(const float* g, float* dst, int32_t O, int32_t I, int32_t T, int32_t C8, int32_t O8, float alpha,
                    int32_t accumulate, void* stream);


/* Per-sample coefficients of the adaptive convolution (AdaptiveConv2DMod.forward, gp.py:378-400) in one launch:
 *   s[b,i] = mod[b,i] + 1 (zero for I <= i < Ip),  a[b,n] = softmax_n(kmod[b,:]) (1 when N == 1, kmod may be NULL),
 *   d[b,o] = rsqrt(max(sum_{i,t} (sum_n a[b,n] w[n,o,i,t] s[b,i])^2, eps)) (zero for O <= o < Op; d NULL: skipped).
 * w fp32 (N, O, I, T); mod (b, I), kmod (b, N), s (b, Ip), a (b, N), d (b, Op) fp32. N <= 4, I, O <= 1024. */
int gg_modcoef_fwd(const float* w, const float* mod, const float* kmod, float* s, float* a, float* d, int32_t b,
                   int32_t N, int32_t O, int32_t I, int32_t T, int32_t Ip, int32_t Op, float eps, void* stream);

/* Backward of gg_modcoef_fwd (what autograd derives from gp.py:378-400): given gs (b, Ip) / ga (b, N) (either may be
 * NULL) and gd (b, Op), writes gmod (b, I) and gkmod (b, N), and ADDS the weights' gradient through the demodulation
 * into gw (N, O, I, T) when gw != NULL. da_acc (b, N) is caller-zeroed scratch. */
int gg_modcoef_bwd(const float* w, const float* kmod, const float* s, const float* d, const float* gs, const float* ga,
                   const float* gd, float* gmod, float* gkmod, float* da_acc, float* gw, int32_t b, int32_t N, int32_t O,
                   int32_t I, int32_t T, int32_t Ip, int32_t Op, float eps, void* stream);

/* The same coefficients through the bank's Gram rows (the tap sum does not depend on the sample): gram (P, O, I) fp32, P = N (N + 1) / 2
 * pairs n <= m row-major, gram[(n,m),o,i] = c sum_t w[n,o,i,t] w[m,o,i,t], c = 1 on the diagonal and 2 off it (gg_modgram; the layout of
 * gg_pack_weights' kind 2). d[b,o] = rsqrt(max(sum_p a_n a_m tsum[b,p,o], eps)) with tsum[b,p,o] = sum_i s^2[b,i] gram[p,o,i] (written for
 * the backward). The backward touches the weights once, in the element-wise update of gw. 17x fewer operations than gg_modcoef_fwd / _bwd on
 * a 512 x 512 x 9 bank at batch 32. b <= 64. da_slots: (O, b, N) fp32 scratch (N > 1), no initialisation needed. */
int gg_modgram(const float* w, float* gram, int32_t N, int32_t O, int32_t I, int32_t T, void* stream);
int gg_modcoef_gram_fwd(const float* gram, const float* mod, const float* kmod, float* s, float* a, float* d, float* tsum, int32_t b,
                        int32_t N, int32_t O, int32_t I, int32_t Ip, int32_t Op, float eps, void* stream);
int gg_modcoef_gram_bwd(const float* w, const float* gram, const float* kmod, const float* s, const float* d, const float* tsum,
                        const float* gs, const float* ga, const float* gd, float* gmod, float* gkmod, float* da_slots, float* gw,
                        int32_t b, int32_t N, int32_t O, int32_t I, int32_t T, int32_t Ip, int32_t Op, float eps, void* stream);

/* dst[c] += alpha * sum_p part[p][c] for c < n: folds the [P][C] fp32 partial column sums written by
 * gg_bias_act_bwd (nn.Conv2d bias gradient, gp.py:1608-1621 autograd) into the bias gradient - `dst` is the
 * parameter's .grad (running sum) or a zeroed buffer; fp32 atomics, one per channel and group of 16 partial rows. */
int gg_colsum_finish(const float* part, float* dst, int32_t P, int32_t C, int32_t n, float alpha, void* stream);

/* many weight-gradient / bias-gradient finishes in ONE launch (a backward pass ends every convolution with one of each: 221
 * launches of 4-8 us per step). kind 0 = gg_wgrad_finish's work (src (T*C8, O8) fp32 -> dst (O, I, T), accumulate as there);
 * kind 1 = gg_colsum_finish's (src (P, C) fp32 partial sums with O = P, I = C, T = n columns -> dst (n,), always accumulated);
 * kind 2 = dst += alpha * src over O fp32 elements (a dense linear-layer weight gradient).
 * The items are copied into kernel arguments (batches of 40): the array may live in pageable host memory and be reused at once.
 * Items of one call must not write overlapping destinations. */
typedef struct gg_finish_item {
    const float* src; float* dst;
    int32_t kind, O, I, T, C8, O8, accumulate;
    float alpha;
    int32_t nsplit;      /* kind 0: src holds nsplit split-K slices [nsplit][T*C8][O8] that are summed while they are read (0 / 1: one) */
    int32_t reserved0;
} gg_finish_item;
int gg_finish_multi(const gg_finish_item* items, int32_t n, void* stream);

/* Many split-K slice stacks folded by ONE launch: for each item, slice 0 += slices 1 .. nsplit-1 in place (fixed order). What a flush of
 * the host's finish queue runs first for weight gradients that were split over more than 16 k-slices (one gg_gemm_bf16 split-K reduce
 * launch each before: ~70 per step); their finishes then read slice 0. src: (nsplit, n) fp32. */
typedef struct {
    float* src;
    int64_t n;
    int32_t nsplit;
    int32_t reserved;
} gg_reduce_item;
int gg_reduce_multi(const gg_reduce_item* items, int32_t n, void* stream);

/* Row softmax over materialised attention logits (replaces sim*scale, masked_fill, softmax and the dtype casts
 * of gp.py:584-588 / :643-649 with one pass):
 *   S[r][j] = softmax_j(alpha * x[r][j] + bias[r / rows_per_batch][j]) for j < n_valid, 0 for n_valid <= j < ld.
 * x fp32 [rows][ld], S bf16 [rows][ld]; ld %% 4 == 0, ld <= 2048; bias optional fp32 [rows/rows_per_batch][ld]. */
int gg_softmax_fwd(const float* x, void* S, const float* bias, int64_t rows, int32_t rows_per_batch, int32_t n_valid,
                   int32_t ld, float alpha, void* stream);
/* backward: u = S*(dS - sum_j S*dS); dx = alpha*u (bf16 [rows][ld]); if dbias != NULL, dbias[batch][j] += sum over
 * the batch's rows of u (fp32, caller zeroes it; rows_per_batch %% 16 == 0). */
int gg_softmax_bwd(const void* S, const void* dS, void* dx, float* dbias, int64_t rows, int32_t rows_per_batch,
                   int32_t n_valid, int32_t ld, float alpha, void* stream);

/* second-order pass of the attention softmax (gradient-penalty steps differentiate gg_softmax_bwd; replaces the
 * autograd of gp.py:584-588's backward): with gt = alpha*g_dx + g_dbias[batch] (either may be NULL = zero),
 * r = sum_j S*dS, gs = sum_j gt*S:  g_dS = S*(gt - gs),  g_S = gt*(dS - r) - dS*gs.  All bf16 [rows][ld] except
 * g_dbias fp32 [rows/rows_per_batch][ld]. */
int gg_softmax_bwd2(const void* S, const void* dS, const void* g_dx, const float* g_dbias, void* g_S, void* g_dS,
                    int64_t rows, int32_t rows_per_batch, int32_t n_valid, int32_t ld, float alpha, void* stream);

/* Backward of "+bias -> leaky-relu" (nn.Conv2d bias + nn.LeakyReLU autograd, gp.py:109, :1608-1621) in one pass:
 * dz = dy * (y > 0 ? 1 : slope) when y != NULL (else dz is not written); when db != NULL, db[w][c] receives workgroup
 * w's partial column sums of dz (fp32 [gg_bias_act_bwd_partials(rows, C)][C]; the caller adds the rows up).
 * dy / y / dz: bf16 [rows][C], C %% 8 == 0. */
int32_t gg_bias_act_bwd_partials(int64_t rows, int32_t C);

/* Exact (erf) GELU over n contiguous bf16 elements (n %% 8 == 0) - nn.GELU() of FeedForward, gp.py:731 / unet.py:388 -
 * and its derivatives, one pass each: mode 0: out0 = gelu(x); mode 1: out0 = dy * gelu'(x) (autograd's gelu backward);
 * mode 2: out0 = g * gelu'(x), out1 = g * dy * gelu''(x) (the backward of that backward, which the gradient penalty's
 * double backward, gp.py:120-155, reaches through the discriminator's attention blocks). */
int gg_gelu(const void* x, const void* dy, const void* g, void* out0, void* out1, int64_t n, int32_t mode, void* stream);
int gg_bias_act_bwd(const void* dy, const void* y, void* dz, float* db, int64_t rows, int32_t C, float slope,
                    void* stream);

/* ---- the bf16 passes around the adaptive / style-modulated convolution (AdaptiveConv2DMod.forward gp.py:344-409,
 * Noise gp.py:925-940, leaky_relu gp.py:109) in its batched form
 *     y[b] = act( d[b,o] * sum_n a[b,n] * conv(W_n, x[b] * s[b,:]) + noise_w[o] * noise[b,p] ).
 * Activations are [b][P pixels][C] bf16, C %% 8 == 0; per-image reductions come back as per-workgroup partial sums
 * [b][chunks][...] (the caller sums over `chunks`). */
/* xs = x * s[b,:]  (the weight modulation of gp.py:394-396 moved onto the activation). */
int gg_modulate_fwd(const void* x, const float* s, void* out, int32_t b, int32_t P, int32_t C, void* stream);
/* dx = g * s[b,:],  ds_part[b][chunk][c] = partial sums over pixels of g * x. */
int gg_modulate_bwd(const void* g, const void* x, const float* s, void* dx, float* ds_part, int32_t b, int32_t P, int32_t C,
                    int32_t chunks, void* stream);
/* y = act(d[b,o] * sum_n a[b,n] * Y[b][p][n*Os + o] + noise_w[o] * noise[b][p]); d / noise optional; act 0 none, 1 leaky
 * relu; N <= 4 stacked kernels with per-kernel pitch Os >= O. */
int gg_modmix_fwd(const void* Y, const float* a, const float* d, const float* noise, const float* noise_w, void* y, int32_t b,
                  int32_t P, int32_t O, int32_t Os, int32_t N, int32_t act, float slope, void* stream);
/* gradient of gg_modmix_fwd: dz = dy * act'(y); dY[b][p][n*Os + o] = a[b,n] * d[b,o] * dz; partial sums
 * da_part[chunk][b][n] = sum dz * d * Y_n, dd_part[chunk][b][o] = sum_p dz * sum_n a_n Y_n (iff d), dnw_part[chunk][b][o] =
 * sum_p dz * noise (iff noise): chunk-major, i.e. slice stacks that gg_reduce_multi folds in one launch. */
int gg_modmix_bwd(const void* dy, const void* y, const void* Y, const float* a, const float* d, const float* noise, void* dY,
                  float* da_part, float* dd_part, float* dnw_part, int32_t b, int32_t P, int32_t O, int32_t Os, int32_t N,
                  int32_t chunks, int32_t act, float slope, void* stream);

/* ---- fused self-attention (replaces SelfAttention.forward's einsum / softmax / einsum, gp.py:573-590, and its autograd)
 * q, k, v, o, d_o, dq, dk, dv: [B][n][h*64] bf16 (the layout the 1x1 projections produce / consume), head dim 64,
 * n %% 128 == 0; k0, v0: the learned null key / value [h][64] bf16 (gp.py:534, :568), prepended to every sequence;
 * logits x_ij = alpha * q_i.k_j + beta * |k_j|^2  (dot product: alpha = scale, beta = 0; squared-L2 distance:
 * alpha = 2*scale, beta = -scale); lse: [B*h][n] fp32 log-sum-exp of each query's logits (saved for the backward).
 * gg_attn_bwd also needs dvec [B*h][n] fp32 scratch and returns the null token's partial gradients in
 * null_part [B*h * n/128][3][64] fp32: [0] = sum_i dS_i0 q_i (multiply by alpha), [1] = sum_i P_i0 dO_i (= dv0),
 * [2][0] = sum_i dS_i0 (d/d bias0); the caller sums over blocks and batch. */
int gg_attn_fwd(const void* q, const void* k, const void* v, const void* k0, const void* v0, void* o, float* lse, int32_t B,
                int32_t n, int32_t h, float alpha, float beta, void* stream);
int gg_attn_bwd(const void* q, const void* k, const void* v, const void* k0, const void* v0, const void* o, const float* lse,
                const void* d_o, float* dvec, void* dq, void* dk, void* dv, float* null_part, int32_t B, int32_t n, int32_t h,
                float alpha, float beta, void* stream);
/* general form of the fused attention (reference attend.py:64-110 `Attend` of the unet, CrossAttention gp.py:617-655, the text
 * transformer's attention gp.py:659-722): n queries [B][n] against m keys / values [B][m] per head of 64 features, q / k / v rows
 * `ld*` elements apart (channel slices of a fused projection are read in place), the null key / value optional (k0 = v0 = NULL),
 * an optional additive per-key bias kbias [B][m] fp32 (key-padding masks: -1e30), any n and m (ragged tails are clamped on load and
 * masked). o / lse / dO / dq as in gg_attn_fwd; dk / dv dense [B][m][h*64]; null_part [ceil(n/128) blocks ...] as gg_attn_bwd when
 * the null token is present. First order (these attentions never sit between the images and the gradient penalty). */
int gg_attn_gen_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* k0, const void* v0,
                    const float* kbias, void* o, float* lse, int32_t B, int32_t n, int32_t m, int32_t h, float alpha, float beta,
                    void* stream);
int gg_attn_gen_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* k0, const void* v0,
                    const float* kbias, const void* o, const float* lse, const void* d_o, float* dvec, void* dq, void* dk, void* dv,
                    float* null_part, int32_t B, int32_t n, int32_t m, int32_t h, float alpha, float beta, void* stream);
/* gg_attn_bwd: `dk` may alias `dq` when the key projection IS the query projection (the L2-distance attention ties them,
 * reference gp.py:566-569): the buffer then receives dq + dk, the gradient of the shared tensor, without a separate add. */

/* ---- ChannelRMSNorm (gp.py:224-232): y = x / max(|x|_2, eps) * sqrt(C) * gamma over the channel axis of [rows][C] bf16
 * (NHWC pixels as rows), fp32 statistics, C %% 8 == 0, C <= 2048. bwd: dx and per-workgroup partial sums of dgamma
 * ([gg_rmsnorm_blocks(rows)][C] fp32, optional). bwd2 differentiates bwd for an incoming gradient v w.r.t. dx:
 * gx (w.r.t. x), gg (w.r.t. g) and the partial sums of that pass's dgamma — gradient-penalty steps only. */
/* LinearAttention's two softmaxes (reference unet.py:338-348) on NHWC bf16 channel slices (C = heads * 64 channels starting at
 * the given pointers, row pitches ld_* in elements: the q / k / v slices of the fused to_qkv output and of its gradient are read
 * and written in place, no head split / transpose copies):
 *   gg_linattn_q_fwd: qs[r, h, :] = scale * softmax(q[r, h, :]) over the 64 features of head h at row (position) r;
 *   gg_linattn_q_bwd: dq from qs and the gradient w.r.t. qs;
 *   gg_linattn_k_fwd: eks[b, p, c] = softmax over the n positions p of k[b, :, c] (fp32 two-stage column statistics, `part`:
 *                     b * gg_linattn_chunks(b, n) * C * 2 floats, `stat`: b * C * 2 floats, both caller-owned);
 *   gg_linattn_k_bwd: dk = eks * (deks - sum_p eks * deks) (`part`: b * chunks * C floats, `stat`: b * C floats).
 * The two contractions between them (context = eks^T v, out = qs context) are gg_gemm_bf16 launches on strided views. */
int gg_linattn_q_fwd(const void* q, int32_t ld_q, void* qs, int32_t ld_qs, int64_t rows, int32_t C, float scale, void* stream);
int gg_linattn_q_bwd(const void* qs, int32_t ld_qs, const void* dqs, int32_t ld_dqs, void* dq, int32_t ld_dq, int64_t rows, int32_t C,
                     float scale, void* stream);
int32_t gg_linattn_chunks(int32_t b, int32_t n);
int gg_linattn_k_fwd(const void* k, int32_t ld_k, void* eks, int32_t ld_eks, float* part, float* stat, int32_t b, int32_t n, int32_t C,
                     void* stream);
int gg_linattn_k_bwd(const void* eks, int32_t ld_eks, const void* deks, int32_t ld_deks, void* dk, int32_t ld_dk, float* part,
                     float* stat, int32_t b, int32_t n, int32_t C, void* stream);

/* Hinge losses of the trainer over one logit tensor (reference gp.py:157-163) and their backward, one launch each (the reference's
 * formulation is ~16 PyTorch launches per tensor, ~95 per step). x: (outer, nb, inner) bf16 or fp32, the batch on the middle axis.
 * mode 0: loss = mean(x) (generator_hinge_loss). mode 1: rows j < split are the fake half, the rest the real half of a merged
 * discriminator batch: loss = mean over a half of relu(1 + real) + relu(1 - fake) (discriminator_hinge_loss). dx == NULL: forward
 * (loss[0] fp32 written); dx != NULL: dx (x's dtype) = gscale[0] * d loss / d x. Up to 64 workgroups; the forward's partial sums are added in
 * a fixed order by the last workgroup to arrive (deterministic). scratch: 65 x 4 bytes, zero before the first use, owned by one stream at a
 * time (the kernel leaves its ticket at zero); NULL = a one-workgroup forward. */
int gg_hinge(const void* x, void* dx, const float* gscale, float* loss, void* scratch, int64_t n, int64_t inner, int32_t nb,
             int32_t split, int32_t mode, int32_t x_is_f32, void* stream);

/* y = (a + b) * c + d over n bf16 elements (b, d may be null): the predictor blocks' residual merge `(x + inner) * 2^-0.5`
 * (reference gp.py:1493), the last one together with the `+ residual` of gp.py:1495, as one pass; with b and d null its backward. */
int gg_scaled_add(const void* a, const void* b, const void* d, void* y, int64_t n, float c, void* stream);

/* The discriminator's multi-scale input merge (reference gp.py:1797-1803): out [2B][n] = cat(x + tile(feats), tile(feats)) for
 * x [B][n] and feats [f][n] bf16 (n elements per sample, B % f == 0, tiled row r = feats[r % f]); backward w.r.t. feats:
 * gfeats [f][n] = sum over the 2 B / f rows of g that came from feats row j (fp32 accumulation). The gradient w.r.t. x is the
 * view g[:B]. */
int gg_addcat_fwd(const void* x, const void* feats, void* out, int32_t B, int32_t f, int64_t n, void* stream);
int gg_addcat_bwd(const void* g, void* gfeats, int32_t B, int32_t f, int64_t n, void* stream);

/* SqueezeExcite's pool (reference gp.py:300, `Reduce('b c h w -> b c', 'mean')`) over an NHWC bf16 activation x [b][P][C]:
 * out [b][C] fp32 = mean over the P pixels (fp32 accumulation, deterministic two-stage sum; `part` is caller-owned scratch of
 * b * gg_pool_chunks(b, P) * C floats). Backward: y = g + gs[b, c] broadcast over the pixels (gs = dL/dmean / P, fp32
 * [b][C]); g (optional) is the gradient that reached x over its other consumer, so the branch merge costs no extra pass;
 * y may alias g. */
int32_t gg_pool_chunks(int32_t b, int32_t P);
int gg_pool_mean_fwd(const void* x, float* part, float* out, int32_t b, int32_t P, int32_t C, void* stream);
int gg_pool_mean_bwd(const void* g, const float* gs, void* y, int32_t b, int32_t P, int32_t C, void* stream);

/* SqueezeExcite's excitation MLP (reference gp.py:297-307: `nn.Linear(dim, dim_hidden), nn.SiLU(), nn.Linear(dim_hidden, dim_out),
 * nn.Sigmoid()` on the pooled rows) in one launch, fp32 throughout, on the parameters where they lie (nn.Linear's [out][in] rows):
 *   h [b][H] = W1 m + b1,  hs = silu(h),  e [b][O] = sigmoid(W2 hs + b2)           (m [b][C]; b1 / b2 may be null)
 * Backward (replaces autograd through the five modules): dz2 = de * e (1 - e), dz1 = (W2^T dz2) * silu'(h), dm [b][C] = W1^T dz1
 * (dm may be null), and - unless gw is null - the parameter gradients, WRITTEN (not accumulated) to
 * gw = [gW1 (H*C) | gb1 (H) | gW2 (O*H) | gb2 (O)], sums over the b samples in sample order. dz2 [b][O] and dz1 [b][H] are
 * caller-owned scratch. Deterministic. Limits: C, O <= 2048, H <= 512. */
int gg_se_mlp_fwd(const float* m, const float* w1, const float* b1, const float* w2, const float* b2, float* h, float* hs, float* e,
                  int32_t b, int32_t C, int32_t H, int32_t O, void* stream);
int gg_se_mlp_bwd(const float* de, const float* e, const float* h, const float* hs, const float* m, const float* w1, const float* w2,
                  float* dz2, float* dz1, float* dm, float* gw, int32_t b, int32_t C, int32_t H, int32_t O, void* stream);

int32_t gg_rmsnorm_blocks(int64_t rows);
/* act = 1: y = silu(norm(x)) in the same pass (the unet Block's norm + activation, reference unet.py:224-234, :268-269); the
 * backward then takes the incoming gradient through silu'(z) with z recomputed from x (first order only) */
int gg_rmsnorm_fwd(const void* x, const float* gamma, void* y, int64_t rows, int32_t C, float eps, int32_t act, void* stream);
/* `carry` (optional, [rows][C] bf16) is added to dx inside the pass: the gradient arriving over the skip connection around
 * the normalised branch (x + f(norm(x)), gp.py:757-758), which autograd would otherwise add in a pass of its own */
int gg_rmsnorm_bwd(const void* x, const void* g, const float* gamma, const void* carry, void* dx, float* dgamma_part,
                   int64_t rows, int32_t C, float eps, int32_t act, void* stream);
int gg_rmsnorm_bwd2(const void* x, const void* g, const void* v, const float* gamma, void* gx, void* gg, float* dgamma_part,
                    int64_t rows, int32_t C, float eps, void* stream);

/* second-order pass of the fused attention (gradient-penalty steps differentiate gg_attn_bwd; reference:
 * torch.autograd.grad(create_graph=True) through SelfAttention, gp.py:120-155 and :538-594). aq/ak/av ([B][n][h*64] bf16)
 * and ak0/av0 ([h][64] bf16) are the incoming gradients w.r.t. gg_attn_bwd's dq/dk/dv/dk0/dv0; lse and dvec are the
 * forward's log-sum-exp and gg_attn_bwd's dvec. Returns gq/gk/gv/gdo = d<A, (dq,dk,dv,dk0,dv0)> / d(q, k, v, dO) and the
 * null token's partial sums null_part [B*h * n/128][3][64]: [0] sum_i (R_i0 q_i + dS_i0 A_q,i) (times alpha),
 * [1] sum_i Pd_i0 dO_i (= gv0), [2][0] = sum_i R_i0, [2][1] = sum_i dS_i0, so that
 * gk0 = alpha*[0] + 2*beta*([2][0]*k0 + [2][1]*ak0). mu, gi: [B*h][n] fp32 scratch. */
int gg_attn_bwd2(const void* q, const void* k, const void* v, const void* k0, const void* v0, const void* d_o, const void* aq,
                 const void* ak, const void* av, const void* ak0, const void* av0, const float* lse, const float* dvec,
                 float* mu, float* gi, void* gq, void* gk, void* gv, void* gdo, float* null_part, int32_t B, int32_t n,
                 int32_t h, float alpha, float beta, void* stream);

/* ---- unet Downsample tail (reference unet_upsampler.py:134-160): pool = max_pool2d(x, 2), hf = x - blur(x) with kornia's
 * normalised [1,2,1]^2/16 filter and 'reflect' border, NHWC bf16, in ONE pass; gg_poolhf_bwd: dx = scatter(g_pool to each window's
 * first maximum, torch's tie rule) + g_hf - blur^T(g_hf) (exact adjoint of the reflect-padded stencil); either incoming gradient may
 * be null. Even H, W; C %% 8 == 0. */
int gg_poolhf_fwd(const void* x, void* pool, void* hf, int32_t b, int32_t H, int32_t W, int32_t C, void* stream);
int gg_poolhf_bwd(const void* x, const void* g_pool, const void* g_hf, void* dx, int32_t b, int32_t H, int32_t W, int32_t C,
                  void* stream);

/* ---- no-grad forward of the adaptive convolution (gp.py:344-409; what the discriminator step's generator pass and generate()
 * run), csrc/gg_modfwd.h ----
 * gg_modw_fwd: ONE launch per layer: s = mod + 1 (b, Ip), a = softmax(kernel_mod) (b, N), d (b, Op) = demodulation coefficients
 *   (ones when demod == 0; any of s / a / d may be null), and optionally the reference's per-sample weights
 *   wmix = d[b,o] s[b,i] sum_n a[b,n] W_n[o,i,t] in bf16: layout 1 = [b][O][T][I] (rows of T*I: the implicit GEMM's weight operand of
 *   image b), layout 2 = [b][T][I/16][32][16] (gg_sconv_fwd's filter bank; rows O..31 are left untouched: zero them once).
 *   xs (optional, (b, I) fp32): an extra per-sample scale of the INPUT activation (the skip-layer excitation `x * excite`,
 *   gp.py:1023-1024) folded into s and wmix but not into d, which the reference derives from mod + 1 alone.
 *   mod (b, I) / kmod (b, N) are fp32 rows `mod_ld` / `kmod_ld` floats apart (column slices of the style network's output).
 *   b <= 64, N <= 4, N*I*T <= 9216, I %% 4 == 0.
 * gg_sconv_fwd: 3x3 / stride 1 / pad 1 convolution of an NHWC bf16 activation with per-image banks (w_bs = elements between
 *   banks, 0 = shared) as a streaming direct convolution: y = act(conv(x * xs) + noise[b][pixel] * noise_w[o]); xs (optional,
 *   [b][C] fp32): a per-sample input-channel scale (the skip-layer excitation), applied to the bank as it is parked in LDS.
 *   W %% 32 == 0, C in {16, 32, 64}, O <= 32, O %% 8 == 0.
 * gg_modulate_bank_fwd: out[b][p][n*Cin + i] = x[b][p][i] * s[b][i] * a[b][n] for the N = Cout / Cin kernels of a bank in one pass
 *   (s is [b][Cin], a is [b][N]). */
int gg_modw_fwd(const float* w, const float* mod, int32_t mod_ld, const float* kmod, int32_t kmod_ld, const float* xs, int32_t xs_ld,
                float* s, float* a, float* d, void* wmix, int32_t layout, int32_t b, int32_t N, int32_t O, int32_t I, int32_t T,
                int32_t Ip, int32_t Op, int32_t demod, float eps, void* stream);
/* gg_modw_multi_fwd: the same work for up to 16 layers in ONE launch. Every adaptive conv of a generator forward is modulated by a
 * column slice of ONE style projection (`style_to_conv_modulations(styles).split(...)`, gp.py:1160-1175, unet_upsampler.py:700-
 * 706), so all coefficients / per-sample weights of a forward are known before its first convolution: one launch at the top of the
 * forward replaces one per layer (15 on config 2). `insc` (optional, (b, N*Ip) fp32 out): a[b,n] * s[b,i] over the stacked channel
 * axis (n, i) - the input scale of the shared-bank convolution with the N kernels stacked along the reduction (gg_gemm_bf16 with
 * in_scale, CV = N*C). Same limits per item as gg_modw_fwd. */
typedef struct gg_modw_item {
    const float* w; const float* mod; const float* kmod; const float* xs;
    const float* gram;      /* optional [pair][O][I] fp32: the bank's Gram rows as refreshed by gg_pack_weights (kind 2) */
    float* s; float* a; float* d; float* insc; void* wmix;
    int32_t mod_ld, kmod_ld, xs_ld, layout;
    int32_t b, N, O, I, T, Ip, Op, demod;
    float eps;
    int32_t reserved;
} gg_modw_item;
int gg_modw_multi_fwd(const gg_modw_item* items, int32_t n_items, void* stream);
int gg_sconv_fwd(const void* x, const void* w, int64_t w_bs, void* y, const float* noise, const float* noise_w, const float* xs,
                 int32_t b, int32_t H, int32_t W, int32_t C, int32_t O, int32_t act, float slope, void* stream);
int gg_modulate_bank_fwd(const void* x, const float* s, const float* a, void* out, int32_t b, int32_t P, int32_t Cin, int32_t Cout,
                         void* stream);

/* ---- gg_spair_fwd (ABI 12): the two adaptive 3x3 convolutions of one generator block in ONE launch (csrc/gg_spair.h): reference
 * Generator.forward's resnet block gp.py:1219-1229 (conv1 -> Noise gp.py:925-940 -> leaky_relu gp.py:109 -> conv2 -> Noise ->
 * leaky_relu) on the per-sample kernels of AdaptiveConv2DMod.forward gp.py:378-409 as gg_modw_fwd / gg_modw_multi_fwd write them
 * (layout 2), at the 128x128 / 256x256 stages where the layers are HBM-bound:
 *     mid = act1(conv3x3(x * xs, w1[b]) + noise1[b][p] * nw1[c])   (rounded to bf16, kept in LDS)
 *     y   = act2(conv3x3(mid,    w2[b]) + noise2[b][p] * nw2[c])
 * x (b, H, W, C0) bf16 NHWC, w1 [b][9][C0/16][32][16], w2 [b][9][C1/16][32][16] (w*_bs = elements between banks, 0 = shared),
 * y (b, H, W, C2) bf16; noise maps [b][H*W] fp32 (16-byte aligned; each goes with its weights or is null), xs optional [b][C0] (the
 * skip-layer excitation, gp.py:1023-1024). W 128: bit-identical to gg_sconv_fwd(gg_sconv_fwd(x)); W 256 with C2 <= 16 runs on
 * v_mfma_f32_16x16x32_bf16 (16 output channels = one row block): the same products in another summation order - equal to bf16 rounding,
 * exact on integer operands. Geometries: gg_spair_supported
 * (W 256 / C0 32 / C1 16 and W 128 / C0 64 / C1 32, C2 <= 32, C2 %% 8 == 0); anything else: -2. */
int gg_spair_supported(int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t C2);
int gg_spair_fwd(const void* x, const void* w1, int64_t w1_bs, const void* w2, int64_t w2_bs, void* y, const float* noise1,
                 const float* nw1, const float* noise2, const float* nw2, const float* xs, int32_t b, int32_t H, int32_t W, int32_t C0,
                 int32_t C1, int32_t C2, int32_t act1, int32_t act2, float slope, void* stream);

/* ---- gg_aconv_fwd (ABI 10): the no-grad adaptive convolution on a SHARED kernel bank in one launch (csrc/gg_aconv.h): reference
 * AdaptiveConv2DMod.forward gp.py:344-409 (softmax-mixed bank, style modulation, demodulation) + Noise gp.py:925-940 + leaky_relu
 * gp.py:109 at the generator's 4x4 .. 64x64 stages (gp.py:1184-1245):
 *   y[b,p,o] = act( d[b,o] * sum_n a[b,n] sum_{i,t} W_n[o,i,t] (s[b,i] xs[b,i] x[b,p+t,i]) + noise[b,p] * noise_w[o] )
 * x (b, H, W, C) / y (b, H, W, O) NHWC bf16; wf = the bank in MFMA-fragment order [O/32][NB][9][C/16][64][8] bf16 as written by
 * gg_pack_weights (entry kind 3); s (b, C) = mod + 1, xs (b, C) optional (skip-layer excitation), a (b, NB) = softmax(kernel_mod)
 * (may be null for NB == 1), d (b, O) optional demodulation coefficients (gg_modw_fwd / gg_modw_multi_fwd produce s, a, d), noise
 * (b, H*W) with noise_w (O) optional - all fp32. H == W a power of two in 4..64, C a power of two in 16..512, O %% 32 == 0, NB 1 or 2.
 * The library picks the tile (TM x 32 pixels, NWN x 32 output channels per workgroup); force_tm / force_nwn (0 = free) pin it for
 * measurements; gg_aconv_plan reports the choice (and is the eligibility test: non-zero = this layer cannot run here). No workspace. */
typedef struct gg_aconv_desc {
    const void* x; const void* wf; void* y;
    const float* s; const float* xs; const float* a; const float* d; const float* noise; const float* noise_w;
    int32_t b, H, W, C, O, NB;
    int32_t act;            /* 0 none, 1 leaky relu */
    float slope;
    int32_t force_tm, force_nwn;
    /* optional: the bank of the NEXT gg_aconv_fwd launch on this stream (fragment order) and that launch's geometry. Wavefronts that
     * have finished their share of this layer request it while the rest of the workgroup finishes, each XCD the slice its own
     * workgroups of the next launch will stream - the next layer then finds its weights in the L2 instead of in HBM. Pure hint. */
    const void* next_wf;
    int32_t next_b, next_H, next_C, next_O, next_NB;
    int32_t reserved;
} gg_aconv_desc;
int gg_aconv_plan(const gg_aconv_desc* d, int32_t* tm, int32_t* nwn, int32_t* nwk, int32_t* lds_bytes, int32_t* grid);
int gg_aconv_fwd(const gg_aconv_desc* d, void* stream);

/* ---- hipGraph repair. The training step is replayed as a captured hipGraph (the reference has no counterpart: it launches eagerly,
 * gp.py:2227-2580). The HIP runtime PyTorch 2.10+rocm7.0 carries (7.0.51831) re-executes a captured hipMemsetAsync with a corrupted
 * value from the second replay on; PyTorch's split reductions clear their semaphores with one (ATen/native/cuda/Reduce.cuh:1301), so a
 * captured column sum over >= 1024 columns is garbage on every replay but the first (tests/gpu_graph_memset_probe.py,
 * tests/gpu_graph_sum_probe.py). gg_graph_patch_memsets walks a captured, not yet instantiated hipGraph_t and puts a kernel node that
 * writes the intended value behind every memset node (dependents of the memset also depend on it). *n_patched = memset nodes found. */
int gg_graph_patch_memsets(void* hip_graph, int32_t* n_patched);

/* ---- data-parallel exchange: RCCL over xGMI (replaces accelerate / DDP's gradient all-reduce, gp.py:1898-1908, :1987, and the
 * reference's all_gather, distributed.py:20-68). One communicator per process (one process per GPU). RCCL is bound at run time
 * from the librccl already mapped into the process (PyTorch-ROCm's); gg_comm_load(path) names a specific one. Collectives are
 * enqueued on `stream` (the host code uses a dedicated side stream with event fences); calls are serialised by the caller.
 * Errors: < 0 argument / state, 1000 + ncclResult_t for RCCL failures. dtype: 0 = fp32, 1 = bf16, 2 = bytes. */
int gg_comm_load(const char* librccl_path);
int gg_comm_unique_id(void* id128);                       /* rank 0: 128-byte ncclUniqueId to hand to every rank (host memory) */
int gg_comm_init(int32_t rank, int32_t world, const void* id128);
int gg_comm_world(void);                                  /* ncclCommCount of the live communicator, 0 when there is none */
int gg_comm_allreduce(void* buf, size_t n, int32_t dtype, void* stream);      /* in-place sum over ranks */
int gg_comm_allgather(const void* send, void* recv, size_t n_per_rank, int32_t dtype, void* stream);
int gg_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* GIGAGAN_AMD_H */
