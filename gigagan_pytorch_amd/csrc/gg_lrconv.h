// gg_lrconv.h — 3x3 / stride 1 / pad 1 convolution of LOW-RESOLUTION feature maps (4x4, 8x8, 16x16 images) against a WIDE shared
// weight bank: the generator's first adaptive convolutions (reference AdaptiveConv2DMod.forward, gp.py:344-409, at the 4x4 .. 16x16
// stages of Generator.forward, gp.py:1184-1245; 512 channels, N kernels stacked along the reduction, plan tile 11).
//
// Why its own kernel: at batch 32 these layers are 512 .. 8192 output pixels against 9.4 MB of weights. The weights are the
// traffic (each element is used by only 2 .. 32 pixel tiles) and the launch is latency, not throughput: the halo-staged kernel
// (gg_conv3.h) streams one weight tile per tap behind a workgroup barrier - 16 MFMAs per wave between barriers, with the tile
// coming from HBM rather than from an L2 shared by hundreds of pixel tiles - and measured 45-47 us on the 8x8 layers (10 GF) and,
// through the implicit GEMM + a modulation pass, 39 us on the 4x4 layers (2.4 GF). Here a workgroup owns 256 pixels (whole images)
// x 64 output channels x a slice of the channel axis and walks it in chunks of 32 channels x ALL NINE TAPS: the chunk's activation
// halo (<= 57 KB, zero border written once) and its 64 x 288 weight block (36 KB) are fetched into registers while the previous
// chunk is multiplied (every load of a chunk in flight together), parked in LDS, and read by 72 MFMAs per wave between two
// barriers. The per-(image, virtual channel) scale a[b,n] * s[b,i] of a stacked bank rides on the halo store as in gg_conv3's
// SCALED form. Partial sums of the channel slices go to the fp32 workspace and are finished by gg_splitk_reduce* (demodulation,
// noise, activation).
//
// LDS layout. Halo: slot = 32 channels + 16 bytes (80-byte pitch: 16 consecutive slots cover the 64 banks exactly once);
// a halo row of W + 2 slots is followed by 96 bytes and an image by (256 - 2 RSB mod 256) bytes, which makes the 32 consecutive
// pixels of a fragment continue the bank sequence across image rows and images as if their slots were consecutive.
// Weights: [tap][64 rows][64 bytes], the four 16-byte pieces of a row XOR-swizzled with (row >> 2) & 3 (rows four apart share
// banks and get different slots).
#pragma once
#include "gg_gemm2.h"

#define GG_LR_NT 256
#define GG_LR_BM 256
#define GG_LR_BN 64
#define GG_LR_KC 32                                   // virtual channels per chunk
#define GG_LR_PITCH 80
#define GG_LR_ROWPAD 96
#define GG_LR_WBYTES (9 * GG_LR_BN * 64)              // one chunk of weights: 36864 bytes

// halo geometry shared by host and device: bytes from one halo row to the next, from one image to the next
GG_HOST_DEVICE int gg_lr_row_bytes(int W) { return (W + 2) * GG_LR_PITCH + GG_LR_ROWPAD; }
GG_HOST_DEVICE int gg_lr_image_bytes(int H, int W) {
    const int rsb = gg_lr_row_bytes(W);
    return (H + 2) * rsb + ((256 - ((2 * rsb) & 255)) & 255);
}

// MIX > 0 (16x16 images: a tile is ONE image): the MIX banks of a stacked bank are combined per image while their weight blocks are
// parked in LDS, w = sum_n bank_mix[img][n] * W_n (one fma per element in the staging registers), the activation carries the
// style modulation s[img][i] alone and the reduction runs over the C physical channels - the adaptive conv's algorithmic flops
// instead of MIX times that (the stacked form multiplies every bank separately).
template <int HB, int SCF, bool FULL_EPI, int MIX = 0>
GG_KERNEL GG_LAUNCH_BOUNDS(GG_LR_NT) void gg_lrconv_kernel(GgGemmParams p) {
    constexpr int BM = GG_LR_BM, BN = GG_LR_BN, KC = GG_LR_KC;
    constexpr int TM = 2, TN = 2;                      // four wavefronts stacked along the pixels: 64 pixels x 64 channels each
    GG_SHARED __attribute__((aligned(1024))) char smem[HB + GG_LR_WBYTES + SCF * 4];
    char* const halo = smem;
    char* const wl = smem + HB;
    float* const scl = (float*)(smem + HB + GG_LR_WBYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // XCD-aware tile order (as gg_gemm2_kernel): flattened grid, output tile fastest, then the channel slice
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_mn = tiles_n * ((p.M + BM - 1) / BM);
    const int ks = wg / tiles_mn, tile = wg - ks * tiles_mn;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    constexpr int NB = MIX > 0 ? MIX : 1;                // weight blocks fetched per chunk
    const int KV = MIX > 0 ? p.C : p.CV;                 // length of the reduction's channel axis
    const int nchunks_all = KV / KC;
    const int cps = p.splitk > 1 ? p.k_per_split / KC : nchunks_all;
    const int c_lo = ks * cps;
    const int c_hi = c_lo + cps < nchunks_all ? c_lo + cps : nchunks_all;
    const int nsc = (c_hi - c_lo) * KC;

    // tile geometry: TI whole images of H x W <= 256 pixels (H, W powers of two: host)
    const int W = p.W, H = p.H, ws = p.w_shift, hs = p.hw_shift;
    const int HW = H * W;
    const int TI = BM >> hs;
    const int RSB = gg_lr_row_bytes(W), IS = gg_lr_image_bytes(H, W);
    const int img0 = m0 >> hs, n_img = p.M >> hs;

    GgBuf bufA = gg_make_buf((const void*)p.A, (unsigned long long)p.a_bytes);
    GgBuf bufB = gg_make_buf((const void*)p.B, (unsigned long long)p.b_bytes);

    // activation loader: thread -> 16-byte piece tid & 3 of the pixels (tid >> 2) + 64 i; weights: tap i, row tid >> 2, piece tid & 3
    const int piece = tid & 3;
    unsigned hvoff[4];
    int hl[4], hsc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 2) + 64 * i;
        const int il = r >> hs, rr = r & (HW - 1);
        const int y = rr >> ws, x = rr & (W - 1);
        hl[i] = il * IS + (y + 1) * RSB + (x + 1) * GG_LR_PITCH + piece * 16;
        hsc[i] = il * nsc + piece * 8;
        hvoff[i] = m0 + r < p.M ? (unsigned)((((long long)(m0 + r)) * p.C + piece * 8) * 2) : 0xFFFFFFFFu;
    }
    const int wrow = tid >> 2;
    const int wbase = wrow * 64 + ((piece ^ ((wrow >> 2) & 3)) << 4);
    unsigned bvoff[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
        bvoff[i] = n0 + wrow < p.N ? (unsigned)((((long long)wrow) * p.ldb + (long long)i * p.CV + piece * 8) * 2) : 0xFFFFFFFFu;

    u16x8 hreg[4], wreg[NB][9];
    auto prefetch = [&](int c) {
        const int cv0 = c * KC;
        const unsigned sa = (unsigned)((KV == p.C ? cv0 : cv0 % p.C) * 2);
        const unsigned sb = (unsigned)((((long long)n0) * p.ldb + cv0) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) hreg[i] = gg_buf_load16(bufA, hvoff[i], sa);
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int i = 0; i < 9; ++i) wreg[n][i] = gg_buf_load16(bufB, bvoff[i], sb + (unsigned)(n * p.C * 2));
    };
    float mixv[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) mixv[n] = 1.f;
    auto stash = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u16x8 h = hreg[i];
            if (p.in_scale) {
                const float* sp = scl + hsc[i] + (c - c_lo) * KC;
                const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = gg_f2bf(gg_bf2f(h[e]) * s0[e]);
                    h[e + 4] = gg_f2bf(gg_bf2f(h[e + 4]) * s1[e]);
                }
            }
            *(u16x8*)(halo + hl[i]) = h;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            u16x8 w = wreg[0][i];
            if constexpr (MIX > 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = gg_bf2f(w[e]) * mixv[0];
#pragma unroll
                    for (int n = 1; n < NB; ++n) f += gg_bf2f(wreg[n][i][e]) * mixv[n];
                    w[e] = gg_f2bf(f);
                }
            }
            *(u16x8*)(wl + i * (BN * 64) + wbase) = w;
        }
    };

    prefetch(c_lo);
    {   // the zero border of the halo (interior slots are rewritten by every chunk) and the scale table of this channel slice
        const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int o = tid * 16; o < TI * IS; o += GG_LR_NT * 16) *(u16x8*)(halo + o) = z;
        if (p.in_scale)
            for (int idx = tid; idx < TI * nsc; idx += GG_LR_NT) {
                const int il = idx / nsc, j = idx - il * nsc;
                const int img = img0 + il;
                scl[idx] = img < n_img ? p.in_scale[(long long)img * KV + c_lo * KC + j] : 0.f;
            }
        if constexpr (MIX > 0) {        // one image per tile (host): its MIX bank weights, zero for a tile beyond the batch
#pragma unroll
            for (int n = 0; n < NB; ++n) mixv[n] = img0 < n_img ? p.bank_mix[(long long)img0 * MIX + n] : 0.f;
        }
    }
    gg_sync();

    // fragment addressing. Activation: pixel r of the tile -> its halo slot at tap (0, 0); a tap adds kh halo rows and kw slots
    const int frow = lane & 31, fhi = lane >> 5;
    int a_addr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wave * 64 + i * 32 + frow;
        const int il = r >> hs, rr = r & (HW - 1);
        a_addr[i] = il * IS + (rr >> ws) * RSB + (rr & (W - 1)) * GG_LR_PITCH + fhi * 16;
    }
    int b_addr[TN], b_sw[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = j * 32 + frow;
        b_addr[j] = row * 64;
        b_sw[j] = (row >> 2) & 3;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int c = c_lo; c < c_hi; ++c) {
        gg_wait_vm<0>();
        stash(c);
        gg_sync();
        if (c + 1 < c_hi) prefetch(c + 1);                  // lands in registers while this chunk's nine taps run
        int toff = 0;
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const char* ta = halo + toff + kw * GG_LR_PITCH;
                const char* tb = wl + (kh * 3 + kw) * (BN * 64);
#pragma unroll
                for (int kk = 0; kk < KC / 16; ++kk) {
                    u16x8 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = *(const u16x8*)(ta + a_addr[i] + kk * 32);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = *(const u16x8*)(tb + b_addr[j] + (((kk * 2 + fhi) ^ b_sw[j]) << 4));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = gg_mfma_32x32x16_bf16(fb[j], fa[i], acc[i][j]);
                }
            }
            toff += RSB;
        }
        gg_sync();                                          // every wave is past its last read of this chunk
    }

    const GgGemmParams e = *gg_late_params(p);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // fp32 partials (slice ks of [splitk][M][N]) or, unsplit, the finished rows: direct quad stores
    gg2_epilogue_step<0, TM, TN, FULL_EPI, false>(acc, e, 0, ks, m0 + wave * 64 + (lane & 31), n0 + 4 * (lane >> 5), nullptr, 0, lane,
                                                  z4, z4);
}
