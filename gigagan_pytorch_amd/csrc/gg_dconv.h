// gg_dconv.h — direct 3x3 / stride-1 / pad-1 convolution for the NARROW high-resolution layers (C_in in {16, 32, 64},
// C_out <= 64, 128x128 and 256x256 feature maps): the generator's last blocks and the discriminator's first blocks
// (reference gp.py:402-409 F.conv2d in AdaptiveConv2DMod, gp.py:1608-1621 DiscriminatorBlock convs, and their data
// gradients, which are the same convolution on flipped weights).
//
// Why not the implicit-GEMM kernels: with K = 9*C <= 576 and N <= 64 a (128 x N) tile performs 2 MFMAs per wave per
// barrier and gathers every input pixel 9 times through L1 - those launches measured 1.1-1.4 TB/s of algorithmic traffic
// (65-165 TFLOP/s), bound by the gather / barrier chain, not by HBM or MFMA. Here a workgroup
//   * keeps the whole filter bank in LDS for its lifetime (persistent workgroups, grid-stride over output tiles),
//   * stages an (8+2) x (32+2) pixel input tile WITH its halo in LDS once (each input byte is fetched once per tile;
//     halos come from L2) and forms all 9 taps from LDS by shifting the pixel index of the A fragment,
//   * prefetches the next tile into registers while the MFMAs of the current one run,
//   * writes the 8 x 32 x C_out outputs back through LDS as row-contiguous 16-byte vectors.
// One barrier pair per tile instead of one per 32 reduction elements. HBM-bound: (C_in + C_out) * 2 B per pixel.
//
// Fragment mapping (same swapped-operand issue as gg_gemm.h: the lane's accumulator registers run along n):
//   A: lane l -> pixel (row r + kh, col (l & 31) + kw) of the halo tile, channels kc*16 + 8*(l >> 5) .. +7
//   B: lane l -> filter row n = j*32 + (l & 31), reduction index tap*C + kc*16 + 8*(l >> 5) .. +7
// LDS pitches: pixel pitch C + 8, filter-row pitch 9*C + 8, output pitch N32 + 8 bf16 - all leave consecutive lanes 16,
// 48 or 80 bytes apart modulo 128: conflict-free ds_read_b128 / ds_write_b64.
#pragma once
#include "gg_gemm.h"

#define GG_DC_TH 8
#define GG_DC_TW 32

template <int C, int TN>
struct GgDconvLds {
    static constexpr int HW = GG_DC_TW + 2;
    static constexpr int NPIX = (GG_DC_TH + 2) * HW;          // 340 pixels with halo
    static constexpr int XP = C + 8;                          // pixel pitch (bf16)
    static constexpr int WP = 9 * C + 8;                      // filter row pitch
    static constexpr int OP = TN * 32 + 8;                    // staged output pitch
    static constexpr int X_ELEMS = NPIX * XP;
    static constexpr int O_ELEMS = GG_DC_TH * GG_DC_TW * OP;
    static constexpr int XO_ELEMS = X_ELEMS > O_ELEMS ? X_ELEMS : O_ELEMS;   // the output staging reuses the tile buffer
    static constexpr int W_ELEMS = TN * 32 * WP;
};

template <int C, int TN, bool FULL_EPI>
GG_KERNEL GG_LAUNCH_BOUNDS(256) void gg_dconv_kernel(GgGemmParams p) {
    using L = GgDconvLds<C, TN>;
    constexpr int TH = GG_DC_TH, TW = GG_DC_TW, HW = L::HW, NPIX = L::NPIX, XP = L::XP, WP = L::WP, OP = L::OP;
    constexpr int CV8 = C / 8;
    constexpr int NVX = (NPIX * CV8 + 255) / 256;
    constexpr int KV = 9 * C / 8;                             // 16-byte vectors per filter row

    GG_SHARED __attribute__((aligned(16))) bf16_t sX[L::XO_ELEMS];
    GG_SHARED __attribute__((aligned(16))) bf16_t sW[L::W_ELEMS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fk = (lane >> 5) * 8, hi = lane >> 5;

    for (int v = tid; v < TN * 32 * KV; v += 256) {           // the filter bank, once per workgroup
        const int n = v / KV, kc = v - n * KV;
        u16x8 x = gg_zero8();
        if (n < p.N) x = *(const u16x8*)(p.B + (long long)n * p.ldb + kc * 8);
        *(u16x8*)&sW[n * WP + kc * 8] = x;
    }

    const int tiles_w = p.W / TW, tiles_h = p.H / TH;
    const int tiles_img = tiles_w * tiles_h;
    const int total = (p.M / (p.H * p.W)) * tiles_img;

    // the halo tile through a buffer descriptor (gg_device.h GgBuf): a lane's byte offset inside the tile ((r * W + c) * C + chunk)
    // is fixed for the kernel's lifetime, the tile origin is one scalar, pixels outside the image get offset 0xFFFFFFFF and come
    // back as zeros (the pointer form spent ~45 vector instructions and a branch per 16-byte load)
    const long long xbias = ((long long)p.W + 1) * C;              // origin of tile (0, 0) is one row and one pixel before the image
    GgBuf bufX = gg_make_buf((const void*)(p.A - xbias), (unsigned long long)(p.a_bytes + xbias * 2));
    unsigned xoff[NVX];
    int xr[NVX], xc[NVX];
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
        const int v = tid + 256 * i;
        const int pix = v / CV8, c8 = v - pix * CV8;
        xr[i] = pix / HW;
        xc[i] = pix - xr[i] * HW;
        xoff[i] = (v < NPIX * CV8) ? (unsigned)((((long long)xr[i] * p.W + xc[i]) * C + c8 * 8) * 2) : 0xFFFFFFFFu;
    }
    u16x8 rx[NVX];
    int rx_img = 0;                 // image of the tile held in rx (for the style modulation applied when it is parked in LDS)
    auto load_tile = [&](int tile) {
        const int img = tile / tiles_img, rem = tile - img * tiles_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const int h0 = th * TH - 1, w0 = tw * TW - 1;
        const unsigned soff = (unsigned)(((((long long)img * p.H + h0) * p.W + w0) * C + xbias) * 2);
        rx_img = img;
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
            const unsigned ok = ((unsigned)(h0 + xr[i]) < (unsigned)p.H && (unsigned)(w0 + xc[i]) < (unsigned)p.W) ? 1u : 0u;
            rx[i] = gg_buf_load16(bufX, xoff[i] | (ok - 1u), soff);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
            const int v = tid + 256 * i;
            if (v < NPIX * CV8) {
                const int pix = v / CV8, c8 = v - pix * CV8;
                u16x8 x = rx[i];
                if (p.in_scale) x = gg_scale8(x, p.in_scale + (long long)rx_img * C + c8 * 8);   // style modulation (zeros stay zeros)
                *(u16x8*)&sX[pix * XP + c8 * 8] = x;
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < total) load_tile(tile);
    for (; tile < total; tile += gridDim.x) {
        gg_sync();                      // the previous tile's staged outputs have been written back (and sW is complete)
        store_tile();
        gg_sync();
        const int next = tile + gridDim.x;
        if (next < total) load_tile(next);          // in flight while the MFMAs below run

        f32x16 acc[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int kc = 0; kc < C / 16; ++kc) {
                u16x8 fa[2], fb[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *(const u16x8*)&sW[(j * 32 + frow) * WP + tap * C + kc * 16 + fk];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[i] = *(const u16x8*)&sX[((2 * wave + i + kh) * HW + frow + kw) * XP + kc * 16 + fk];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = gg_mfma_32x32x16_bf16(fb[j], fa[i], acc[i][j]);
            }
        }
        gg_sync();                      // every wave is done reading the input tile: reuse it as the output staging area

        const int img = tile / tiles_img, rem = tile - img * tiles_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 2 * wave + i;
            const int m = (img * p.H + th * TH + r) * p.W + tw * TW + frow;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = j * 32 + 8 * g + 4 * hi;
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = 0.f;
                        if (n + e < p.N) v = FULL_EPI ? gg_epilogue(p, acc[i][j][g * 4 + e], m, n + e) : acc[i][j][g * 4 + e] * p.alpha;
                        o[e] = gg_f2bf(v);
                    }
                    *(u16x4*)&sX[(r * TW + frow) * OP + n] = o;
                }
            }
        }
        gg_sync();
        const int n8 = p.N >> 3;                    // 16-byte vectors per output pixel (N % 8 == 0)
        for (int v = tid; v < TH * TW * n8; v += 256) {
            const int pix = v / n8, c8 = v - pix * n8;
            const int r = pix / TW, c = pix - r * TW;
            const long long m = ((long long)img * p.H + th * TH + r) * p.W + tw * TW + c;
            *(u16x8*)((bf16_t*)p.Cout + m * p.ldc + c8 * 8) = *(const u16x8*)&sX[pix * OP + c8 * 8];
        }
    }
}
