// Weight gradient of a 3x3 / stride 1 / pad 1 convolution (autograd of the F.conv2d call sites gigagan_pytorch.py:1608-1621,
// :1661-1668), all nine taps of a 32-input-channel slice in ONE workgroup:
//     dW[tap][ci][co] = sum over pixels of x[pixel + tap][ci] * dy[pixel][co]
// Output tile: (9 taps x 32 input channels) x 256 output channels, reduction over 64-pixel k-tiles (whole image rows), split-K over
// the pixels. Why: the implicit-GEMM weight gradient (gg_gemm2.h, reduction-major operands) gives every tap its own 256 x 256 tile,
// so each workgroup streams a 32 KB x tile AND a 32 KB dy tile per k-tile — 64 KB L2 -> LDS per 8.4 MFLOP, the same L2-rate
// bound the forward kernel had (gg_conv3.h). Here the dy tile is shared by the nine taps and the x operand is the k-tile's
// one-pixel halo of 32 channels (<= 198 slots x 64 bytes), from which a tap's fragments are transpose reads at a tap-uniform
// offset: ~40-45 KB per 9.4 MFLOP.
// Wave w owns output channels 32 w .. 32 w + 31 of the tile and all nine (tap, 32 ci) row blocks: 144 accumulator registers,
// per 16-pixel step one dy fragment and nine x fragments (two ds_read_b64_tr_b16 each) for nine MFMAs.
// Output layout = the implicit GEMM's: fp32 [tap * C + ci][co] (or split-K partials of it), so gg_splitk_reduce / gg_wgrad_finish
// are unchanged.
#pragma once
#include "gg_gemm2.h"

#define GG_W9_MAX_SLOTS 198                   // (64 / W + 2) * (W + 2) for W = 64
#define GG_W9_SLOT 64                         // bytes per halo slot: 32 channels
#define GG_W9_HBYTES 12800                    // GG_W9_MAX_SLOTS * GG_W9_SLOT rounded up to 256
#define GG_W9_HNV ((GG_W9_MAX_SLOTS * 4 + GG2_NT - 1) / GG2_NT)      // 16-byte halo vectors per thread: 2

GG_KERNEL GG_LAUNCH_BOUNDS(GG2_NT) void gg_wgrad9_kernel(GgGemmParams p) {
    constexpr int BN = 256;
    constexpr int BNV = Gg2KRow<BN>::NV, BBYTES = Gg2KRow<BN>::BYTES;
    static_assert(GG_W9_MAX_SLOTS * GG_W9_SLOT <= GG_W9_HBYTES, "halo area");

    GG_SHARED __attribute__((aligned(16))) char smem[2 * (BBYTES + GG_W9_HBYTES)];
    auto tileB = [&](int buf) { return smem + buf * (BBYTES + GG_W9_HBYTES); };
    auto haloT = [&](int buf) { return smem + buf * (BBYTES + GG_W9_HBYTES) + BBYTES; };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // XCD-aware flattened grid (as gg_gemm2_kernel): output tile fastest, so the tiles of one k-slice share an XCD's L2
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_mn = tiles_n * (p.C >> 5);
    const int bz = wg / tiles_mn, tile = wg - bz * tiles_mn;
    const int ci0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * BN;
    const int kbeg = bz * p.k_per_split;
    int kend = kbeg + p.k_per_split;
    if (kend > p.K) kend = p.K;

    // a k-tile = 64 consecutive pixels = PH full image rows (W <= 64, H * W >= 64: host); its halo: (PH + 2) x (W + 2) slots
    const int W = p.W, H = p.H, ws = p.w_shift, HW = H * W;
    const int PH = 64 >> ws, HWp = W + 2, NS = (PH + 2) * HWp;

    // the descriptor base sits (W + 1) pixels before the tensor: halo slot (hy, hx) of a tile whose first pixel is k0 is the
    // element offset (k0 + hy * W + hx) * C from it (the tile starts on an image row)
    const long long abias = (long long)(W + 1) * p.C;
    GgBuf bufA = gg_make_buf((const void*)(p.A - abias), (unsigned long long)(p.a_bytes + abias * 2));
    GgBuf bufB = gg_make_buf((const void*)p.B, (unsigned long long)p.b_bytes);

    unsigned hrel[GG_W9_HNV];
    int hyx[GG_W9_HNV];                      // hy | hx << 8 ; -1: no such slot
#pragma unroll
    for (int i = 0; i < GG_W9_HNV; ++i) {
        const int v = tid + GG2_NT * i;
        const int slot = v >> 2, ch = v & 3;
        const int hy = slot / HWp, hx = slot - hy * HWp;
        hyx[i] = slot < NS ? (hy | (hx << 8)) : -1;
        hrel[i] = (unsigned)((((long long)hy * W + hx) * p.C + ch * 8) * 2);
    }
    unsigned bvoff[BNV];
    gg2_bkrow_init<BN>(bvoff, p.ldb, p.N, n0);

    u16x8 ra[GG_W9_HNV], rb[BNV];
    auto load_tiles = [&](int k0) {
        const int row0 = (k0 & (HW - 1)) >> ws;              // image row of the tile's first pixel
        const unsigned soff = (unsigned)(((long long)k0 * p.C + ci0) * 2);
#pragma unroll
        for (int i = 0; i < GG_W9_HNV; ++i) {
            const int hy = hyx[i] & 255, hx = (hyx[i] >> 8) & 255;
            const bool in = hyx[i] >= 0 && (unsigned)(row0 + hy - 1) < (unsigned)H && (unsigned)(hx - 1) < (unsigned)W && k0 < kend;
            const unsigned ok = in ? 1u : 0u;
            ra[i] = gg_buf_load16(bufA, hrel[i] | (ok - 1u), soff);
        }
        gg2_bload_krow_dense<BN>(rb, bufB, bvoff, p.ldb, n0, kend, k0);
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < GG_W9_HNV; ++i) {
            const int v = tid + GG2_NT * i;
            if (hyx[i] >= 0) *(u16x8*)(haloT(buf) + (v >> 2) * GG_W9_SLOT + (v & 3) * 16) = ra[i];
        }
        gg2_store_krow<BN>(tileB(buf), rb);
    };

    // x fragment addressing (the transpose read of gg2_frag_krow with the k-row -> halo slot map): lane group g of 16 lanes
    // points at 4 pixels x 4 channel quads; pixel r of the tile sits at slot ((r >> ws) + kh) * HWp + (r & (W - 1)) + kw
    const int li = lane & 15, lg = lane >> 4;
    int pa[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = (q >> 1) * 16 + (lg >> 1) * 8 + (li >> 2) + 4 * (q & 1);
        pa[q] = ((r >> ws) * HWp + (r & (W - 1))) * GG_W9_SLOT + ((lg & 1) * 16 + 4 * (li & 3)) * 2;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nk = (kend > kbeg) ? (kend - kbeg + GG2_BK - 1) / GG2_BK : 0;
    if (nk > 0) {
        load_tiles(kbeg);
        store_tiles(0);
        if (nk > 1) load_tiles(kbeg + GG2_BK);
    }
    gg_sync();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        if (kt + 2 < nk) load_tiles(kbeg + (kt + 2) * GG2_BK);
        const char* hb = haloT(buf);
#pragma unroll
        for (int kk = 0; kk < GG2_BK / 16; ++kk) {
            const u16x8 fb = gg2_frag_krow<BN>(tileB(buf), wave * 32, kk, lane);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * HWp + (t % 3)) * GG_W9_SLOT;
                const u16x4 a = gg_lds_read_tr16((const bf16_t*)(hb + pa[2 * kk] + toff));
                const u16x4 b = gg_lds_read_tr16((const bf16_t*)(hb + pa[2 * kk + 1] + toff));
                const u16x8 fa = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                acc[t] = gg_mfma_32x32x16_bf16(fb, fa, acc[t]);       // rows: input channels, lane registers run along co
            }
        }
        gg_sync();
    }

    // lane owns row ci0 + (lane & 31) of every tap block; register r holds column (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int n_lane = n0 + wave * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const long long m = (long long)t * p.C + ci0 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n_lane + 8 * g;
            if (n < p.N) {                                     // N % 4 == 0 (host)
                if (p.splitk > 1) {
                    const f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                    *(f32x4*)(p.partial + ((long long)bz * p.M + m) * p.N + n) = v;
                } else {
                    const f32x4 v = {acc[t][4 * g] * p.alpha, acc[t][4 * g + 1] * p.alpha, acc[t][4 * g + 2] * p.alpha,
                                     acc[t][4 * g + 3] * p.alpha};
                    *(f32x4*)((float*)p.Cout + m * p.ldc + n) = v;
                }
            }
        }
    }
}
