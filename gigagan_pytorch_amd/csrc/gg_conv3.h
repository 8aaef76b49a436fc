// 3x3 / stride 1 / pad 1 convolution (forward and data gradient of the discriminator's and the plain generator convolutions,
// reference gigagan_pytorch.py:1597-1618 block convs, :1661-1668 predictor convs) on 64-channel-multiple inputs, 256 pixels x 256
// output channels per workgroup, with the activation staged ONCE per 64-channel chunk for all nine taps.
//
// Why: the implicit GEMM (gg_gemm2.h) reloads the 256 x 64 activation tile for every tap — 64 KB per workgroup and k-tile
// through L2 -> LDS. At 256 workgroups that is 8.6-10.9 TB/s when the kernel runs at 1100 TFLOP/s, which is what the L2 delivers
// (the 256x128 tile tops out at 930 TFLOP/s = the same byte rate; register staging vs. LDS-DMA staging measured the same,
// profiles/r02_dma_ab_{off,on}.log): the tile is fed at the L2's rate, not the matrix pipe's. Here the workgroup's 256 pixels are whole
// image rows (or whole images), their one-pixel halo is parked in LDS per channel chunk (<= 400 slots x 128 bytes instead of
// 9 x 256 x 128), and the nine taps read their fragments from it at a tap-uniform offset; only the weight tiles stream per
// tap (LDS-DMA into XOR-swizzled rows: Gg2Dma in gg_gemm2.h). L2 -> LDS bytes per 9 k-tiles: 339 KB instead of 576 KB.
//
// Reduction order: (channel chunk, tap, channel) instead of (tap, channel): fp32 sums differ from the implicit GEMM in the last
// bits.
//
// Round 3 — what the generator's no-grad adaptive convolutions (gp.py:344-409) need from it:
//  * split-K over the channel chunks (p.splitk slices of p.k_per_split channels, fp32 partials finished by gg_splitk_reduce*):
//    at batch 32 the 8x8 / 16x16 layers have 8 / 32 row tiles - a chip of 256 CUs needs the reduction spread as well;
//  * a virtual channel axis CV = N * C (the N kernels of the bank stacked along the reduction, weights [co][tap][n][ci]) with a
//    per-(image, virtual channel) scale a[b,n] * s[b,i] applied ONCE per staged halo chunk (SCALED): the style modulation and
//    kernel mix ride on the operand load instead of a separate pass that writes the N-fold modulated activation (the implicit
//    GEMM applies such a scale per tap - nine times the work, measured slower than the separate pass; here it is 1/9 of that);
//  * per-image weight operands (p.b_img_stride: the reference's per-sample weights) when a tile lies inside one image.
#pragma once
#include "gg_gemm2.h"

#define GG_C3_MAX_SLOTS 400
#define GG_C3_MAX_ROWS 40                                  // halo rows of a tile (host-checked with the slot count)
#define GG_C3_PITCH 144                                   // bytes per halo slot: 64 channels + 16 bytes pad (consecutive slots: conflict-free)
// Every halo row is followed by 224 bytes: a fragment's 32 lanes are 32 consecutive pixels, which on 8- and 16-wide images wrap
// into the next image row, W + 2 slots further. (W + 2) * 144 + 224 = 144 W mod 256, so the wrapped lanes continue the bank
// sequence of the lanes before them as if the slots were consecutive.
#define GG_C3_ROWPAD 224
#define GG_C3_HBYTES 67584                                // largest halo (8x8 images: 400 slots, 40 rows) rounded up to 1 KB
#define GG_C3_NVH ((GG_C3_MAX_SLOTS * 8 + GG2_NT - 1) / GG2_NT)    // 16-byte halo vectors per thread: 7
#define GG_C3_SC_FLOATS 4096                              // SCALED: (images of a tile) x (channels of a k-slice) scale values in LDS

// PAIR (round 6; the 64-column tile on images of 32 / 64 pixels a side, instantiated as BN_ = GG_C3_BN64_PAIR so that the other
// instantiations keep their names and code): two workgroups per CU - the halo area cut to what ten halo rows need (59 KB instead of
// 66), one tap per weight buffer (16 KB instead of 48): 75 KB of LDS, <= 128 registers. One workgroup's prologue, epilogue and barriers
// overlap the other's tap loop.
#define GG_C3_HBYTES_WIDE 60416                           // 400 slots x 144 + 10 rows x 224, rounded up to 1 KB
#define GG_C3_BN64_PAIR 1064
#define GG_C3_SC_FLOATS_PAIR 512
template <int BN_, int WM, int WN, bool FULL_EPI, bool SCALED = false>
GG_KERNEL GG_LAUNCH_BOUNDS2(GG2_NT, BN_ >= 1000 ? 4 : 2) void gg_conv3_kernel(GgGemmParams p) {
    constexpr bool PAIR = BN_ >= 1000;
    constexpr int BN = BN_ % 1000;
    static_assert(!PAIR || BN == 64, "the paired form carries the 64-column tile");
    static_assert(WM * WN == 8, "8 wavefronts per workgroup");
    constexpr int BM = 256;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int BNV = Gg2Dma<BN>::NV, BBYTES = Gg2Dma<BN>::BYTES;
    constexpr int SP = WTN * 2 + 8;
    constexpr int SC_BYTES = SCALED ? (BN_ >= 1000 ? GG_C3_SC_FLOATS_PAIR : GG_C3_SC_FLOATS) * 4 : 0;     // (PAIR: one image, <= 512 channels per k-slice: host)
    // taps per barrier interval: the 64-column tile does 8 MFMAs per wave and tap - a whole kernel row (3 taps, 24 MFMAs) rides on one
    // weight transfer + barrier there (round 6, same results bit for bit; the adaptive 64x64 layers on per-sample weights 48.5-50.4 -> 47.0 us
    // and 35.7 -> 32.1 us across boxes, profiles/r06_conv3_row_of_taps.log: the barrier count was not what bounds this tile)
    constexpr int U = (BN == 64 && !PAIR) ? 3 : 1;
    constexpr int HBYTES = PAIR ? GG_C3_HBYTES_WIDE : GG_C3_HBYTES;
    constexpr int WBUF = U * BBYTES;
    constexpr int TILE_BYTES = HBYTES + 2 * WBUF + SC_BYTES, STAGE_BYTES = 8 * WTM * SP;
    static_assert(GG_C3_MAX_SLOTS * GG_C3_PITCH + GG_C3_MAX_ROWS * GG_C3_ROWPAD <= GG_C3_HBYTES, "halo area");

    GG_SHARED __attribute__((aligned(1024))) char smem[TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES];
    char* const halo = smem;
    auto tileB = [&](int buf) { return smem + HBYTES + buf * WBUF; };
    float* const scl = (float*)(smem + HBYTES + 2 * WBUF);      // SCALED only

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: as gg_gemm2_kernel (flattened grid, output tile fastest, then the k-slice)
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_mn = tiles_n * ((p.M + BM - 1) / BM);
    const int ks = wg / tiles_mn, tile = wg - ks * tiles_mn;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // k-slice: channel chunks [c_lo, c_hi) of the virtual channel axis (CV = C, or N * C for a stacked bank)
    const int nchunks_all = p.CV / GG2_BK;
    const int cps = p.splitk > 1 ? p.k_per_split / GG2_BK : nchunks_all;
    const int c_lo = ks * cps;
    const int c_hi = c_lo + cps < nchunks_all ? c_lo + cps : nchunks_all;

    // tile geometry: PH full-width rows of TI images (H, W powers of two, 8 <= W <= 64: host)
    const int W = p.W, H = p.H, ws = p.w_shift, hs = p.hw_shift;
    const int HW = H * W;
    const int PH = HW >= BM ? BM >> ws : H;
    const int TI = HW >= BM ? 1 : BM >> hs;
    const int HWp = W + 2, SPI = (PH + 2) * HWp;                 // halo row length, halo slots per image
    const int NS = TI * SPI;
    const int RSB = HWp * GG_C3_PITCH + GG_C3_ROWPAD;            // bytes from one halo row to the next
    const int img0 = m0 >> hs, row0 = HW >= BM ? (m0 & (HW - 1)) >> ws : 0;
    const int n_img = p.M >> hs;

    GgBuf bufA = gg_make_buf((const void*)p.A, (unsigned long long)p.a_bytes);
    GgBuf bufB = gg_make_buf((const void*)p.B, (unsigned long long)p.b_bytes);

    // halo loader: vector v = t + 512 i -> slot v / 8, 8-channel piece v % 8 (loop invariant; 0xFFFFFFFF = padding -> zero fill)
    unsigned hvoff[GG_C3_NVH];
#pragma unroll
    for (int i = 0; i < GG_C3_NVH; ++i) {
        const int v = tid + GG2_NT * i;
        const int slot = v >> 3, ch = v & 7;
        hvoff[i] = 0xFFFFFFFFu;
        if (slot < NS) {
            const int il = slot / SPI, rem = slot - il * SPI;
            const int hy = rem / HWp, hx = rem - hy * HWp;
            const int ih = row0 + hy - 1, iw = hx - 1, img = img0 + il;
            if (ih >= 0 && ih < H && iw >= 0 && iw < W && img < n_img)
                hvoff[i] = (unsigned)(((((long long)img * H + ih) * W + iw) * p.C + ch * 8) * 2);
        }
    }
    u16x8 hreg[GG_C3_NVH];
    // the physical channel of virtual channel cv is cv % C: a 64-wide chunk never wraps (C % 64 == 0, host)
    auto load_halo = [&](int c) {
        const int cv0 = c * GG2_BK;
        const unsigned soff = (unsigned)((p.CV == p.C ? cv0 : cv0 % p.C) * 2);
#pragma unroll
        for (int i = 0; i < GG_C3_NVH; ++i) hreg[i] = gg_buf_load16(bufA, hvoff[i], soff);
    };
    // SCALED: offset of the vector's 8 scale values inside `scl` for chunk c_lo (image-in-tile * channels of the slice + piece)
    const int nsc = (c_hi - c_lo) * GG2_BK;
    int hsc[SCALED ? GG_C3_NVH : 1];
    if constexpr (SCALED) {
#pragma unroll
        for (int i = 0; i < GG_C3_NVH; ++i) {
            const int v = tid + GG2_NT * i;
            const int slot = v >> 3;
            hsc[i] = (slot < NS ? (slot / SPI) * nsc : 0) + (v & 7) * 8;
        }
        for (int idx = tid; idx < TI * nsc; idx += GG2_NT) {
            const int il = idx / nsc, j = idx - il * nsc;
            const int img = img0 + il;
            scl[idx] = img < n_img ? p.in_scale[(long long)img * p.CV + c_lo * GG2_BK + j] : 0.f;
        }
        gg_sync();
    }
    auto store_halo = [&](int c) {
#pragma unroll
        for (int i = 0; i < GG_C3_NVH; ++i) {
            const int v = tid + GG2_NT * i;
            const int slot = v >> 3, hrow = slot / HWp;         // (halo rows run on across the tile's images)
            if (slot < NS) {
                u16x8 h = hreg[i];
                if constexpr (SCALED) {
                    const float* sp = scl + hsc[i] + (c - c_lo) * GG2_BK;
                    const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] = gg_f2bf(gg_bf2f(h[e]) * s0[e]);
                        h[e + 4] = gg_f2bf(gg_bf2f(h[e + 4]) * s1[e]);
                    }
                }
                *(u16x8*)(halo + hrow * RSB + (slot - hrow * HWp) * GG_C3_PITCH + (v & 7) * 16) = h;
            }
        }
    };

    // weight tiles: LDS-DMA into XOR-swizzled 128-byte rows (Gg2Dma, gg_gemm2.h)
    unsigned bvoff[BNV];
#pragma unroll
    for (int i = 0; i < BNV; ++i) {
        const int row = gg2d_row<BN>(i), kc = gg2d_chunk<BN>(i);
        bvoff[i] = (n0 + row < p.N) ? (unsigned)(((long long)row * p.ldb + kc * 8) * 2) : 0xFFFFFFFFu;
    }
    // per-image weight operand (the adaptive conv's per-sample weights): a tile lies inside one image then (TI == 1, host)
    const long long b_img = p.b_img_stride ? (long long)img0 * p.b_img_stride : 0;
    auto dma_b = [&](int buf, int tap, int c) {              // the weight tiles of taps tap .. tap + U - 1
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned soff = (unsigned)((b_img + (long long)n0 * p.ldb + (tap + u) * p.CV + c * GG2_BK) * 2);
            char* lb = tileB(buf) + u * BBYTES + wave * (BNV * 1024);
#pragma unroll
            for (int i = 0; i < BNV; ++i) gg_buf_load_lds16(bufB, bvoff[i], soff, lb + i * 1024);
        }
    };

    // fragment addressing. A: pixel row r of the tile -> its halo slot at tap (0, 0); a tap adds kh halo rows and kw slots
    const int frow = lane & 31, fhi = lane >> 5;
    int a_addr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WTM + i * 32 + frow;
        const int il = HW >= BM ? 0 : r >> hs;
        const int rr = HW >= BM ? r : r & (HW - 1);
        a_addr[i] = (il * (PH + 2) + (rr >> ws)) * RSB + (rr & (W - 1)) * GG_C3_PITCH + fhi * 16;
    }
    const int fx = (fhi ^ ((frow >> 1) & 7)) * 16;
    const int b_lane = (wn * WTN + frow) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = c_hi;
    load_halo(c_lo);
    dma_b(0, 0, c_lo);
    store_halo(c_lo);
    gg_wait_vm<0>();
    gg_sync();

    int step = 0;
    for (int c = c_lo; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_halo(c + 1);              // lands in registers while this chunk's nine taps run
        int toff = 0;                                        // byte offset of the tap inside the halo
        for (int kh = 0; kh < 3; ++kh) {
            for (int kw0 = 0; kw0 < 3; kw0 += U, ++step) {
                const int buf = step & 1;
                const int tap = kh * 3 + kw0;
                // the other weight buffer was released by the barrier that ended the previous interval
                if (tap + U < 9) dma_b(buf ^ 1, tap + U, c);
                else if (c + 1 < nchunks) dma_b(buf ^ 1, 0, c + 1);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const char* ta = halo + toff;
                    const char* tb = tileB(buf) + u * BBYTES + b_lane;
#pragma unroll
                    for (int kk = 0; kk < GG2_BK / 16; ++kk) {
                        u16x8 fa[TM], fb[TN];
                        const int fo = fx ^ (kk * 32);
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i] = *(const u16x8*)(ta + a_addr[i] + kk * 32);
#pragma unroll
                        for (int j = 0; j < TN; ++j) fb[j] = *(const u16x8*)(tb + j * 32 * 128 + fo);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = gg_mfma_32x32x16_bf16(fb[j], fa[i], acc[i][j]);
                    }
                    toff += GG_C3_PITCH;
                }
                gg_wait_vm<0>();
                gg_sync();
            }
            toff += RSB - 3 * GG_C3_PITCH;
        }
        if (c + 1 < nchunks) {      // every wave is past its last read of this chunk's halo (barrier above)
            store_halo(c + 1);
            gg_sync();
        }
    }

    const GgGemmParams e = *gg_late_params(p);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int m_wave = m0 + wm * WTM, n_wave = n0 + wn * WTN;
    const bool staged = e.splitk == 1 && !e.c_f32 && !e.d2s && (e.N & 3) == 0 && (e.ldc & 3) == 0 &&
                        (!e.residual || (e.ldr & 3) == 0);
    if (staged) {
        char* stage = smem + wave * (WTM * SP);
        gg2_epilogue_step<0, TM, TN, FULL_EPI, true>(acc, e, 0, 0, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), stage, SP, lane,
                                                     z4, z4);
        gg_sync();
        gg2_stage_writeback<WTM, WTN>(e, 0, stage, SP, m_wave, n_wave, lane);
    } else {      // fp32 outputs and split-K partials (slice ks of [splitk][M][N]): direct quad stores
        gg2_epilogue_step<0, TM, TN, FULL_EPI, false>(acc, e, 0, ks, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), nullptr, 0, lane,
                                                      z4, z4);
    }
}

// Round 5, measured and not shipped: the same kernel with the weight tiles in a RING of 3 - 4 buffers (transfers issued from inline
// assembly R - 1 taps ahead, counted vmcnt waits, raw barriers - the gg_wgrads recipe). Bit-identical results, 7 - 10 % SLOWER on every
// discriminator shape (D4.conv2 231 -> 249 us, D3.conv2 139 -> 160 us: profiles/r05_conv3_ring_ab.log): the tap loop is not waiting for
// its weight tile's latency, so more lead buys nothing and the extra buffer and wait bookkeeping cost. The code was deleted.
