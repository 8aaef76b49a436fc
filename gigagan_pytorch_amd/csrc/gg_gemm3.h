// gg_gemm3.h — EXPERIMENTAL k-loop for the 8-wave 256x256 tile (plan tile 7: only ever chosen by `force_tile = 7`, the
// launch planner never selects it): the structure the phase probe of gg_gemm2.h calls for. Written after
// this round's GPU budget was spent: index math and synchronisation ORDER are verified on the host emulator through the
// normal C ABI (tests/test_emulator_kernels.py), the asynchronous behaviour (counted vmcnt across raw barriers, LDS-DMA
// landing order) has not run on a GPU yet and must be race-screened before the planner may use it.
//
// What changes against gg_gemm2_kernel (row-major x row-major operands, dense or conv gather, K % 32 == 0):
//   * operand tiles go global -> LDS directly (`global_load_lds_dwordx4`, 1 KiB per wave instruction): no staging VGPRs,
//     no ds_write pass (830 LDS cycles per 64-k tile in gg_gemm2), and loads stay in flight ACROSS barriers;
//   * a ring of 4 stages of (256 + 256) rows x 32 k (32 KiB each): three stages = 96 KiB per CU in flight while the
//     fourth is consumed (gg_gemm2 holds one 64 KiB tile in flight, issued as a burst);
//   * waits are counted: a wave issues exactly 4 DMA instructions per stage, so `s_waitcnt vmcnt(8)` retires the stage
//     about to be read while two younger stages keep streaming; barriers are raw `s_barrier` (no vmcnt(0) drain);
//   * LDS rows are 64 bytes, unpadded (the DMA writes lane i at base + 16 i); bank conflicts are avoided by an XOR
//     swizzle of the 16-byte chunk index with (row >> 2) & 3, applied to the GLOBAL source address of the DMA and to the
//     ds_read_b128 address: the 16 lanes of every ds_read_b128 service group then hit 16 distinct 16-byte bank slots.
// Accumulator ownership, XCD-aware tile order, split-K contract and epilogue are gg_gemm2's (shared code).
#pragma once
#include "gg_gemm2.h"

#define GG3_BK 32
#define GG3_NS 4
#define GG3_ROWB 64                                  // bytes per LDS row (32 bf16)
#define GG3_STAGE ((256 + 256) * GG3_ROWB)           // 32 KiB
#define GG3_LDS 139264                               // max(ring = 4 * 32 KiB, gg_gemm2's epilogue staging: 8 * 128 * 136 B)

// STAGGER: the two wave rows (waves 0-3 / 4-7: one wave of each per SIMD) run half a stage apart - while one half issues its 16
// MFMAs of a stage (fragments already in registers, raised priority) the other half reads its fragments of the next stage from
// LDS, issues its DMA share and waits for the share it needs next, so the matrix pipe of every SIMD always has one wave feeding
// it (the ping-pong of the guide's 8-phase template). Two raw barriers per stage; the halves are offset by ONE barrier (the
// second half executes an extra barrier before the loop, the first half one after it). Hazards, with R(s) / M(s) the barriers
// ending a half's read / MFMA interval of stage s (physical barrier = first half's R(s) = second half's M(s-1)):
//   RAW  a wave waits (counted vmcnt) for ITS share of stage s+1 before R(s); every wave's R(s) precedes every read of s+1.
//   WAR  slot (s+3)%4 held stage s-1; the last reads of s-1 are issued before the second half's R(s-1) (lgkmcnt(0) precedes the
//        barrier) = the first half's M(s-1), and no wave issues the DMA of stage s+3 before its read interval of stage s.
template <bool A_CONV, bool FULL_EPI, bool STAGGER>
GG_KERNEL GG_LAUNCH_BOUNDS(GG2_NT) void gg_gemm3_kernel(GgGemmParams p) {
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    GG_SHARED __attribute__((aligned(16))) char smem[GG3_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware flattened grid, as gg_gemm2_kernel
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_mn = tiles_n * ((p.M + BM - 1) / BM);
    const int bz = wg / tiles_mn, tile = wg - bz * tiles_mn;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int b = bz / p.splitk, ks = bz % p.splitk;
    const int kbeg = ks * p.k_per_split;
    int kend = kbeg + p.k_per_split;
    if (kend > p.K) kend = p.K;
    const int nk = (kend > kbeg) ? (kend - kbeg) / GG3_BK : 0;          // whole stages only (K % 32 == 0)

    const bf16_t* Ab = p.A + (long long)b * p.a_bs;
    const bf16_t* Bb = p.B + (long long)b * p.b_bs;

    // DMA plan of this lane: stage = 512 rows x 4 chunks of 16 B = 32 wave instructions of 64 chunks; wave w owns
    // instructions 4w .. 4w+3. Chunk id -> stage row id / 4 (0..255: A rows, 256..511: B rows), LDS slot id % 4 of that
    // row; the slot holds source chunk (slot ^ swizzle(row)). Rows beyond M / N are clamped to the last valid row: they
    // only feed output rows / columns that are never stored.
    // Conv gather (A_CONV): the A rows are output pixels and k = (tap, cv), CV % 32 == 0, so a stage lies inside one tap:
    // per A row the element offset of its window corner and a tap-validity bitmask are fixed, the tap offset is a scalar per
    // stage; a padding tap loads the caller's zero page instead (same instruction count every stage, no LDS zero fill).
    // Waves 0..3 own the A instructions (stage rows 0..255), waves 4..7 the B (weight) instructions.
    const bf16_t* src[4];
    long long corner[4];
    unsigned int tapmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = (wave * 4 + i) * 64 + lane;
        const int row = id >> 2, slot = id & 3;
        const int chunk = slot ^ ((row >> 2) & 3);
        corner[i] = 0;
        tapmask[i] = 0;
        if (row < BM) {
            int r = m0 + row;
            if (r > p.M - 1) r = p.M - 1;
            if (A_CONV) {
                const int hw = p.OH * p.OW;
                const int img = r / hw, rem = r - img * hw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
                corner[i] = (((long long)img * p.H + ih0) * p.W + iw0) * p.C + chunk * 8;
                for (int kh = 0; kh < p.R; ++kh)
                    for (int kw = 0; kw < p.S; ++kw) {
                        const int ih = ih0 + kh, iw = iw0 + kw;
                        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) tapmask[i] |= 1u << (kh * p.S + kw);
                    }
                src[i] = p.A;
            } else {
                src[i] = Ab + (long long)r * p.lda + chunk * 8;
            }
        } else {
            int r = n0 + row - BM;
            if (r > p.N - 1) r = p.N - 1;
            src[i] = Bb + (long long)r * p.ldb + chunk * 8;
        }
    }
    auto issue_stage = [&](int kt) {                    // the 4 DMA instructions of this wave for k-stage kt
        char* base = smem + (kt % GG3_NS) * GG3_STAGE + wave * 4 * 1024;
        const int k0 = kbeg + kt * GG3_BK;
        if (A_CONV && wave < 4) {
            const int tap = k0 / p.CV, cv0 = k0 - tap * p.CV;
            const int kh = tap / p.S, kw = tap - kh * p.S;
            const long long off = ((long long)kh * p.W + kw) * p.C + ((p.CV == p.C) ? cv0 : cv0 % p.C);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16_t* g = ((tapmask[i] >> tap) & 1u) ? p.A + corner[i] + off : p.zero_page;
                gg_load_lds16(g, base + i * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) gg_load_lds16(src[i] + k0, base + i * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int s = 0; s < GG3_NS - 1 && s < nk; ++s) issue_stage(s);

    const int frow = lane & 31, hi = lane >> 5;
    if (STAGGER) {
        const int half = wave >> 2;
        if (nk > 2) gg_wait_vm<8>(); else if (nk > 1) gg_wait_vm<4>(); else gg_wait_vm<0>();
        gg_barrier_raw();                   // every wave's share of stage 0 has landed
        if (half == 1) gg_barrier_raw();    // the second half runs one barrier behind
        for (int kt = 0; kt < nk; ++kt) {
            // ---- read interval: DMA of stage kt+3 into the slot stage kt-1 left, fragments of stage kt into registers
            if (kt + GG3_NS - 1 < nk) issue_stage(kt + GG3_NS - 1);
            const char* stA = smem + (kt % GG3_NS) * GG3_STAGE;
            const char* stB = stA + BM * GG3_ROWB;
            u16x8 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = wm * WTM + i * 32 + frow;
                    fa[kk][i] = *(const u16x8*)(stA + r * GG3_ROWB + (((kk * 2 + hi) ^ ((r >> 2) & 3)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = wn * WTN + j * 32 + frow;
                    fb[kk][j] = *(const u16x8*)(stB + r * GG3_ROWB + (((kk * 2 + hi) ^ ((r >> 2) & 3)) << 4));
                }
            }
            // this wave's share of stage kt+1 must be in LDS before anybody reads it (shares kt+2, kt+3 may stay in flight)
            if (kt + 3 < nk) gg_wait_vm<8>();
            else if (kt + 2 < nk) gg_wait_vm<4>();
            else gg_wait_vm<0>();
            gg_barrier_raw();               // R(kt): lgkmcnt(0) first - the fragments are in registers, the slot may be refilled
            // ---- MFMA interval
            gg_setprio<1>();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = gg_mfma_32x32x16_bf16(fb[kk][j], fa[kk][i], acc[i][j]);
            gg_setprio<0>();
            gg_barrier_raw();               // M(kt)
        }
        if (half == 0) gg_barrier_raw();    // pairs with the second half's last M
    } else
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most the two younger stages (4 instructions each) are outstanding
        if (kt + 2 < nk) gg_wait_vm<8>();
        else if (kt + 1 < nk) gg_wait_vm<4>();
        else gg_wait_vm<0>();
        gg_barrier_raw();       // every wave's share of stage kt is in LDS; every wave has finished reading stage kt-1
        if (kt + GG3_NS - 1 < nk) issue_stage(kt + GG3_NS - 1);        // refills the slot that stage kt-1 occupied
        const char* stA = smem + (kt % GG3_NS) * GG3_STAGE;
        const char* stB = stA + BM * GG3_ROWB;
#pragma unroll
        for (int kk = 0; kk < GG3_BK / 16; ++kk) {
            u16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * WTM + i * 32 + frow;
                fa[i] = *(const u16x8*)(stA + r * GG3_ROWB + (((kk * 2 + hi) ^ ((r >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WTN + j * 32 + frow;
                fb[j] = *(const u16x8*)(stB + r * GG3_ROWB + (((kk * 2 + hi) ^ ((r >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = gg_mfma_32x32x16_bf16(fb[j], fa[i], acc[i][j]);
        }
    }
    gg_barrier_raw();           // the ring is free: the epilogue stages through it

    const GgGemmParams e = *gg_late_params(p);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int m_wave = m0 + wm * WTM, n_wave = n0 + wn * WTN;
    const bool staged = e.splitk == 1 && !e.c_f32 && !e.d2s && (e.N & 3) == 0 && (e.ldc & 3) == 0 &&
                        (!e.residual || (e.ldr & 3) == 0);
    if (staged) {
        constexpr int SP = WTN * 2 + 8;
        static_assert(8 * WTM * SP <= GG3_LDS, "staging area must fit");
        char* stage = smem + wave * (WTM * SP);
        gg2_epilogue_step<0, TM, TN, FULL_EPI, true>(acc, e, b, bz, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), stage, SP,
                                                     lane, z4, z4);
        gg_sync();
        gg2_stage_writeback<WTM, WTN>(e, b, stage, SP, m_wave, n_wave, lane);
    } else {
        gg2_epilogue_step<0, TM, TN, FULL_EPI, false>(acc, e, b, bz, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), nullptr, 0,
                                                      lane, z4, z4);
    }
}

// ---- reduction-major x reduction-major (weight gradients): out[m][n] = sum_k A[k][m] * B[k][n] -------------------------
// Stage = 32 k-rows x 256 columns per operand, rows kept as they sit in HBM (512 bytes, unpadded: DMA destination is
// linear); the k-contiguous MFMA fragments come from ds_read_b64_tr_b16 as in gg_gemm2 (gg2_frag_krow). The four k-rows a
// transpose read touches must fall on different 64-byte bank quarters: LDS chunk position pos of k-row r holds source
// chunk pos ^ ((r & 3) << 2). A_CONV: column = (tap, cv) of the im2col matrix, k-row = output pixel; out-of-image taps
// load the zero page. Plain epilogue only (what the step uses: fp32 split-K partials / fp32 outputs).
#define GG3K_ROWB 512

GG_DEVICE u16x8 gg3_frag_krow(const char* tile, int col0, int kk, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int col = col0 + (g & 1) * 16 + 4 * (i & 3);
    const int row = kk * 16 + (g >> 1) * 8 + (i >> 2);
    const int phys = ((col >> 3) ^ ((row & 3) << 2)) * 16 + (col & 7) * 2;         // rows row and row + 4 share row & 3
    u16x4 a = gg_lds_read_tr16((const bf16_t*)(tile + row * GG3K_ROWB + phys));
    u16x4 b = gg_lds_read_tr16((const bf16_t*)(tile + (row + 4) * GG3K_ROWB + phys));
    u16x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
}

template <bool A_CONV>
GG_KERNEL GG_LAUNCH_BOUNDS(GG2_NT) void gg_gemm3k_kernel(GgGemmParams p) {
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    GG_SHARED __attribute__((aligned(16))) char smem[GG3_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_mn = tiles_n * ((p.M + BM - 1) / BM);
    const int bz = wg / tiles_mn, tile = wg - bz * tiles_mn;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int b = bz / p.splitk, ks = bz % p.splitk;
    const int kbeg = ks * p.k_per_split;
    int kend = kbeg + p.k_per_split;
    if (kend > p.K) kend = p.K;
    const int nk = (kend > kbeg) ? (kend - kbeg) / GG3_BK : 0;

    const bf16_t* Ab = p.A + (long long)b * p.a_bs;
    const bf16_t* Bb = p.B + (long long)b * p.b_bs;

    // DMA plan: per operand 32 k-rows x 32 chunks = 16 wave instructions; waves 0..3 own A's, waves 4..7 B's. Chunk id ->
    // k-row id / 32, LDS position id % 32, source chunk = position ^ swizzle(k-row). Columns beyond M / N are clamped to the
    // last valid 8-column group (they only feed outputs that are never stored).
    int krow[4], colv[4];                       // this lane's k-row inside a stage and first column of its chunk
    int ckh[4], ckw[4], cci[4];                 // A_CONV: the column's tap and physical channel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = ((wave & 3) * 4 + i) * 64 + lane;
        krow[i] = id >> 5;
        const int chunk = (id & 31) ^ ((krow[i] & 3) << 2);
        const bool isA = wave < 4;
        const int lim = isA ? p.M : p.N;
        int c = (isA ? m0 : n0) + chunk * 8;
        if (c > lim - 8) c = lim - 8 > 0 ? lim - 8 : 0;
        colv[i] = c;
        ckh[i] = ckw[i] = cci[i] = 0;
        if (A_CONV && isA) {
            const int tap = c / p.CV, cv = c - tap * p.CV;
            ckh[i] = tap / p.S;
            ckw[i] = tap - ckh[i] * p.S;
            cci[i] = (p.CV == p.C) ? cv : cv % p.C;
        }
    }
    const int hw = p.OH * p.OW;
    auto issue_stage = [&](int kt) {
        char* base = smem + (kt % GG3_NS) * GG3_STAGE + wave * 4 * 1024;       // A: bytes 0..16383 of the stage, B: the rest
        const int k0 = kbeg + kt * GG3_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + krow[i];
            const bf16_t* g;
            if (wave >= 4) {
                g = Bb + (long long)k * p.ldb + colv[i];
            } else if (!A_CONV) {
                g = Ab + (long long)k * p.lda + colv[i];
            } else {
                const int img = k / hw, rem = k - img * hw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const int ih = oh * p.stride - p.pad + ckh[i], iw = ow * p.stride - p.pad + ckw[i];
                const bool in = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                g = in ? p.A + (((long long)img * p.H + ih) * p.W + iw) * p.C + cci[i] : p.zero_page;
            }
            gg_load_lds16(g, base + i * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int s = 0; s < GG3_NS - 1 && s < nk; ++s) issue_stage(s);

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) gg_wait_vm<8>();
        else if (kt + 1 < nk) gg_wait_vm<4>();
        else gg_wait_vm<0>();
        gg_barrier_raw();
        if (kt + GG3_NS - 1 < nk) issue_stage(kt + GG3_NS - 1);
        const char* stA = smem + (kt % GG3_NS) * GG3_STAGE;
        const char* stB = stA + 32 * GG3K_ROWB;
#pragma unroll
        for (int kk = 0; kk < GG3_BK / 16; ++kk) {
            u16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = gg3_frag_krow(stA, wm * WTM + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = gg3_frag_krow(stB, wn * WTN + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = gg_mfma_32x32x16_bf16(fb[j], fa[i], acc[i][j]);
        }
    }
    gg_barrier_raw();

    const GgGemmParams e = *gg_late_params(p);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int m_wave = m0 + wm * WTM, n_wave = n0 + wn * WTN;
    const bool staged = e.splitk == 1 && !e.c_f32 && !e.d2s && (e.N & 3) == 0 && (e.ldc & 3) == 0 && !e.residual;
    if (staged) {
        constexpr int SP = WTN * 2 + 8;
        char* stage = smem + wave * (WTM * SP);
        gg2_epilogue_step<0, TM, TN, false, true>(acc, e, b, bz, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), stage, SP, lane,
                                                  z4, z4);
        gg_sync();
        gg2_stage_writeback<WTM, WTN>(e, b, stage, SP, m_wave, n_wave, lane);
    } else {
        gg2_epilogue_step<0, TM, TN, false, false>(acc, e, b, bz, m_wave + (lane & 31), n_wave + 4 * (lane >> 5), nullptr, 0, lane,
                                                   z4, z4);
    }
}
