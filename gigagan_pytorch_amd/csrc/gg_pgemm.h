// gg_pgemm.h — PERSISTENT, loader-fed contraction for the short-K row-major launches of the step (plan tile 15): the 1x1 convolutions and
// linear layers around attention and FeedForward (gigagan_pytorch.py:726-740, :620-700), their data gradients, the discriminator's
// 1x1 residual projections. C[m][n] = epilogue(sum_k A[m][k] * B[n][k]) with M = 32K..512K pixels, K = 64..1024, N = 64..2048.
// These launches move 2 * (K + N) bytes per output row against 2 * K * N flops: at K <= 512 the output store is most of the traffic
// and the HBM floor sits 2.4-4x under what gg_gemm2 measures on them (profiles/r05_shortk_probe.log: 8 us per 256 x 256 tile spent
// OUTSIDE the k-loop - first-load latency of a fresh workgroup, its store drain, its exit). This kernel keeps one workgroup per CU
// alive over a sequence of 128 x 128 output tiles and never lets it wait on memory:
//   * two LOADER waves (wave 8: A tiles + the tile's bias row, wave 9: B tiles) stream 64-wide k-stages HBM / L2 -> LDS by LDS-DMA into
//     a ring of four 32 KB slots, three stages ahead, ACROSS tile boundaries: while the compute waves run a tile's epilogue, up to
//     three stages of the next tile land. Their only wait is a counted s_waitcnt vmcnt in front of the stage's raw s_barrier
//     (17 / 16 transfers per stage and wave: 51 outstanding at most, the counter holds 63);
//   * eight COMPUTE waves (2 x 4, 64 x 32 outputs each, two per SIMD: one wave alone on a SIMD ran the epilogue's dependent VALU chain
//     at 1.8 us per tile, profiles/r05_pgemm_phases_v1.log) never load through vmcnt in the k-loop: fragments come from the
//     XOR-swizzled 128-byte LDS rows (Gg2Dma layout), one barrier per stage. Their stores drain during the next tile's k-loop;
//   * epilogue: bias from LDS (no global load behind the stores), residual / GELU-aux operands of the WHOLE wave tile requested before
//     the tile's first store (a load issued behind stores waits for them: shared vmcnt), 32 x 32 sub-tiles parked in a wave-private
//     staging area and written back as 64-byte row halves (the neighbouring wave writes the other half: the L2 merges them).
//   * tile order (n fastest): round-robin over the workgroups - the workgroups of an XCD work on consecutive tiles at any time, an A
//     tile is fetched from HBM once and shared through that L2 while hot - or a contiguous run per workgroup, chosen by shape (below);
//     B (<= 2 MB) lives in every L2.
// Algorithmic bytes: 2 * M * (K + N) (+ 2 * M * N per residual / aux operand) + 2 * N * K.
#pragma once
#include "gg_gemm2.h"

#define GG_PG_NT 640                           // eight compute waves + the two loaders
#define GG_PG_RING 4
#define GG_PG_STAGE 32768                     // ring slot: A tile | B tile, 128 rows x 128 bytes each
#define GG_PG_SP 72                           // staging pitch: 32 bf16 + 8 bytes (conflict-free ds_write_b64)
#define GG_PG_BIAS (GG_PG_RING * GG_PG_STAGE)
#define GG_PG_STAGING (GG_PG_BIAS + 4 * 1024)
#define GG_PG_LDS (GG_PG_STAGING + 8 * 32 * GG_PG_SP)

template <bool FULL_EPI>
GG_KERNEL GG_LAUNCH_BOUNDS(GG_PG_NT) void gg_pgemm_kernel(GgGemmParams p) {
    GG_SHARED __attribute__((aligned(1024))) char smem[GG_PG_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = gg_uniform(tid >> 6);

    // XCD-aware workgroup numbering (block b runs on XCD b % 8): the workgroups of XCD x are wg = 32 x .. 32 x + 31
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    const int tiles_n = (p.N + 127) >> 7, tiles_m = (p.M + 127) >> 7;
    const long long T = (long long)tiles_m * tiles_n;
    // Two tile orders (n fastest inside a row of tiles in both):
    //  * round-robin (p.pg_order 1): in round r workgroup wg takes tile r * nwg + wg, so that at any time the 32 workgroups of an XCD
    //    (contiguous wg, above) work on 32 CONSECUTIVE tiles - the n-tiles of a few m-tiles - and an A tile is fetched from HBM once and
    //    shared through that XCD's L2 while it is hot: rocprofv3 FETCH_SIZE = the A operand, WRITE_SIZE = the output, exactly
    //    (profiles/r05_pmc_pgemm_v3_roundrobin.log);
    //  * contiguous runs (0): workgroup wg takes tiles [wg T / nwg, (wg + 1) T / nwg): it re-reads its A tile once per n-tile, 4-5 us apart -
    //    by then the output stream has pushed it out of the L2 (FETCH_SIZE 3.4x the A operand, r05_pmc_pgemm_v2_nt_vs_plain.log) - but
    //    the re-reads come from the memory-side cache and a workgroup meets a cold A tile only once per row of tiles.
    // The host picks by shape (gg_api.hip).
    const bool rr_order = p.pg_order != 0;
    const int t0 = rr_order ? wg : (int)(wg * T / nwg);
    const int ntw = rr_order ? (wg < T ? (int)((T - wg + nwg - 1) / nwg) : 0) : (int)((wg + 1) * T / nwg) - t0;      // tiles of this workgroup
    const int dtm = rr_order ? nwg / tiles_n : 0, dtn = rr_order ? nwg - dtm * tiles_n : 1;                        // tile step as (rows of tiles, tiles)
    const int KT = p.K >> 6;
    const int Q = ntw * KT;
    // probe builds only (-DGG_PROBE, GG_PGEMM_DBG in the environment, tests/gpu_r5_pgemm_probe.py): 1 no MFMAs, 2 no epilogue, 4 no transfers,
    // 8 no stores, 16 plain stores. The product library compiles the constant 0: no switch in it can turn results into garbage
#if defined(GG_PROBE)
    const int dbg = p.xcd_slices;
#else
    constexpr int dbg = 0;
#endif

    if (wave >= 8) {
        // ---------------------------------------------------------------- loaders
        const bool isB = wave == 9;
        const int pitch = isB ? p.ldb : p.lda;
        const int lim = isB ? p.N : p.M;
        const GgBufS buf = gg_make_bufs(isB ? (const void*)p.B : (const void*)p.A, (unsigned long long)(isB ? p.b_bytes : p.a_bytes));
        const GgBufS bufb = gg_make_bufs((const void*)p.bias, p.bias ? (unsigned long long)p.N * 4 : 0ull);
        unsigned voff[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {           // chunk c of row r lands in slot c ^ ((r >> 1) & 7): lane -> (row 8 i + lane / 8, slot lane % 8)
            const int row = 8 * i + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
            voff[i] = (unsigned)((row * pitch + chunk * 8) * 2);
        }
        int itm = t0 / tiles_n, itn = t0 - itm * tiles_n, ik = 0, head = 0, itk = 0;
        auto issue = [&]() {
            if (dbg & 4) return;
            const int r0 = (isB ? itn : itm) << 7;
            const unsigned soff = (unsigned)(((long long)r0 * pitch + ik * 64) * 2);
            char* dst = smem + head * GG_PG_STAGE + (isB ? 16384 : 0);
            const int left = lim - r0;
            if (left >= 128) {
#pragma unroll
                for (int i = 0; i < 16; ++i) gg_bufs_load_lds16(buf, voff[i], soff, dst + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) gg_bufs_load_lds16(buf, (8 * i + (lane >> 3)) < left ? voff[i] : 0xFFFFFFFFu, soff, dst + i * 1024);
            }
            if (!isB) {                          // the tile's 128 bias values (zeros without a bias / past N), slot = (the workgroup's tile count) % 4
                const int bo = ((itn << 7) + lane * 4) * 4;
                gg_bufs_load_lds16(bufb, (lane < 32 && bo + 16 <= p.N * 4) ? (unsigned)bo : 0xFFFFFFFFu, 0u,
                                   smem + GG_PG_BIAS + (itk & 3) * 1024);
            }
            head = (head + 1) & (GG_PG_RING - 1);
            if (++ik == KT) {
                ik = 0;
                ++itk;
                itm += dtm; itn += dtn;
                if (itn >= tiles_n) { itn -= tiles_n; ++itm; }
            }
        };
        static_assert(GG_PG_RING == 4, "the literals below: two stages may stay in flight behind the one waited for");
        for (int s = 0; s < GG_PG_RING - 1 && s < Q; ++s) issue();
        for (int q = 0; q < Q; ++q) {
            // literal waits (gg_wait_vm_le's jump table costs a scalar-memory round trip: 0.2 us per stage here)
            const int ahead = (dbg & 4) ? 0 : Q - 1 - q;
            if (isB) {
                if (ahead >= 2) gg_wait_vm<32>();
                else if (ahead == 1) gg_wait_vm<16>();
                else gg_wait_vm<0>();
            } else {
                if (ahead >= 2) gg_wait_vm<34>();
                else if (ahead == 1) gg_wait_vm<17>();
                else gg_wait_vm<0>();
            }
            gg_barrier_lds();                    // stage q is in LDS; every compute wave is done with stage q - 1
            if (q + GG_PG_RING - 1 < Q) issue();
        }
        gg_wait_vm<0>();
        return;
    }

    // -------------------------------------------------------------------- compute waves
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, hi = lane >> 5;
    int fo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = frow * 128 + (((2 * kk + hi) ^ ((frow >> 1) & 7)) << 4);
    const int aoff = wm * 64 * 128, boff = 16384 + wn * 32 * 128;
    char* stage = smem + GG_PG_STAGING + wave * 32 * GG_PG_SP;
    const int qc = lane & 3, rr = lane >> 2;      // write-back: lane -> (row rr + 16 it, 16-byte chunk qc) of a parked 32 x 32 sub-tile
    bf16_t* const cbase = (bf16_t*)p.Cout;
    const bool has_res = p.residual != nullptr, aux1 = p.aux_mode == 1, aux2 = p.aux_mode == 2;

    // residual / GELU-aux rows of the wave tile are requested BEFORE the tile's k-loop and settled after it: a load issued behind the
    // previous stores waits for them (shared vmcnt), and a register hipcc still counts as pending costs a vmcnt(0) - a full store
    // drain - in front of every later use (host: the two operands never come together)
    const bf16_t* const pre = aux2 ? (const bf16_t*)p.aux : p.residual;
    const int ld_pre = aux2 ? p.ld_aux : p.ldr;

    // (initialised ONCE: a per-tile `rpre = 0` made hipcc put s_waitcnt vmcnt(0) - a drain of the previous tile's stores - at the top of every tile)
    u16x8 rpre[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int it = 0; it < 2; ++it) rpre[i][it] = gg_zero8();
    int tm = t0 / tiles_n, tn = t0 - tm * tiles_n, head = 0;
    for (int t = 0; t < ntw; ++t) {
        const int m_wave = (tm << 7) + wm * 64, n_wave = (tn << 7) + wn * 32;
        const int n = n_wave + qc * 8;
        const bool inner = m_wave + 64 <= p.M && n_wave + 32 <= p.N;       // (wave-uniform)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int it = 0; it < 2; ++it) {     // (rows outside the matrix keep a stale value: their stores are masked)
                const int m = m_wave + i * 32 + it * 16 + rr;
                if (pre && (inner || (m < p.M && n < p.N))) rpre[i][it] = *(const u16x8*)(pre + (long long)m * ld_pre + n);
            }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        for (int kt = 0; kt < KT; ++kt) {
            gg_barrier_lds();                    // the loaders have seen this stage land
            const char* ta = smem + head * GG_PG_STAGE + aoff;
            const char* tb = smem + head * GG_PG_STAGE + boff;
            head = (head + 1) & (GG_PG_RING - 1);
            if (dbg & 1) continue;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const u16x8 fb = *(const u16x8*)(tb + fo[kk]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const u16x8 fa = *(const u16x8*)(ta + i * 32 * 128 + fo[kk]);
                    acc[i] = gg_mfma_32x32x16_bf16(fb, fa, acc[i]);        // swapped: lane registers run along n
                }
            }
        }

        if (dbg & 2) {
            tm += dtm; tn += dtn;
            if (tn >= tiles_n) { tn -= tiles_n; ++tm; }
            continue;
        }
        // ---- epilogue: lane owns row m = ... + (lane & 31); register r holds column (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        // (unconditional: hipcc's wait placement is path-insensitive, a settle under `if (pre)` leaves the registers pending on the
        // other path and every later use waits for vmcnt(0) again)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int it = 0; it < 2; ++it) gg_settle(rpre[i][it]);
        const float* bl = (const float*)(smem + GG_PG_BIAS + (t & 3) * 1024) + wn * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 8 * g + 4 * hi;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[i][4 * g + q] * p.alpha;
                if (FULL_EPI) {
                    const f32x4 cb = *(const f32x4*)(bl + nl);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += cb[q] * p.bias_scale;
                    if (p.act != GG_ACT_NONE)
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = gg_apply_act(v[q], p.act, p.act_slope);
                }
                const u16x4 o = {gg_f2bf(v[0]), gg_f2bf(v[1]), gg_f2bf(v[2]), gg_f2bf(v[3])};
                *(u16x4*)(stage + frow * GG_PG_SP + nl * 2) = o;
            }
            gg_wave_sync();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + rr, m = m_wave + i * 32 + row;
                const u16x4 lo = *(const u16x4*)(stage + row * GG_PG_SP + qc * 16), hh = *(const u16x4*)(stage + row * GG_PG_SP + qc * 16 + 8);
                u16x8 o = {lo[0], lo[1], lo[2], lo[3], hh[0], hh[1], hh[2], hh[3]};
                if ((inner || (m < p.M && n < p.N)) && !(dbg & 8)) {
                    if (aux1) {                   // FeedForward up-projection: keep the pre-activation, emit gelu of its bf16 value
                        gg_store_nt16(p.aux + (long long)m * p.ld_aux + n, o);
#pragma unroll
                        for (int q = 0; q < 8; ++q) o[q] = gg_f2bf(gg_gelu_f(gg_bf2f(o[q])));
                    } else if (aux2) {            // data gradient of the down-projection: times gelu'(h) = Phi(h) + h phi(h)
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            float c, d;
                            const float x = gg_bf2f(rpre[i][it][q]);
                            gg_normal_cdf_pdf(x, c, d);
                            o[q] = gg_f2bf(gg_bf2f(o[q]) * (c + x * d));
                        }
                    }
                    if (has_res) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) o[q] = gg_f2bf(gg_bf2f(o[q]) + gg_bf2f(rpre[i][it][q]) * p.res_scale);
                    }
                    if (dbg & 16) *(u16x8*)(cbase + (long long)m * p.ldc + n) = o;
                    else gg_store_nt16(cbase + (long long)m * p.ldc + n, o);
                }
            }
            gg_wave_sync();                      // (the next sub-tile's staging writes follow this one's reads)
        }
        tm += dtm; tn += dtn;
        if (tn >= tiles_n) { tn -= tiles_n; ++tm; }
    }
}
