// gg_spair.h — TWO adaptive 3x3 convolutions of one generator block in ONE launch, the intermediate map never leaving the CU:
//
//   mid = act1( conv3x3(x * xs, w1[b]) + noise1[b,p] * nw1[c] )       rounded to bf16 (what the unfused path writes to HBM)
//   y   = act2( conv3x3(mid,    w2[b]) + noise2[b,p] * nw2[c] )
//
// reference: Generator.forward's resnet block at 128x128 / 256x256 (gp.py:1219-1229: conv1 -> Noise -> leaky_relu -> conv2 -> Noise ->
// leaky_relu, nothing in between) on the per-sample kernels of AdaptiveConv2DMod.forward (gp.py:378-409), which gg_modw_multi_fwd has
// already written (layout 2, demodulation folded in). These layers are HBM problems (2 (C0 + C1) + 2 (C1 + C2) bytes per pixel when
// run as two launches against 18 (C0 C1 + C1 C2) flops): fused, the 2 * 2 * C1 bytes per pixel of the intermediate map (201 of the
// 502 MB the four layers of config 2 move at batch 32) stay in LDS.
//
// Skeleton (gg_sfwd.h's, with a second stage):
//   * a workgroup owns `rows` output rows of one image, full width (W = 32 * NCW * PT: NCW compute waves x PT 32-pixel blocks);
//   * a LOADER wave streams whole x rows HBM -> LDS ring by LDS-DMA, NSX - 1 rows ahead (the two noise maps ride along: 1 KB per row each); its only wait is a counted vmcnt in
//     front of the iteration's raw s_barrier (the compute waves store: their vmcnt cannot count transfers);
//   * iteration r: conv1 of mid row r + 1 from x rows r .. r + 2 -> noise, leaky-relu, bf16 -> LDS mid ring (4 rows); conv2 of
//     output row r - 1 from mid rows r - 2 .. r (complete since the iteration's barrier) -> noise, leaky-relu -> the wave's staging
//     (8-byte-per-lane row stores from the registers). ONE barrier per row; mid rows above / below the image are zeros (conv2's padding), the
//     x rows there arrive as zeros from the DMA's range check; one zero pixel left and right of every ring row;
//   * both rings are XOR-swizzled by the pixel's x coordinate so that the 16 lanes a ds_read_b128 serves per cycle hit 16 distinct
//     bank quads (the loader picks which global chunk each DMA lane fetches; the mid row is written swizzled);
//   * conv1's weights live in registers as MFMA A fragments (scaled by the skip-layer excitation xs there: conv(x * e, w) =
//     conv(x, w * e)), conv2's in registers (C0 = 32) or in LDS (C0 = 64: 144 + 72 registers do not fit beside the pipeline);
//   * every accumulator is ONE chain in gg_sconv's order (tap row, tap column, channel step), the epilogues are gg_sconv's
//     expressions: the result is bit-identical to gg_sconv(gg_sconv(x)) - the test of this kernel is torch.equal.
// Halo cost: rows + 2 conv1 rows and rows + 4 x rows per `rows` output rows.
// Algorithmic bytes: 2 * (C0 + C2) per pixel + 8 bytes of noise.
#pragma once
#include "gg_gemm.h"


struct GgSpairParams {
    const bf16_t* x;            // [b][H][W][C0]
    const bf16_t* w1;           // [b][9][C0/16][32][16] (gg_modw layout 2; rows >= C1 zero)
    const bf16_t* w2;           // [b][9][C1/16][32][16]
    bf16_t* y;                  // [b][H][W][C2]
    long long w1_bs, w2_bs;     // elements between the banks of consecutive images (0: shared)
    const float* noise1;        // [b][H*W] or null
    const float* nw1;           // [C1]
    const float* noise2;        // [b][H*W] or null
    const float* nw2;           // [C2]
    const float* xs;            // optional [b][C0]
    int b, H, C2;
    int act1, act2;             // 0 none, 1 leaky relu
    float slope;
    int rows, strips;           // output rows per workgroup, workgroups per image
#if defined(GG_SP_PROBE)        // tests/probes/spair_probe.hip only: phase time stamps of one workgroup, pieces switched off
    long long* stamps;          // [wave 0 | loader][iteration][8]
    int probe_wg, probe_off;    // off bits: 1 no conv1 MFMAs, 2 no conv2 MFMAs, 4 no output stores, 8 no DMA, 16 no mid-row writes
#endif
};
#if defined(GG_SP_PROBE)
#define GG_SP_STAMP(slot) do { if (stamp_on && lane == 0) p.stamps[((wave != 0 ? 64 : 0) + (r - r_first)) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define GG_SP_OFF(bit) (p.probe_off & (bit))
#else
#define GG_SP_STAMP(slot) do {} while (0)
#define GG_SP_OFF(bit) 0
#endif

// NCW compute waves (one per SIMD, or two: the second hides the first one's epilogue and LDS latencies - a lone wave pays every one of
// them: profiles/r06_spair_probe_v1.log) x PT 32-pixel blocks each = one image row
// M16 (C0 = 32, C1 = 16, C2 <= 16: the 256x256 block): the convolutions run on v_mfma_f32_16x16x32_bf16 - 16 output channels are a
// whole MFMA row block there, where the 32x32x16 form spends half its matrix-pipe time on zero weight rows
template <int C0, int C1, int PT, int NCW, bool M16 = false>
struct GgSpGeom {
    static constexpr int W = 32 * NCW * PT;
    static constexpr int NT = (NCW + 1) * 64;                      // + the loader wave
    static constexpr int NSX = C0 <= 32 ? 6 : 5;                   // x ring rows: three being read, the rest in flight (what LDS allows)
    static constexpr int P0 = C0 * 2, P1 = C1 * 2;                 // bytes per pixel in the rings
    static constexpr int XS = (W + 2) * P0, MS = (W + 2) * P1;     // bytes per ring row
    static constexpr int XSLOT = XS + 2048;                        // + the noise rows that travel with an x row (1 KB each: conv1's, conv2's)
    static constexpr int KC0 = C0 / 16, KC1 = C1 / 16;
    static constexpr bool W2REG = M16 || (C0 <= 32 && NCW == 4);    // (nine waves: 168 registers each; M16: 9 + 5 fragments in all)
    static constexpr int xring = 0;
    static constexpr int mring = NSX * XSLOT;
    static constexpr int epi = mring + 4 * MS;                     // nw1 [32] | nw2 [32] floats
    static constexpr int w2l(int) { return epi + 256; }            // conv2's bank (9 * KC1 KB) when it does not live in registers
    static constexpr int bytes(int C2) { return w2l(C2) + (W2REG ? 0 : 9 * KC1 * 1024); }
};

// swizzle key of pixel x for a ring with P bytes per pixel: chunk c of the pixel sits at position c ^ key. The 32x32x16 fragments read 32
// consecutive pixels at one chunk index per half wave; the 16x16x32 ones (M16) read 16 pixels at chunk index lane >> 4, and the 16 lanes a
// ds_read_b128 serves per cycle mix two chunk indices (lanes {0-3, 12-15} of one, {20-27} of the next): 64-byte pixels then want the
// key table {0, 2, 3, 1}[(x >> 2) & 3], 32-byte pixels none
template <int P, bool M16 = false>
GG_DEVICE int gg_sp_key(int X) {
    if (M16) return P == 64 ? (0x78 >> (2 * ((X >> 2) & 3))) & 3 : 0;
    return P == 128 ? (X >> 1) & 7 : (P == 64 ? (X >> 2) & 3 : (X >> 3) & 1);
}

// the instruction order of one convolution row, pinned: the LDS reads of step s + AH are issued BEHIND the MFMAs of step s (hipcc sinks
// them to one read in flight otherwise: a lone wave on its SIMD then waits an LDS round trip per MFMA). RD = reads per pixel block
// and step (2: the A fragment comes from LDS as well)
template <int NS, int AH, int PT, int RD>
GG_DEVICE void gg_sp_pipeline() {
#if !defined(GG_HOST_EMULATION)
    __builtin_amdgcn_sched_group_barrier(0x100, (AH < NS ? AH : NS) * PT * RD - (RD == 2 ? (AH < NS ? AH : NS) * (PT - 1) : 0), 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, PT, 0);
        if (s + AH < NS) __builtin_amdgcn_sched_group_barrier(0x100, RD == 2 ? PT + 1 : PT, 0);
    }
#endif
}

// the interleaved form: slot k is a conv1 step (k % 3 < 2) or a conv2 step, whose A fragment is an LDS read as well when AL == 1
template <int NSL, int AH, int PT, int AL>
GG_DEVICE void gg_sp_pipeline_both() {
#if !defined(GG_HOST_EMULATION)
    constexpr int pre = AH < NSL ? AH : NSL;
    __builtin_amdgcn_sched_group_barrier(0x100, pre * PT + AL * (pre / 3), 0);
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, PT, 0);
        if (k + AH < NSL) {
            if ((k + AH) % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x100, PT + AL, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, PT, 0);
        }
    }
#endif
}

template <int C0, int C1, int PT, int NCW, bool M16 = false>
GG_KERNEL GG_LAUNCH_BOUNDS((NCW + 1) * 64) void gg_spair_kernel(GgSpairParams p) {
    typedef GgSpGeom<C0, C1, PT, NCW, M16> G;
    static_assert(!M16 || (C0 == 32 && C1 == 16 && PT == 1), "the 16-row form is written for the 32 -> 16 -> <= 16 block");
    constexpr int W = G::W, P0 = G::P0, P1 = G::P1, XS = G::XS, XSLOT = G::XSLOT, MS = G::MS, KC0 = G::KC0, KC1 = G::KC1, NSX = G::NSX;
    constexpr bool W2REG = G::W2REG;
    GG_DYN_SHARED(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = gg_uniform(tid >> 6);
    const int H = p.H;
    const int img = blockIdx.x / p.strips, strip = blockIdx.x - img * p.strips;
    const int y0 = strip * p.rows;
    const int y1 = y0 + p.rows < H ? y0 + p.rows : H;
    const int r_first = y0 - 2, r_last = y1;              // iterations; x rows y0 - 2 .. y1 + 1, mid rows y0 - 1 .. y1
    const int x_last = y1 + 1;
    constexpr int D = NSX - 1;

#if defined(GG_SP_PROBE)
    const bool stamp_on = (int)blockIdx.x == p.probe_wg && (wave == 0 || wave == NCW);
#endif
    if (wave == NCW) {
        // ---------------------------------------------------------------- loader
        GgBufS buf = gg_make_bufs((const void*)(p.x + (long long)img * H * W * C0), (unsigned long long)H * W * C0 * 2);
        // the noise maps of this image: row g of conv1's map and row g - 2 of conv2's travel with x row g (iteration r needs rows r + 1
        // and r - 1 of them: both sit in the slot of x row r + 1); W floats = 16 bytes from W / 4 lanes, the others deposit zeros
        GgBufS bufn1 = gg_make_bufs((const void*)((p.noise1 ? p.noise1 : (const float*)p.x) + (p.noise1 ? (long long)img * H * W : 0)),
                                    (unsigned long long)H * W * 4);
        GgBufS bufn2 = gg_make_bufs((const void*)((p.noise2 ? p.noise2 : (const float*)p.x) + (p.noise2 ? (long long)img * H * W : 0)),
                                    (unsigned long long)H * W * 4);
        const bool n1 = p.noise1 != nullptr, n2 = p.noise2 != nullptr;
        constexpr int NL = W * P0 / 1024;                 // 1 KB transfers per row
        constexpr int CH = P0 / 16;                       // chunks per pixel
        const int per_row = NL + (n1 ? 1 : 0) + (n2 ? 1 : 0);
        unsigned voff[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int j = i * 64 + lane, px = j / CH, pos = j % CH;
            voff[i] = (unsigned)((px * CH + (pos ^ gg_sp_key<P0, M16>(px))) * 16);      // chunk (pos ^ key) of pixel px lands at position pos
        }
        const unsigned nvoff = lane * 16 < W * 4 ? (unsigned)(lane * 16) : 0xFFFFFFFFu;
        int head = 0;
        auto issue_row = [&](int gr) {
            const bool ok = (unsigned)gr < (unsigned)H && !GG_SP_OFF(8);
            char* slot = smem + G::xring + head * XSLOT;
#pragma unroll
            for (int i = 0; i < NL; ++i)
                gg_bufs_load_lds16(buf, ok ? voff[i] : 0xFFFFFFFFu, ok ? (unsigned)gr * (unsigned)(W * P0) : 0u, slot + P0 + i * 1024);
            if (n1) gg_bufs_load_lds16(bufn1, ok ? nvoff : 0xFFFFFFFFu, ok ? (unsigned)gr * (unsigned)(W * 4) : 0u, slot + XS);
            if (n2) {
                const bool ok2 = (unsigned)(gr - 2) < (unsigned)H;
                gg_bufs_load_lds16(bufn2, ok2 ? nvoff : 0xFFFFFFFFu, ok2 ? (unsigned)(gr - 2) * (unsigned)(W * 4) : 0u, slot + XS + 1024);
            }
            head = head + 1 == NSX ? 0 : head + 1;
        };
        for (int g = r_first; g < r_first + D && g <= x_last; ++g) issue_row(g);
        for (int r = r_first; r <= r_last; ++r) {
            // rows issued so far: up to min(r + D - 1, x_last); rows r .. r + 2 must have landed
            int newest = r + D - 1 < x_last ? r + D - 1 : x_last;
            int ahead = newest - (r + 2);
            if (ahead < 0) ahead = 0;
            GG_SP_STAMP(0);
            gg_wait_vm_le(ahead * per_row);
            GG_SP_STAMP(1);
            gg_barrier_lds();
            GG_SP_STAMP(2);
            if (r + D <= x_last) issue_row(r + D);
            GG_SP_STAMP(3);
        }
        gg_wait_vm_le(0);
        return;
    }

    // -------------------------------------------------------------------- compute waves
    const int pl = lane & 31, hi = lane >> 5;
    const int xw = wave * PT * 32;                        // first pixel column of this wave
    const int C2 = p.C2;

    // zero pixels left and right of every ring row (the DMA and the mid-row writes never touch them); epilogue constants
    {
        constexpr int NZ0 = NSX * 2 * (P0 / 16), NZ1 = 4 * 2 * (P1 / 16);
        for (int v = tid; v < NZ0 + NZ1; v += NCW * 64) {
            int off;
            if (v < NZ0) {
                const int row = v / (2 * (P0 / 16)), rem = v - row * (2 * (P0 / 16));
                const int side = rem / (P0 / 16), c = rem - side * (P0 / 16);
                off = G::xring + row * XSLOT + side * (W + 1) * P0 + c * 16;
            } else {
                const int u = v - NZ0;
                const int row = u / (2 * (P1 / 16)), rem = u - row * (2 * (P1 / 16));
                const int side = rem / (P1 / 16), c = rem - side * (P1 / 16);
                off = G::mring + row * MS + side * (W + 1) * P1 + c * 16;
            }
            *(u16x8*)(smem + off) = gg_zero8();
        }
        if (tid < 64) {
            const int n = tid & 31;
            float v = 0.f;
            if (tid < 32) v = (p.noise1 && n < C1) ? p.nw1[n] : 0.f;
            else v = (p.noise2 && n < C2) ? p.nw2[n] : 0.f;
            ((float*)(smem + G::epi))[tid] = v;
        }
    }

    if constexpr (M16) {
        // ================================================================ 16x16x32 form (32 -> 16 -> <= 16 channels)
        // lane = (pixel n = lane & 15 of a 16-pixel half block, k-group g = lane >> 4); a wave's 32 pixels are two half blocks.
        // conv1: K = 32 is one tap's 32 input channels (chunk g of the pixel): 9 MFMAs per half block.
        // conv2: K = 32 is TWO taps' 16 channels (g < 2: tap 2 j, chunk g; g >= 2: tap 2 j + 1, chunk g - 2): 5 MFMAs per half block,
        //        the ninth tap paired with zero weights.
        // accumulator register r of lane (n, g): channel 4 g + r of pixel n.
        const int n16 = lane & 15, g = lane >> 4;
        u16x8 w1f[9], w2f[5];
        {
            const bf16_t* wb = p.w1 + (long long)img * p.w1_bs;
#pragma unroll
            for (int t = 0; t < 9; ++t) w1f[t] = *(const u16x8*)(wb + ((t * 2 + (g >> 1)) * 32 + n16) * 16 + (g & 1) * 8);
            if (p.xs) {
                const float* sp = p.xs + (long long)img * C0 + g * 8;
                const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    u16x8 v = w1f[t];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = gg_f2bf(gg_bf2f(v[e]) * s0[e]);
                        v[e + 4] = gg_f2bf(gg_bf2f(v[e + 4]) * s1[e]);
                    }
                    w1f[t] = v;
                }
            }
            const bf16_t* wc = p.w2 + (long long)img * p.w2_bs;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int t = 2 * j + (g >> 1);
                u16x8 v = *(const u16x8*)(wc + ((t < 9 ? t : 8) * 32 + n16) * 16 + (g & 1) * 8);
                if (t >= 9) v = gg_zero8();
                w2f[j] = v;
            }
        }
        // fragment byte offsets inside a ring row, per lane: conv1 [half block][tap column]; conv2 [half block][tap pair] (the pair's row
        // and column differ per lane half: lanes g < 2 read tap 2 j, lanes g >= 2 tap 2 j + 1)
        int fo1[2][3], fo2[2][5], ky2[5];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int X = xw + h * 16 + n16 + dx - 1;
                fo1[h][dx] = (X + 1) * P0 + ((g ^ gg_sp_key<P0, true>(X)) << 4);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int t = 2 * j + (g >> 1), tc = t < 9 ? t : 8;
                const int X = xw + h * 16 + n16 + tc % 3 - 1;
                fo2[h][j] = (X + 1) * P1 + ((g & 1) << 4);
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int t = 2 * j + (g >> 1), tc = t < 9 ? t : 8;
            ky2[j] = tc / 3;                                 // (per lane half)
        }
        int mo[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) mo[h] = (xw + h * 16 + n16 + 1) * P1 + 8 * g;     // channels 4 g .. + 3: byte 8 g of the pixel's 32
        const float* epi = (const float*)(smem + G::epi);
        const float slope1 = p.act1 == 1 ? p.slope : 1.f, slope2 = p.act2 == 1 ? p.slope : 1.f;
        const bool has_n1 = p.noise1 != nullptr, has_n2 = p.noise2 != nullptr;
        const long long img_pix0 = (long long)img * H * W;
        const bool own2 = 4 * g < C2;                        // this lane group's four output channels exist (C2 is a multiple of 8)

        int xs0 = 0, mi = 0;
        for (int r = r_first; r <= r_last; ++r) {
            GG_SP_STAMP(0);
            gg_barrier_lds();             // x rows r .. r + 2 have landed (the loader waited), mid rows up to r are written
            GG_SP_STAMP(1);
            int eo = 0;
#if !defined(GG_HOST_EMULATION)
            asm volatile("" : "+v"(eo));
#endif
            const int m = r + 1, o = r - 1;
            const bool do1 = r <= y1 - 1, do2 = r >= y0 + 1;
            const bool m_in = (unsigned)m < (unsigned)H;
            int xslot[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int sl = xs0 + ky;
                sl = sl >= NSX ? sl - NSX : sl;
                xslot[ky] = G::xring + sl * XSLOT;
            }
            int rbv[5];                                      // conv2: the ring row of each tap pair's lane half
#pragma unroll
            for (int j = 0; j < 5; ++j) rbv[j] = G::mring + ((mi + 1 + ky2[j]) & 3) * MS;
            const bool c1 = do1 && m_in && !GG_SP_OFF(1), c2 = do2 && !GG_SP_OFF(2);
            f32x4 acc1[2], acc2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) acc1[h] = acc2[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (plain loops: pinning the 28 fragment reads six slots ahead of their MFMAs cut this phase from 1,670 to 1,300 cycles and the
            // kernel not at all - 56.7 vs 55.9 us: with half the MFMA time gone the row is bound by the stream, r06_spair_probe_v7_m16.log)
            if (c1) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        acc1[h] = gg_mfma_16x16x32_bf16(w1f[t], *(const u16x8*)(smem + xslot[t / 3] + fo1[h][t % 3]), acc1[h]);
            }
            if (c2) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        acc2[h] = gg_mfma_16x16x32_bf16(w2f[j], *(const u16x8*)(smem + rbv[j] + fo2[h][j]), acc2[h]);
            }
            GG_SP_STAMP(2);
            // conv1's epilogue -> mid ring row of m (zeros outside the image: conv2's padding)
            if (do1 && !GG_SP_OFF(16)) {
                char* mrow = smem + G::mring + mi * MS;
                if (m_in) {
                    const f32x4 w4 = *(const f32x4*)(epi + eo + 4 * g);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float a = *(const float*)(smem + xslot[1] + XS + (xw + h * 16 + n16) * 4);
                        const float nz = has_n1 ? a : 0.f;
                        u16x4 o4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc1[h][e] + nz * w4[e];
                            o4[e] = gg_f2bf(fmaxf(v, v * slope1));
                        }
                        *(u16x4*)(mrow + mo[h]) = o4;
                    }
                } else {
                    const u16x4 z4 = {0, 0, 0, 0};
#pragma unroll
                    for (int h = 0; h < 2; ++h) *(u16x4*)(mrow + mo[h]) = z4;
                }
            }
            GG_SP_STAMP(3);
            // conv2's epilogue -> row stores (8 bytes per lane: four channels of one pixel)
            if (do2) {
                const f32x4 w4 = *(const f32x4*)(epi + eo + 32 + 4 * g);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float c = *(const float*)(smem + xslot[1] + XS + 1024 + (xw + h * 16 + n16) * 4);
                    const float nz = has_n2 ? c : 0.f;
                    u16x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc2[h][e] + nz * w4[e];
                        o4[e] = gg_f2bf(fmaxf(v, v * slope2));
                    }
                    if (own2 && !GG_SP_OFF(4)) *(u16x4*)(p.y + (img_pix0 + (long long)o * W + xw + h * 16 + n16) * C2 + 4 * g) = o4;
                }
                GG_SP_STAMP(5);
            }
            GG_SP_STAMP(6);
            xs0 = xs0 + 1 == NSX ? 0 : xs0 + 1;
            mi = (mi + 1) & 3;
        }
        return;
    }

    // weights as MFMA A fragments: [tap][k-step] 8 bf16 of output channel pl, input channels kc * 16 + 8 * hi .. + 7
    u16x8 w1f[9 * KC0];
    {
        const bf16_t* wb = p.w1 + (long long)img * p.w1_bs + pl * 16 + hi * 8;
#pragma unroll
        for (int s = 0; s < 9 * KC0; ++s) w1f[s] = *(const u16x8*)(wb + s * 512);
        if (p.xs) {
#pragma unroll
            for (int kc = 0; kc < KC0; ++kc) {
                const float* sp = p.xs + (long long)img * C0 + kc * 16 + hi * 8;
                const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    u16x8 v = w1f[t * KC0 + kc];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = gg_f2bf(gg_bf2f(v[e]) * s0[e]);
                        v[e + 4] = gg_f2bf(gg_bf2f(v[e + 4]) * s1[e]);
                    }
                    w1f[t * KC0 + kc] = v;
                }
            }
        }
    }
    u16x8 w2f[W2REG ? 9 * KC1 : 1];
    const int w2l = G::w2l(C2);
    if (W2REG) {
        const bf16_t* wb = p.w2 + (long long)img * p.w2_bs + pl * 16 + hi * 8;
#pragma unroll
        for (int s = 0; s < 9 * KC1; ++s) w2f[s] = *(const u16x8*)(wb + s * 512);
    } else {
        const u16x8* src = (const u16x8*)(p.w2 + (long long)img * p.w2_bs);
        constexpr int NV = 9 * KC1 * 64;
        for (int v = tid; v < NV; v += NCW * 64) *(u16x8*)(smem + w2l + v * 16) = src[v];
    }

    // fragment byte offsets inside a ring row, per lane: [pixel block][tap column] for k-step 0; chunk 2 kc + hi of pixel X sits at
    // position (2 kc + hi) ^ key(X) = (hi ^ key(X)) ^ 2 kc, and the pixel's base has those address bits clear: k-step kc = offset ^ (kc << 5)
    int fo1[PT][3], fo2[PT][3];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int X = xw + t * 32 + pl + dx - 1;          // -1 / W: the zero pixels
            fo1[t][dx] = (X + 1) * P0 + ((hi ^ gg_sp_key<P0>(X)) << 4);
            fo2[t][dx] = (X + 1) * P1 + ((hi ^ gg_sp_key<P1>(X)) << 4);
        }
    // mid-row write: register quad q of a block holds channels 8 q + 4 hi .. + 3 of pixel pl: chunk q of the pixel, half hi
    int mo[PT], mkey[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        mo[t] = (xw + t * 32 + pl + 1) * P1 + 8 * hi;
        mkey[t] = gg_sp_key<P1>(xw + t * 32 + pl);
    }

    const float* epi = (const float*)(smem + G::epi);
    const int cpp = C2 >> 3;
    const float slope1 = p.act1 == 1 ? p.slope : 1.f, slope2 = p.act2 == 1 ? p.slope : 1.f;
    const bool has_n1 = p.noise1 != nullptr, has_n2 = p.noise2 != nullptr;
    const long long img_pix0 = (long long)img * H * W;

    // one convolution row: 9 * KC steps of (tap row ky, tap column dx, channel step kc), every pixel block one accumulator chain in
    // that order; the B fragments of step s + AH are requested before the MFMAs of step s (one wave per SIMD: nothing else hides the
    // LDS latency)
    constexpr int AH = (NCW == 4 && PT == 1) ? 3 : 4;

    // ring slots by iteration count it = r - r_first: x row g sits in slot (g - r_first) mod NSX; mid row m in slot (m - (y0 - 1)) & 3,
    // i.e. iteration `it` writes slot it & 3 and reads the rows r - 2 .. r from slots (it + 1 + ky) & 3
    int xs0 = 0, mi = 0;
    for (int r = r_first; r <= r_last; ++r) {
        GG_SP_STAMP(0);
        gg_barrier_lds();                 // x rows r .. r + 2 have landed (the loader waited), mid rows up to r are written
        GG_SP_STAMP(1);
        int eo = 0;                       // (as for the bank below: keeps the eight ds_read_b128 of the noise weights in the loop, not 32 registers)
#if !defined(GG_HOST_EMULATION)
        asm volatile("" : "+v"(eo));
#endif
        const int m = r + 1, o = r - 1;
        const bool do1 = r <= y1 - 1, do2 = r >= y0 + 1;
        const bool m_in = (unsigned)m < (unsigned)H;
        int xslot[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int s = xs0 + ky;
            s = s >= NSX ? s - NSX : s;
            xslot[ky] = G::xring + s * XSLOT;
        }
        // ---- the MFMAs of the iteration. Both rows' chains are independent (conv2 reads mid rows finished before the barrier), so where
        // both run they are interleaved - two conv1 steps, one conv2 step (C0 = 2 C1: conv1 has twice the steps): no chain waits for its own
        // previous MFMA, and one fill / drain of the fragment pipeline instead of two
        f32x16 acc1[PT], acc2[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc1[t][q] = acc2[t][q] = 0.f;
        int rb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) rb[ky] = G::mring + ((mi + 1 + ky) & 3) * MS;
        // (the bank in LDS is loop invariant: an offset the compiler cannot see through keeps its reads in the loop instead of 72
        // registers for the whole strip; an integer, not the pointer - see gg_sconv)
        int opaque = 0;
#if !defined(GG_HOST_EMULATION)
        asm volatile("" : "+v"(opaque));
#endif
        const char* wl = smem + w2l + pl * 32 + hi * 16 + opaque;
        constexpr int NS1 = 9 * KC0, NS2 = 9 * KC1;
        static_assert(NS1 == 2 * NS2, "the interleave assumes C0 == 2 * C1");
        auto loadB1 = [&](int s, int t) {
            const int ky = s / (3 * KC0), dx = (s / KC0) % 3, kc = s % KC0;
            return *(const u16x8*)(smem + xslot[ky] + (fo1[t][dx] ^ (kc << 5)));
        };
        auto loadB2 = [&](int s, int t) {
            const int ky = s / (3 * KC1), dx = (s / KC1) % 3, kc = s % KC1;
            return *(const u16x8*)(smem + rb[ky] + (fo2[t][dx] ^ (kc << 5)));
        };
        const bool c1 = do1 && m_in && !GG_SP_OFF(1), c2 = do2 && !GG_SP_OFF(2);
        if (c1 && c2) {
            constexpr int NSL = NS1 + NS2;                 // slot k: k % 3 < 2 -> conv1 step 2 (k / 3) + k % 3, else conv2 step k / 3
            u16x8 fb[AH + 1][PT];
            u16x8 fa[W2REG ? 1 : AH + 1];
            auto load_slot = [&](int k) {
                const int bi = k % (AH + 1);
                if (k % 3 < 2) {
#pragma unroll
                    for (int t = 0; t < PT; ++t) fb[bi][t] = loadB1(2 * (k / 3) + k % 3, t);
                } else {
#pragma unroll
                    for (int t = 0; t < PT; ++t) fb[bi][t] = loadB2(k / 3, t);
                    if (!W2REG) fa[W2REG ? 0 : bi] = *(const u16x8*)(wl + (k / 3) * 1024);
                }
            };
#pragma unroll
            for (int k = 0; k < AH; ++k) load_slot(k);
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                if (k + AH < NSL) load_slot(k + AH);
                const int bi = k % (AH + 1);
                if (k % 3 < 2) {
#pragma unroll
                    for (int t = 0; t < PT; ++t) acc1[t] = gg_mfma_32x32x16_bf16(w1f[2 * (k / 3) + k % 3], fb[bi][t], acc1[t]);
                } else {
#pragma unroll
                    for (int t = 0; t < PT; ++t)
                        acc2[t] = gg_mfma_32x32x16_bf16(W2REG ? w2f[W2REG ? k / 3 : 0] : fa[W2REG ? 0 : bi], fb[bi][t], acc2[t]);
                }
            }
            gg_sp_pipeline_both<NSL, AH, PT, (W2REG ? 0 : 1)>();
        } else {
            if (c1) {
                u16x8 fb[AH + 1][PT];
#pragma unroll
                for (int s = 0; s < AH; ++s)
#pragma unroll
                    for (int t = 0; t < PT; ++t) fb[s][t] = loadB1(s, t);
#pragma unroll
                for (int s = 0; s < NS1; ++s) {
                    if (s + AH < NS1) {
#pragma unroll
                        for (int t = 0; t < PT; ++t) fb[(s + AH) % (AH + 1)][t] = loadB1(s + AH, t);
                    }
#pragma unroll
                    for (int t = 0; t < PT; ++t) acc1[t] = gg_mfma_32x32x16_bf16(w1f[s], fb[s % (AH + 1)][t], acc1[t]);
                }
                gg_sp_pipeline<NS1, AH, PT, 1>();
            }
            if (c2) {
                u16x8 fb[AH + 1][PT];
                u16x8 fa[W2REG ? 1 : AH + 1];
#pragma unroll
                for (int s = 0; s < AH && s < NS2; ++s) {
#pragma unroll
                    for (int t = 0; t < PT; ++t) fb[s][t] = loadB2(s, t);
                    if (!W2REG) fa[W2REG ? 0 : s] = *(const u16x8*)(wl + s * 1024);
                }
#pragma unroll
                for (int s = 0; s < NS2; ++s) {
                    if (s + AH < NS2) {
#pragma unroll
                        for (int t = 0; t < PT; ++t) fb[(s + AH) % (AH + 1)][t] = loadB2(s + AH, t);
                        if (!W2REG) fa[W2REG ? 0 : (s + AH) % (AH + 1)] = *(const u16x8*)(wl + (s + AH) * 1024);
                    }
#pragma unroll
                    for (int t = 0; t < PT; ++t)
                        acc2[t] = gg_mfma_32x32x16_bf16(W2REG ? w2f[W2REG ? s : 0] : fa[W2REG ? 0 : s % (AH + 1)], fb[s % (AH + 1)][t], acc2[t]);
                }
                gg_sp_pipeline<NS2, AH, PT, (W2REG ? 1 : 2)>();
            }
        }
        GG_SP_STAMP(2);

        // this iteration's noise values: rows m of conv1's map and o of conv2's, both in the slot of x row r + 1
        float nz1[PT], nz2[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const float a = *(const float*)(smem + xslot[1] + XS + (xw + t * 32 + pl) * 4);
            const float c = *(const float*)(smem + xslot[1] + XS + 1024 + (xw + t * 32 + pl) * 4);
            nz1[t] = has_n1 ? a : 0.f;        // (without a map nothing was deposited there: never multiplied)
            nz2[t] = has_n2 ? c : 0.f;
        }

        // ---- conv1's epilogue: noise, activation, bf16 -> mid ring row of m (zeros outside the image: conv2's padding). leaky-relu as
        // max(v, v * slope) (0 < slope <= 1; slope 1 = no activation): the same values as gg_sconv's select, one instruction less each
        if (do1 && !GG_SP_OFF(16)) {
            char* mrow = smem + G::mring + mi * MS;
            if (m_in) {
#pragma unroll
                for (int t = 0; t < PT; ++t) {
#pragma unroll
                    for (int q = 0; q < C1 / 8; ++q) {
                        const f32x4 w4 = *(const f32x4*)(epi + eo + 8 * q + 4 * hi);
                        u16x4 o4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc1[t][q * 4 + e] + nz1[t] * w4[e];
                            o4[e] = gg_f2bf(fmaxf(v, v * slope1));
                        }
                        *(u16x4*)(mrow + mo[t] + ((q ^ mkey[t]) << 4)) = o4;
                    }
                }
            } else {
                const u16x4 z4 = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < PT; ++t)
#pragma unroll
                    for (int q = 0; q < C1 / 8; ++q) *(u16x4*)(mrow + mo[t] + ((q ^ mkey[t]) << 4)) = z4;
            }
        }
        GG_SP_STAMP(3);

        // ---- conv2's epilogue -> row stores
        if (do2) {
#pragma unroll
            for (int t = 0; t < PT; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < cpp) {            // (wave uniform: C2 is a multiple of 8, so both halves of quad q exist or neither does)
                        const f32x4 w4 = *(const f32x4*)(epi + eo + 32 + 8 * q + 4 * hi);
                        u16x4 o4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc2[t][q * 4 + e] + nz2[t] * w4[e];
                            o4[e] = gg_f2bf(fmaxf(v, v * slope2));
                        }
                        // (8 bytes per lane straight from the registers, as gg_sconv stores: parking the row in LDS for 16-byte row
                        // stores cost a wave hand-over and an LDS round trip per row - 73.2 -> 70.2 us at 256x256, 49.1 -> 46.6 us at
                        // 128x128 same-box, profiles/r06_spair_direct_stores_ab.txt)
                        if (!GG_SP_OFF(4)) *(u16x4*)(p.y + (img_pix0 + (long long)o * W + xw + t * 32 + pl) * C2 + 8 * q + 4 * hi) = o4;
                    }
                }
            }
            GG_SP_STAMP(5);
        }

        GG_SP_STAMP(6);
        xs0 = xs0 + 1 == NSX ? 0 : xs0 + 1;
        mi = (mi + 1) & 3;
    }
}
