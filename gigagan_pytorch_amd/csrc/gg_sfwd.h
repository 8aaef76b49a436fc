// gg_sfwd.h — STREAMING forward / data-gradient convolution of the narrow high-resolution layers (plan tile 14): 3x3 / stride 1 / pad 1,
// C_in in {8, 16, 32, 64}, C_out <= 64, 64..256-wide power-of-two images, shared OR per-image weights — the discriminator's stem and
// first blocks with their data gradients (gigagan_pytorch.py:1608-1621), the plain and the adaptive convolutions of the generator's last
// blocks in the no-grad pass (per-sample weights, gp.py:390-409, noise + leaky-relu of gp.py:1030-1046 fused).
// These layers are HBM problems: 2 * (C_in + C_out) bytes per pixel against 18 * C_in * C_out flops. gg_dconv (LDS tile per 256 pixels,
// four barriers and a load -> stage -> multiply -> stage -> store chain per tile) measured 2.0-2.4 TB/s on them, the 4-wave implicit
// GEMM 1.1 TB/s (stem, 64 -> 64), gg_sconv 2.9-5.5 TB/s. Same skeleton as gg_wgrads.h:
//   * a LOADER wave (wave 4) streams whole image rows HBM -> LDS ring by LDS-DMA `depth` steps ahead; its only wait is a counted
//     s_waitcnt vmcnt in front of the step's raw s_barrier. The four COMPUTE waves never wait on vector memory in the loop (their
//     stores and the loader's transfers live in different per-wave counters: on gfx9 loads and stores share vmcnt and may retire out
//     of order relative to each other, so a wave that also stored could not count its transfers);
//   * a row is fetched once and serves the three row taps of three consecutive steps; one zero slot left and right of every row, a zero
//     row for the rows above / below an image;
//   * weights live in REGISTERS as MFMA A fragments ([co][tap][ci] rows: 16 bytes per lane and fragment straight from global memory,
//     no LDS), reloaded when the image changes (per-image weights) and multiplied by the per-image input scale there (the skip-layer
//     excitation / style modulation: conv(x * s, w) = conv(x, w * s));
//   * MFMA orientation D[co][pixel]: a lane owns one pixel and 16 output channels per 32 x 32 block: bias / noise / activation in
//     registers, then the wave parks its 64 (128) pixels in a private LDS area and writes them back as whole 16-byte-per-lane rows
//     (+ residual read the same way).
// Algorithmic bytes: 2 * (C_in + C_out) per pixel (+ 4 B noise, + 2 * C_out residual).
#pragma once
#include "gg_wgrads.h"

#define GG_SF_NT 320                          // four compute waves + the loader
GG_HOST_DEVICE int gg_sf_stage_bytes(int N, int spx) {      // output staging per compute wave: its pixels x (its channels + 8) bf16
    return N > 32 ? (spx / 2) * 80 : (spx / 4) * ((N < 32 ? N : 32) * 2 + 16);
}

struct GgSfGeom {
    int sbx, npx, xpp, xp, rs, nrx;
    long long bytes;
};
GG_HOST_DEVICE GgSfGeom gg_sf_geom(int W, int C, int N, int spx, int depth) {
    GgSfGeom g;
    g.sbx = (C < 32 ? C : 32) * 2; g.npx = C <= 32 ? 1 : C >> 5;
    g.xpp = (W + 2) * g.sbx; g.xp = g.npx * g.xpp;
    g.rs = spx / W;
    g.nrx = g.rs * (depth + 1) + 2;
    g.bytes = (long long)(g.nrx + 1) * g.xp + 4 * gg_sf_stage_bytes(N, spx) + 512 + GG_WS_SLACK;
    return g;
}

// CK: 16-channel k-steps per tap (C / 16, 1 for C = 8); NSPLIT: the output channels are split over wave pairs (C_out = 64: waves
// (w & 1) take 32 channels each of 128 pixels; else every wave takes all (<= 32) channels of 64 pixels)
template <int CK, bool NSPLIT>
GG_KERNEL GG_LAUNCH_BOUNDS(GG_SF_NT) void gg_sfwd_kernel(GgGemmParams p) {
    constexpr int PB = (CK == 4 ? 2 : 4) / (NSPLIT ? 1 : 2);   // 32-pixel blocks per wave and step: 64-channel inputs run 128-pixel steps (host)
    GG_SHARED __attribute__((aligned(16))) char smem[GG_WS_LDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = gg_uniform(tid >> 6);
    const int W = p.W, H = p.H, C = p.C, N = p.N, ws = p.w_shift;
    const int SPX = p.ws_spx, D = p.ws_depth;
    const GgSfGeom g = gg_sf_geom(W, C, N, SPX, D);
    const int RS = g.rs, XP = g.xp, NRx = g.nrx;
    const int zrow = NRx * XP, stage0 = zrow + XP, epi0 = stage0 + 4 * gg_sf_stage_bytes(N, SPX);

    const int total_steps = p.M / SPX, total_rows = p.M >> ws;
    const int spw = p.k_per_split / SPX;      // (steps per workgroup travel in k_per_split)
    const int s0 = blockIdx.x * spw;
    int s1 = s0 + spw;
    if (s1 > total_steps) s1 = total_steps;

    for (int v = tid; v < stage0 / 16; v += GG_SF_NT) *(u16x8*)(smem + v * 16) = gg_zero8();
    if (tid < 128) {                          // epilogue constants: bias * bias_scale [64] | noise_w [64]
        const int n = tid & 63;
        float v = 0.f;
        if (n < N) v = tid < 64 ? (p.bias ? p.bias[n] * p.bias_scale : 0.f) : (p.noise ? p.noise_w[n] : 0.f);
        ((float*)(smem + epi0))[tid] = v;
    }
    gg_barrier_lds();

    if (wave == 4) {
        // ---------------------------------------------------------------- loader
        GgBufS bufA = gg_make_bufs((const void*)p.A, (unsigned long long)p.a_bytes);
        const int kbx = (W * g.sbx) >> 10;                                 // 1 KB transfers per row-plane
        const int npr = g.npx * kbx;                                       // ... per row
        const int cps = g.sbx >> 4, csh = g.sbx >> 5;
        const unsigned xrowb = (unsigned)(W * C * 2);
        constexpr int MAXR = 16;                                           // transfers per row (host: W * C <= 8192)
        unsigned voff[MAXR];
        int ldso[MAXR];
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int pl = i / kbx, kb = i - pl * kbx;
            const int j = kb * 64 + lane, slot = j >> csh, pos = j & (cps - 1);
            const int part = pos ^ ((slot >> (g.sbx == 64 ? 2 : 3)) & (cps - 1));     // chunk `part` of pixel `slot` lands at position `pos` (see the reader)
            voff[i] = (unsigned)((slot * C + pl * 32) * 2 + part * 16);
            ldso[i] = pl * g.xpp + g.sbx + kb * 1024;
        }
        int head = 0;
        auto issue_row = [&](int gr) {
            const bool ok = (unsigned)gr < (unsigned)total_rows;
#pragma unroll
            for (int i = 0; i < MAXR; ++i)
                if (i < npr) gg_bufs_load_lds16(bufA, ok ? voff[i] : 0xFFFFFFFFu, ok ? (unsigned)gr * xrowb : 0u, smem + head * XP + ldso[i]);
            head += 1;
            if (head >= NRx) head -= NRx;
        };
        issue_row(s0 * RS - 1);
        issue_row(s0 * RS);
        for (int t = s0; t < s0 + D && t < s1; ++t)
            for (int r = 0; r < RS; ++r) issue_row(t * RS + 1 + r);
        const int per_step = RS * npr;
        for (int s = s0; s < s1; ++s) {
            int ahead = s1 - 1 - s;
            if (ahead > D - 1) ahead = D - 1;
            gg_wait_vm_le(ahead * per_step);
            gg_barrier_lds();
            if (s + D < s1)
                for (int r = 0; r < RS; ++r) issue_row((s + D) * RS + 1 + r);
        }
        gg_wait_vm_le(0);
        return;
    }

    // -------------------------------------------------------------------- compute waves
    const int nbw = NSPLIT ? (wave & 1) : 0;              // 32-channel output block of this wave
    const int pq = NSPLIT ? (wave >> 1) : wave;           // pixel share: PB * 32 pixels starting at pq * PB * 32
    const int frow = lane & 31, hi = lane >> 5;
    const int co_a = nbw * 32 + frow;                     // A-fragment row (output channel) of this lane

    // weights as A fragments: [tap][k-step] 8 bf16 of row co_a, channels kc*16 + 8*hi .. + 7 of tap t (zeros beyond C / N)
    u16x8 wf[9][CK];
    int w_img = -1;
    auto load_weights = [&](int img) {
        const bf16_t* wb = p.B + (long long)img * p.b_img_stride + (long long)co_a * p.ldb;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < CK; ++kc) {
                const int c0 = kc * 16 + 8 * hi;
                u16x8 v = gg_zero8();
                if (co_a < N && c0 < C) v = *(const u16x8*)(wb + t * C + c0);
                if (p.in_scale && co_a < N && c0 < C) v = gg_scale8(v, p.in_scale + (long long)img * C + c0);
                wf[t][kc] = v;
            }
        w_img = img;
    };

    // epilogue operands of the lane's output channels live in LDS (bias * bias_scale | noise_w, 64 floats each, filled before the first
    // barrier): register quad q of a block holds channels 8 q + 4 hi .. + 3 of the wave's 32
    const float* epi = (const float*)(smem + epi0);
    const int sb = g.sbx;
    const int NB2 = (NSPLIT ? 32 : (N < 32 ? N : 32)) * 2;      // staged bytes per pixel (this wave's channels)
    char* stage = smem + stage0 + wave * gg_sf_stage_bytes(N, SPX);
    const int hw_shift = p.hw_shift;
    const float alpha = p.alpha, slope = p.act == GG_ACT_LRELU ? p.act_slope : 1.f;
    // a slot's 16-byte chunks are stored XOR-swizzled by the pixel's x coordinate (the loader picks which global chunk each DMA lane
    // fetches): chunk c of pixel x sits at position c ^ key(x), key = (x >> 2) & 3 for 64-byte slots, (x >> 3) & 1 for 32-byte ones. A
    // ds_read_b128 serves 16 lanes = 16 consecutive pixels per cycle group; unswizzled they hit only 4 (8) of the 16 bank quads
    const int kmask = (sb >> 4) - 1, kshift = sb == 64 ? 2 : 3;           // (16-byte slots: kmask 0)

    // the step's work in UNITS of (pixel block b, tap row kh, channel group): FPU fragments each; the reads of unit u + 1 are issued
    // before the MFMAs of unit u (one or two waves per SIMD: nothing else hides the LDS latency), and a block's MFMAs alternate between
    // two accumulator chains
    constexpr int KG = CK > 2 ? 2 : 1;                    // channel groups of <= 2 k-steps per tap
    constexpr int KPG = CK / KG;                          // k-steps per group
    constexpr int FPU = 3 * KPG, NU = PB * 3 * KG;
    constexpr int NACC = (CK == 4 || NSPLIT) ? 1 : 2;      // (register budget: two waves share SIMD 0's file, 256 registers each)

    // everything of a fragment address that does not change from step to step, per lane: [pixel block][kw][k-step of the tap] ->
    // byte offset inside a ring row (slot of pixel x + kw - 1, plane, swizzled chunk)
    int fo[PB][3][CK];
    int ryb[PB];                                          // image row of the block inside the step
#pragma unroll
    for (int b = 0; b < PB; ++b) {
        const int px = (pq * PB + b) * 32, cx = px & (W - 1);
        ryb[b] = px >> ws;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int X = cx + frow + kw - 1;                              // pixel x coordinate (-1 / W: the zero slots)
            const int key = (X >> kshift) & kmask;
#pragma unroll
            for (int kc = 0; kc < CK; ++kc)                                // channels kc*16 + 8*hi ..: plane kc >> 1, chunk 2 * (kc & 1) + hi
                fo[b][kw][kc] = (X + 1) * sb + (kc >> 1) * g.xpp + (((2 * (kc & 1) + hi) ^ key) << 4);
            // 8-channel inputs: the upper half of the 16-channel k-step does not exist. Those lanes read the row's left halo slot
            // (constant zeros) - NOT the neighbouring slot: its bytes times the zero weight fragment are zero only while they are
            // finite, and past the zero row they are whatever the previous kernel left in LDS (NaN under hipGraph replay: found
            // as a non-finite generator loss in the text-conditional bench, profiles/r04_text_nan.log)
            if (C < 16 && hi) fo[b][kw][0] = 0;
        }
    }
    const int SPITCH = NB2 + 16;                          // staging pitch per pixel (bank spread)
    const int st_wr = frow * SPITCH + 8 * hi;             // + b * 32 * SPITCH + q * 16
    // write-back: the wave's PB * 32 pixels x NB2 bytes as 16-byte chunks, chunk c = lane + 64 it -> (pixel, chunk of the pixel)
    constexpr int WBI = PB * 2;                           // iterations at 4 chunks per pixel (fewer chunks: fewer iterations)
    const int cpp = NB2 >> 4, total_chunks = PB * 32 * cpp;
    int wb_lds[WBI];
    int wb_out[WBI];                        // element offsets inside the step's output rows (< 2^31; the residual has the same pitch: host)
#pragma unroll
    for (int it = 0; it < WBI; ++it) {
        const int c = lane + 64 * it, pix = c / cpp, ch = c - pix * cpp;
        wb_lds[it] = c < total_chunks ? pix * SPITCH + ch * 16 : -1;
        wb_out[it] = (pq * PB * 32 + pix) * p.ldc + nbw * 32 + ch * 8;
    }

    int tail = 0;                                         // ring row of image row R0 - 1
    float nz[PB];
    auto fetch_noise = [&](int s) {
#pragma unroll
        for (int b = 0; b < PB; ++b) nz[b] = (p.noise && s < s1) ? p.noise[(long long)s * SPX + (pq * PB + b) * 32 + frow] : 0.f;
    };
    fetch_noise(s0);
    for (int s = s0; s < s1; ++s) {
        const int R0 = s * RS;
        const int img = (R0 << ws) >> hw_shift;
        if (img != w_img && (p.b_img_stride || p.in_scale || w_img < 0)) load_weights(img);
        gg_barrier_lds();                                 // the loader has seen this step's rows land
        f32x16 acc[PB][NACC];
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][a][r] = 0.f;

        auto load_unit = [&](int u, u16x8 (&f)[FPU]) {
            const int b = u / (3 * KG), kh = (u / KG) % 3, kg = u % KG;
            const int y = (R0 + ryb[b]) & (H - 1);
            int xr = tail + ryb[b] + kh;
            if (xr >= NRx) xr -= NRx;
            const bool in = (unsigned)(y + kh - 1) < (unsigned)H;
            const char* row = smem + (in ? xr * XP : zrow);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int k = 0; k < KPG; ++k) f[kw * KPG + k] = *(const u16x8*)(row + fo[b][kw][kg * KPG + k]);
        };
        auto mul_unit = [&](int u, const u16x8 (&f)[FPU]) {
            const int b = u / (3 * KG), kh = (u / KG) % 3, kg = u % KG;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int k = 0; k < KPG; ++k) {
                    const int a = (kw * KPG + k) % NACC;
                    acc[b][a] = gg_mfma_32x32x16_bf16(wf[kh * 3 + kw][kg * KPG + k], f[kw * KPG + k], acc[b][a]);   // D[co][pixel]
                }
        };
        u16x8 fA[FPU], fB[FPU];
        load_unit(0, fA);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 1 < NU) {
                if (u & 1) load_unit(u + 1, fA);
                else load_unit(u + 1, fB);
            }
            if (u & 1) mul_unit(u, fB);
            else mul_unit(u, fA);
        }

        // epilogue in registers, then through the wave's staging area: [pixel][NB2 bytes], pitch NB2 + 16
#pragma unroll
        for (int b = 0; b < PB; ++b) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 cb = *(const f32x4*)(epi + nbw * 32 + 8 * q + 4 * hi);
                const f32x4 cw = *(const f32x4*)(epi + 64 + nbw * 32 + 8 * q + 4 * hi);
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    float v = acc[b][0][r];
                    if (NACC == 2) v += acc[b][NACC - 1][r];
                    v = v * alpha + cb[e] + nz[b] * cw[e];
                    v = v > 0.f ? v : v * slope;          // leaky-relu, or the identity (slope 1): branch-free (host: no other activation here)
                    o[e] = gg_f2bf(v);
                }
                if ((8 * q + 4 * hi) * 2 < NB2) *(u16x4*)(stage + st_wr + b * 32 * SPITCH + q * 16) = o;
            }
        }
        gg_wave_sync();
        // the next step's noise is requested BEFORE this step's stores: loads and stores share the wave's vmcnt, and a wait for a load
        // issued behind the stores would be a wait for the stores (their round trip, every step)
        fetch_noise(s + 1);
        // write-back: 16 bytes per lane, rows contiguous in memory (pixel pitch N * 2 bytes)
        bf16_t* outp = (bf16_t*)p.Cout + (long long)s * SPX * p.ldc;
        const bf16_t* resp = p.residual ? p.residual + (long long)s * SPX * p.ldc : nullptr;
#pragma unroll
        for (int it = 0; it < WBI; ++it) {
            if (wb_lds[it] >= 0) {
                u16x8 o = *(const u16x8*)(stage + wb_lds[it]);
                if (resp) {
                    const u16x8 rr = *(const u16x8*)(resp + wb_out[it]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = gg_f2bf(gg_bf2f(o[e]) + gg_bf2f(rr[e]) * p.res_scale);
                }
                *(u16x8*)(outp + wb_out[it]) = o;
            }
        }
        gg_wave_sync();                                   // (the next step's staging writes follow this step's reads)
        tail += RS;
        if (tail >= NRx) tail -= NRx;
    }
}
