// gg_aconv.h — the no-grad adaptive (style-modulated, demodulated) 3x3 convolution on a SHARED kernel bank, one launch per layer:
// reference AdaptiveConv2DMod.forward (gp.py:344-409) + Noise (gp.py:925-940) + leaky_relu (gp.py:109) for the generator's
// 4x4 .. 64x64 stages (Generator.forward, gp.py:1184-1245) at batch sizes where the bank, not the activation, is the traffic.
//
//   y[b,o,p] = act( d[b,o] * sum_n a[b,n] * sum_{i,t} W_n[o,i,t] * (s[b,i] * x[b,i,p+t]) + noise[b,p] * nw[o] )
//
// What the earlier kernels of this path paid for (profiles/r04_*, r5_pitch_probe.log): a 256-pixel tile needs split-K over 4 - 16
// workgroups to fill 256 CUs when a whole layer is 512 .. 8192 pixels, i.e. 8 - 16 MB of fp32 partials written and re-read by a
// second launch (gg_lrconv + gg_splitk_reduce: 26 us on a 4x4 layer whose bank streams in 2 us), and every 32 / 64-channel chunk of
// the reduction sits between two workgroup barriers behind a register-staged weight tile (4 - 5 us per chunk measured).
// Here a workgroup owns a SMALL pixel tile (32 / 64 / 128 pixels: whole images, or whole rows of one image) and ALL of its reduction:
//   * the tile's one-pixel halo is parked in LDS ONCE for every input channel, already multiplied by s[b,i] (and by the skip-layer
//     excitation where the layer has one, gp.py:1023-1024) - no chunk loop, no barrier in the main loop;
//   * the bank is stored in MFMA-fragment order by gg_pack_weights (kind 3: [O/32][n][tap][C/16][lane][8]), so a wavefront's weight
//     operand of one k-step is ONE coalesced 1 KB load straight into registers - no LDS round trip, no swizzle, any prefetch depth;
//   * the N kernels of the bank keep SEPARATE fp32 accumulators and are mixed with a[b,n] after the reduction (exact in fp32; the
//     stacked form rounds a*s*x to bf16) - the activation fragment of a k-step is read once for all banks;
//   * the 8 (or 4) wavefronts split the tile's output channels (NWN) and the reduction (NWK); the K-slices are summed through LDS in
//     a fixed order, then demodulation, noise and leaky-relu run on the accumulators: one launch, no workspace, no partials in HBM.
// Bound: the bank streams L2 -> registers once per workgroup: BN * 9 C * N * 2 bytes per BMT * BN * 9 C * N * 2 flops, i.e. 1 / BMT bytes
// per flop - 64 B/clk/CU at the matrix pipe's full rate for a 64-pixel tile (the L2's delivery rate), half that for 128 pixels.
// Algorithmic work: 2 b O I 9 H W flops (the stacked banks multiply N times that); bytes: the bank once per XCD + x + y.
#pragma once
#include "gg_device.h"

struct GgAconvParams {
    const bf16_t* x;        // [b][H][W][C]
    const bf16_t* wf;       // fragment-ordered bank [O/32][NB][9][C/16][64][8] (gg_pack_weights kind 3)
    bf16_t* y;              // [b][H][W][O]
    const float* s;         // [b][C] input scale (mod + 1)
    const float* xs;        // optional [b][C]: a second input scale (skip-layer excitation)
    const float* a;         // [b][NB] bank weights (softmax of kernel_mod); null: all ones (NB == 1)
    const float* d;         // [b][O] demodulation coefficients, or null
    const float* noise;     // [b][H*W] or null
    const float* noise_w;   // [O] (with noise)
    int b, H, W, C, O;
    int w_shift, hw_shift;  // log2 W, log2 (H * W)
    int c8_shift;           // log2 (C / 8)
    int act;                // 0 none, 1 leaky relu
    float slope;
    int mt;                 // pixel tiles (gridDim.x = mt * O / BN)
    int inv_spi, inv_hwp;   // 16.16 reciprocals (rounded up) of the halo slots per image / the halo row length
    // the NEXT launch's bank (optional): its workgroups wg' = tile_n' * pf_mt + tile_m' (pf_grid of them, dealt to the XCDs as this
    // kernel's are) stream pf_tn_bytes each from pf_wf + tile_n' * pf_tn_bytes
    const bf16_t* pf_wf;
    long long pf_bytes;
    int pf_tn_bytes, pf_mt, pf_grid;
    int dbg;                // probe builds only (-DGG_PROBE; gg_aconv_desc.reserved): 1 = no reduction loop, 2 = no halo staging; ignored by the product library
    long long x_bytes, wf_bytes;
#if defined(GG_AC_PROBE)        // tests/probes/aconv_probe.hip only: phase time stamps (s_memtime) of every workgroup's first and last wavefront
    long long* stamps;          // [workgroup][2][8]
#endif
};
#if defined(GG_AC_PROBE)
#define GG_AC_STAMP(slot) do { if (lane == 0 && (wave == 0 || wave == NW - 1)) p.stamps[((long long)blockIdx.x * 2 + (wave != 0)) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GG_AC_STAMP(slot) do {} while (0)
#endif

// k-steps of weight fragments in flight per wavefront (multiples of 3: they divide every slice length the host picks). What the
// first measurements said (profiles/r5_aconv_probe*.log): the weight stream runs at what the L2 delivers (64 B/clk per CU, ~25-30 TB/s
// chip-wide) whatever the depth; what dominated was a FIXED ~15 us per workgroup: three dependent round trips in front of the loop
// (halo loads in batches behind integer divisions, then the first weight fragments) and one behind it (demodulation / noise operands).
// (Round 6, last day: the two-kernel bank on a 32-pixel tile - the 4x4 layers - ran twelve deep with 256 registers and 68-100 bytes of scratch
// per lane: its finishing operands were spilled across the loop, the spill stores waiting for their loads in FRONT of it. Nine deep: 244
// registers, no scratch, 22.6 -> 19.8 us per layer same-box, profiles/r06_aconv_4x4_spill_ab.log.)
template <int NB, int TM> struct GgAcDepth { static constexpr int PD = TM == 1 ? (NB == 1 ? 12 : 9) : (TM == 2 ? 6 : 3) * (NB == 1 ? 2 : 1); };

#define GG_AC_XV 16         // halo vectors a thread keeps in flight (all of them on every shape the model has)

template <int NB, int TM, int NWN, int NWK>
GG_KERNEL GG_LAUNCH_BOUNDS(64 * NWN * NWK) void gg_aconv_kernel(GgAconvParams p) {
    constexpr int NW = NWN * NWK, NT = 64 * NW;
    constexpr int BMT = 32 * TM, BN = 32 * NWN;
    constexpr int PD = GgAcDepth<NB, TM>::PD;
    GG_DYN_SHARED(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = gg_uniform(tid >> 6);
    const int wn = wave % NWN, wk = wave / NWN;
    GG_AC_STAMP(0);

    // XCD-aware tile order (workgroup -> XCD is round-robin in blockIdx): the pixel tiles of one output-channel tile - the readers of
    // one weight stream - are consecutive in `wg`, so they share an XCD's L2 and the bank leaves HBM once per XCD slice
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + pos;
    const int tile_n = wg / p.mt, tile_m = wg - tile_n * p.mt;
    const int m0 = tile_m * BMT, n0 = tile_n * BN;

    // tile geometry: RT whole rows of one image (H * W >= BMT) or TI <= 2 whole images
    const int W = p.W, H = p.H, ws = p.w_shift, hs = p.hw_shift, HW = H * W;
    const bool rows = HW >= BMT;
    const int RT = rows ? BMT >> ws : H;
    const int TI = rows ? 1 : BMT >> hs;
    const int HWp = W + 2, SPI = (RT + 2) * HWp;            // halo row length, halo slots per image
    const int NS = TI * SPI;
    const int PITCH = p.C * 2 + 16;                         // bytes per halo slot (+16: consecutive slots rotate through the banks)
    const int img0 = m0 >> hs;
    const int row0 = rows ? (m0 & (HW - 1)) >> ws : 0;

    GgBuf bufX = gg_make_buf((const void*)p.x, (unsigned long long)p.x_bytes);
    GgBuf bufW = gg_make_buf((const void*)p.wf, (unsigned long long)p.wf_bytes);

    // this wavefront's slice of the (tap, channel-block) steps and its 32-row block of the bank
    const int cbs = p.c8_shift - 1;                          // log2 (C / 16)
    const int KS = 9 << cbs;                                 // k-steps of one bank
    const int per = (KS + NWK - 1) / NWK;
    const int k_lo = wk * per;
    const int k_hi = k_lo + per < KS ? k_lo + per : KS;
    const int ot = tile_n * NWN + wn;
    const unsigned wlane = (unsigned)lane * 16u;
    auto wsoff = [&](int n, int k) {                         // byte offset of (bank n, k-step k) of this wavefront's block: scalar
        const int kc = k < KS ? k : KS - 1;                  // (prefetches past the slice stay inside the bank)
        return (unsigned)((((long long)ot * NB + n) * KS + kc) * 1024);
    };

    // ---- ONE round trip in front of the loop: the thread's halo vectors, its scales and the first weight fragments all in flight ----
    // A thread's vectors v = tid + NT u share their channel group j (NT is a multiple of C / 8), so its 8 scale values depend on the
    // image only: fetched once for the tile's (at most two) images.
    const int c8s = p.c8_shift, nvec = NS << c8s;
    const int j8 = (tid & ((1 << c8s) - 1)) * 8;
    f32x4 sc[2][2];
#pragma unroll
    for (int il = 0; il < 2; ++il) {
        const int img = img0 + il < p.b ? img0 + il : p.b - 1;
        const float* sp = p.s + (long long)img * p.C + j8;
        sc[il][0] = *(const f32x4*)sp;
        sc[il][1] = *(const f32x4*)(sp + 4);
        if (p.xs) {
            const float* ep = p.xs + (long long)img * p.C + j8;
            sc[il][0] = sc[il][0] * *(const f32x4*)ep;
            sc[il][1] = sc[il][1] * *(const f32x4*)(ep + 4);
        }
    }
    // the bank weights a[img][n] of this lane's pixels (needed only after the reduction: requested here, they cost no round trip there)
    float av[TM][NB];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int img = img0 + (rows ? 0 : (i * 32 + (lane & 31)) >> hs);
        const int ic = img < p.b ? img : p.b - 1;
#pragma unroll
        for (int n = 0; n < NB; ++n) av[i][n] = (p.a ? p.a : p.s)[p.a ? (long long)ic * NB + n : 0];
    }
    u16x8 bq[PD][NB];
#if defined(GG_PROBE)
    const int nvec_run = (p.dbg & 2) ? 0 : nvec;
#else
    const int nvec_run = nvec;
#endif
    for (int v0 = tid; v0 < nvec_run || v0 == tid; v0 += NT * GG_AC_XV) {     // (at least once for EVERY thread: the first pass starts the weight stream)
        u16x8 xv[GG_AC_XV];
        int dst[GG_AC_XV];
#pragma unroll
        for (int u = 0; u < GG_AC_XV; ++u) {
            const int v = v0 + NT * u;
            const int slot = v >> c8s;
            // slot -> (image, halo row, halo column) by 16.16 reciprocals (exact below 1024 slots: host-checked)
            const int il = TI > 1 ? (slot * p.inv_spi) >> 16 : 0, rem = slot - il * SPI;
            const int hy = (rem * p.inv_hwp) >> 16, hx = rem - hy * HWp;
            const int ih = row0 + hy - 1, iw = hx - 1, img = img0 + il;
            const bool ok = v < nvec && ih >= 0 && ih < H && iw >= 0 && iw < W && img < p.b;
            const unsigned off = ok ? (unsigned)(((((long long)img * H + ih) * W + iw) * p.C + j8) * 2) : 0xFFFFFFFFu;
            xv[u] = gg_buf_load16(bufX, off, 0);             // (padding and tail vectors: hardware zero fill)
            dst[u] = v < nvec ? (slot * PITCH + j8 * 2) | (il << 24) : -1;
        }
        if (v0 == tid) {                                     // behind the first batch of halo loads: the weight stream starts now
#pragma unroll
            for (int u = 0; u < PD; ++u)
#pragma unroll
                for (int n = 0; n < NB; ++n) bq[u][n] = gg_buf_load16(bufW, wlane, wsoff(n, k_lo + u));
        }
#pragma unroll
        for (int u = 0; u < GG_AC_XV; ++u)
            if (dst[u] >= 0) {
                const int il = dst[u] >> 24;
                const f32x4 s0 = il ? sc[1][0] : sc[0][0], s1 = il ? sc[1][1] : sc[0][1];
                u16x8 h = xv[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = gg_f2bf(gg_bf2f(h[e]) * s0[e]);
                    h[e + 4] = gg_f2bf(gg_bf2f(h[e + 4]) * s1[e]);
                }
                *(u16x8*)(smem + (dst[u] & 0xFFFFFF)) = h;
            }
    }
    GG_AC_STAMP(1);
    gg_sync();
    GG_AC_STAMP(2);

    // ---- the reduction: this wavefront's 32 output channels x BMT pixels x its k-steps; no barrier, no LDS write ------------------
    int a_base[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = i * 32 + (lane & 31);
        const int il = rows ? 0 : r >> hs;
        const int rr = rows ? r : r & (HW - 1);
        a_base[i] = (il * SPI + (rr >> ws) * HWp + (rr & (W - 1))) * PITCH + (lane >> 5) * 16;
    }

    f32x16 acc[NB][TM];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][i][r] = 0.f;

    // Who finishes what (round 6): the K-slices used to meet in the wk == 0 wavefronts, which then ran the whole epilogue (NWN of eight
    // wavefronts: 6 k cycles behind the last MFMA on a 128-pixel tile, profiles/r06_aconv_probe_hot_cold_mall.log). Now every wavefront
    // owns the register quads Q = 4 i + g (pixel block i, channels chb + 8 g .. + 3) with Q % NWK == wk of its column block: it sums
    // THOSE over the slices (slice order 0, 1, .. as before: the same bits) and stores them - the sum and the epilogue spread over all
    // wavefronts. The finishing operands (demodulation, noise) of the owned quads are requested in front of the reduction (always
    // issued, from clamped addresses: loads under a condition are waited for on the spot).
    constexpr int NQ = TM * 4, OWN = (NQ + NWK - 1) / NWK;    // quads of a column block, quads a wavefront owns at most
    const int chb = n0 + wn * 32 + 4 * (lane >> 5);           // output channel of accumulator register q: chb + (q & 3) + 8 * (q >> 2)
    f32x4 dv[OWN], nwv[OWN];
    float nzv[OWN];
#pragma unroll
    for (int j = 0; j < OWN; ++j) {
        const int Q = (wk + j * NWK) % NQ, i = Q >> 2, g = Q & 3;          // (beyond NQ: a clamped duplicate, never used)
        const long long pix = (long long)m0 + i * 32 + (lane & 31);
        const int img = (int)(pix >> hs);
        const int ic = img < p.b ? img : p.b - 1;
        const long long pc = img < p.b ? pix : 0;
        nzv[j] = (p.noise ? p.noise : p.s)[p.noise ? pc : 0];
        nwv[j] = *(const f32x4*)((p.noise ? p.noise_w : p.s) + (p.noise ? chb + 8 * g : 0));
        dv[j] = *(const f32x4*)((p.d ? p.d + (long long)ic * p.O + chb + 8 * g : p.s));
    }

#if defined(GG_PROBE)
    const int k_end = (p.dbg & 1) ? k_lo : k_hi;
#else
    const int k_end = k_hi;
#endif
    for (int k0 = k_lo; k0 < k_end; k0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int k = k0 + u;
            u16x8 bw[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) bw[n] = bq[u][n];
#pragma unroll
            for (int n = 0; n < NB; ++n) bq[u][n] = gg_buf_load16(bufW, wlane, wsoff(n, k + PD));
            if (k < k_hi) {                                  // (wave-uniform; slices are multiples of the depth on the model's shapes)
                const int t = k >> cbs, cb = k & ((1 << cbs) - 1);
                const int kh = (t * 11) >> 5, kw = t - 3 * kh;
                const int toff = (kh * HWp + kw) * PITCH + cb * 32;
                u16x8 fa[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *(const u16x8*)(smem + a_base[i] + toff);
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[n][i] = gg_mfma_32x32x16_bf16(bw[n], fa[i], acc[n][i]);
            }
        }
    }

    GG_AC_STAMP(3);
    // ---- mix the banks (fp32), sum the K-slices through LDS in slice order, finish -----------------------------------------------
    const GgAconvParams e = *gg_late_params(p);
    f32x16 out[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float v = (e.a ? av[i][0] : 1.f) * acc[0][i][q];
#pragma unroll
            for (int n = 1; n < NB; ++n) v = gg_fmaf(e.a ? av[i][n] : 1.f, acc[n][i][q], v);
            out[i][q] = v;
        }
    }
    auto finish = [&](int j, int i, int g, const f32x4& sum) {
        const long long pix = (long long)m0 + i * 32 + (lane & 31);              // global pixel index (images are contiguous)
        if ((int)(pix >> hs) >= e.b) return;
        const float nz = e.noise ? nzv[j] : 0.f;
        u16x4 o4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float dd = e.d ? dv[j][c] : 1.f, nn = e.noise ? nwv[j][c] : 0.f;
            float v = gg_fmaf(sum[c], dd, nz * nn);
            if (e.act == 1) v = v > 0.f ? v : v * e.slope;
            o4[c] = gg_f2bf(v);
        }
        *(u16x4*)(e.y + pix * e.O + chb + 8 * g) = o4;
    };
    if (NWK > 1) {
        gg_sync();                                           // every wavefront is past its last read of the halo
        f32x4* red = (f32x4*)smem;                           // [wk][wn][Q][64 lanes] quads
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q)
            if (Q % NWK != wk) {                             // (wave uniform) somebody else's quad: hand this slice's share over
                const f32x4 v = {out[Q >> 2][(Q & 3) * 4], out[Q >> 2][(Q & 3) * 4 + 1], out[Q >> 2][(Q & 3) * 4 + 2], out[Q >> 2][(Q & 3) * 4 + 3]};
                red[((wk * NWN + wn) * NQ + Q) * 64 + lane] = v;
            }
        gg_sync();
        GG_AC_STAMP(4);
#pragma unroll
        for (int j = 0; j < OWN; ++j)
#pragma unroll
            for (int Q = j * NWK; Q < (j + 1) * NWK && Q < NQ; ++Q)
                if (Q % NWK == wk) {                         // (wave uniform) this wavefront's quad: the slices in order, its own from registers
                    const f32x4 mine = {out[Q >> 2][(Q & 3) * 4], out[Q >> 2][(Q & 3) * 4 + 1], out[Q >> 2][(Q & 3) * 4 + 2], out[Q >> 2][(Q & 3) * 4 + 3]};
                    f32x4 sum = wk == 0 ? mine : red[((0 * NWN + wn) * NQ + Q) * 64 + lane];
#pragma unroll
                    for (int sl = 1; sl < NWK; ++sl) {
                        const f32x4 t = sl == wk ? mine : red[((sl * NWN + wn) * NQ + Q) * 64 + lane];
                        sum = sum + t;
                    }
                    finish(j, Q >> 2, Q & 3, sum);
                }
        if (wk > 0 && e.pf_wf) {
            // request the slice of the NEXT layer's bank that the workgroups of the next launch on THIS XCD will stream (same blockIdx ->
            // XCD dealing), split over this XCD's workgroups - it lands in this XCD's L2 behind the tile's stores. The loads are never
            // waited for (s_endpgm retires them).
            const int g2 = e.pf_grid, q2 = g2 >> 3, r2 = g2 & 7;
            const int lo = xcd < r2 ? xcd * (q2 + 1) : r2 * (q2 + 1) + (xcd - r2) * q2;
            const int cnt = q2 + (xcd < r2 ? 1 : 0);
            if (cnt > 0) {
                const long long b_lo = (long long)(lo / e.pf_mt) * e.pf_tn_bytes;
                long long b_hi = (long long)((lo + cnt - 1) / e.pf_mt + 1) * e.pf_tn_bytes;
                b_hi = b_hi < e.pf_bytes ? b_hi : e.pf_bytes;
                const int mine = (nwg + 7 - xcd) >> 3;                     // workgroups of THIS launch on this XCD; `pos` is ours
                const long long share = (((b_hi - b_lo) / mine + 1023) >> 10) << 10;
                const long long s_lo = b_lo + share * pos;
                const long long s_hi = s_lo + share < b_hi ? s_lo + share : b_hi;
                GgBuf bufP = gg_make_buf((const void*)e.pf_wf, (unsigned long long)e.pf_bytes);
                constexpr int NPT = 64 * NWN * (NWK - 1);
                for (long long o = s_lo + (long long)(tid - 64 * NWN) * 16; o < s_hi; o += NPT * 16) gg_buf_touch16(bufP, (unsigned)o);
            }
        }
    } else {
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q) {
            const f32x4 mine = {out[Q >> 2][(Q & 3) * 4], out[Q >> 2][(Q & 3) * 4 + 1], out[Q >> 2][(Q & 3) * 4 + 2], out[Q >> 2][(Q & 3) * 4 + 3]};
            finish(Q, Q >> 2, Q & 3, mine);
        }
    }
    GG_AC_STAMP(5);
}
