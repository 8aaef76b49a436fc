// gg_wgrads.h — STREAMING weight gradient of the narrow high-resolution convolutions (plan tile 13): 3x3 / stride 1 / pad 1, 1x1, and the
// two stride-2 windows without overlap (2x2 = space-to-depth + 1x1, and the 1x1 / stride 2 residual projection), C_in and C_out <= 64
// (at most four 32 x 32 blocks), 64..256-wide power-of-two output maps — the discriminator's stem and first blocks and
// the generator's last blocks (autograd of the F.conv2d call sites gigagan_pytorch.py:402-409, :1608-1621, and of the to/from-rgb 1x1s
// :1066-1070, :1648):
//     dW[tap][ci][co] = sum over pixels of x[pixel + tap][ci] * dy[pixel][co]        fp32 [tap * C + ci][co]
// These launches are HBM problems (2 * (C + N) bytes per pixel against 18 * C * N flops: 537 MB and 77 GFLOP for 32 -> 32 at 256 x 256,
// batch 64): the implicit-GEMM weight gradient on the 4-wave kernel ran them at 0.5-1.1 TB/s (a 128 x 32 output tile per workgroup,
// thousands of k-slices of a few k-tiles each, every input pixel gathered nine times through L1, 18-75 MB of fp32 partials).
//
// Here a workgroup (4 waves) streams a contiguous run of image rows ONCE:
//   * x and dy rows go HBM -> LDS by LDS-DMA (buffer_load ... lds from inline assembly, gg_device.h: no staging registers, no compiler-inserted waits) into two rings, `depth` steps ahead of the
//     multiplication (a step = 256 or 128 pixels = whole image rows); the only wait is a counted s_waitcnt vmcnt that leaves the newest
//     (depth - 1) steps in flight, and the workgroup barrier is the raw s_barrier (__syncthreads would drain the prefetch);
//   * a row lives in LDS for three steps: the nine taps of a step read rows y - 1, y, y + 1 of the ring at a tap-uniform slot offset
//     (one zero slot left and right of every row, a zero row standing in for the rows above / below an image);
//   * both MFMA operands want the reduction (pixels) contiguous per lane while NHWC memory has the channels contiguous: the fragments are
//     ds_read_b64_tr_b16 transpose reads (as gg_wgrad9.h); channels >= C / >= N of a 32-row block read neighbouring bytes and produce
//     output rows / columns that are never stored (MFMA rows are independent);
//   * every wave keeps all taps of ONE (32 ci x 32 co) block in registers (144 accumulators) and takes a share of the step's 16-pixel
//     k-steps; waves sharing a block are summed through LDS at the end; one fp32 partial [9C][N] per workgroup (<= 256 of them), finished
//     by gg_splitk_reduce / gg_wgrad_finish like every other split weight gradient.
// The stride-2 windows are the same kernel over SUPER-PIXELS: an input row of 2 W pixels x C channels is W slots of 2 C channels, so
// the 2x2 window is two vertically stacked 1-tap reductions over slots (tap = input row parity; [ty][tx * C + c] is exactly the
// [tap][ci] order of the output) and the 1x1 / stride 2 one reads the even rows and stores only the first C of the 2 C slot channels.
// Algorithmic bytes: 2 * (C + N) per pixel (+ (W + 2) / W halo columns never fetched: the halo slots are constants).
#pragma once
#include "gg_gemm2.h"

#define GG_WS_NT 256
#define GG_WS_LDS 155648                      // 152 KB: ring budget (one workgroup per CU)
#define GG_WS_SLACK 128                       // bytes past the last ring row that over-wide fragment reads may touch

// host + device: the ring geometry of a (window, W, slot channels, N, pixels per step, depth) choice. `cs`: channels per x slot (C, or
// 2 C for the stride-2 windows); `rpr`: x ring rows per output row (2 for the 2x2 window: both input rows of the pair are kept)
struct GgWsGeom {
    int sbx, npx, xpp, xp, sby, npy, ypp, yp, rs, rsx, nrx, nry, halo;
    long long bytes;
};
GG_HOST_DEVICE GgWsGeom gg_ws_geom(int halo, int rpr, int W, int cs, int N, int spx, int depth) {
    GgWsGeom g;
    g.halo = halo;
    g.sbx = (cs < 32 ? cs : 32) * 2; g.npx = cs <= 32 ? 1 : cs >> 5;
    g.sby = (N < 32 ? N : 32) * 2; g.npy = N <= 32 ? 1 : N >> 5;
    g.xpp = (W + 2 * halo) * g.sbx; g.xp = g.npx * g.xpp;
    g.ypp = W * g.sby; g.yp = g.npy * g.ypp;
    g.rs = spx / W; g.rsx = g.rs * rpr;
    g.nrx = g.rsx * (depth + 1) + 2 * halo; g.nry = g.rs * (depth + 1);
    g.bytes = (long long)(g.nrx + halo) * g.xp + (long long)g.nry * g.yp + GG_WS_SLACK;      // (+ the zero row of the 3x3 window)
    return g;
}

// KH x KW: 3 x 3 (stride 1, pad 1), 1 x 1 (stride 1, or stride 2 over super-pixels), 2 x 1 (the 2x2 / stride 2 window over super-pixels).
// Geometry arrives in the OUTPUT map's terms: p.OW x p.OH pixels per image, p.ws_cs channels per x slot, p.ws_cstore rows stored per tap.
template <int KH, int KW>
GG_KERNEL GG_LAUNCH_BOUNDS(GG_WS_NT) void gg_wgrads_kernel(GgGemmParams p) {
    constexpr int TAPS = KH * KW, HALO = KW == 3 ? 1 : 0;
    constexpr int RPR = (KH == 2) ? 2 : 1;            // x ring rows per output row
    GG_SHARED __attribute__((aligned(16))) char smem[GG_WS_LDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = gg_uniform(tid >> 6);
    const int W = p.OW, H = p.OH, C = p.ws_cs, N = p.N, ws = p.w_shift;
    const int SPX = p.ws_spx, D = p.ws_depth;
    const int GMUL = p.ws_gmul;                       // memory row of ring row r: r * GMUL (2: the 1x1 / stride 2 window reads even rows)
    const GgWsGeom g = gg_ws_geom(HALO, RPR, W, C, N, SPX, D);
    const int RS = g.rs, RSX = g.rsx, XP = g.xp, YP = g.yp, NRx = g.nrx, NRy = g.nry;
    const int xring = 0, zrow = NRx * XP, yring = zrow + HALO * XP;

    // work of this workgroup: steps [s0, s1) of SPX pixels each (a step never straddles an image: H * W >= SPX, powers of two)
    const int total_steps = p.K / SPX, total_xrows = (p.K >> ws) * RPR;       // (ring-row units)
    const int spw = p.k_per_split / SPX;
    const int s0 = blockIdx.x * spw;
    int s1 = s0 + spw;
    if (s1 > total_steps) s1 = total_steps;

    // constants of the rings: everything that is never DMA'd must read as zero (halo slots, the zero row)
    for (int v = tid; v < yring / 16; v += GG_WS_NT) *(u16x8*)(smem + v * 16) = gg_zero8();

    GgBufS bufA = gg_make_bufs((const void*)p.A, (unsigned long long)p.a_bytes);
    GgBufS bufB = gg_make_bufs((const void*)p.B, (unsigned long long)p.b_bytes);

    // DMA plan: a wave instruction moves 64 x 16 bytes = 1 KB of one row-plane. x: SPX * C * 2 / 1024 instructions per step, dy:
    // SPX * N * 2 / 1024, dealt round-robin to the four waves (host: both divisible by 4, so every wave issues the same count)
    const int kbx = (W * g.sbx) >> 10, kby = (W * g.sby) >> 10;            // instructions per row-plane
    const int nix = RSX * g.npx * kbx, niy = RS * g.npy * kby;             // per step, whole workgroup
    const int npw = (nix + niy) >> 2;                                      // per wave and step
    const int cpsx = g.sbx >> 4, cpsy = g.sby >> 4;                        // 16-byte chunks per slot (1, 2 or 4)
    const int cshx = g.sbx >> 5, cshy = g.sby >> 5;                        // ... and their log2 (16 -> 0, 32 -> 1, 64 -> 2)

    // This wave's share of a group, decoded ONCE: instruction q of the wave is instruction i = wave + 4 q of the group = (row rr of the
    // group, plane, kilobyte). Per step only the ring row and the row's byte offset change (the decode costs integer divisions: done
    // per instruction and step it was ~600 scalar / vector instructions in front of every step's multiplication: 1.7 us of 2.7)
    constexpr int MAXI = 8;                   // instructions per wave, step and operand (host: SPX * C <= 16384)
    const int nxw = nix >> 2, nyw = niy >> 2;
    const int xrowb = W * C * 2 * GMUL, yrowb = W * p.ldb * 2;              // bytes per ring row in memory
    unsigned xvoff[MAXI], yvoff[MAXI];
    int xlds[MAXI], xrr[MAXI], ylds[MAXI], yrr[MAXI];
#pragma unroll
    for (int q = 0; q < MAXI; ++q) {
        const int i = wave + 4 * q;
        {
            const int rr = i / (g.npx * kbx), rem = i - rr * (g.npx * kbx);
            const int pl = rem / kbx, kb = rem - pl * kbx;
            const int j = kb * 64 + lane, slot = j >> cshx, part = j & (cpsx - 1);
            xrr[q] = rr;
            xlds[q] = xring + pl * g.xpp + HALO * g.sbx + kb * 1024;
            xvoff[q] = (unsigned)((slot * C + pl * 32) * 2 + part * 16);
        }
        {
            const int rr = i / (g.npy * kby), rem = i - rr * (g.npy * kby);
            const int pl = rem / kby, kb = rem - pl * kby;
            const int j = kb * 64 + lane, slot = j >> cshy, part = j & (cpsy - 1);
            yrr[q] = rr;
            ylds[q] = yring + pl * g.ypp + kb * 1024;
            yvoff[q] = (unsigned)((slot * p.ldb + pl * 32) * 2 + part * 16);
        }
    }

    int xhead = 0, yhead = 0;                 // ring rows the next issued x / dy row lands in
    auto issue_x = [&](int gr, int ring_row, int q) {     // this wave's instruction q of x row gr; rows outside the tensor land as zeros
        const bool ok = (unsigned)gr < (unsigned)total_xrows;
        gg_bufs_load_lds16(bufA, ok ? xvoff[q] : 0xFFFFFFFFu, ok ? (unsigned)gr * (unsigned)xrowb : 0u, smem + ring_row * XP + xlds[q]);
    };
    auto issue_group = [&](int t) {           // the rows step t adds to the rings: x rows t*RSX + HALO .. + RSX - 1, dy rows t*RS ..
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
            if (q < nxw) {
                int ring_row = xhead + xrr[q];
                if (ring_row >= NRx) ring_row -= NRx;
                issue_x(t * RSX + HALO + xrr[q], ring_row, q);
            }
        }
        xhead += RSX;
        if (xhead >= NRx) xhead -= NRx;
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
            if (q < nyw) {
                int ring_row = yhead + yrr[q];
                if (ring_row >= NRy) ring_row -= NRy;
                gg_bufs_load_lds16(bufB, yvoff[q], (unsigned)(t * RS + yrr[q]) * (unsigned)yrowb, smem + ring_row * YP + ylds[q]);
            }
        }
        yhead += RS;
        if (yhead >= NRy) yhead -= NRy;
    };

    gg_barrier_lds();                         // the zero fill is complete before any transfer may land next to it
    if (HALO) {                               // rows s0*RS - 1 and s0*RS: the window of the first step (dealt like a group's rows)
        for (int r = 0; r < 2; ++r) {         // (the first row of a group is instructions 0 .. npx * kbx - 1: the same decode applies)
#pragma unroll
            for (int q = 0; q < MAXI; ++q)
                if (wave + 4 * q < g.npx * kbx) issue_x(s0 * RS - 1 + r, xhead, q);
            xhead += 1;
            if (xhead >= NRx) xhead -= NRx;
        }
    }
    for (int t = s0; t < s0 + D && t < s1; ++t) issue_group(t);

    // block and k-step share of this wave
    const int nbx = g.npx, nby = g.npy, nblk = nbx * nby;                  // 1, 2 or 4 blocks of 32 x 32
    const int blk = wave % nblk, kpart = wave / nblk, kparts = 4 / nblk;
    const int cb = blk / nby, nb = blk - cb * nby;
    const int li = lane & 15, lg = lane >> 4;
    const int lpix = (lg >> 1) * 8 + (li >> 2), lch = ((lg & 1) * 16 + 4 * (li & 3)) * 2;
    const int xlane = lpix * g.sbx + lch + cb * g.xpp, ylane = lpix * g.sby + lch + nb * g.ypp;
    const int KSTEPS = SPX >> 4;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int xtail = 0, ytail = 0;                 // ring rows of image row R0 - HALO (x) and R0 (dy) of the current step
    for (int s = s0; s < s1; ++s) {
        int ahead = s1 - 1 - s;               // groups younger than this step's that are in flight
        if (ahead > D - 1) ahead = D - 1;
        gg_wait_vm_le(ahead * npw);           // this step's rows have landed (transfers retire in order)
        gg_barrier_lds();                     // ... in every wave; and every wave is done with the rows the next group overwrites
        if (s + D < s1) issue_group(s + D);
        const int R0 = s * RS;
        // fragments of one 16-pixel k-step: the dy fragment and one x fragment per tap (two transpose reads each). Two sets alternate:
        // the reads of the next k-step are in flight while the MFMAs of the current one run (one wave per SIMD: nothing else hides
        // the LDS latency; left to itself the compiler issued each tap's two reads right in front of its MFMA: ~110 cycles per tap)
        auto load = [&](int kk, u16x8& fb, u16x8 (&fa)[TAPS]) {
            const int px = kk << 4, ry = px >> ws, cx = px & (W - 1);
            const int y = (R0 + ry) & (H - 1);
            int yr = ytail + ry;
            if (yr >= NRy) yr -= NRy;
            const int yo = yring + yr * YP + cx * g.sby + ylane;
            const u16x4 b0 = gg_lds_read_tr16((const bf16_t*)(smem + yo));
            const u16x4 b1 = gg_lds_read_tr16((const bf16_t*)(smem + yo + 4 * g.sby));
            fb = u16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                int xr = xtail + ry * RPR + kh;
                if (xr >= NRx) xr -= NRx;
                const bool in = (unsigned)(y + kh - HALO) < (unsigned)H;
                const int xo = ((!HALO || in) ? xring + xr * XP : zrow) + cx * g.sbx + xlane;
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const u16x4 a0 = gg_lds_read_tr16((const bf16_t*)(smem + xo + kw * g.sbx));
                    const u16x4 a1 = gg_lds_read_tr16((const bf16_t*)(smem + xo + (kw + 4) * g.sbx));
                    fa[kh * KW + kw] = u16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                }
            }
        };
        auto mul = [&](const u16x8& fb, const u16x8 (&fa)[TAPS]) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = gg_mfma_32x32x16_bf16(fb, fa[t], acc[t]);   // lane: ci = lane & 31, registers run along co
        };
        u16x8 fb0, fb1, fa0[TAPS], fa1[TAPS];
        load(kpart, fb0, fa0);
        for (int kk = kpart; kk < KSTEPS; kk += 2 * kparts) {      // (KSTEPS / kparts is even: 16 or 8 k-steps, 1 / 2 / 4 shares)
            load(kk + kparts, fb1, fa1);
            mul(fb0, fa0);
            if (kk + 2 * kparts < KSTEPS) load(kk + 2 * kparts, fb0, fa0);
            mul(fb1, fa1);
        }
        xtail += RSX;
        if (xtail >= NRx) xtail -= NRx;
        ytail += RS;
        if (ytail >= NRy) ytail -= NRy;
    }

    // waves that share a block: summed through LDS (fixed order: deterministic)
    gg_wait_vm_le(0);
    gg_barrier_lds();
    if (kparts > 1) {
        float* red = (float*)smem;
        if (kpart > 0) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave - nblk) * TAPS * 16 + t * 16 + r) * 64 + lane] = acc[t][r];
        }
        gg_barrier_lds();
        if (kpart == 0) {
            for (int kp = 1; kp < kparts; ++kp)
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += red[((blk + nblk * kp - nblk) * TAPS * 16 + t * 16 + r) * 64 + lane];
        }
    }
    if (kpart != 0) return;

    // lane owns row ci = cb*32 + (lane & 31) of every tap block; register r holds column nb*32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int ci = cb * 32 + (lane & 31);
    if (ci >= p.ws_cstore) return;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const long long m = (long long)t * p.ws_cstore + ci;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = nb * 32 + 4 * (lane >> 5) + 8 * q;
            if (n < N) {                                           // N % 8 == 0 (host)
                if (p.splitk > 1) {
                    const f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                    *(f32x4*)(p.partial + ((long long)blockIdx.x * p.M + m) * N + n) = v;
                } else {
                    const f32x4 v = {acc[t][4 * q] * p.alpha, acc[t][4 * q + 1] * p.alpha, acc[t][4 * q + 2] * p.alpha,
                                     acc[t][4 * q + 3] * p.alpha};
                    *(f32x4*)((float*)p.Cout + m * p.ldc + n) = v;
                }
            }
        }
    }
}
