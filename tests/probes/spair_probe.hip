// spair_probe.hip — where an iteration of gg_spair_kernel spends its time: s_memtime stamps of compute wave 0 and of the loader wave
// of one workgroup at the phase boundaries of every row, and the kernel's wall time with pieces switched off (GG_SP_PROBE hooks in
// csrc/gg_spair.h: not in the product build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DGG_SP_PROBE -Igigagan_pytorch_amd/csrc \
//       tests/probes/spair_probe.hip -o /tmp/spair_probe && /tmp/spair_probe
#include "gg_device.h"
#include "gg_spair.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

template <int C0, int C1, int PT, int NCW, bool M16 = false>
static void run(int C2, int b, int H, int rows) {
    typedef GgSpGeom<C0, C1, PT, NCW, M16> G;
    const int W = G::W;
    const size_t nx = (size_t)b * H * W * C0, ny = (size_t)b * H * W * C2, nw1 = (size_t)b * 9 * (C0 / 16) * 512, nw2 = (size_t)b * 9 * (C1 / 16) * 512;
    std::vector<unsigned short> hx(nx), hw1(nw1), hw2(nw2);
    srand(1);
    for (auto& v : hx) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    for (auto& v : hw1) v = (unsigned short)(0x3a00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
    for (auto& v : hw2) v = (unsigned short)(0x3a00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
    std::vector<float> hn((size_t)b * H * W, 0.25f), hnw(32, 0.1f), hxs((size_t)b * C0, 1.25f);
    unsigned short *x, *w1, *w2, *y; float *n1, *n2, *nw, *xs; long long* st;
    hipMalloc(&x, nx * 2); hipMalloc(&w1, nw1 * 2); hipMalloc(&w2, nw2 * 2); hipMalloc(&y, ny * 2);
    hipMalloc(&n1, hn.size() * 4); hipMalloc(&n2, hn.size() * 4); hipMalloc(&nw, 128); hipMalloc(&xs, hxs.size() * 4);
    hipMalloc(&st, 128 * 8 * 8);
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w1, hw1.data(), nw1 * 2, hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), nw2 * 2, hipMemcpyHostToDevice); hipMemcpy(n1, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(n2, hn.data(), hn.size() * 4, hipMemcpyHostToDevice); hipMemcpy(nw, hnw.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(xs, hxs.data(), hxs.size() * 4, hipMemcpyHostToDevice);
    GgSpairParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w1 = w1; p.w2 = w2; p.y = y; p.w1_bs = 9 * (C0 / 16) * 512; p.w2_bs = 9 * (C1 / 16) * 512;
    p.noise1 = n1; p.nw1 = nw; p.noise2 = n2; p.nw2 = nw; p.xs = xs; p.b = b; p.H = H; p.C2 = C2; p.act1 = p.act2 = 1; p.slope = 0.2f;
    p.rows = rows; p.strips = (H + rows - 1) / rows; p.stamps = st;
    const int grid = b * p.strips, lds = G::bytes(C2);
    hipFuncSetAttribute((const void*)gg_spair_kernel<C0, C1, PT, NCW, M16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double mb = (nx + ny) * 2 / 1e6;
    printf("== %d compute waves x %d blocks  C0 %d C1 %d C2 %d  b %d  %dx%d  rows/wg %d  grid %d  lds %d  (%.0f MB algorithmic)\n", NCW, PT, C0, C1, C2, b, H, W, rows, grid, lds, mb);
    const int offs[] = {0, 1, 2, 3, 4, 8, 16, 1 | 2 | 16, 1 | 2 | 4 | 16, 1 | 2 | 4 | 8 | 16};
    for (int off : offs) {
        p.probe_off = off; p.probe_wg = -1;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gg_spair_kernel<C0, C1, PT, NCW, M16>), dim3(grid), dim3(G::NT), lds, 0, p);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((gg_spair_kernel<C0, C1, PT, NCW, M16>), dim3(grid), dim3(G::NT), lds, 0, p);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  off %2d (%s%s%s%s%s): %7.1f us  %.2f TB/s\n", off, off & 1 ? "-conv1 " : "", off & 2 ? "-conv2 " : "", off & 4 ? "-stores " : "",
               off & 8 ? "-dma " : "", off & 16 ? "-midwrite " : "", ms * 100, mb / (ms * 100));
    }
    // stamps of one workgroup in the middle of the grid, full kernel
    p.probe_off = 0; p.probe_wg = grid / 2 + 1;
    hipMemset(st, 0, 128 * 8 * 8);
    hipLaunchKernelGGL((gg_spair_kernel<C0, C1, PT, NCW, M16>), dim3(grid), dim3(G::NT), lds, 0, p);
    hipDeviceSynchronize();
    std::vector<long long> hs(128 * 8);
    hipMemcpy(hs.data(), st, 128 * 8 * 8, hipMemcpyDeviceToHost);
    const int iters = rows + 3;
    printf("  compute wave 0 (cycles): it | barrier wait | mfma (both) | conv1 epi | conv2 epi | stores | total ; loader: wait | barrier | issue\n");
    for (int it = 0; it < iters && it < 64; ++it) {
        const long long* c = &hs[it * 8];
        const long long* l = &hs[(64 + it) * 8];
        printf("    %2d | %6lld | %6lld | %6lld | %6lld | %6lld | %6lld ; %6lld | %6lld | %6lld\n", it, c[1] - c[0], c[2] - c[1], c[3] - c[2],
               c[5] ? c[5] - c[3] : 0, c[5] ? c[6] - c[5] : 0, c[6] - c[0], l[1] - l[0], l[2] - l[1], l[3] - l[2]);
    }
    hipFree(x); hipFree(w1); hipFree(w2); hipFree(y); hipFree(n1); hipFree(n2); hipFree(nw); hipFree(xs); hipFree(st);
}

int main() {
    run<32, 16, 1, 8>(16, 32, 256, 32);
    printf("(next: the 16x16x32 form)\n");
    run<32, 16, 1, 8, true>(16, 32, 256, 32);
    run<64, 32, 1, 4>(32, 32, 128, 16);
    return 0;
}
