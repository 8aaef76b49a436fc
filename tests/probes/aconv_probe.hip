// aconv_probe.hip — where a gg_aconv_kernel launch spends its time on the low-resolution layers: s_memtime stamps of every workgroup's
// first and last wavefront at the phase boundaries (start | halo + first fragments loaded, LDS written | barrier | reduction loop |
// K-slice sum | stores), on a bank that is HOT (same bank again) or COLD (a walk over 48 banks: nothing of it in L2 / MALL).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DGG_AC_PROBE -Igigagan_pytorch_amd/csrc \
//       tests/probes/aconv_probe.hip -o tests/probes/_bin/aconv_probe
#include "gg_device.h"
#include "gg_aconv.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

// reads `bytes` at p (16 bytes per lane and step) and keeps nothing: what is left behind is the data in the memory-side cache (and, for the
// last few MB, in the L2s)
__global__ void touch_kernel(const char* p, size_t bytes, int* sink) {
    int acc = 0;
    for (size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o + 16 <= bytes; o += (size_t)gridDim.x * blockDim.x * 16) {
        const int4 v = *(const int4*)(p + o);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) *sink = acc;
}

static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

template <int NB, int TM, int NWN, int NWK>
static void run(const char* name, int b, int R, int C, int O) {
    const int NBANKS = 48;
    const size_t wbytes = (size_t)O * NB * 9 * C * 2, xb = (size_t)b * R * R * C * 2, yb = (size_t)b * R * R * O * 2;
    char* wf; unsigned short *x, *y; float *s, *a, *d, *nz, *nw; long long* st;
    hipMalloc(&wf, wbytes * NBANKS); hipMalloc(&x, xb); hipMalloc(&y, yb);
    hipMalloc(&s, b * C * 4); hipMalloc(&a, b * NB * 4); hipMalloc(&d, b * O * 4); hipMalloc(&nz, (size_t)b * R * R * 4); hipMalloc(&nw, O * 4);
    hipMemset(wf, 0x3c, wbytes * NBANKS); hipMemset(x, 0x3c, xb); hipMemset(s, 0, b * C * 4); hipMemset(a, 0, b * NB * 4); hipMemset(d, 0, b * O * 4);
    hipMemset(nz, 0, (size_t)b * R * R * 4); hipMemset(nw, 0, O * 4);
    const int bmt = 32 * TM, BN = 32 * NWN, hw = R * R;
    const int mt = (b * hw + bmt - 1) / bmt, grid = mt * (O / BN);
    const int rt = hw >= bmt ? bmt / R : R, ti = hw >= bmt ? 1 : bmt / hw;
    const int lds_halo = ti * (rt + 2) * (R + 2) * (C * 2 + 16);
    const int lds_red = NWK * NWN * TM * 16 * 64 * 4;
    const int lds = std::max(lds_halo, lds_red);
    hipMalloc(&st, (size_t)grid * 2 * 8 * 8);
    GgAconvParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.y = y; p.s = s; p.a = a; p.d = d; p.noise = nz; p.noise_w = nw;
    p.b = b; p.H = R; p.W = R; p.C = C; p.O = O; p.w_shift = ilog2(R); p.hw_shift = ilog2(hw); p.c8_shift = ilog2(C / 8);
    p.act = 1; p.slope = 0.2f; p.mt = mt;
    p.inv_spi = 65536 / ((rt + 2) * (R + 2)) + 1; p.inv_hwp = 65536 / (R + 2) + 1;
    p.x_bytes = xb; p.wf_bytes = wbytes; p.stamps = st;
    hipFuncSetAttribute((const void*)gg_aconv_kernel<NB, TM, NWN, NWK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("== %s: %d->%d @%dx%d b %d  tile %d px x %d ch, %d K-slices  grid %d  lds %d  bank %.1f MB\n", name, C, O, R, R, b, bmt, BN, NWK, grid, lds, wbytes / 1e6);
    int* sink; hipMalloc(&sink, 4);
    hipMemset(st, 0, (size_t)grid * 2 * 8 * 8);
    for (int cold = 0; cold < 3; ++cold) {
        // chains of 12 launches (one after the other on the stream, as the layers of a forward are): hot = one bank, cold = a walk over the banks
        for (int rep = 0; rep < 2; ++rep) {
            // mode 2: the chain's 12 banks (12 x the bank size: beyond the L2s, inside the 256 MB memory-side cache) are read once up front
            if (cold == 2) hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, 0, wf + (size_t)((rep * 12) % NBANKS) * wbytes, wbytes * 12, sink);
            hipEventRecord(e0);
            for (int i = 0; i < 12; ++i) {
                p.wf = (const bf16_t*)(wf + (cold ? (size_t)((rep * 12 + i) % NBANKS) * wbytes : 0));
                hipLaunchKernelGGL((gg_aconv_kernel<NB, TM, NWN, NWK>), dim3(grid), dim3(64 * NWN * NWK), lds, 0, p);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> hs((size_t)grid * 16);
        hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        // per phase: median and max over the workgroups (first wavefront), and the span of the launch on the s_memtime clock
        long long t_first = hs[0], t_last = 0;
        for (int w = 0; w < grid; ++w) { t_first = std::min(t_first, hs[(size_t)w * 16]); for (int k = 0; k < 2; ++k) t_last = std::max(t_last, std::max(hs[((size_t)w * 2 + k) * 8 + 5], hs[((size_t)w * 2 + k) * 8 + 4])); }
        printf("  %s: %.1f us per launch in a chain of 12 (events); last launch spans %lld ticks first start -> last end\n", cold == 2 ? "MALL" : (cold ? "COLD" : "HOT "), ms * 1000 / 12, t_last - t_first);
        const char* names[] = {"start skew", "loads->LDS", "barrier", "loop", "mix+K-sum", "finish"};
        for (int k = 0; k < 2; ++k) {
            printf("    wavefront %s:", k ? "last " : "first");
            for (int ph = 0; ph < 6; ++ph) {
                std::vector<long long> v;
                for (int w = 0; w < grid; ++w) {
                    const long long* c = &hs[((size_t)w * 2 + k) * 8];
                    long long dt = ph == 0 ? 0 : (ph == 5 ? (c[5] > c[4] ? c[5] - c[4] : 0) : c[ph] - c[ph - 1]);
                    v.push_back(dt);
                }
                std::sort(v.begin(), v.end());
                printf("  %s med %lld max %lld |", names[ph], v[v.size() / 2], v.back());
            }
            printf("\n");
        }
    }
    hipFree(wf); hipFree(x); hipFree(y); hipFree(s); hipFree(a); hipFree(d); hipFree(nz); hipFree(nw); hipFree(st);
}

int main() {
    run<2, 1, 1, 8>("4x4", 32, 4, 512, 512);
    run<2, 2, 2, 4>("8x8", 32, 8, 512, 512);
    run<2, 2, 4, 2>("16x16 a", 32, 16, 512, 256);
    run<2, 4, 2, 4>("16x16 b", 32, 16, 256, 256);
    run<2, 4, 4, 2>("32x32 a", 32, 32, 256, 128);
    return 0;
}
