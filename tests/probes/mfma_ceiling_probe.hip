// mfma_ceiling_probe.hip — what bounds "MFMAs + barrier only" in gg_gemm2's k-loop (the phase probe read 1400 TF,
// the guide's back-to-back v_mfma_f32_32x32x16_bf16 figure is 2495 TF). Same accumulator order as gg_gemm2_kernel (a 4 x 2
// grid of 32x32 tiles per wave, 32 MFMAs per 64-deep k-tile, fragments reused across the grid) with the pieces switched on
// one at a time:
//   waves per workgroup 4 (one per SIMD) or 8 (two per SIMD)   | barrier per k-tile: none / __syncthreads
//   operands: constants in registers / re-read from LDS every k-tile (24 ds_read_b128, the kernel's fragment traffic)
//   data: zeros / random bf16 (the matrix pipe's power draw, hence the clock, depends on the operand bits: DVFS)
//   hipcc --offload-arch=gfx950 -O3 tests/probes/mfma_ceiling_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WAVES, bool BARRIER, bool LDS_READS>
__global__ __launch_bounds__(WAVES * 64) void probe(const unsigned short* src, float* sink, int ktiles) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[(256 + 256) * 72];     // gg_gemm2's 144-byte rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (256 + 256) * 72; i += WAVES * 64) tile[i] = src[i];
    __syncthreads();
    const int wm = (wave & 7) / 4, wn = (wave & 7) % 4, frow = lane & 31, fk = (lane >> 5) * 8;
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u16x8 fa[4], fb[2];
    for (int i = 0; i < 4; ++i) fa[i] = *(const u16x8*)&tile[(wm * 128 + i * 32 + frow) * 72 + fk];
    for (int j = 0; j < 2; ++j) fb[j] = *(const u16x8*)&tile[(256 + wn * 64 + j * 32 + frow) * 72 + fk];
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (LDS_READS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *(const u16x8*)&tile[(wm * 128 + i * 32 + frow) * 72 + kk * 16 + fk];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *(const u16x8*)&tile[(256 + wn * 64 + j * 32 + frow) * 72 + kk * 16 + fk];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fb[j]), __builtin_bit_cast(bf8, fa[i]),
                                                                        acc[i][j], 0, 0, 0);
        }
        if (BARRIER) __syncthreads();
        if (!LDS_READS) asm volatile("" : "+v"(fa[0]), "+v"(fb[0]));    // keep the loop from being folded
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[0] = s;
}

template <int WAVES, bool BARRIER, bool LDS_READS>
static void run(const char* name, const unsigned short* src, float* sink, const char* data) {
    const int ktiles = 2048, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) probe<WAVES, BARRIER, LDS_READS><<<blocks, WAVES * 64>>>(src, sink, ktiles);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) probe<WAVES, BARRIER, LDS_READS><<<blocks, WAVES * 64>>>(src, sink, ktiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * WAVES * ktiles * 32 * (2.0 * 32 * 32 * 16);
    const double tf = flops / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz nominal: waves per SIMD = WAVES / 4
    const double cyc = (ms * 1e-3 / reps) * 2.4e9 / ((double)ktiles * 32 * (WAVES / 4.0));
    printf("%-58s %-7s %8.1f TF  %6.2f nominal-clock cycles per MFMA per SIMD\n", name, data, tf, cyc);
    fflush(stdout);
}

int main() {
    const int n = (256 + 256) * 72;
    std::vector<unsigned short> h(n);
    unsigned short* d; float* sink;
    hipMalloc(&d, n * 2); hipMalloc(&sink, 4);
    for (int pass = 0; pass < 2; ++pass) {
        const char* data = pass ? "random" : "zeros";
        srand(1);
        for (int i = 0; i < n; ++i) {           // random: N(0,1)-like bf16 values of both signs
            float f = pass ? ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 1.5f : 0.f;
            unsigned int u; memcpy(&u, &f, 4);
            h[i] = (unsigned short)(u >> 16);
        }
        hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
        run<4, false, false>("4 waves (1/SIMD), no barrier, register operands", d, sink, data);
        run<8, false, false>("8 waves (2/SIMD), no barrier, register operands", d, sink, data);
        run<8, true, false>("8 waves, barrier per k-tile, register operands", d, sink, data);
        run<4, false, true>("4 waves, no barrier, 24 ds_read_b128 per k-tile", d, sink, data);
        run<8, false, true>("8 waves, no barrier, 24 ds_read_b128 per k-tile", d, sink, data);
        run<8, true, true>("8 waves, barrier + LDS reads (= gg_gemm2 minus staging)", d, sink, data);
        run<4, true, true>("4 waves, barrier + LDS reads", d, sink, data);
    }
    return 0;
}
