// how fast does ONE wavefront per SIMD issue v_mfma_f32_32x32x16_bf16 back to back (16 independent accumulators), against two
// wavefronts per SIMD with 8 accumulators each - and with LDS fragment reads interleaved?  (round 6; test infrastructure)
//   hipcc --offload-arch=gfx950 -O3 tests/probes/mfma_issue_probe.hip -o tests/probes/_bin/mfma_issue_probe && tests/probes/_bin/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int NACC, bool LDS>
__global__ __launch_bounds__(NACC == 16 ? 256 : 512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) ((u4*)smem)[i] = u4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f16v acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u4 fa[2][4], fb[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[s][i] = ((u4*)smem)[lane + 64 * i]; fb[s][i] = ((u4*)smem)[lane + 64 * (i + 4)]; }
    const char* base = smem + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int nx = (kk + 1) & 1, cu = kk & 1;
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[nx][i] = *(const u4*)(base + ((kk * 8 + i) * 1024 & 65535));
#pragma unroll
                for (int i = 0; i < (NACC == 16 ? 4 : 2); ++i) fb[nx][i] = *(const u4*)(base + ((kk * 8 + 4 + i) * 1024 & 65535));
            }
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fb[cu][a & (NACC == 16 ? 3 : 1)]),
                                                                 __builtin_bit_cast(bf8, fa[cu][a >> (NACC == 16 ? 2 : 1)]), acc[a], 0, 0, 0);
            if (LDS) {
#pragma unroll
                for (int r = 0; r < (NACC == 16 ? 8 : 6); ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
static void run(const char* name, int threads, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, LDS>), dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 4 * NACC * (threads / 64);          // MFMAs per workgroup
    const double per_simd = mf / 4;
    printf("%-60s %8.3f ms  %6.1f ns per MFMA and SIMD  (%.0f TFLOP/s)\n", name, ms, ms * 1e6 / per_simd, mf * 256 * 32768 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<16, false>("1 wave/SIMD, 16 accumulators, no LDS reads", 256, out);
    run<8, false>("2 waves/SIMD, 8 accumulators each, no LDS reads", 512, out);
    run<16, true>("1 wave/SIMD, 16 accumulators, 8 ds_read_b128 per 16 MFMAs", 256, out);
    run<8, true>("2 waves/SIMD, 8 accumulators, 6 ds_read_b128 per 8 MFMAs", 512, out);
    return 0;
}
