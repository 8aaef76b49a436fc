// Under the power cap, does v_mfma_f32_16x16x32_bf16 sustain more than v_mfma_f32_32x32x16_bf16 on the same operands? (The two move
// the same operand bits per flop, but the 16x16x32 form reads / writes a quarter of the accumulator values per flop.) Two wavefronts per
// SIMD, operands constant in registers (N(0,1) bf16, or zeros), 256 workgroups, no memory traffic in the loop.   (round 6)
//   hipcc --offload-arch=gfx950 -O3 tests/probes/mfma_shape_power_probe.hip -o tests/probes/_bin/mfma_shape_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>      // 0: 32x32x16 with 8 accumulator tiles; 1: 16x16x32 with 8 accumulator tiles
__global__ __launch_bounds__(512) void probe(const unsigned short* src, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    u16x8 fa[4], fb[2];
    for (int i = 0; i < 4; ++i) fa[i] = *(const u16x8*)(src + (i * 64 + lane) * 8);
    for (int j = 0; j < 2; ++j) fb[j] = *(const u16x8*)(src + ((4 + j) * 64 + lane) * 8);
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fb[j]), __builtin_bit_cast(bf8, fa[i]), acc[i][j], 0, 0, 0);
            asm volatile("" : "+v"(fa[0]), "+v"(fb[0]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)       // same flops per iteration: 64 x 16384 MACs
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fb[j]), __builtin_bit_cast(bf8, fa[i]), acc[i][j], 0, 0, 0);
            asm volatile("" : "+v"(fa[0]), "+v"(fb[0]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    if (s == 12345.678f) sink[0] = s;
}


// the same comparison inside gg_conv3's loop shape: a 128 x 64 wave tile, the fragments re-read from LDS every k-step (ds_read_b128, the
// kernel's 144-byte row pitch), eight wavefronts, a barrier per 64-deep k-tile
template <int SHAPE>
__global__ __launch_bounds__(512) void loop_probe(const unsigned short* src, float* sink, int ktiles) {
    constexpr int P = SHAPE == 0 ? 72 : 80;       // bf16 row pitch: 144 B is conflict-free for 32-row fragments, 160 B for 16-row ones
    __shared__ __attribute__((aligned(16))) unsigned short tile[(256 + 256) * P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (256 + 256) * P; i += 512) tile[i] = src[i % (6 * 64 * 8)];
    __syncthreads();
    const int wm = wave / 4, wn = wave % 4;
    float s = 0.f;
    if (SHAPE == 0) {
        const int frow = lane & 31, fk = (lane >> 5) * 8;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                u16x8 fa[4], fb[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *(const u16x8*)&tile[(wm * 128 + i * 32 + frow) * P + kk * 16 + fk];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *(const u16x8*)&tile[(256 + wn * 64 + j * 32 + frow) * P + kk * 16 + fk];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fb[j]), __builtin_bit_cast(bf8, fa[i]), acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        const int frow = lane & 15, fk = (lane >> 4) * 8;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u16x8 fa[8], fb[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) fa[i] = *(const u16x8*)&tile[(wm * 128 + i * 16 + frow) * P + kk * 32 + fk];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *(const u16x8*)&tile[(256 + wn * 64 + j * 16 + frow) * P + kk * 32 + fk];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fb[j]), __builtin_bit_cast(bf8, fa[i]), acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    if (s == 12345.678f) sink[0] = s;
}

template <int SHAPE>
static void run_loop(const char* name, const unsigned short* src, float* sink, const char* data) {
    const int ktiles = 4096, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) loop_probe<SHAPE><<<blocks, 512>>>(src, sink, ktiles);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) loop_probe<SHAPE><<<blocks, 512>>>(src, sink, ktiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * ktiles * (2.0 * 256 * 256 * 64);
    printf("%-28s %-8s %8.1f TFLOP/s   (fragments from LDS every k-step, barrier per k-tile)\n", name, data, flops / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

template <int SHAPE>
static void run(const char* name, const unsigned short* src, float* sink, const char* data) {
    const int iters = 4096, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) probe<SHAPE><<<blocks, 512>>>(src, sink, iters);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) probe<SHAPE><<<blocks, 512>>>(src, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 8 * iters * 32 * (2.0 * 16384);      // (32 MFMAs of 32x32x16 = 64 of 16x16x32 per iteration)
    printf("%-28s %-8s %8.1f TFLOP/s\n", name, data, flops / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const int n = 6 * 64 * 8;
    std::vector<unsigned short> h(n);
    unsigned short* d; float* sink;
    hipMalloc(&d, n * 2); hipMalloc(&sink, 4);
    for (int pass = 0; pass < 3; ++pass) {
        srand(1);
        for (int i = 0; i < n; ++i) {
            float f = 0.f;
            if (pass == 1) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f); f = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
            if (pass == 2) f = 1.f;
            unsigned int u; memcpy(&u, &f, 4);
            h[i] = (unsigned short)(u >> 16);
        }
        hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
        const char* data = pass == 0 ? "zeros" : (pass == 1 ? "N(0,1)" : "ones");
        run<0>("v_mfma_f32_32x32x16_bf16", d, sink, data);
        run<1>("v_mfma_f32_16x16x32_bf16", d, sink, data);
        run_loop<0>("v_mfma_f32_32x32x16_bf16", d, sink, data);
        run_loop<1>("v_mfma_f32_16x16x32_bf16", d, sink, data);
    }
    return 0;
}
