// gg_device_emu.h — host-side stand-in for gg_device.h's device vocabulary (TEST INFRASTRUCTURE ONLY).
//
// Compiles the unmodified kernel sources in gigagan_pytorch_amd/csrc for the CPU: every HIP thread of a
// workgroup is a ucontext fiber, `gg_sync()` is a fiber barrier, wave-collective operations (MFMA,
// shuffles) rendezvous the 64 fibers of a wave and then apply the documented gfx950 lane mappings
// (guide: cdna_hip_programming.md §3). This lets `pytest -m "not gpu"` check the kernels' index math,
// LDS layouts, masking and epilogues through the same C ABI the GPU build exports. It is never loaded by
// the product path (gigagan_pytorch_amd/_C.py loads only the hipcc-built library).
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <functional>

#define GG_DEVICE static inline
#define GG_HOST_DEVICE static inline
#define GG_KERNEL static
#define GG_SHARED static
#define GG_LAUNCH_BOUNDS(n)
#define GG_LAUNCH_BOUNDS2(n, w)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct gg_emu_dim3 {
    unsigned x, y, z;
    gg_emu_dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef gg_emu_dim3 dim3;
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

extern gg_emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

void gg_emu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
void gg_emu_syncthreads();
f32x16 gg_emu_mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c);
f32x4 gg_emu_mfma_16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c);
float gg_emu_shfl(float v, int src_lane);
u16x4 gg_emu_lds_read_tr16(const bf16_t* p);

#define GG_LAUNCH(kernel, grid, block, stream, ...) \
    gg_emu_launch(grid, block, [=]() { kernel(__VA_ARGS__); })
// launch-sized LDS (gg_aconv.h): the emulator runs one workgroup at a time, so one static 160 KB array per kernel stands in
#define GG_DYN_SHARED(name) static __attribute__((aligned(1024))) char name[160 * 1024]
#define GG_LAUNCH_DYN(kernel, grid, block, lds_bytes, stream, ...) \
    gg_emu_launch(grid, block, [=]() { kernel(__VA_ARGS__); })

// LDS-DMA stand-ins. Two landing models bracket what the hardware may do (the emulator itself is sequential):
//   early (default)      the 16 bytes land at issue time: a slot that is refilled before its last reader ran (WAR) shows up
//                        whenever the refilling wave runs before the reading wave inside a barrier interval (fibers run in
//                        thread order; GG_EMU_REVERSE=1 runs them in the opposite order to catch the other direction);
//   late (GG_EMU_DMA=late)  a transfer lands only when its issuing thread executes the counted wait that retires it
//                        (gg_wait_vm<N>: all but the N newest), a __syncthreads (which drains vmcnt on the hardware) or the
//                        kernel end: a missing or mis-counted wait before a read (RAW) then reads stale LDS bytes.
void gg_emu_dma_issue(const void* g, void* dst);
void gg_emu_dma_wait(int keep_newest);
static inline void gg_sync() { gg_emu_dma_wait(0); gg_emu_syncthreads(); }
template <int N>
static inline void gg_wait_vm() { gg_emu_dma_wait(N); }
static inline void gg_wait_vm_le(int n) { gg_emu_dma_wait(n); }
static inline void gg_barrier_lds() { gg_emu_syncthreads(); }       // the raw barrier: transfers stay in flight (late model: not landed)
// buffer addressing stand-in: the hardware range-checks the per-lane offset (not the scalar one) and returns zeros beyond `bytes`
struct GgBuf { const char* base; unsigned long long bytes; };
static inline GgBuf gg_make_buf(const void* base, unsigned long long bytes) {
    GgBuf r = {(const char*)base, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes};
    return r;
}
static inline u16x8 gg_buf_load16(GgBuf r, unsigned voff, unsigned soff) {
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((unsigned long long)voff + 16 <= r.bytes) memcpy(&v, r.base + voff + soff, 16);
    return v;
}
// (a cache hint on the hardware; here: the touched bytes must lie inside the buffer, so that a wrong range shows up in the CPU suite)
extern unsigned long long gg_emu_touched_bytes;
static inline void gg_buf_touch16(GgBuf r, unsigned voff) {
    if ((unsigned long long)voff + 16 > r.bytes) __builtin_trap();
    volatile char c = r.base[voff + 15];
    (void)c;
    gg_emu_touched_bytes += 16;
}
static inline void gg_buf_load_lds16(GgBuf r, unsigned voff, unsigned soff, void* lds_wave_base) {
    static const char zeros[16] = {0};
    const void* src = ((unsigned long long)voff + 16 <= r.bytes) ? (const void*)(r.base + voff + soff) : (const void*)zeros;
    gg_emu_dma_issue(src, (char*)lds_wave_base + 16 * (threadIdx.x & 63u));
}
typedef GgBuf GgBufS;
static inline GgBufS gg_make_bufs(const void* base, unsigned long long bytes) { return gg_make_buf(base, bytes); }
static inline void gg_bufs_load_lds16(GgBufS r, unsigned voff, unsigned soff, void* lds_wave_base) { gg_buf_load_lds16(r, voff, soff, lds_wave_base); }
static inline void gg_settle(u16x8&) {}
static inline void gg_store_nt16(void* p, u16x8 v) { *(u16x8*)p = v; }
static inline int gg_uniform(int v) { return v; }
static inline void gg_wave_sync() { (void)gg_emu_shfl(0.f, (int)(threadIdx.x & 63u)); }      // a wave collective: every lane arrives before any leaves
template <typename T>
static inline const T* gg_late_params(const T& by_value) { return &by_value; }
static inline f32x16 gg_mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
    return gg_emu_mfma_32x32x16_bf16(a, b, c);
}
static inline f32x4 gg_mfma_16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c) { return gg_emu_mfma_16x16x32_bf16(a, b, c); }
static inline u16x4 gg_lds_read_tr16(const bf16_t* p) { return gg_emu_lds_read_tr16(p); }
static inline float gg_shfl_xor(float v, int mask) {
    int lane = (int)(threadIdx.x & 63u);
    return gg_emu_shfl(v, lane ^ mask);
}
static inline float gg_shfl(float v, int src) { return gg_emu_shfl(v, src & 63); }
static inline float gg_readlane(float v, int src) { return gg_emu_shfl(v, src & 63); }
static inline float gg_row16_sum(float v) {        // wave collective; the device version's association: quads, 8, 16
    v += gg_shfl_xor(v, 1); v += gg_shfl_xor(v, 2);
    { int lane = (int)(threadIdx.x & 63u); v += gg_emu_shfl(v, (lane & ~7) | (7 - (lane & 7))); }
    { int lane = (int)(threadIdx.x & 63u); v += gg_emu_shfl(v, (lane & ~15) | (15 - (lane & 15))); }
    return v;
}
static inline float gg_wave_sum_all(float v) {     // wave collective; the device version's association: quads, 8, 16, rows
    v += gg_shfl_xor(v, 1); v += gg_shfl_xor(v, 2);
    { int lane = (int)(threadIdx.x & 63u); v += gg_emu_shfl(v, (lane & ~7) | (7 - (lane & 7))); }
    { int lane = (int)(threadIdx.x & 63u); v += gg_emu_shfl(v, (lane & ~15) | (15 - (lane & 15))); }
    return (gg_emu_shfl(v, 0) + gg_emu_shfl(v, 16)) + (gg_emu_shfl(v, 32) + gg_emu_shfl(v, 48));
}
static inline void gg_atomic_add(float* p, float v) { *p += v; }
static inline unsigned gg_ticket_take(unsigned* p) { return (*p)++; }      // workgroups run one after another on the host
static inline void gg_ticket_reset(unsigned* p) { *p = 0; }

static inline float gg_expf(float x) { return expf(x); }
static inline float gg_exp2f(float x) { return exp2f(x); }
static inline bool gg_wave_any(bool pred) {     // wave collective: every fiber of the wave calls it
    float f = pred ? 1.f : 0.f;
    for (int o = 1; o < 64; o <<= 1) f = fmaxf(f, gg_shfl_xor(f, o));
    return f > 0.f;
}
static inline float gg_rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float gg_rcpf(float x) { return 1.0f / x; }
