// gg_device.h — the device-side vocabulary every kernel in csrc/ is written against (gfx950 / CDNA4).
//
// Kernels use: 64-lane wavefronts, `gg_mfma_32x32x16_bf16` (v_mfma_f32_32x32x16_bf16), 16-byte global
// loads, LDS tiles, wave shuffles. bf16 is stored as raw `unsigned short` so that vector loads are plain
// integer vectors and rounding is explicit (round-to-nearest-even, as torch's bf16 cast).
//
// The only conditional in this file selects the host-side kernel emulator used by tests/emu (fibers on
// the CPU, same kernel sources, same C ABI) so that index math can be verified without a GPU. The
// emulator is test infrastructure; the product library is always built with hipcc for gfx950.
#pragma once
#include <stdint.h>

#if defined(GG_HOST_EMULATION)
#include "gg_device_emu.h"
#else

#include <hip/hip_runtime.h>

#define GG_DEVICE __device__ __forceinline__
#define GG_HOST_DEVICE __host__ __device__ __forceinline__
#define GG_KERNEL __global__
#define GG_SHARED __shared__
#define GG_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define GG_LAUNCH_BOUNDS2(n, waves_per_simd) __launch_bounds__(n, waves_per_simd)   // caps the register allocation for that occupancy
#define GG_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
// kernels whose LDS need depends on the launch geometry (gg_aconv.h): one `extern __shared__` array, sized at launch. More than 64 KB
// of it must be allowed per kernel once (hipFuncAttributeMaxDynamicSharedMemorySize: a host-side attribute, no stream work)
#define GG_DYN_SHARED(name) extern __shared__ __attribute__((aligned(1024))) char name[]
// (the attribute is per device: one flag per device ordinal; a refused attribute is left unset so that the launch below reports it)
#define GG_LAUNCH_DYN(kernel, grid, block, lds_bytes, stream, ...)                                                      \
    do {                                                                                                                \
        static bool gg_lds_allowed_[64] = {false};                                                                      \
        int gg_dev_ = 0;                                                                                                \
        if (hipGetDevice(&gg_dev_) != hipSuccess || gg_dev_ < 0 || gg_dev_ >= 64) gg_dev_ = 0;                          \
        if (!gg_lds_allowed_[gg_dev_]) {                                                                                \
            if (hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess) \
                gg_lds_allowed_[gg_dev_] = true;                                                                        \
            else (void)hipGetLastError();                                                                               \
        }                                                                                                               \
        hipLaunchKernelGGL(kernel, grid, block, (size_t)(lds_bytes), stream, __VA_ARGS__);                              \
    } while (0)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 gg_bf16x8_native;

GG_DEVICE void gg_sync() { __syncthreads(); }

// The kernel-argument struct re-read through the kernarg segment pointer behind a compiler barrier: fields that only
// the epilogue needs are then fetched AFTER the main loop instead of sitting in SGPRs (and, once those run out, in
// scratch) for the whole kernel. `by_value` is the kernel's by-value parameter (first and only argument).
template <typename T>
GG_DEVICE const T* gg_late_params(const T& by_value) {
    (void)by_value;
    const T* kp = (const T*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

// D[i][j] += sum_k Aop[i][k] * Bop[k][j], 32x32x16, one wave.
//   Aop: lane l holds Aop[i = l&31][k = 8*(l>>5) + e], e = 0..7
//   Bop: lane l holds Bop[k = 8*(l>>5) + e][j = l&31]
//   D  : lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
GG_DEVICE f32x16 gg_mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gg_bf16x8_native, a),
                                                   __builtin_bit_cast(gg_bf16x8_native, b), c, 0, 0, 0);
}

// D[i][j] += sum_k Aop[i][k] * Bop[k][j], 16x16x32, one wave (v_mfma_f32_16x16x32_bf16: the matrix-pipe time of HALF a 32x32x16).
//   Aop: lane l holds Aop[i = l&15][k = 8*(l>>4) + e], e = 0..7;   Bop: lane l holds Bop[k = 8*(l>>4) + e][j = l&15]
//   D  : lane l, reg r holds D[i = 4*(l>>4) + r][j = l&15]
// (both operands share the k labelling: element e of lane group l>>4 meets element e of the same group, whatever the hardware calls it)
GG_DEVICE f32x4 gg_mfma_16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gg_bf16x8_native, a), __builtin_bit_cast(gg_bf16x8_native, b), c, 0, 0, 0);
}

// ds_read_b64_tr_b16: every lane supplies the LDS address of 4 contiguous bf16 (8-byte aligned); inside each group
// of 16 lanes, lane i receives element (i & 3) of the quads addressed by lanes 4*j + (i >> 2), j = 0..3 (measured on
// gfx950 with tests/probes/tr_probe.hip). With lane s pointing at row (s >> 2), columns 4*(s & 3).. of a
// [4 k][16 m] block this hands lane i the 4 k-values of column i: a k-contiguous MFMA fragment out of a tile that
// is stored reduction-major.
typedef __attribute__((ext_vector_type(4))) short gg_s16x4_native;
GG_DEVICE u16x4 gg_lds_read_tr16(const bf16_t* p) {
    gg_s16x4_native r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gg_s16x4_native __attribute__((address_space(3)))*)p);
    return __builtin_bit_cast(u16x4, r);
}

template <int N>
GG_DEVICE void gg_wait_vm() {           // s_waitcnt vmcnt(N): at most N vector-memory operations of this wave outstanding
    static_assert(N >= 0 && N < 64, "six counter bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a WAVE-UNIFORM run-time n (the immediate must be a literal: a scalar jump over the literals a streaming
// kernel's prefetch depths produce). At most n vector-memory operations of this wave stay outstanding; they retire in order.
GG_DEVICE void gg_wait_vm_le(int n) {
#define GG_VMCASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (__builtin_amdgcn_readfirstlane(n)) {
        GG_VMCASE(1) GG_VMCASE(2) GG_VMCASE(3) GG_VMCASE(4) GG_VMCASE(5) GG_VMCASE(6) GG_VMCASE(7) GG_VMCASE(8) GG_VMCASE(9) GG_VMCASE(10)
        GG_VMCASE(11) GG_VMCASE(12) GG_VMCASE(13) GG_VMCASE(14) GG_VMCASE(15) GG_VMCASE(16) GG_VMCASE(17) GG_VMCASE(18) GG_VMCASE(19)
        GG_VMCASE(20) GG_VMCASE(21) GG_VMCASE(22) GG_VMCASE(23) GG_VMCASE(24) GG_VMCASE(25) GG_VMCASE(26) GG_VMCASE(27) GG_VMCASE(28)
        GG_VMCASE(29) GG_VMCASE(30) GG_VMCASE(31) GG_VMCASE(32) GG_VMCASE(33) GG_VMCASE(34) GG_VMCASE(35) GG_VMCASE(36) GG_VMCASE(37)
        GG_VMCASE(38) GG_VMCASE(39) GG_VMCASE(40) GG_VMCASE(41) GG_VMCASE(42) GG_VMCASE(43) GG_VMCASE(44) GG_VMCASE(45) GG_VMCASE(46)
        GG_VMCASE(47) GG_VMCASE(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;        // 0, and anything the table does not carry: wait for all
    }
#undef GG_VMCASE
}

// workgroup barrier that leaves vector-memory operations (LDS-DMA prefetches) in flight: __syncthreads() waits for vmcnt(0) first. This
// wave's LDS reads / writes are retired (lgkmcnt(0)) before it arrives; data another wave's DMA deposited is visible after that wave's
// own gg_wait_vm_le + this barrier.
GG_DEVICE void gg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Buffer addressing (SRSRC): a wave-uniform 128-bit descriptor {base, bytes} plus a 32-bit per-lane byte offset and a scalar byte
// offset. What it buys the convolution gather: no 64-bit per-lane address arithmetic in the k-loop (the per-lane part of an
// operand address is loop invariant, the per-k-tile part is one scalar), and out-of-range rows / padding taps are zero-filled by
// the hardware bounds check (voffset >= bytes -> 0; the scalar offset is not range-checked) instead of by branches around loads.
typedef __amdgpu_buffer_rsrc_t GgBuf;
GG_DEVICE GgBuf gg_make_buf(const void* base, unsigned long long bytes) {
    // the descriptor must be PROVABLY wave-uniform or hipcc wraps every buffer op in a waterfall loop: readfirstlane its inputs
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)nb, 0x00020000);
}
GG_DEVICE u16x8 gg_buf_load16(GgBuf r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(u16x8, v);
}

// a 16-byte load whose result nobody reads: it pulls the line into this XCD's L2 (and the memory-side cache) for a LATER kernel. Issued
// from assembly so that the compiler neither drops it nor waits for it; a wave may end with such loads outstanding.
GG_DEVICE void gg_buf_touch16(GgBuf r, unsigned voff) {
    u32x4 sink;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(sink) : "v"(voff), "s"(r) : "memory");
}

// buffer-addressed LDS-DMA (buffer_load_dwordx4 ... lds): lane i's 16 bytes land at lds_wave_base + 16 i (wave-uniform base, it
// travels in M0); out-of-range lanes deposit zeros. Counts on vmcnt; NOT ordered with ds_* operations: a reader needs
// gg_wait_vm<0>() in the issuing wave and a barrier.
GG_DEVICE void gg_buf_load_lds16(GgBuf r, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

// The same transfer issued from inline assembly, for kernels that keep SEVERAL steps of LDS-DMA in flight (gg_wgrads.h): hipcc's
// waitcnt pass knows that the builtin above writes LDS and, without alias scopes, puts `s_waitcnt vmcnt(0)` in front of EVERY later
// LDS read of the kernel - the prefetch depth collapses to zero (measured: 3.5 us per 32 KB step). The assembly form is invisible to
// that pass; ordering is then entirely the kernel's own gg_wait_vm_le + gg_barrier_lds. The descriptor is a plain SGPR quad
// {base lo, base hi, bytes, flags}; M0 (the LDS base of the wave's 1 KB) is written inside the statement and is not live across it
// in such kernels (they use no other M0 consumer: check the ISA when adding one).
typedef u32x4 GgBufS;
GG_DEVICE GgBufS gg_make_bufs(const void* base, unsigned long long bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned nb = (unsigned)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes);
    GgBufS r = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xFFFFu,
                (unsigned)__builtin_amdgcn_readfirstlane(nb), 0x00020000u};
    return r;
}
GG_DEVICE void gg_bufs_load_lds16(GgBufS r, unsigned voff, unsigned soff, void* lds_wave_base) {
    // (the low half of a flat LDS address is the LDS byte offset: the shared aperture is 4 GiB aligned. A generic -> local pointer cast
    // here trips hipcc 7.2's instruction verifier: V_CMP_NE_U32 against src_shared_base)
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_wave_base);
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(r), "s"(m), "s"(so) : "memory");
}
// hand-over of LDS data between the lanes of ONE wave (a wave-private staging area): the hardware executes a wave's LDS instructions
// in order, so nothing is emitted - the builtin only pins the compiler's schedule; the emulator, whose lanes are independent fibers,
// makes it a wave rendezvous
GG_DEVICE void gg_wave_sync() { __builtin_amdgcn_wave_barrier(); }
// a 16-byte store marked non-temporal (global_store_dwordx4 ... nt): a streamed output row does not displace the operand tiles that the
// same workgroups keep re-reading from this XCD's L2 (gg_pgemm.h)
GG_DEVICE void gg_store_nt16(void* p, u16x8 v) { __builtin_nontemporal_store(v, (u16x8*)p); }
// a loaded value declared complete HERE: the empty statement reads the register, so hipcc places the load's s_waitcnt in front of it,
// and redefines it, so that no later use carries a pending-load state across stores and branches (gg_pgemm.h)
GG_DEVICE void gg_settle(u16x8& v) { asm volatile("" : "+v"(v)); }
// a value the program knows to be wave-uniform, moved to a scalar register (loop bounds, LDS bases, branch conditions derived from the
// wave index would otherwise live in vector registers: divergent-loop code, waterfall loops around scalar operands)
GG_DEVICE int gg_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

GG_DEVICE float gg_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
GG_DEVICE float gg_shfl(float v, int src) { return __shfl(v, src, 64); }
// the value lane `src` holds, for a WAVE-UNIFORM src (v_readlane_b32: a few cycles; gg_shfl is a ds_bpermute round trip)
GG_DEVICE float gg_readlane(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(src)));
}
// sum over the 64 lanes, returned to every lane (wave-uniform): four DPP adds inside the 16-lane rows (quad swaps, half-row and row
// mirrors: ~4 cycles each, no LDS crossbar) and one readlane per row. The butterfly of gg_shfl_xor costs six ds_bpermute round trips
// (~100 cycles each) - the adaptive-conv coefficient kernel does one reduction per (sample, channel).
GG_DEVICE float gg_wave_sum_all(float v) {
    int x = __builtin_bit_cast(int, v);
#define GG_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    (void)x;
    GG_DPP_ADD(0xB1);        // quad_perm [1,0,3,2]
    GG_DPP_ADD(0x4E);        // quad_perm [2,3,0,1]
    GG_DPP_ADD(0x141);       // row_half_mirror
    GG_DPP_ADD(0x140);       // row_mirror
#undef GG_DPP_ADD
    const int vi = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48)));
}
// sum over the 16 lanes of one DPP row, returned to every lane of that row (the first four steps of gg_wave_sum_all: four
// independent sums per wavefront)
GG_DEVICE float gg_row16_sum(float v) {
#define GG_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    GG_DPP_ADD(0xB1);        // quad_perm [1,0,3,2]
    GG_DPP_ADD(0x4E);        // quad_perm [2,3,0,1]
    GG_DPP_ADD(0x141);       // row_half_mirror
    GG_DPP_ADD(0x140);       // row_mirror
#undef GG_DPP_ADD
    return v;
}
GG_DEVICE void gg_atomic_add(float* p, float v) { atomicAdd(p, v); }
// "last workgroup to arrive" ticket: release this workgroup's global writes, take a ticket; the taker of the last ticket sees every
// other workgroup's writes (acquire) and puts the counter back to zero when it is done
GG_DEVICE unsigned gg_ticket_take(unsigned* p) {
    __threadfence();
    const unsigned v = atomicAdd(p, 1u);
    __threadfence();
    return v;
}
GG_DEVICE void gg_ticket_reset(unsigned* p) { atomicExch(p, 0u); }
GG_DEVICE float gg_expf(float x) { return __expf(x); }
GG_DEVICE float gg_exp2f(float x) { return __builtin_amdgcn_exp2f(x); }     // bare v_exp_f32 (no range fix-ups: x <= 128 here)
GG_DEVICE bool gg_wave_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0; }    // wave-uniform result
GG_DEVICE float gg_rsqrtf(float x) { return rsqrtf(x); }
GG_DEVICE float gg_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32 (1 ulp): `1.f / x` compiles to a ~10-instruction IEEE division

#endif  // GG_HOST_EMULATION

// ---- shared scalar helpers (same code on device and in the emulator) ------------------------------

// a * b + c with ONE rounding, spelled out: where several unrolled copies of a reduction must give bit-identical results (the same
// sample in another batch slot), `a * b + c` leaves the contraction into an fma to the compiler, which may decide differently per copy
GG_HOST_DEVICE float gg_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

GG_HOST_DEVICE float gg_bf2f(bf16_t h) {
    union { unsigned int u; float f; } x;
    x.u = ((unsigned int)h) << 16;
    return x.f;
}

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet), identical to torch's .to(torch.bfloat16). The device pass uses
// the gfx950 converter (v_cvt_pk_bf16_f32: same rounding, one instruction per PAIR of values instead of ~6 integer
// ops per value — the softmax / epilogue / elementwise kernels are VALU-bound on exactly this); the host pass and the
// emulator keep the integer formulation.
GG_HOST_DEVICE bf16_t gg_f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GG_HOST_EMULATION)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
#endif
    union { unsigned int u; float f; } x;
    x.f = f;
    if ((x.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x.u >> 16) | 0x40u);
    unsigned int lsb = (x.u >> 16) & 1u;
    x.u += 0x7fffu + lsb;
    return (bf16_t)(x.u >> 16);
}

