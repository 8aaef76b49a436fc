// gg_emu.cpp — fiber-based workgroup emulator behind gg_device_emu.h (TEST INFRASTRUCTURE ONLY).
#include "gg_device_emu.h"
#include <ucontext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <deque>
#include <utility>

gg_emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {

constexpr size_t kStack = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    unsigned tx = 0, ty = 0, tz = 0;
};

struct WaveState {
    int arrived = 0;
    unsigned gen = 0;
    u16x8 a[64], b[64];
    float f[64];
    u16x4 q[64];
};

ucontext_t g_sched;
std::vector<Fiber> g_fibers;
std::vector<WaveState> g_waves;
int g_cur = -1;
int g_nthreads = 0;
int g_block_arrived = 0;
unsigned g_block_gen = 0;
const std::function<void()>* g_body = nullptr;
std::vector<std::deque<std::pair<const void*, void*>>> g_dma;     // per thread: LDS-DMA transfers issued and not yet landed
bool g_dma_late = false, g_reverse = false;

void yield_to_sched() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

void dma_flush(int t, size_t keep) {
    auto& q = g_dma[t];
    while (q.size() > keep) {
        memcpy(q.front().second, q.front().first, 16);
        q.pop_front();
    }
}

void fiber_entry() {
    (*g_body)();
    dma_flush(g_cur, 0);
    g_fibers[g_cur].done = true;
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

void wave_sync() {
    WaveState& w = g_waves[g_cur / 64];
    int nlanes = g_nthreads - (g_cur / 64) * 64;
    if (nlanes > 64) nlanes = 64;
    unsigned my_gen = w.gen;
    if (++w.arrived == nlanes) {
        w.arrived = 0;
        w.gen++;
        return;
    }
    while (w.gen == my_gen) yield_to_sched();
}

}  // namespace

void gg_emu_dma_issue(const void* g, void* dst) {
    if (!g_dma_late) { memcpy(dst, g, 16); return; }
    g_dma[g_cur].push_back({g, dst});
}

void gg_emu_dma_wait(int keep_newest) {
    if (g_dma_late) dma_flush(g_cur, (size_t)keep_newest);
}

void gg_emu_syncthreads() {
    unsigned my_gen = g_block_gen;
    if (++g_block_arrived == g_nthreads) {
        g_block_arrived = 0;
        g_block_gen++;
        return;
    }
    while (g_block_gen == my_gen) yield_to_sched();
}

static inline float bf2f_(unsigned short h) {
    union { unsigned u; float f; } x;
    x.u = ((unsigned)h) << 16;
    return x.f;
}

// Lane mappings of v_mfma_f32_32x32x16_bf16 as documented in the CDNA4 guide:
//   A: lane l -> row i = l&31, k = 8*(l>>5)+e ; B: lane l -> col j = l&31, k = 8*(l>>5)+e
//   D: lane l, reg r -> row i = (r&3) + 8*(r>>2) + 4*(l>>5), col j = l&31
f32x16 gg_emu_mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
    WaveState& w = g_waves[g_cur / 64];
    int lane = g_cur & 63;
    w.a[lane] = a;
    w.b[lane] = b;
    wave_sync();
    f32x16 out;
    int j = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float s = c[r];
        for (int k = 0; k < 16; ++k) {
            float av = bf2f_(w.a[(k >> 3) * 32 + i][k & 7]);
            float bv = bf2f_(w.b[(k >> 3) * 32 + j][k & 7]);
            s += av * bv;
        }
        out[r] = s;
    }
    wave_sync();
    return out;
}

// v_mfma_f32_16x16x32_bf16 (gg_device.h: lane l holds A[i = l & 15][k = 8 (l >> 4) + e] / B[k][j = l & 15]; D[i = 4 (l >> 4) + r][j = l & 15])
f32x4 gg_emu_mfma_16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c) {
    WaveState& w = g_waves[g_cur / 64];
    int lane = g_cur & 63;
    w.a[lane] = a;
    w.b[lane] = b;
    wave_sync();
    f32x4 out;
    int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * (lane >> 4) + r;
        float s = c[r];
        for (int k = 0; k < 32; ++k) {
            float av = bf2f_(w.a[(k >> 3) * 16 + i][k & 7]);
            float bv = bf2f_(w.b[(k >> 3) * 16 + j][k & 7]);
            s += av * bv;
        }
        out[r] = s;
    }
    wave_sync();
    return out;
}

float gg_emu_shfl(float v, int src_lane) {
    WaveState& w = g_waves[g_cur / 64];
    int lane = g_cur & 63;
    w.f[lane] = v;
    wave_sync();
    float r = w.f[src_lane & 63];
    wave_sync();
    return r;
}

// ds_read_b64_tr_b16 as measured on gfx950 (tests/probes/tr_probe.hip): inside each group of 16 lanes, lane i
// receives element (i & 3) of the 4-element quads addressed by lanes 4*j + (i >> 2), j = 0..3.
u16x4 gg_emu_lds_read_tr16(const bf16_t* p) {
    WaveState& w = g_waves[g_cur / 64];
    int lane = g_cur & 63;
    u16x4 mine = {p[0], p[1], p[2], p[3]};
    w.q[lane] = mine;
    wave_sync();
    int g = lane >> 4, i = lane & 15;
    u16x4 out;
    for (int j = 0; j < 4; ++j) out[j] = w.q[g * 16 + 4 * j + (i >> 2)][i & 3];
    wave_sync();
    return out;
}

void gg_emu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads % 64 != 0) {
        fprintf(stderr, "gg_emu: block size %d is not a multiple of the 64-lane wavefront\n", nthreads);
        abort();
    }
    static std::vector<char*> stacks;
    while ((int)stacks.size() < nthreads) stacks.push_back((char*)malloc(kStack));
    gridDim = grid;
    blockDim = block;
    g_body = &body;
    g_nthreads = nthreads;
    const char* dm = getenv("GG_EMU_DMA");
    g_dma_late = dm && dm[0] == 'l';
    const char* rv = getenv("GG_EMU_REVERSE");
    g_reverse = rv && rv[0] == '1';
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_fibers.assign(nthreads, Fiber());
                g_waves.assign((nthreads + 63) / 64, WaveState());
                g_dma.assign(nthreads, {});
                g_block_arrived = 0;
                g_block_gen = 0;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    f.tx = t % block.x;
                    f.ty = (t / block.x) % block.y;
                    f.tz = t / (block.x * block.y);
                    f.stack = stacks[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &g_sched;
                    makecontext(&f.ctx, fiber_entry, 0);
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    int progressed = 0;
                    for (int tt = 0; tt < nthreads; ++tt) {
                        const int t = g_reverse ? nthreads - 1 - tt : tt;
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        g_cur = t;
                        threadIdx = gg_emu_dim3(f.tx, f.ty, f.tz);
                        blockIdx = gg_emu_dim3(bx, by, bz);
                        swapcontext(&g_sched, &f.ctx);
                        ++progressed;
                        if (f.done) --remaining;
                    }
                    if (!progressed) break;
                }
            }
    g_body = nullptr;
}

unsigned long long gg_emu_touched_bytes = 0;
extern "C" unsigned long long gg_emu_touched(void) { unsigned long long v = gg_emu_touched_bytes; gg_emu_touched_bytes = 0; return v; }
