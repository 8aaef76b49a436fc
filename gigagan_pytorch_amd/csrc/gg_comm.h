// gg_comm.h — the data-parallel exchange of the GigaGAN step behind the C ABI (SURVEY.md §8b `gg_comm_*`): RCCL over xGMI, bound at
// run time from the librccl the process already carries (PyTorch-ROCm bundles one; two RCCL copies in one process would each
// bootstrap their own transport). Replaces the reference's accelerate / DDP gradient all-reduce (gp.py:1898-1908, :1987) and its
// hand-written all_gather (distributed.py:20-68). One communicator per process (one process per GPU); collectives are enqueued
// on the caller's stream, never synchronise the host, and are issued by the Python host code on a dedicated side stream with
// event fences to / from the compute stream.
#pragma once
#include <dlfcn.h>
#include <stdio.h>
#include <stdint.h>
#include <stddef.h>

namespace gg_comm {

struct UniqueId { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
enum { kSum = 0 };                                // ncclRedOp_t
enum { kFloat32 = 7, kBfloat16 = 9, kInt8 = 0 };  // ncclDataType_t

struct Api {
    void* handle = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

static Api g_api;
static Comm g_comm = nullptr;
static int g_world = 0, g_rank = -1;

static bool load(const char* path, char* err, size_t errlen) {
    if (g_api.handle) return true;
    const char* candidates[] = {path, "librccl.so.1", "librccl.so", nullptr};
    void* h = nullptr;
    for (int i = 0; i < 3 && !h; ++i) {
        if (!candidates[i] || !candidates[i][0]) continue;
        h = dlopen(candidates[i], RTLD_NOW | RTLD_NOLOAD);       // the copy that is already mapped (torch's), if any
        if (!h && i == 0) h = dlopen(candidates[i], RTLD_NOW | RTLD_GLOBAL);   // an explicit path may be loaded fresh
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        snprintf(err, errlen, "gg_comm: cannot load librccl (%s)", dlerror());
        return false;
    }
#define GG_SYM(field, name)                                                      \
    *(void**)(&g_api.field) = dlsym(h, name);                                    \
    if (!g_api.field) { snprintf(err, errlen, "gg_comm: librccl lacks %s", name); return false; }
    GG_SYM(GetUniqueId, "ncclGetUniqueId")
    GG_SYM(CommInitRank, "ncclCommInitRank")
    GG_SYM(CommDestroy, "ncclCommDestroy")
    GG_SYM(CommCount, "ncclCommCount")
    GG_SYM(AllReduce, "ncclAllReduce")
    GG_SYM(AllGather, "ncclAllGather")
    GG_SYM(GetErrorString, "ncclGetErrorString")
#undef GG_SYM
    g_api.handle = h;
    return true;
}

static int dtype_of(int32_t dtype, size_t* elem) {
    if (dtype == 0) { *elem = 4; return kFloat32; }
    if (dtype == 1) { *elem = 2; return kBfloat16; }
    if (dtype == 2) { *elem = 1; return kInt8; }
    return -1;
}

}  // namespace gg_comm
