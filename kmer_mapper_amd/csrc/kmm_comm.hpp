// kmm_comm.hpp — part of libkmm (MI355X / gfx950); included by kmm.hip.
// The one exchange step of the path: the sum of the per-GPU uint32 node-count vectors (RCCL over xGMI).
// Replaces the additive reduce of per-chunk vectors in the reference
// (kmer_mapper/command_line_interface.py:124-130, shared_memory_wrapper's additative_shared_array_map_reduce).
//
// RCCL is loaded at first use (dlopen), not linked: libkmm.so stays loadable on a box without RCCL, and inside
// a PyTorch process the loader hands back the librccl that torch has already mapped.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int rccl_load()
{
    if (g_rccl.lib)
        return KMM_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    // first a copy the process has already mapped (PyTorch's wheel carries its own librccl.so, SONAME librccl.so.1: two
    // RCCL instances in one process would each own a set of channels and proxy threads), then the system's
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD)))
            break;
    for (const char *n : names)
        if (h || (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!h)
        return fail(KMM_ERR_HIP, "RCCL is not available (dlopen librccl.so.1: %s)", dlerror());
    RcclApi a;
    a.lib = h;
#define KMM_SYM(field, name)                                                                     \
    *(void **)(&a.field) = dlsym(h, name);                                                       \
    if (!a.field)                                                                                \
        return fail(KMM_ERR_HIP, "RCCL symbol %s is missing", name);
    KMM_SYM(GetUniqueId, "ncclGetUniqueId")
    KMM_SYM(CommInitRank, "ncclCommInitRank")
    KMM_SYM(CommInitAll, "ncclCommInitAll")
    KMM_SYM(CommDestroy, "ncclCommDestroy")
    KMM_SYM(Reduce, "ncclReduce")
    KMM_SYM(AllReduce, "ncclAllReduce")
    KMM_SYM(GroupStart, "ncclGroupStart")
    KMM_SYM(GroupEnd, "ncclGroupEnd")
    KMM_SYM(GetErrorString, "ncclGetErrorString")
#undef KMM_SYM
    g_rccl = a;
    return KMM_OK;
}

#define RCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(KMM_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                        g_rccl.GetErrorString(r_));                                                \
    } while (0)

} // namespace
