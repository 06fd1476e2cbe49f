// Shard exchange inside the library: RCCL over xGMI, one communicator per context (one process per GPU).
// The RCCL entry points are looked up at run time (dlopen), so the library neither links against librccl
// nor pulls a second copy in beside the one a host framework may already have loaded.
//
// What travels:
//   * per iteration ONE all-gather of 4 + 2 Dc doubles per rank (log-sum-exp and moment partials),
//     B of them batched for a fused block -- latency-bound, payloads of a few hundred bytes;
//   * on the (rare) resampling events: the tile totals of the shard's weight scan (N_local / 1024 doubles),
//     then an all-to-all of the resampling keys (8 B per particle) and of ONLY the ancestor rows each rank
//     needs (8 D bytes per particle) -- point-to-point ncclSend / ncclRecv pairs in one group, which is the
//     natural shape on xGMI's fully connected point-to-point links.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

namespace smcn {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    bool ok = false;
    RcclApi() {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { why = "librccl.so not found"; return; }
#define SMCN_SYM(field, sym)                                                 \
        field = (decltype(field))dlsym(h, sym);                              \
        if (!field) { why = std::string("librccl: missing ") + sym; return; }
        SMCN_SYM(GetUniqueId, "ncclGetUniqueId");
        SMCN_SYM(CommInitRank, "ncclCommInitRank");
        SMCN_SYM(CommDestroy, "ncclCommDestroy");
        SMCN_SYM(AllGather, "ncclAllGather");
        SMCN_SYM(Send, "ncclSend");
        SMCN_SYM(Recv, "ncclRecv");
        SMCN_SYM(GroupStart, "ncclGroupStart");
        SMCN_SYM(GroupEnd, "ncclGroupEnd");
        SMCN_SYM(CommCount, "ncclCommCount");
        SMCN_SYM(CommUserRank, "ncclCommUserRank");
        SMCN_SYM(GetVersion, "ncclGetVersion");
        SMCN_SYM(GetErrorString, "ncclGetErrorString");
#undef SMCN_SYM
        ok = true;
    }
};
inline RcclApi& rccl() { static RcclApi a; return a; }

}  // namespace smcn
