// vg_rccl.hip — the all-reduce hook of the large-window path (include/vinsgpu.h "Large windows and landmark shards") bound to
// RCCL over xGMI in C, inside the library: one communicator per handle, two in-place ncclAllReduce(ncclDouble, ncclSum) per
// trust-region round, enqueued on the handle's launch stream between two kernel launches — no host round trip, no Python in
// the loop.  librccl is opened at run time (dlopen): a single-GPU process never loads it and libvinsgpu.so carries no
// link-time dependency on it.  The rendezvous (who is rank 0, how the 128-byte unique id reaches the other ranks) is the
// caller's: torch.distributed.broadcast_object_list in bench.py / the tests, MPI_Bcast or a file elsewhere.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>
#include <rccl/rccl.h>
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that the process already holds (torch ships its own librccl) wins over a second one from /opt/rocm
        for (int pass = 0; pass < 2 && !a.lib; ++pass)
            for (const char* name : {"librccl.so", "librccl.so.1"}) {
                a.lib = dlopen(name, RTLD_LAZY | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (a.lib) break;
            }
        if (!a.lib) return;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    });
    return a;
}
int fail(vg_handle* h, const char* what, ncclResult_t r) {
    RcclApi& a = api();
    h->err = std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error");
    return VG_ERR_HIP;
}
// the hook itself: (user = the communicator)
int rccl_allreduce(void* user, double* buf, size_t count, void* stream) {
    return api().AllReduce(buf, buf, count, ncclDouble, ncclSum, (ncclComm_t)user, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}
}  // namespace

extern "C" int vg_rccl_unique_id(char* id128) {
    static_assert(sizeof(ncclUniqueId) == VG_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    RcclApi& a = api();
    if (!id128) return VG_ERR_BAD_ARG;
    if (!a.ok) return VG_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (a.GetUniqueId(&id) != ncclSuccess) return VG_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return VG_OK;
}

extern "C" int vg_ba_rccl_init(vg_handle* h, int nranks, int rank, const char* id128) {
    if (!h || nranks < 1 || rank < 0 || rank >= nranks || !id128) return VG_ERR_BAD_ARG;
    RcclApi& a = api();
    if (!a.ok) { h->err = "librccl.so could not be opened (dlopen)"; return VG_ERR_UNSUPPORTED; }
    if (h->rccl_comm) { h->err = "vg_ba_rccl_init: this handle already has a communicator"; return VG_ERR_BAD_ARG; }
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return VG_ERR_HIP; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = a.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return fail(h, "ncclCommInitRank", r);
    h->rccl_comm = comm;
    return vg_ba_set_allreduce(h, rccl_allreduce, comm);
}

extern "C" int vg_ba_rccl_finalize(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    if (!h->rccl_comm) return VG_OK;
    (void)hipStreamSynchronize(h->stream);
    vg_ba_set_allreduce(h, nullptr, nullptr);
    const ncclResult_t r = api().CommDestroy((ncclComm_t)h->rccl_comm);
    h->rccl_comm = nullptr;
    return r == ncclSuccess ? VG_OK : fail(h, "ncclCommDestroy", r);
}
