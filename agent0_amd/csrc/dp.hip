// a0_dp_*: the data-parallel gradient exchange — an in-place fp32 SUM all-reduce over RCCL (xGMI between the GPUs of one node), enqueued
// on the CALLER's stream like every other entry point, so that the exchange is part of the update's hipGraph instead of a host-side call
// between graphs (the reference has no counterpart: its only inter-process traffic is Launchpad's gRPC,
// launch.py:166-176).
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): a process that already carries an RCCL — PyTorch ships one with that soname — keeps
// exactly one copy, and libagent0_hip.so has no link-time dependency on it.  Rendezvous stays with the caller: rank 0 obtains the 128-byte
// unique id (a0_dp_unique_id) and distributes it by whatever means it has (torch.distributed broadcast in agent0_amd/deepq/dist.py).
#include "a0_internal.h"

#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>

// The five RCCL entry points this file uses, declared here so that the library builds without the RCCL development headers (the
// symbols are looked up with dlsym at run time anyway).  Values as in rccl.h of ROCm 7.2: ncclSuccess = 0, ncclFloat32 = 7, ncclSum = 0,
// ncclUniqueId = 128 opaque bytes, ncclComm_t = opaque pointer.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclFloat32 = 7;
static const ncclRedOp_t ncclSum = 0;

namespace {
struct rccl_api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

rccl_api& api() {
    static rccl_api a;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string why = "?";
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);       // the copy the process already has, if any
            if (a.lib) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (a.lib) break;
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!a.lib) { const char* e = dlerror(); why = e ? e : "?"; }     // dlerror() clears the state it returns: read it once
        }
        if (!a.lib) { a.error = "a0_dp: cannot load librccl.so.1: " + why; return; }
        auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); if (!p && a.error.empty()) a.error = std::string("a0_dp: librccl lacks ") + n; return p; };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
        a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
        a.CommCuDevice = (decltype(a.CommCuDevice))sym("ncclCommCuDevice");
    });
    return a;
}

int fail_rccl(ncclResult_t r, const char* what) {
    rccl_api& a = api();
    return a0_fail(A0_EHIP, (std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "rccl error")).c_str());
}
}  // namespace

extern "C" int a0_dp_unique_id(void* host_id128) {
    rccl_api& a = api();
    if (!a.error.empty()) return a0_fail(A0_EINVAL, a.error.c_str());
    if (!host_id128) return a0_fail(A0_EINVAL, "a0_dp_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "rendezvous blob is 128 bytes");
    ncclResult_t r = a.GetUniqueId((ncclUniqueId*)host_id128);
    return r == ncclSuccess ? A0_OK : fail_rccl(r, "ncclGetUniqueId");
}

extern "C" long long a0_dp_init(const void* host_id128, int rank, int world) {
    rccl_api& a = api();
    if (!a.error.empty()) { a0_fail(A0_EINVAL, a.error.c_str()); return 0; }
    if (!host_id128 || world < 1 || rank < 0 || rank >= world) { a0_fail(A0_EINVAL, "a0_dp_init: bad argument"); return 0; }
    ncclUniqueId id;
    std::memcpy(&id, host_id128, sizeof id);
    ncclComm_t comm = nullptr;
    ncclResult_t r = a.CommInitRank(&comm, world, id, rank);      // binds the communicator to the calling thread's current device
    if (r != ncclSuccess) { fail_rccl(r, "ncclCommInitRank"); return 0; }
    return (long long)(intptr_t)comm;
}

extern "C" int a0_dp_allreduce(long long comm, float* buf, long long n, void* stream) {
    rccl_api& a = api();
    if (!a.error.empty()) return a0_fail(A0_EINVAL, a.error.c_str());
    if (!comm || !buf || n < 1) return a0_fail(A0_EINVAL, "a0_dp_allreduce: bad argument");
    a0_trace_scope range("exchange");
    ncclResult_t r = a.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)(intptr_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? A0_OK : fail_rccl(r, "ncclAllReduce");
}

// What RCCL itself says about a communicator — ncclCommCount, ncclCommUserRank, ncclCommCuDevice — so that a multi-rank record (bench.py's "rccl" object) carries the
// library's own statement of the rank count instead of the launcher's environment.  host_out3 = {nranks, rank, device}.
extern "C" int a0_dp_info(long long comm, int* host_out3) {
    rccl_api& a = api();
    if (!a.error.empty()) return a0_fail(A0_EINVAL, a.error.c_str());
    if (!comm || !host_out3) return a0_fail(A0_EINVAL, "a0_dp_info: bad argument");
    ncclComm_t c = (ncclComm_t)(intptr_t)comm;
    ncclResult_t r = a.CommCount(c, &host_out3[0]);
    if (r == ncclSuccess) r = a.CommUserRank(c, &host_out3[1]);
    if (r == ncclSuccess) r = a.CommCuDevice(c, &host_out3[2]);
    return r == ncclSuccess ? A0_OK : fail_rccl(r, "ncclCommCount / ncclCommUserRank / ncclCommCuDevice");
}

extern "C" int a0_dp_destroy(long long comm) {
    rccl_api& a = api();
    if (!a.error.empty()) return a0_fail(A0_EINVAL, a.error.c_str());
    if (!comm) return A0_OK;
    ncclResult_t r = a.CommDestroy((ncclComm_t)(intptr_t)comm);
    return r == ncclSuccess ? A0_OK : fail_rccl(r, "ncclCommDestroy");
}
