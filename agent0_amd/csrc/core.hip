// Error plumbing and device queries for the C-ABI.
#include "a0_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>

static thread_local std::string a0_tls_error;

int a0_fail(int code, const char* msg) {
    a0_tls_error = msg ? msg : "";
    return code;
}

int a0_fail_hip(int hip_error, const char* what) {
    if (hip_error == (int)hipSuccess) return A0_OK;
    a0_tls_error = std::string(what ? what : "hip") + ": " + hipGetErrorString((hipError_t)hip_error);
    return A0_EHIP;
}

extern "C" const char* a0_last_error(void) { return a0_tls_error.c_str(); }
extern "C" int a0_abi_version(void) { return A0_ABI_VERSION; }
// "default" for the product build (agent0_amd/csrc/build.sh); tools/build_variant.sh stamps its name and extra -D flags here, and the
// product loader (agent0_amd/_abi.py) refuses any library that does not say "default".
#ifndef A0_BUILD_VARIANT
#define A0_BUILD_VARIANT "default"
#endif
extern "C" const char* a0_build_info(void) { return A0_BUILD_VARIANT; }

extern "C" int a0_device_info(int* cu_count, long long* hbm_bytes, char* arch_name64) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return a0_fail_hip((int)e, "a0_device_info");
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return a0_fail_hip((int)e, "a0_device_info");
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (long long)p.totalGlobalMem;
    if (arch_name64) { std::strncpy(arch_name64, p.gcnArchName, 63); arch_name64[63] = 0; }
    return A0_OK;
}

// ------------------------------------------------------------------------------------------------ roctx ranges (SURVEY.md §5 tracing; the reference times wall clock only, trainer.py:176-180)
// Named host-side ranges for rocprofv3 --marker-trace: `rollout`, `update_block`, `update`, `exchange` ..., prefixed with the rank ("r3:exchange").  Off unless
// A0_ROCTX=1 (one getenv + a branch otherwise); the marker library is resolved at run time (dlopen "librocprofiler-sdk-roctx.so", no link-time dependency).  A range
// brackets the ENQUEUE of its launches on the host — the kernels themselves are in the kernel trace of the same run, on the streams named there.
namespace {
struct roctx_api {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    void (*mark)(const char*) = nullptr;
    bool on = false;
    char prefix[16] = "";
};
roctx_api& roctx() {
    static roctx_api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("A0_ROCTX");
        if (!e || e[0] != '1') return;
        void* lib = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return;
        a.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        a.pop = (int (*)())dlsym(lib, "roctxRangePop");
        a.mark = (void (*)(const char*))dlsym(lib, "roctxMarkA");
        a.on = a.push && a.pop;
    });
    return a;
}
}  // namespace

void a0_trace_push_internal(const char* name) {
    roctx_api& a = roctx();
    if (!a.on) return;
    char buf[96];
    snprintf(buf, sizeof buf, "%s%s", a.prefix, name ? name : "?");
    a.push(buf);
}
void a0_trace_pop_internal() {
    roctx_api& a = roctx();
    if (a.on) a.pop();
}
extern "C" int a0_trace_enabled(void) { return roctx().on ? 1 : 0; }
extern "C" int a0_trace_rank(int rank) {
    roctx_api& a = roctx();
    if (rank >= 0) snprintf(a.prefix, sizeof a.prefix, "r%d:", rank); else a.prefix[0] = 0;
    return A0_OK;
}
extern "C" int a0_trace_push(const char* name) { a0_trace_push_internal(name); return A0_OK; }
extern "C" int a0_trace_pop(void) { a0_trace_pop_internal(); return A0_OK; }
