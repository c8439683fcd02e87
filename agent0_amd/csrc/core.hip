// Error plumbing and device queries for the C-ABI.
#include "a0_internal.h"

#include <cstring>
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
