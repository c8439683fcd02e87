// Internal helpers shared by the .hip translation units: error plumbing for the C-ABI.
#pragma once
#include "../../include/agent0_hip.h"
#include "a0_defs.h"

#include <exception>
#include <stdexcept>
#include <string>

int a0_fail(int code, const char* msg);          // records the thread-local message, returns code
int a0_fail_hip(int hip_error, const char* what);

struct a0_hip_error : std::runtime_error {
    int err;
    a0_hip_error(int e, const std::string& w) : std::runtime_error(w), err(e) {}
};

#define A0_STR2(x) #x
#define A0_STR(x) A0_STR2(x)
#define A0_HIP_THROW(expr)                                                                                  \
    do {                                                                                                    \
        hipError_t a0_e_ = (expr);                                                                          \
        if (a0_e_ != hipSuccess) throw a0_hip_error((int)a0_e_, std::string(__FILE__ ":" A0_STR(__LINE__) ": ") + hipGetErrorString(a0_e_)); \
    } while (0)

#define A0_TRY try {
#define A0_CATCH                                                   \
    }                                                              \
    catch (const a0_hip_error& e) { return a0_fail_hip(e.err, e.what()); } \
    catch (const std::exception& e) { return a0_fail(A0_EINVAL, e.what()); } \
    catch (...) { return a0_fail(A0_EINVAL, "unknown C++ exception"); }

// profiler probe (net.hip): HIP events around launches tagged `tag`
#if defined(__HIPCC__)
bool a0_probe_start(int tag, hipStream_t st);
// conv1_wgrad.hip: per-observation conv1 weight gradient on the bf16 pipe; returns the slab count (0 = unsupported shape)
int a0_conv1_wgrad_fused_launch(const a0_frames_arg* f, int C, int H, int W, int B, const float* d1, float* slabs, hipStream_t st);
void a0_probe_stop(hipStream_t st, double flops);
// conv23_wgrad.hip: per-observation conv2 + conv3 weight gradients on the bf16 pipe (one launch); returns 1 if it ran, 0 = unsupported shape
struct a0_net_core;
int a0_conv23_wgrad_fused_launch(const a0_net_core& n, int B, const float* act1, const float* act2, const float* d2, const float* d3, float* slab2, float* slab3,
                                 hipStream_t st);
#endif
#define A0_TAG_ENCODER_FUSED 12
#define A0_TAG_ENCODER_DGRAD_FUSED 13
#define A0_TAG_ACTOR_STEP_ENC 14      // the actor step's tail + env step + next observation's encoder in one kernel (its own family: the encoder's roofline keeps the pure launches)
