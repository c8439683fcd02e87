// Internal helpers shared by the .hip translation units: error plumbing for the C-ABI.
#pragma once
#include "../../include/agent0_hip.h"
#include "a0_defs.h"

#include <exception>
#if defined(__HIPCC__)
#include <hip/hip_ext.h>
#endif
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

// roctx ranges (core.hip): no-ops unless A0_ROCTX=1
void a0_trace_push_internal(const char* name);
void a0_trace_pop_internal();
struct a0_trace_scope {
    explicit a0_trace_scope(const char* name) { a0_trace_push_internal(name); }
    ~a0_trace_scope() { a0_trace_pop_internal(); }
    a0_trace_scope(const a0_trace_scope&) = delete;
    a0_trace_scope& operator=(const a0_trace_scope&) = delete;
};

// profiler probe (net.hip): HIP events around launches tagged `tag`
#if defined(__HIPCC__)
bool a0_probe_start(int tag, hipStream_t st);
// conv1_wgrad.hip: per-observation conv1 weight gradient on the bf16 pipe; returns the slab count (0 = unsupported shape)
int a0_conv1_wgrad_fused_launch(const a0_frames_arg* f, int C, int H, int W, int B, const float* d1, float* slabs, hipStream_t st);
void a0_probe_stop(hipStream_t st, double flops);
// Kernel-exact form for launches of ONE kernel: the probe hands out its next event pair and the launch carries it (hipExtLaunchKernelGGL: the events take the
// dispatch's own begin / end timestamps, what rocprofv3's kernel trace reports) — an event recorded in front of a launch also counts the ~2.5 us between the
// previous kernel's end and this one's first wave.
bool a0_probe_events(int tag, hipEvent_t* start, hipEvent_t* stop);
void a0_probe_commit(double flops);
int a0_x9_products_now();      // net.hip: 6 or 9 cross products in the split-operand kernels (a0_x9_products)
#define A0_LAUNCH_PROBED(tag, flops, kern, grid, block, lds, st, ...)                                                              \
    do {                                                                                                                           \
        hipEvent_t a0_e0_ = nullptr, a0_e1_ = nullptr;                                                                             \
        if (a0_probe_events((tag), &a0_e0_, &a0_e1_)) {                                                                            \
            hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(lds), st, a0_e0_, a0_e1_, 0, __VA_ARGS__);                         \
            a0_probe_commit(flops);                                                                                                \
        } else                                                                                                                     \
            hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                           \
    } while (0)
// conv23_wgrad.hip: per-observation conv2 + conv3 weight gradients on the bf16 pipe (one launch); returns 1 if it ran, 0 = unsupported shape
struct a0_net_core;
int a0_conv23_wgrad_fused_launch(const a0_net_core& n, int B, const float* act1, const float* act2, const float* d2, const float* d3, float* slab2, float* slab3,
                                 hipStream_t st);
#endif
#define A0_TAG_ENCODER_FUSED 12
#define A0_TAG_ENCODER_DGRAD_FUSED 13
#define A0_TAG_ACTOR_STEP_ENC 14      // the actor step's tail + env step + next observation's encoder in one kernel (its own family: the encoder's roofline keeps the pure launches)
