// Philox4x32-10 counter RNG fills (gfx950).  Contract: oracle/philox.c — element i of stream (seed, stream) is word
// (i & 3) of philox(ctr = (i>>2 lo, i>>2 hi, stream, 0), key = (seed lo, seed hi)); the u32 / uniform streams are
// bit-exact against the oracle, the normal stream (Box-Muller) to transcendental rounding.
// Stands in for the host RNGs the reference draws from (numpy global MT19937 in agent0/deepq/agent.py:29-36, torch CPU
// generator in agent0/deepq/model.py:74-76,238); parity tests inject draws instead of comparing streams.
#include "a0_internal.h"
#include "philox.h"

#pragma clang fp contract(off)

__global__ void a0_rng_u32_kernel(unsigned long long seed, uint32_t stream, unsigned long long offset, uint32_t* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = a0_philox_word(seed, stream, offset + (unsigned long long)i);
}

__global__ void a0_rng_uniform_kernel(unsigned long long seed, uint32_t stream, unsigned long long offset, float* __restrict__ out, long long n,
                                      const long long* __restrict__ ctrl, int ctrl_idx) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ctrl) offset += (unsigned long long)ctrl[ctrl_idx];
    out[i] = (float)(a0_philox_word(seed, stream, offset + (unsigned long long)i) >> 8) * 0x1.0p-24f;
}

__global__ void a0_rng_randint_kernel(unsigned long long seed, uint32_t stream, unsigned long long offset, int hi, int* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (int)(a0_philox_word(seed, stream, offset + (unsigned long long)i) % (uint32_t)hi);
}

__global__ void a0_rng_normal_kernel(unsigned long long seed, uint32_t stream, unsigned long long offset, float stdv, float* __restrict__ out, long long n,
                                     const long long* __restrict__ ctrl, int ctrl_idx) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ctrl) offset += (unsigned long long)ctrl[ctrl_idx];
    const unsigned long long pos = offset + (unsigned long long)i;
    const unsigned long long pair = pos & ~1ull;
    const float u1 = (float)((a0_philox_word(seed, stream, pair) >> 8) + 1u) * 0x1.0p-24f;
    const float u2 = (float)(a0_philox_word(seed, stream, pair + 1) >> 8) * 0x1.0p-24f;
    const float rad = sqrtf(-2.0f * logf(u1));
    const float ang = 6.283185307179586f * u2;
    out[i] = stdv * rad * ((pos & 1) ? sinf(ang) : cosf(ang));
}

#define A0_RNG_LAUNCH(kernel, ...)                                                                       \
    if (!out || n < 1) return a0_fail(A0_EINVAL, "a0_rng: bad argument");                               \
    hipLaunchKernelGGL(kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_h, __VA_ARGS__); \
    return a0_fail_hip((int)hipGetLastError(), "a0_rng")

extern "C" int a0_rng_u32(unsigned long long seed, unsigned int stream, unsigned long long offset, unsigned int* out, long long n, void* stream_h) {
    A0_RNG_LAUNCH(a0_rng_u32_kernel, seed, stream, offset, out, n);
}
extern "C" int a0_rng_uniform(unsigned long long seed, unsigned int stream, unsigned long long offset, float* out, long long n, void* stream_h) {
    A0_RNG_LAUNCH(a0_rng_uniform_kernel, seed, stream, offset, out, n, (const long long*)nullptr, 0);
}
extern "C" int a0_rng_uniform_ctrl(unsigned long long seed, unsigned int stream, unsigned long long offset, float* out, long long n, const long long* ctrl,
                                   int ctrl_idx, void* stream_h) {
    if (ctrl_idx < 0 || ctrl_idx >= A0_CTRL_WORDS) return a0_fail(A0_EINVAL, "a0_rng_uniform_ctrl: bad ctrl index");
    A0_RNG_LAUNCH(a0_rng_uniform_kernel, seed, stream, offset, out, n, ctrl, ctrl_idx);
}
extern "C" int a0_rng_randint(unsigned long long seed, unsigned int stream, unsigned long long offset, int hi, int* out, long long n, void* stream_h) {
    if (hi < 1) return a0_fail(A0_EINVAL, "a0_rng_randint: hi < 1");
    A0_RNG_LAUNCH(a0_rng_randint_kernel, seed, stream, offset, hi, out, n);
}
extern "C" int a0_rng_normal(unsigned long long seed, unsigned int stream, unsigned long long offset, float stdv, float* out, long long n, void* stream_h) {
    A0_RNG_LAUNCH(a0_rng_normal_kernel, seed, stream, offset, stdv, out, n, (const long long*)nullptr, 0);
}
extern "C" int a0_rng_normal_ctrl(unsigned long long seed, unsigned int stream, unsigned long long offset, float stdv, float* out, long long n,
                                  const long long* ctrl, int ctrl_idx, void* stream_h) {
    if (ctrl_idx < 0 || ctrl_idx >= A0_CTRL_WORDS) return a0_fail(A0_EINVAL, "a0_rng_normal_ctrl: bad ctrl index");
    A0_RNG_LAUNCH(a0_rng_normal_kernel, seed, stream, offset, stdv, out, n, ctrl, ctrl_idx);
}
