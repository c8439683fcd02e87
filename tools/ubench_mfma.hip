// Diagnostics: sustained fp32 MFMA rate on this GPU (operands in registers, no memory).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC> __global__ void k16(float* out, int iters, float a0, float b0) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void k32(float* out, int iters, float a0, float b0) {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> double run(F f, const char* name, int blocks, int threads, double flop_per_iter_per_wave, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(out, blocks, threads, 10);
    hipDeviceSynchronize();
    hipEventRecord(s); f(out, blocks, threads, iters); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double waves = (double)blocks * threads / 64;
    double tf = waves * iters * flop_per_iter_per_wave / (ms * 1e-3) / 1e12;
    printf("%-34s blocks %4d x %4d thr: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, threads, ms, tf);
    hipFree(out); return tf;
}
int main() {
    const int it = 20000;
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k16<8>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "16x16x4 8acc 1 wave/SIMD", 256, 256, 8 * 2048.0, it);
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k16<8>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "16x16x4 8acc 2 waves/SIMD", 256, 512, 8 * 2048.0, it);
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k16<2>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "16x16x4 2acc 1 wave/SIMD", 256, 256, 2 * 2048.0, it);
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k16<1>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "16x16x4 1acc 1 wave/SIMD", 256, 256, 1 * 2048.0, it);
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "32x32x2 2acc 1 wave/SIMD", 256, 256, 2 * 4096.0, it);
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k32<1>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "32x32x2 1acc 1 wave/SIMD", 256, 256, 1 * 4096.0, it);
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "32x32x2 2acc 2 waves/SIMD", 256, 512, 2 * 4096.0, it);
    run([](float* o, int b, int t, int n) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(t), 0, 0, o, n, 1.f, 2.f); }, "32x32x2 2acc 162 blocks", 162, 256, 2 * 4096.0, it);
    return 0;
}
