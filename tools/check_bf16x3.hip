// Diagnostics: v_mfma_f32_16x16x32_bf16 with an exact 3-way bf16 split of fp32 weights and u8 (exact in bf16) activations.
// Verifies the fragment layouts and compares against an fp64 reference and the fp32 fmaf chain.   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* A /*[16][32] bf16*/, const uint16_t* B /*[3][16 n][32 k] bf16*/, float* C /*[16][16]*/) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    uint4 av = *(const uint4*)(A + r * 32 + 8 * g);
    f4 acc = {0, 0, 0, 0};
    for (int s = 0; s < 3; ++s) {
        uint4 bv = *(const uint4*)(B + (s * 16 + r) * 32 + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&av, *(bf16x8*)&bv, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];   // row = 4*(lane>>4)+reg, column = lane&15
}
static uint16_t trunc_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    std::vector<uint16_t> A(16 * 32), B(3 * 16 * 32);
    std::vector<float> W(16 * 32), X(16 * 32);
    srand(1);
    int bad_split = 0;
    for (int i = 0; i < 16 * 32; ++i) {
        X[i] = (float)(rand() % 256);
        A[i] = trunc_bf16(X[i]);
        float w = ((rand() / (float)RAND_MAX) - 0.5f) * 0.2f / 255.0f;
        W[i] = w;
        uint16_t h = trunc_bf16(w); float r1 = w - from_bf16(h);
        uint16_t m = trunc_bf16(r1); float r2 = r1 - from_bf16(m);
        uint16_t lo = trunc_bf16(r2);
        if (from_bf16(h) + from_bf16(m) + from_bf16(lo) != w || from_bf16(lo) != r2) ++bad_split;
        B[0 * 512 + i] = h; B[1 * 512 + i] = m; B[2 * 512 + i] = lo;
    }
    uint16_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    double e_mfma = 0, e_chain = 0, scale = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double ref = 0; float chain = 0.f;
            for (int kk = 0; kk < 32; ++kk) { ref += (double)X[m * 32 + kk] * (double)W[n * 32 + kk]; chain = fmaf(X[m * 32 + kk], W[n * 32 + kk], chain); }
            e_mfma = fmax(e_mfma, fabs(C[m * 16 + n] - ref)); e_chain = fmax(e_chain, fabs(chain - ref)); scale = fmax(scale, fabs(ref));
        }
    printf("split exact: %s; max|mfma_bf16x3 - fp64| = %.3e, max|fp32 chain - fp64| = %.3e, scale %.3e\n", bad_split ? "NO" : "yes", e_mfma, e_chain, scale);
    return 0;
}
