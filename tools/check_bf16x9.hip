// Diagnostics: fp32 x fp32 products on v_mfma_f32_16x16x32_bf16 with BOTH operands split exactly into three bf16 terms (nine MFMAs per
// 32 k, every partial product exact in fp32, fp32 accumulation).  Compares against fp64 and the fp32 fmaf chain.   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int KK = 512;
__global__ void k(const uint16_t* A /*[3][16][KK]*/, const uint16_t* B /*[3][16][KK]*/, float* C) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < KK; k0 += 32)
        for (int ta = 0; ta < 3; ++ta)
            for (int tb = 0; tb < 3; ++tb) {
                uint4 av = *(const uint4*)(A + (ta * 16 + r) * KK + k0 + 8 * g), bv = *(const uint4*)(B + (tb * 16 + r) * KK + k0 + 8 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&av, *(bf16x8*)&bv, acc, 0, 0, 0);
            }
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}
static uint16_t tr(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float up(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split(float w, uint16_t* o) { o[0] = tr(w); float r1 = w - up(o[0]); o[1] = tr(r1); o[2] = tr(r1 - up(o[1])); }
int main() {
    std::vector<uint16_t> A(3 * 16 * KK), B(3 * 16 * KK);
    std::vector<float> X(16 * KK), W(16 * KK);
    srand(3);
    int bad = 0;
    for (int i = 0; i < 16 * KK; ++i) {
        float x = (rand() % 3 == 0) ? 0.f : (rand() / (float)RAND_MAX) * 2.0f;      // post-ReLU-like activations
        float w = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
        X[i] = x; W[i] = w;
        uint16_t s[3]; split(x, s); if (up(s[0]) + up(s[1]) + up(s[2]) != x) ++bad; for (int t = 0; t < 3; ++t) A[t * 16 * KK + i] = s[t];
        split(w, s); if (up(s[0]) + up(s[1]) + up(s[2]) != w) ++bad; for (int t = 0; t < 3; ++t) B[t * 16 * KK + i] = s[t];
    }
    uint16_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    double e9 = 0, ec = 0, sc = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double ref = 0; float ch = 0.f;
            for (int kk = 0; kk < KK; ++kk) { ref += (double)X[m * KK + kk] * (double)W[n * KK + kk]; ch = fmaf(X[m * KK + kk], W[n * KK + kk], ch); }
            e9 = fmax(e9, fabs(C[m * 16 + n] - ref)); ec = fmax(ec, fabs(ch - ref)); sc = fmax(sc, fabs(ref));
        }
    printf("K=%d split exact: %s; max|bf16x9 - fp64| = %.3e, max|fp32 chain - fp64| = %.3e, scale %.3e\n", KK, bad ? "NO" : "yes", e9, ec, sc);
    return 0;
}
