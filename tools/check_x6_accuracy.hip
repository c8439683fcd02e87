// Accuracy record for the six-product form of the split-operand kernels (a0_x9_products, include/agent0_hip.h): fp32 x fp32 dot products on
// v_mfma_f32_16x16x32_bf16 with both operands split exactly into three bf16 terms (truncating splits, as the kernels make them), forming all nine cross
// products or only the six with term orders i + j <= 2, against fp64 — beside the sequential fp32 fmaf chain (what an fp32-input kernel computes) and a
// pairwise fp32 sum of rounded products.  Reduction lengths of the path: 256 (conv1 as fp32), 512 (conv2), 576 (conv3), 3136 (fc1), 8192 / 32768 (fc1 weight
// gradients over B x N quantile rows); operand distributions: post-ReLU activations x small weights (forward), signed gradients x weights (data gradients),
// unit gaussians.   hipcc --offload-arch=gfx950 -O2 tools/check_x6_accuracy.hip -o /tmp/x6 && /tmp/x6 > profiles/r06_x6_accuracy.txt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// one wave per 16 x 16 tile: A [tiles][3][16][K], B [tiles][3][16][K] (bf16 term planes), C [tiles][16][16]
__global__ void dot_kernel(const uint16_t* A, const uint16_t* B, float* C, int K, int maxord) {
    const int t = blockIdx.x, l = threadIdx.x, r = l & 15, g = l >> 4;
    const uint16_t* a = A + (size_t)t * 3 * 16 * K;
    const uint16_t* b = B + (size_t)t * 3 * 16 * K;
    f4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32)
        for (int ta = 0; ta < 3; ++ta)
            for (int tb = 0; tb < 3; ++tb) {
                if (ta + tb > maxord) continue;
                uint4 av = *(const uint4*)(a + (size_t)(ta * 16 + r) * K + k0 + 8 * g), bv = *(const uint4*)(b + (size_t)(tb * 16 + r) * K + k0 + 8 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8*)&av, *(bf16x8*)&bv, acc, 0, 0, 0);
            }
    for (int i = 0; i < 4; ++i) C[(size_t)t * 256 + (4 * g + i) * 16 + r] = acc[i];
}
// the same dot products on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32: what the fp32-input kernels of igemm.h / A0_GEMM=fp32 issue): X [tiles][16][K], W [tiles][16][K] fp32
__global__ void dot_f32_kernel(const float* X, const float* W, float* C, int K) {
    const int t = blockIdx.x, l = threadIdx.x, r = l & 15, g = l >> 4;
    const float* x = X + (size_t)t * 16 * K;
    const float* w = W + (size_t)t * 16 * K;
    f4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[(size_t)r * K + k0 + g], w[(size_t)r * K + k0 + g], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[(size_t)t * 256 + (4 * g + i) * 16 + r] = acc[i];
}
static uint16_t tr(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float up(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split(float w, uint16_t* o) { o[0] = tr(w); float r1 = w - up(o[0]); o[1] = tr(r1); o[2] = tr(r1 - up(o[1])); }

struct Err { double mx = 0, sq = 0, bias = 0; long n = 0; void add(double e, double scale) { e /= scale; mx = fmax(mx, fabs(e)); sq += e * e; bias += e; ++n; } };

int main() {
    const int TILES = 48;
    const int Ks[] = {256, 512, 576, 3136, 8192, 32768};
    const char* dists[] = {"relu_x_weights", "signed_grad_x_weights", "unit_gaussians"};
    printf("# six- vs nine-product split-operand dot products against fp64 (v_mfma_f32_16x16x32_bf16, fp32 accumulation); errors relative to sum|a*b| of the dot product\n");
    printf("# %d tiles x 256 dot products per row; fmaf = sequential fp32 fmaf chain; pair = fp32 products summed pairwise in fp32\n", TILES);
    printf("# mfma32 = the same dot product on v_mfma_f32_16x16x4_f32, the fp32 matrix instruction (the fp32-input kernels this family replaced: igemm.h, A0_GEMM=fp32)\n");
    printf("%-22s %6s | %10s %10s %10s | %10s %10s %10s | %10s %10s %10s | %10s %10s %10s | %10s %s\n", "distribution", "K", "x9 max", "x9 rms", "x9 bias", "x6 max", "x6 rms", "x6 bias", "fmaf max",
           "fmaf rms", "fmaf bias", "mfma32 max", "mfma32 rms", "mfma32 bias", "pair rms", "x6 <= mfma32 (max and rms)");
    int all_ok = 1;
    double r69 = 0, r6f_lo = 1e9, r6f_hi = 0, r6m_lo = 1e9, r6m_hi = 0;
    for (int d = 0; d < 3; ++d)
        for (int K : Ks) {
            std::mt19937_64 rng(1234 + 17 * d + K);
            std::normal_distribution<float> nrm(0.f, 1.f);
            std::uniform_real_distribution<float> uni(0.f, 1.f);
            std::vector<float> X((size_t)TILES * 16 * K), W((size_t)TILES * 16 * K);
            std::vector<uint16_t> A((size_t)TILES * 3 * 16 * K), B((size_t)TILES * 3 * 16 * K);
            int bad = 0;
            for (int t = 0; t < TILES; ++t)
                for (int i = 0; i < 16 * K; ++i) {
                    float x, w;
                    if (d == 0) { x = uni(rng) < 0.4f ? 0.f : fabsf(nrm(rng)) * 1.3f; w = nrm(rng) * 0.03f; }
                    else if (d == 1) { x = nrm(rng) * 1e-3f * (uni(rng) < 0.5f ? 0.f : 1.f); w = nrm(rng) * 0.03f; }
                    else { x = nrm(rng); w = nrm(rng); }
                    X[(size_t)t * 16 * K + i] = x; W[(size_t)t * 16 * K + i] = w;
                    uint16_t s[3];
                    split(x, s); if (up(s[0]) + up(s[1]) + up(s[2]) != x) ++bad;
                    for (int q = 0; q < 3; ++q) A[((size_t)t * 3 + q) * 16 * K + i] = s[q];
                    split(w, s); if (up(s[0]) + up(s[1]) + up(s[2]) != w) ++bad;
                    for (int q = 0; q < 3; ++q) B[((size_t)t * 3 + q) * 16 * K + i] = s[q];
                }
            uint16_t *dA, *dB; float* dC;
            (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, (size_t)TILES * 1024);
            (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
            std::vector<float> C9((size_t)TILES * 256), C6((size_t)TILES * 256);
            hipLaunchKernelGGL(dot_kernel, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, K, 4);
            (void)hipMemcpy(C9.data(), dC, C9.size() * 4, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(dot_kernel, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, K, 2);
            (void)hipMemcpy(C6.data(), dC, C6.size() * 4, hipMemcpyDeviceToHost);
            (void)hipFree(dA); (void)hipFree(dB);
            std::vector<float> CM((size_t)TILES * 256);
            float *dX, *dW;
            (void)hipMalloc(&dX, X.size() * 4); (void)hipMalloc(&dW, W.size() * 4);
            (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(dot_f32_kernel, dim3(TILES), dim3(64), 0, 0, dX, dW, dC, K);
            (void)hipMemcpy(CM.data(), dC, CM.size() * 4, hipMemcpyDeviceToHost);
            (void)hipFree(dX); (void)hipFree(dW); (void)hipFree(dC);
            Err e9, e6, ef, ep, em;
            std::vector<float> tmp(K);
            for (int t = 0; t < TILES; ++t)
                for (int m = 0; m < 16; ++m)
                    for (int n = 0; n < 16; ++n) {
                        const float* x = &X[((size_t)t * 16 + m) * K];
                        const float* w = &W[((size_t)t * 16 + n) * K];
                        double ref = 0, sabs = 0; float ch = 0.f;
                        for (int k = 0; k < K; ++k) { const double p = (double)x[k] * (double)w[k]; ref += p; sabs += fabs(p); ch = fmaf(x[k], w[k], ch); tmp[k] = x[k] * w[k]; }
                        for (int len = K; len > 1;) { const int h = (len + 1) / 2; for (int k = 0; k + h < len; ++k) tmp[k] += tmp[k + h]; len = h; }
                        if (sabs == 0) continue;
                        e9.add(C9[(size_t)t * 256 + m * 16 + n] - ref, sabs); e6.add(C6[(size_t)t * 256 + m * 16 + n] - ref, sabs);
                        ef.add(ch - ref, sabs); ep.add(tmp[0] - ref, sabs); em.add(CM[(size_t)t * 256 + m * 16 + n] - ref, sabs);
                    }
            auto rms = [](const Err& e) { return sqrt(e.sq / (double)e.n); };
            auto bias = [](const Err& e) { return e.bias / (double)e.n; };
            const int ok = rms(e6) <= rms(em) && e6.mx <= em.mx;
            all_ok &= ok;
            r69 = fmax(r69, rms(e6) / rms(e9)); r6f_lo = fmin(r6f_lo, rms(e6) / rms(ef)); r6f_hi = fmax(r6f_hi, rms(e6) / rms(ef));
            r6m_lo = fmin(r6m_lo, rms(e6) / rms(em)); r6m_hi = fmax(r6m_hi, rms(e6) / rms(em));
            printf("%-22s %6d | %10.3e %10.3e %+10.2e | %10.3e %10.3e %+10.2e | %10.3e %10.3e %+10.2e | %10.3e %10.3e %+10.2e | %10.3e %s%s\n", dists[d], K, e9.mx, rms(e9), bias(e9), e6.mx, rms(e6),
                   bias(e6), ef.mx, rms(ef), bias(ef), em.mx, rms(em), bias(em), rms(ep), ok ? "yes" : "NO", bad ? "  (SPLIT NOT EXACT)" : "");
        }
    printf("# 2^-24 = %.3e.  six-product rms error / nine-product rms error: at most %.4f over all rows (the dropped cross terms do not show).\n", ldexp(1.0, -24), r69);
    printf("# six-product rms error / sequential scalar fmaf chain's: %.3f .. %.3f; / the fp32 matrix instruction's (v_mfma_f32_16x16x4_f32): %.3f .. %.3f.\n", r6f_lo, r6f_hi, r6m_lo, r6m_hi);
    printf("# six-product error no larger than the fp32 MATRIX-instruction chain's (max and rms) on every row: %s\n", all_ok ? "yes" : "NO");
    return 0;
}
