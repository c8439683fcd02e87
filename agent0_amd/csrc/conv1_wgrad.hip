// conv1 weight gradient on the bf16 matrix pipe, one workgroup per observation (84 x 84 x 4 geometry).
//
// dW1[n][c,kh,kw] = (1/255) * sum_{oh,ow} d1[oh,ow,n] * byte[c][4oh+kh][4ow+kw]   — autograd's backward-weight of the first conv of
// ConvEncoder (reference agent0/deepq/model.py:93-101 with the /255 of agent.py:132 in front; called from agent.py:153-155).
// As an implicit GEMM this is M = 32, N = 256, K = 204 800 with a u8 im2col gather: the slowest GEMM of the update (45 TFLOP/s).
// Here the byte operand makes the bf16 pipe exact, as in the forward kernel (encoder_fused.hip): bytes are exact in bf16, d1 is
// split exactly into three bf16 terms (8 + 8 + 8 mantissa bits), every product is exact in fp32, and three
// v_mfma_f32_16x16x32_bf16 per 32 positions accumulate in fp32.  The 1/255 is applied once, to the sum.
//
// Layout tricks (all in LDS, rebuilt per observation):
//   * the observation as 64 stride-4 PHASE PLANES  Xp[c][kh%4][kw%4][21][24] (bf16): for a fixed tap (c,kh,kw) consecutive output
//     columns ow are consecutive elements, so a B fragment (8 consecutive positions of one tap) is one aligned 20-byte read
//     plus a per-lane byte shift (v_alignbyte) for taps with kw >= 4;
//   * positions are indexed m' = 24*oh + ow (rows padded from 20 to 24 with zeros in the d1 planes), so groups of 8 never cross a
//     row and BOTH operand addresses are linear in the group index: +16 bytes per group;
//   * d1 transposed into T[term][n][m'] (pitch 488: conflict-free 16-byte fragment reads).
// Every wave owns all 32 output channels x 32 taps (four 16x16 accumulators) for the whole launch; a workgroup adds up its
// observations in registers and writes one slab [32][256] + bias[32]; the slabs meet in a0_reduce_slabs_kernel (deterministic).
#include "a0_internal.h"

#include <cstdint>

typedef __bf16 a0w_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t a0w_u32x4 __attribute__((ext_vector_type(4)));
typedef float a0w_acc4 __attribute__((ext_vector_type(4)));

constexpr int A0W_THREADS = 512;
constexpr int A0W_XP_ROW = 24, A0W_XP_PLANE = 21 * 24;          // elements
constexpr int A0W_XP_ELEMS = 64 * A0W_XP_PLANE;                 // 32 256 bf16
constexpr int A0W_T_PITCH = 488, A0W_T_TERM = 32 * A0W_T_PITCH; // elements
constexpr int A0W_GROUPS = 60;                                  // 20 rows x 3 groups of 8 positions
constexpr int A0W_STEPS = 15;                                   // 32 positions per MFMA step

struct a0_c1w_args {
    const uint8_t* frames; const int* slot; long long sample_stride; int chan_off;
    const float* d1;        // [B][400][32], ReLU-masked
    float* slabs;           // [gridDim.x][32*256 + 32]
    int B;
};

A0_D uint32_t a0w_trunc(float f) { return __float_as_uint(f) >> 16; }
A0_D float a0w_up(uint32_t h) { return __uint_as_float(h << 16); }

__global__ __launch_bounds__(A0W_THREADS) void a0_conv1_wgrad_fused_kernel(a0_c1w_args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* Xp = (uint16_t*)smem;
    uint16_t* T = Xp + A0W_XP_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    // zero once: plane columns 21..23 / d1 pads (ow >= 20, pitch tail) are never written afterwards and must stay finite / zero
    for (int i = tid; i < (A0W_XP_ELEMS + 3 * A0W_T_TERM) / 2; i += A0W_THREADS) ((uint32_t*)smem)[i] = 0u;
    // this lane's two taps (k = (2*wave + t)*16 + r16): plane base (aligned part) and byte shift
    int bbase[2], bshift[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = (2 * wave + t) * 16 + r16;
        const int c = k >> 6, kh = (k >> 3) & 7, kw = k & 7;
        const int plane = (c * 4 + (kh & 3)) * 4 + (kw & 3);
        bbase[t] = plane * A0W_XP_PLANE + (kh >> 2) * A0W_XP_ROW;        // + 8*G; the tap's column offset kw>>2 is the byte shift
        bshift[t] = 2 * (kw >> 2);
    }
    a0w_acc4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[i][t] = a0w_acc4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                                    // bias partial of column n = tid % 32 over this thread's position groups
    __syncthreads();
    // raw global data of one observation, held in registers: requested for observation b + gridDim.x BEFORE the MFMA loop of
    // observation b, so the HBM latency of the next observation hides behind the matrix work of the current one
    constexpr int NDW = 4 * 84 * 21, TR = (NDW + A0W_THREADS - 1) / A0W_THREADS;      // 7056 dwords of pixels, 14 per thread
    constexpr int GT = (A0W_GROUPS + A0W_THREADS / 32 - 1) / (A0W_THREADS / 32);      // 4 position groups per thread
    uint32_t w[TR];
    float v[GT][8];
    auto load_raw = [&](int b) {
        const long long s = P.slot ? (long long)P.slot[b] : (long long)b;
        const uint32_t* src = (const uint32_t*)(P.frames + s * P.sample_stride + P.chan_off);
#pragma unroll
        for (int j = 0; j < TR; ++j) { const int i = tid + j * A0W_THREADS; w[j] = src[i < NDW ? i : NDW - 1]; }
        const float* g = P.d1 + (long long)b * 400 * 32 + (tid & 31);
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            int G = (tid >> 5) + t * (A0W_THREADS / 32);
            G = G < A0W_GROUPS ? G : A0W_GROUPS - 1;
            const int oh = G / 3, ow0 = 8 * (G - 3 * oh);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int m = oh * 20 + ow0 + j; v[t][j] = g[(m < 400 ? m : 399) * 32]; }       // pads are zeroed at use
        }
    };
    load_raw(blockIdx.x);
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        // ---- observation -> phase planes: one dword = 4 horizontally adjacent pixels = the four kw%4 planes at one (row, col)
#pragma unroll
        for (int j = 0; j < TR; ++j) {
            int i = tid + j * A0W_THREADS;
            i = i < NDW ? i : NDW - 1;             // threads past the end redo the last dword (same value, same address)
            const int c = i / (84 * 21), rem = i - c * (84 * 21), y = rem / 21, xc = rem - y * 21;
            uint16_t* d = Xp + ((c * 4 + (y & 3)) * 4) * A0W_XP_PLANE + (y >> 2) * A0W_XP_ROW + xc;
            d[0 * A0W_XP_PLANE] = (uint16_t)(__float_as_uint((float)(w[j] & 0xffu)) >> 16);
            d[1 * A0W_XP_PLANE] = (uint16_t)(__float_as_uint((float)((w[j] >> 8) & 0xffu)) >> 16);
            d[2 * A0W_XP_PLANE] = (uint16_t)(__float_as_uint((float)((w[j] >> 16) & 0xffu)) >> 16);
            d[3 * A0W_XP_PLANE] = (uint16_t)(__float_as_uint((float)(w[j] >> 24)) >> 16);
        }
        // ---- d1 -> three exact bf16 terms, transposed: item (n, G) = 8 positions of one row for one channel
        {
            const int n = tid & 31;
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int G = (tid >> 5) + t * (A0W_THREADS / 32);
                if (G < A0W_GROUPS) {
                    const int oh = G / 3, ow0 = 8 * (G - 3 * oh);
                    uint32_t t0[4], t1[4], t2[4];
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        uint32_t h[2], m[2], l[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float x = (ow0 + j + e < 20) ? v[t][j + e] : 0.f;
                            bsum += x;
                            h[e] = a0w_trunc(x);
                            const float r1 = x - a0w_up(h[e]);
                            m[e] = a0w_trunc(r1);
                            l[e] = a0w_trunc(r1 - a0w_up(m[e]));
                        }
                        t0[j >> 1] = h[0] | (h[1] << 16); t1[j >> 1] = m[0] | (m[1] << 16); t2[j >> 1] = l[0] | (l[1] << 16);
                    }
                    uint16_t* d = T + n * A0W_T_PITCH + 8 * G;
                    *(uint4*)(d) = uint4{t0[0], t0[1], t0[2], t0[3]};
                    *(uint4*)(d + A0W_T_TERM) = uint4{t1[0], t1[1], t1[2], t1[3]};
                    *(uint4*)(d + 2 * A0W_T_TERM) = uint4{t2[0], t2[1], t2[2], t2[3]};
                }
            }
        }
        __syncthreads();
        if (b + (int)gridDim.x < P.B) load_raw(b + gridDim.x);
        // ---- 15 MFMA steps of 32 positions: A = d1 terms (2 channel blocks), B = two taps blocks; lane group q owns position group 4*st + q
        {
            const uint16_t* ap = T + r16 * A0W_T_PITCH + 8 * q;
            const uint16_t* bp0 = Xp + bbase[0] + 8 * q;
            const uint16_t* bp1 = Xp + bbase[1] + 8 * q;
#pragma unroll 3
            for (int st = 0; st < A0W_STEPS; ++st) {
                a0w_u32x4 a[2][3], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) {
                        const uint4 x = *(const uint4*)(ap + s3 * A0W_T_TERM + i * 16 * A0W_T_PITCH + 32 * st);
                        a[i][s3] = a0w_u32x4{x.x, x.y, x.z, x.w};
                    }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint16_t* bp = (t == 0 ? bp0 : bp1) + 32 * st;
                    const uint4 lo = *(const uint4*)bp;
                    const uint32_t hi = *(const uint32_t*)(bp + 8);
                    const int sh = bshift[t];
                    bf[t] = a0w_u32x4{__builtin_amdgcn_alignbyte(lo.y, lo.x, sh), __builtin_amdgcn_alignbyte(lo.z, lo.y, sh),
                                      __builtin_amdgcn_alignbyte(lo.w, lo.z, sh), __builtin_amdgcn_alignbyte(hi, lo.w, sh)};
                }
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(a0w_bf16x8, a[i][s3]), __builtin_bit_cast(a0w_bf16x8, bf[t]), acc[i][t], 0, 0, 0);
            }
        }
        __syncthreads();          // the planes are rebuilt for the next observation
    }
    // ---- slab: dW (x 1/255) in the packed [32][256] layout, then the bias row sums (16 partials per channel, fixed order)
    float* out = P.slabs + (long long)blockIdx.x * (32 * 256 + 32);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = (2 * wave + t) * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(i * 16 + 4 * q + r) * 256 + k] = acc[i][t][r] / 255.0f;
        }
    float* red = (float*)smem;
    red[tid] = bsum;
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < A0W_THREADS / 32; ++j) s += red[j * 32 + tid];
        out[32 * 256 + tid] = s;
    }
}

// slabs: gridDim.x x (32*256 + 32) floats; returns the number of slabs written (0 = shape not supported, nothing launched)
int a0_conv1_wgrad_fused_launch(const a0_frames_arg* f, int C, int H, int W, int B, const float* d1, float* slabs, hipStream_t st) {
    if (C != 4 || H != 84 || W != 84 || B < 1 || (f->sample_stride & 3) || (f->chan_off & 3) || (((uintptr_t)f->frames) & 3)) return 0;
    a0_c1w_args P;
    P.frames = f->frames; P.slot = f->slot; P.sample_stride = f->sample_stride; P.chan_off = f->chan_off;
    P.d1 = d1; P.slabs = slabs; P.B = B;
    const int grid = B < 256 ? B : 256;
    const size_t lds = (size_t)(A0W_XP_ELEMS + 3 * A0W_T_TERM) * 2;
    static bool configured = false;
    if (!configured) {
        A0_HIP_THROW(hipFuncSetAttribute((const void*)a0_conv1_wgrad_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(a0_conv1_wgrad_fused_kernel, dim3(grid), dim3(A0W_THREADS), lds, st, P);
    A0_HIP_THROW(hipGetLastError());
    return grid;
}
