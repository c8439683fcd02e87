// Fused Nature-CNN encoder forward: one workgroup per observation, all three convolutions back to back with the
// activations resident in LDS (gfx950).
//
// Replaces ConvEncoder.forward (reference agent0/deepq/model.py:93-105) plus the uint8 -> fp32 /255 in front of it
// (agent.py:27, agent.py:129-134) for the shapes the actor-learner loop uses.  Why: as three separate implicit GEMMs the
// actor's 256-observation forward launches 162-800 workgroups per layer and is latency-bound (profiles/r01); here every
// CU owns one observation (28 KB u8 -> LDS once), reads its im2col operands straight out of LDS with computed addresses
// (no staging pass, no HBM round trip for act1/act2) and only streams the shared weights (k-major copies, L2-resident)
// through a double-buffered 32-deep LDS tile.  The MFMA is v_mfma_f32_16x16x4_f32: 16-row tiles fit M = 400 / 81 / 49
// rows with little padding and split evenly over the eight waves.  conv2 / conv3 are the same k-ascending fp32 fmaf chains as in
// igemm.h.  conv1 feeds the RAW byte values to the MFMA and folds the reference's x/255 into the weight copy (w/255): its inner
// loop is instruction-issue bound and the exact three-instruction division per operand element cost more than the MFMA itself;
// sum_k x_k*fl(w_k/255) and sum_k fl(x_k/255)*w_k carry the same two roundings per term, so the result agrees with the unfused
// path (and the reference) to fp32 rounding — tests/test_gpu_engine.py::test_fused_encoder_matches_unfused, rtol 2e-6.
#include "a0_internal.h"
#include "net_tables.h"
#include "operands.h"

#include <cstdlib>

typedef float a0_acc4 __attribute__((ext_vector_type(4)));

struct a0_fused_args {
    const uint8_t* frames; const int* slot; long long sample_stride; int chan_off;
    const float *wt1, *wt2, *wt3;        // k-major weights [K][N]
    const float *b1, *b2, *b3;
    float *act1, *act2, *act3;           // act1/act2 optional (needed only when a backward pass follows)
    int B;
    int C, H, W, H1, W1, H2, W2, H3, W3;
    int off_act1, off_act2, off_bs;      // float offsets of the LDS regions behind the u8 observation
    int stage_mask;                      // diagnostics only (A0_FUSED_STAGES env): bit i enables conv(i+1); 7 = everything
};

// Eight waves per workgroup = two per SIMD: while one wave waits for its LDS operands the other keeps the matrix pipe busy.
constexpr int A0_FUSED_WAVES = 8;
constexpr int A0_FUSED_THREADS = 64 * A0_FUSED_WAVES;
constexpr int A0_P1 = 33;   // LDS pixel pitch of act1 (32 channels + 1: 2*33 = 66 = 2 mod 32 -> conflict-free stride-2 row reads)
constexpr int A0_P2 = 65;   // LDS pixel pitch of act2 (64 channels + 1)

// ---- A-operand fetchers: element (row m, k = 32*kt + 4*s + q) of the im2col matrix, read directly from LDS.
// address = row(m) + tile_off(kt) + step_off(s) + q: the tile part is added once per 32-deep tile, the step part is a
// compile-time constant that folds into the DS instruction's immediate offset (WC > 0: width known at compile time).
template <int WC> struct AF1 {   // conv1 8x8/4 over the u8 observation [C][H][W]; k = c*64 + kh*8 + kw
    const uint8_t* obs; int HW, W, W1;
    A0_D int width() const { return WC > 0 ? WC : W; }
    A0_D int row(int m) const { const int oh = m / W1, ow = m - oh * W1; return (4 * oh) * width() + 4 * ow; }
    A0_D int tile_off(int kt) const { return (kt >> 1) * HW + 4 * (kt & 1) * width(); }
    A0_D int step_off(int s) const { return (s >> 1) * width() + 4 * (s & 1); }
    // raw byte value 0..255: the 1/255 of the reference's normalisation (agent.py:27,132) is folded into the k-major weight copy
    A0_D float fetch(int addr) const { return (float)obs[addr]; }
};
struct AF2 {   // conv2 4x4/2 over act1 [H1][W1][P1]; k = (kh*4 + kw)*32 + c
    const float* act; int W1, W2;
    A0_D int row(int m) const { const int oh = m / W2, ow = m - oh * W2; return ((2 * oh) * W1 + 2 * ow) * A0_P1; }
    A0_D int tile_off(int kt) const { return ((kt >> 2) * W1 + (kt & 3)) * A0_P1; }
    A0_D int step_off(int s) const { return 4 * s; }
    A0_D float fetch(int addr) const { return act[addr]; }
};
struct AF3 {   // conv3 3x3/1 over act2 [H2][W2][P2]; k = (kh*3 + kw)*64 + c
    const float* act; int W2, W3;
    A0_D int row(int m) const { const int oh = m / W3, ow = m - oh * W3; return (oh * W2 + ow) * A0_P2; }
    A0_D int tile_off(int kt) const { const int kk = kt >> 1; return ((kk / 3) * W2 + (kk % 3)) * A0_P2 + (kt & 1) * 32; }
    A0_D int step_off(int s) const { return 4 * s; }
    A0_D float fetch(int addr) const { return act[addr]; }
};

// One convolution: C[M x N] = relu(A[M x K] * Wt[K x N] + bias), A read through AF.  Waves: WN along N, 4/WN along M; a wave
// owns MBW 16-row blocks (interleaved) x NBW 16-column blocks.  (8 waves: conv1 2 x 4, conv2 / conv3 4 x 2.)
template <int N, int WN, int MBW, class AF>
A0_D void a0_conv_stage(const AF& af, int M, int K, const float* __restrict__ wt, const float* __restrict__ bias, float* Bs,
                        float* out_lds, int out_pitch, float* __restrict__ out_glb) {
    constexpr int NB = N / 16, NBW = NB / WN, WMG = A0_FUSED_WAVES / WN, NS = N + 16;
    constexpr int WF4 = N * 32 / 4;             // float4 of one weight tile
    constexpr int WREG = (WF4 + A0_FUSED_THREADS - 1) / A0_FUSED_THREADS;
    static_assert(NBW >= 1, "tile shape");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the per-block skips below become scalar branches
    const int wn = wave % WN, wmg = wave / WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int MB = (M + 15) >> 4, KT = K >> 5;

    int rows[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int m = (wmg + i * WMG) * 16 + r16;
        rows[i] = af.row(m < M ? m : 0) + q;
    }
    a0_acc4 acc[MBW][NBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = a0_acc4{0.f, 0.f, 0.f, 0.f};

    // weight tile kt: rows k = 32*kt .. +31 of Wt, N floats each, copied as float4 into Bs[buf][k][NS]
    a0_f4 wreg[WREG];
    auto wload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < WREG; ++j) {
            const int f = tid + A0_FUSED_THREADS * j;
            if (f < WF4) wreg[j] = *(const a0_f4*)(wt + (long long)(32 * kt + f / (N / 4)) * N + 4 * (f % (N / 4)));
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WREG; ++j) {
            const int f = tid + A0_FUSED_THREADS * j;
            if (f < WF4) *(a0_f4*)&Bs[buf * 32 * NS + (f / (N / 4)) * NS + 4 * (f % (N / 4))] = wreg[j];
        }
    };
    wload(0);
    wstore(0);
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) wload(kt + 1);
        const float* bb = Bs + buf * 32 * NS + q * NS + wn * (NBW * 16) + r16;
        // operands of k-step s+1 are read from LDS while the MFMAs of step s run; no branch inside (16-row blocks beyond M
        // recompute row 0 and are never stored), so the compiler is free to keep all MBW + NBW reads in flight
        float a[2][MBW], b[2][NBW];
        int rk[MBW];
        const int toff = af.tile_off(kt);
#pragma unroll
        for (int i = 0; i < MBW; ++i) rk[i] = rows[i] + toff;
#pragma unroll
        for (int j = 0; j < NBW; ++j) b[0][j] = bb[j * 16];
#pragma unroll
        for (int i = 0; i < MBW; ++i) a[0][i] = af.fetch(rk[i] + af.step_off(0));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < 8) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) b[nxt][j] = bb[4 * (s + 1) * NS + j * 16];
#pragma unroll
                for (int i = 0; i < MBW; ++i) a[nxt][i] = af.fetch(rk[i] + af.step_off(s + 1));
            }
#pragma unroll
            for (int i = 0; i < MBW; ++i)
#pragma unroll
                for (int j = 0; j < NBW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) wstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // C/D layout of v_mfma_f32_16x16x4_f32: column = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int mb = wmg + i * WMG;
        if (mb < MB) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const int n = (wn * NBW + j) * 16 + r16;
                const float bv = bias[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb * 16 + 4 * q + r;
                    if (m < M) {
                        float v = acc[i][j][r] + bv;
                        v = (v < 0.f) ? 0.f : v;
                        if (out_lds) out_lds[m * out_pitch + n] = v;
                        if (out_glb) out_glb[(long long)m * N + n] = v;
                    }
                }
            }
        }
    }
    __syncthreads();
}

template <int MBW1, int MBW2, int MBW3, int WC>
__global__ __launch_bounds__(A0_FUSED_THREADS) void a0_encoder_fused_kernel(a0_fused_args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* obs = smem;
    float* fl = (float*)smem;
    float* act1 = fl + P.off_act1;
    float* act2 = fl + P.off_act2;
    float* Bs = fl + P.off_bs;
    const int obs_bytes = P.C * P.H * P.W;
    const int M1 = P.H1 * P.W1, M2 = P.H2 * P.W2, M3 = P.H3 * P.W3;
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        // ---- observation -> LDS (16 B per lane)
        const long long s = P.slot ? (long long)P.slot[b] : (long long)b;
        const uint4* src = (const uint4*)(P.frames + s * P.sample_stride + P.chan_off);
        for (int i = threadIdx.x; i < (obs_bytes >> 4); i += A0_FUSED_THREADS) ((uint4*)obs)[i] = src[i];
        __syncthreads();
        AF1<WC> f1{obs, P.H * P.W, P.W, P.W1};
        if (P.stage_mask & 1) a0_conv_stage<32, 2, MBW1, AF1<WC>>(f1, M1, P.C * 64, P.wt1, P.b1, Bs, act1, A0_P1, P.act1 ? P.act1 + (long long)b * M1 * 32 : nullptr);
        AF2 f2{act1, P.W1, P.W2};
        if (P.stage_mask & 2) a0_conv_stage<64, 4, MBW2, AF2>(f2, M2, 512, P.wt2, P.b2, Bs, act2, A0_P2, P.act2 ? P.act2 + (long long)b * M2 * 64 : nullptr);
        AF3 f3{act2, P.W2, P.W3};
        if (P.stage_mask & 4) a0_conv_stage<64, 4, MBW3, AF3>(f3, M3, 576, P.wt3, P.b3, Bs, nullptr, 0, P.act3 + (long long)b * M3 * 64);
    }
}

// ---- k-major weight copies: wt1 [K1][32] (pre-divided by 255), wt2 [512][64], wt3 [576][64] from the packed [N][K] blocks
__global__ void a0_conv_wt_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3, float* __restrict__ wt, int K1) {
    const int n1 = 32 * K1, n2 = 64 * 512, n3 = 64 * 576;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) { const int k = i / 32, n = i % 32; wt[i] = w1[n * K1 + k] / 255.0f; return; }   // conv1 reads raw bytes: fold the /255 here
    i -= n1;
    if (i < n2) { const int k = i / 64, n = i % 64; wt[n1 + i] = w2[n * 512 + k]; return; }
    i -= n2;
    if (i < n3) { const int k = i / 64, n = i % 64; wt[n1 + n2 + i] = w3[n * 576 + k]; }
}

extern "C" long long a0_net_conv_wt_floats(int C) { return 32LL * C * 64 + 64LL * 512 + 64LL * 576; }

extern "C" int a0_net_conv_wt_refresh(const a0_encoder_weights* w, int C, float* wt, void* stream) {
    if (!w || !w->w1 || !w->w2 || !w->w3 || !wt || C < 1) return a0_fail(A0_EINVAL, "a0_net_conv_wt_refresh: bad argument");
    const long long n = a0_net_conv_wt_floats(C);
    hipLaunchKernelGGL(a0_conv_wt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w->w1, w->w2, w->w3, wt, C * 64);
    return a0_fail_hip((int)hipGetLastError(), "a0_net_conv_wt_refresh");
}

static bool a0_fused_layout(int C, int H, int W, a0_fused_args& P, size_t& lds_bytes) {
    a0_net_core n;
    if (!a0_net_core_init(n, C, H, W)) return false;
    const int M1 = n.H1 * n.W1, M2 = n.H2 * n.W2, M3 = n.H3 * n.W3;
    if (M1 > 7 * 4 * 16 || M2 > 4 * 2 * 16 || M3 > 2 * 2 * 16) return false;        // tile capacity of the three stages
    const int obs_bytes = C * H * W;
    if (obs_bytes % 16) return false;
    P.C = C; P.H = H; P.W = W; P.H1 = n.H1; P.W1 = n.W1; P.H2 = n.H2; P.W2 = n.W2; P.H3 = n.H3; P.W3 = n.W3;
    P.off_act1 = obs_bytes / 4;
    P.off_act2 = P.off_act1 + ((M1 * A0_P1 + 3) & ~3);
    P.off_bs = P.off_act2 + ((M2 * A0_P2 + 3) & ~3);
    lds_bytes = (size_t)(P.off_bs + 2 * 32 * (64 + 16)) * 4;
    return lds_bytes <= 160 * 1024;
}

extern "C" int a0_net_encoder_fused_supported(int C, int H, int W) {
    a0_fused_args P; size_t lds;
    return a0_fused_layout(C, H, W, P, lds) ? 1 : 0;
}

// Same contract as a0_net_encoder_fwd (bit-identical outputs); wt from a0_net_conv_wt_refresh; act1 / act2 may be NULL.
extern "C" int a0_net_encoder_fwd_fused(int C, int H, int W, const float* wt, const a0_encoder_weights* w, const a0_frames_arg* f, int B,
                                        float* act1, float* act2, float* act3, void* stream) {
    A0_TRY
    if (!wt || !w || !w->b1 || !w->b2 || !w->b3 || !f || !f->frames || !act3 || B < 1) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused: bad argument");
    a0_fused_args P;
    size_t lds = 0;
    if (!a0_fused_layout(C, H, W, P, lds)) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused: observation shape not supported by the fused kernel");
    if ((f->sample_stride % 16) || (f->chan_off % 16) || (((uintptr_t)f->frames) % 16)) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused: frames must be 16-byte aligned");
    P.frames = f->frames; P.slot = f->slot; P.sample_stride = f->sample_stride; P.chan_off = f->chan_off;
    P.wt1 = wt; P.wt2 = wt + 32LL * C * 64; P.wt3 = P.wt2 + 64LL * 512;
    P.b1 = w->b1; P.b2 = w->b2; P.b3 = w->b3;
    P.act1 = act1; P.act2 = act2; P.act3 = act3; P.B = B;
    static const int stage_mask = getenv("A0_FUSED_STAGES") ? atoi(getenv("A0_FUSED_STAGES")) : 7;
    P.stage_mask = stage_mask;
    // 16-row blocks per wave: conv1 ceil(MB1/4), conv2 ceil(MB2/2), conv3 ceil(MB3/2); exact for 84x84, generous otherwise
    const int mb1 = (P.H1 * P.W1 + 15) / 16, mb2 = (P.H2 * P.W2 + 15) / 16, mb3 = (P.H3 * P.W3 + 15) / 16;
    const bool standard = ((mb1 + 3) / 4 == 7) && ((mb2 + 1) / 2 == 3) && ((mb3 + 1) / 2 == 2) && W == 84;
    static size_t configured[2] = {0, 0};
    const void* fn = standard ? (const void*)a0_encoder_fused_kernel<7, 3, 2, 84> : (const void*)a0_encoder_fused_kernel<7, 4, 2, 0>;
    if (lds > configured[standard]) {
        A0_HIP_THROW(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[standard] = lds;
    }
    const bool probed = a0_probe_start(A0_TAG_ENCODER_FUSED, (hipStream_t)stream);
    if (standard) hipLaunchKernelGGL((a0_encoder_fused_kernel<7, 3, 2, 84>), dim3(B), dim3(A0_FUSED_THREADS), lds, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((a0_encoder_fused_kernel<7, 4, 2, 0>), dim3(B), dim3(A0_FUSED_THREADS), lds, (hipStream_t)stream, P);
    if (probed) {   // algorithmic FLOP of the three convolutions: 2 * (M1*32*K1 + M2*64*512 + M3*64*576) per observation
        const double per_obs = 2.0 * ((double)P.H1 * P.W1 * 32 * (P.C * 64) + (double)P.H2 * P.W2 * 64 * 512 + (double)P.H3 * P.W3 * 64 * 576);
        a0_probe_stop((hipStream_t)stream, per_obs * B);
    }
    A0_HIP_THROW(hipGetLastError());
    return A0_OK;
    A0_CATCH
}
