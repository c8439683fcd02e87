// Nature-CNN encoder + dense layers on gfx950: the HIP backend of net_impl.h plus the exported C entry points.
// Replaces the ATen -> cuDNN/cuBLAS dispatches behind reference agent0/deepq/model.py:93-101 (ConvEncoder),
// model.py:112-114 / 144-146 / 203-216 (dense layers of the heads) and their autograd backward (agent.py:153-155).
#include "igemm_x9.h"
#include "short_k_fwd.h"
#include "a0_internal.h"
#include "net_impl.h"

#include <vector>

// ------------------------------------------------------------------------------------------------ small kernels
// out[i] = sum_z slabs[z][i].  One workgroup per 32 consecutive outputs; its 8 row-groups stride over z and are combined in a
// fixed order through LDS, so the sum is deterministic and 8 slab rows are in flight per output column.
__global__ __launch_bounds__(256) void a0_reduce_slabs_kernel(const float* __restrict__ slabs, long long slab_stride, int nslab,
                                                               float* __restrict__ out, long long count) {
    __shared__ float red[8][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + c;
    float s = 0.f;
    if (i < count)
        for (int z = g; z < nslab; z += 8) s += slabs[(long long)z * slab_stride + i];
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j][c];
        out[i] = t;
    }
}

__global__ void a0_reduce_bias_act_kernel(const float* __restrict__ slabs, long long slab_stride, int nslab,
                                          const float* __restrict__ bias, float* __restrict__ out, int rows, int N, int relu) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long count = (long long)rows * N;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        float s = 0.f;
        // eight slabs requested before any is added (additions still in slab order): 32 dependent-looking loads in a row cost a short, wide reduction 10 us
        for (int z = 0; z < nslab; z += 8) {
            float t[8];
#pragma unroll
            for (int zz = 0; zz < 8; ++zz) t[zz] = (z + zz < nslab) ? slabs[(long long)(z + zz) * slab_stride + i] : 0.f;
#pragma unroll
            for (int zz = 0; zz < 8; ++zz)
                if (z + zz < nslab) s += t[zz];
        }
        s += bias[i % N];
        if (relu) s = (s < 0.f) ? 0.f : s;
        out[i] = s;
    }
}

// a0_reduce_bias_act_kernel for up to four layers of the same width in ONE launch (round 4: the three fc1 passes of a distributional update — online on s,
// online on s', target on s' — each leave their own split-K slabs; their reductions are independent and each is a 4.7 us launch of pure latency on
// its own).  Same arithmetic per element as the single-layer kernel: slabs added in slab order, eight requested at a time, then the bias, then the ReLU.
struct a0_rba_seg { const float* slabs; long long slab_stride; int nslab; const float* bias; float* out; int rows; };
struct a0_rba_multi_args { a0_rba_seg seg[4]; int n, N, relu; };
__global__ __launch_bounds__(256) void a0_reduce_bias_act_multi_kernel(a0_rba_multi_args A) {
    const int si = blockIdx.y;
    const a0_rba_seg S = A.seg[si];
    const int N4 = A.N >> 2;
    const long long count4 = (long long)S.rows * N4, st4 = S.slab_stride >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += stride) {
        a0_f4 s = a0_zero4();
        const a0_f4* p = (const a0_f4*)S.slabs + i;
        for (int z = 0; z < S.nslab; z += 8) {
            a0_f4 t[8];
#pragma unroll
            for (int zz = 0; zz < 8; ++zz) t[zz] = (z + zz < S.nslab) ? p[(long long)(z + zz) * st4] : a0_zero4();
#pragma unroll
            for (int zz = 0; zz < 8; ++zz)
                if (z + zz < S.nslab) { s.x += t[zz].x; s.y += t[zz].y; s.z += t[zz].z; s.w += t[zz].w; }
        }
        const a0_f4 b = ((const a0_f4*)S.bias)[i % N4];
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        if (A.relu) { s.x = s.x < 0.f ? 0.f : s.x; s.y = s.y < 0.f ? 0.f : s.y; s.z = s.z < 0.f ? 0.f : s.z; s.w = s.w < 0.f ? 0.f : s.w; }
        ((a0_f4*)S.out)[i] = s;
    }
}

// Up to four slab reductions in one launch: workgroup g serves 128 outputs of the segment its index falls into — 32 lanes x 16 bytes wide,
// eight row groups striding over the slabs and combined in a fixed order through LDS (deterministic); a segment whose base, stride or
// count is not a multiple of four floats takes the scalar path (32 outputs per workgroup).
struct a0_reduce_multi_args { a0_reduce_seg seg[8]; int first_block[9]; int vec[8]; };
__global__ __launch_bounds__(256) void a0_reduce_segments_kernel(a0_reduce_multi_args A) {
    __shared__ a0_f4 red4[8][33];
    int si = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) si += ((int)blockIdx.x >= A.first_block[k]) ? 1 : 0;
    const a0_reduce_seg S = A.seg[si];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long blk = (long long)((int)blockIdx.x - A.first_block[si]);
    if (A.vec[si]) {
        const long long i4 = blk * 32 + c;                    // float4 index
        const long long n4 = S.count >> 2, st4 = S.slab_stride >> 2;
        a0_f4 s = a0_zero4();
        if (i4 < n4) {
            const a0_f4* p = (const a0_f4*)S.slabs + i4;
            // four of this row group's slabs requested before any is added (same order of additions): the loads of a 72-slab segment overlap instead of queueing
            for (int z = g; z < S.nslab; z += 32) {
                a0_f4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (z + 8 * u < S.nslab) ? p[(long long)(z + 8 * u) * st4] : a0_zero4();
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (z + 8 * u < S.nslab) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
        }
        red4[g][c] = s;
        __syncthreads();
        if (g == 0 && i4 < n4) {
            a0_f4 t = a0_zero4();
#pragma unroll
            for (int j = 0; j < 8; ++j) { const a0_f4 v = red4[j][c]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            ((a0_f4*)S.out)[i4] = t;
        }
        return;
    }
    float* red = (float*)red4;                                 // [8][33] floats
    const long long i = blk * 32 + c;
    float s = 0.f;
    if (i < S.count)
        for (int z = g; z < S.nslab; z += 8) s += S.slabs[(long long)z * S.slab_stride + i];
    red[g * 33 + c] = s;
    __syncthreads();
    if (g == 0 && i < S.count) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j * 33 + c];
        S.out[i] = t;
    }
}

static inline int a0_grid_for(long long count, int block = 256, int cap = 2048) {
    long long g = (count + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------ profiler probe
// bench.py brackets every launch of ONE tagged GEMM with HIP events on the stream the kernel runs on, so that its
// average duration (and hence achieved FLOP/s against the fp32 MFMA roofline) is measured live inside the timed region.
struct a0_probe_t {
    int tag = 0;
    std::vector<hipEvent_t> ev;     // pairs: start, stop
    size_t used = 0;
    double flops = 0.0;
};
static a0_probe_t g_probe;

// used by encoder_fused.hip: returns true if this launch is being probed (start event recorded)
bool a0_probe_start(int tag, hipStream_t st) {
    if (g_probe.tag == 0 || g_probe.tag != tag || g_probe.used + 2 > g_probe.ev.size()) return false;
    return hipEventRecord(g_probe.ev[g_probe.used], st) == hipSuccess;
}
void a0_probe_stop(hipStream_t st, double flops) {
    (void)hipEventRecord(g_probe.ev[g_probe.used + 1], st);
    g_probe.used += 2;
    g_probe.flops += flops;
}

bool a0_probe_events(int tag, hipEvent_t* start, hipEvent_t* stop) {
    if (g_probe.tag == 0 || g_probe.tag != tag || g_probe.used + 2 > g_probe.ev.size()) return false;
    *start = g_probe.ev[g_probe.used]; *stop = g_probe.ev[g_probe.used + 1];
    return true;
}
void a0_probe_commit(double flops) { g_probe.used += 2; g_probe.flops += flops; }

extern "C" int a0_probe_begin(int tag, int max_launches) {
    A0_TRY
    for (hipEvent_t e : g_probe.ev) (void)hipEventDestroy(e);
    g_probe.ev.clear();
    g_probe.used = 0; g_probe.flops = 0.0; g_probe.tag = 0;
    if (tag <= 0 || max_launches <= 0) return A0_OK;
    g_probe.ev.resize(2 * (size_t)max_launches);
    for (auto& e : g_probe.ev) A0_HIP_THROW(hipEventCreate(&e));
    g_probe.tag = tag;
    return A0_OK;
    A0_CATCH
}

// host_out3: [launches, total milliseconds, total algorithmic FLOP]; synchronises on the recorded events
extern "C" int a0_probe_end(double* host_out3) {
    A0_TRY
    if (!host_out3) return a0_fail(A0_EINVAL, "a0_probe_end: null");
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g_probe.used; i += 2) {
        A0_HIP_THROW(hipEventSynchronize(g_probe.ev[i + 1]));
        float t = 0.f;
        A0_HIP_THROW(hipEventElapsedTime(&t, g_probe.ev[i], g_probe.ev[i + 1]));
        ms += t;
    }
    host_out3[0] = (double)(g_probe.used / 2); host_out3[1] = ms; host_out3[2] = g_probe.flops;
    g_probe.tag = 0;
    return A0_OK;
    A0_CATCH
}

static int g_gemm_x9 = (getenv("A0_GEMM") && std::string(getenv("A0_GEMM")) == "fp32") ? 0 : 1;
static const long long g_x9_big_min = getenv("A0_X9_BIG_MIN") ? atoll(getenv("A0_X9_BIG_MIN")) : 256;     // tuning aid
extern "C" int a0_gemm_mode(int mode) {
    const int prev = g_gemm_x9;
    if (mode >= 0) g_gemm_x9 = mode ? 1 : 0;
    return prev;
}

// Cross products of the split-operand kernels (igemm_x9.h, encoder_fused.hip): 6 (default) or 9 (strict: every partial product of the fp32 chain)
static int g_x9_products = (getenv("A0_X9_PRODUCTS") && atoi(getenv("A0_X9_PRODUCTS")) == 9) ? 9 : 6;
int a0_x9_products_now() { return g_x9_products; }
extern "C" int a0_x9_products(int n) {
    const int prev = g_x9_products;
    if (n == 6 || n == 9) g_x9_products = n;
    return prev;
}

static const long long g_x9_huge_min = getenv("A0_X9_HUGE_MIN") ? atoll(getenv("A0_X9_HUGE_MIN")) : 250;   // tuning aid (a huge value switches the 256 x 128 tile off)
template <class OP> struct a0_is_mat { static constexpr bool value = false; };
template <> struct a0_is_mat<OpMatKC> { static constexpr bool value = true; };
template <> struct a0_is_mat<OpMatXC> { static constexpr bool value = true; };
// im2col-gathered B operands (conv weight gradients): their address arithmetic already fills the issue slots the splits would need
template <class OP> struct a0_is_gather { static constexpr bool value = false; };
template <> struct a0_is_gather<OpActXC> { static constexpr bool value = true; };

struct a0_hip_backend {
    hipStream_t st;
    int tag = 0;
    template <class OA, class OB, class EP, int WM, int WN, int MT, int NT>
    void igemm(const typename OA::Params& pa, const typename OB::Params& pb, const typename EP::Params& pe, int X, int Y, int K, int splits) {
        // the probe's event pair travels IN the split-operand launches (the dispatch's own timestamps, as for the fused kernels); the fp32-chain kernel is bracketed
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool probe = a0_probe_events(tag, &e0, &e1);
        bool carried = false;
        // fp32 operands: the split-operand kernel on the bf16 matrix pipe (igemm_x9.h); A0_GEMM=fp32 keeps the fmaf-chain kernel
        const bool x9 = g_gemm_x9 != 0;
        // the fmaf-chain kernel has four waves: an eight-wave tile shape falls back to the same tile on four (twice the blocks per wave along N)
        constexpr bool eight = WM * WN == 8;
        constexpr int FWM = eight ? WM : WM, FWN = eight ? WN / 2 : WN, FMT = MT, FNT = eight ? NT * 2 : NT;
        static_assert(!eight || (WN % 2) == 0, "eight-wave shapes: even wave count along N");
        if constexpr (a0_x9_ok<OA>::value && a0_x9_ok<OB>::value) {
            // large problems: 128 x 128 tiles on eight waves (two per SIMD: the splits of one wave issue in the shadow of the other's MFMAs)
            const int sp = splits < 1 ? 1 : splits;
            const long long big = (long long)((X + 127) / 128) * ((Y + 127) / 128) * sp;
            // weight gradients (x-contiguous A): the short, heavily split reductions of a 512-row batch are staging-bound, and there
            // the fp32 kernel's plain 16-byte LDS commits win (conv2/conv3 36 vs 44 us, fc1 17.7 vs 19.4 us); the split kernel takes
            // them only when each split is at least 512 deep (the quantile networks' B*N-row reductions: fc1 1.12 vs 1.34 ms)
            constexpr bool wgrad_family = OA::MODE == A0_XC;
            const bool deep = K / sp >= 512;
            const bool large = X >= 128 && Y >= 128 && big >= g_x9_big_min && (deep || !wgrad_family);
            // very large dense problems (the quantile networks' 32 768-row layers): 256 x 128 tiles of 16 k on eight waves, 64 x 64 per wave — a quarter less
            // staging and a third fewer LDS fragment reads per MFMA than the 128 x 128 x 32 tile, the same 36 MFMAs per wave between barriers; taken when
            // there is about a full round of such tiles (the actor's 8 192 rows would fill half the CUs and keep the smaller tile)
            constexpr bool mats = a0_is_mat<OA>::value && a0_is_mat<OB>::value;
            const long long huge = (long long)((X + 255) / 256) * ((Y + 127) / 128) * sp;
            // ... and when they fill their rounds of 256 workgroups about as well as the smaller tiles fill theirs (the tile is worth ~5 %; relaxing this for
            // fqf's 16 384-row data gradient, 1600 tiles = 6.25 rounds, or its 15 872-row pass, 248 tiles, gained nothing: 384.7 vs 385.3 us, 347.9 vs 342.8 us)
            const double fill_huge = (double)huge / (256.0 * (double)((huge + 255) / 256)), fill_big = (double)big / (256.0 * (double)((big + 255) / 256));
            if constexpr (mats) {
                if (x9 && X >= 256 && Y >= 128 && huge >= g_x9_huge_min && fill_huge >= 0.93 * fill_big && (deep || !wgrad_family)) {
                    A0_HIP_THROW((a0_igemm_x9_launch<OA, OB, EP, 4, 2, 2, 2, 1>(st, pa, pb, pe, X, Y, K, splits, e0, e1)));
                    if (probe) a0_probe_commit(2.0 * (double)X * (double)Y * (double)K);
                    return;
                }
            }
            // (round 6, six products: the same 128 x 128 tile on FOUR waves of 64 x 64 — a third fewer LDS fragment reads per MFMA, one wave per SIMD — measured 139.9 vs
            // 137.7 us on the actor's 8 192-row fc1, iqn 86.4 vs 85.9 ms: not kept, profiles/r06_experiments.md)
            // (round 6: TWO four-wave workgroups per CU on 128 x 64 x 16 tiles — independent barriers, so the SIMD partners are not in lockstep — measured 211 vs 177 us at
            // 8 192 rows, 319 vs 322 at 16 384: not kept, profiles/r06_experiments.md)
            if (x9 && large) { A0_HIP_THROW((a0_igemm_x9_launch<OA, OB, EP, 4, 2, 1, 2>(st, pa, pb, pe, X, Y, K, splits, e0, e1))); carried = true; }
            else if (x9 && (!wgrad_family || (deep && !a0_is_gather<OB>::value))) { A0_HIP_THROW((a0_igemm_x9_launch<OA, OB, EP, WM, WN, MT, NT>(st, pa, pb, pe, X, Y, K, splits, e0, e1))); carried = true; }
            else {
                if (probe) A0_HIP_THROW(hipEventRecord(e0, st));
                A0_HIP_THROW((a0_igemm_launch<OA, OB, EP, FWM, FWN, FMT, FNT>(st, pa, pb, pe, X, Y, K, splits)));
            }
        } else {
            if (probe) A0_HIP_THROW(hipEventRecord(e0, st));
            A0_HIP_THROW((a0_igemm_launch<OA, OB, EP, FWM, FWN, FMT, FNT>(st, pa, pb, pe, X, Y, K, splits)));
        }
        if (probe) {
            if (!carried) A0_HIP_THROW(hipEventRecord(e1, st));
            a0_probe_commit(2.0 * (double)X * (double)Y * (double)K);
        }
    }
    // (not part of the timing probe's dense_fwd family: bound by its output stream, not by the matrix pipe)
    void short_k_fwd(const float* X, int ldx, const float* W, const float* b, const float* M, int group, float* Y, float* Y2, int R, int N, int relu) {
        A0_HIP_THROW(a0_short_k_fwd_launch(st, X, ldx, W, b, M, group, Y, Y2, R, N, relu));
    }
    int conv1_wgrad_fused(const a0_net_core& n, const a0_frames_arg& f, int B, const float* d1, float* slabs) {
        static const bool off = getenv("A0_NO_CONV1_WGRAD_FUSED") != nullptr;
        if (off || !slabs) return 0;
        const bool probe = g_probe.tag != 0 && g_probe.tag == tag && g_probe.used + 2 <= g_probe.ev.size();
        if (probe) A0_HIP_THROW(hipEventRecord(g_probe.ev[g_probe.used], st));
        const int g = a0_conv1_wgrad_fused_launch(&f, n.C, n.H, n.W, B, d1, slabs, st);
        if (probe && g > 0) {
            A0_HIP_THROW(hipEventRecord(g_probe.ev[g_probe.used + 1], st));
            g_probe.used += 2;
            g_probe.flops += 2.0 * 32.0 * (double)n.K1 * (double)B * n.H1 * n.W1;
        }
        return g;
    }
    bool conv23_wgrad_fused(const a0_net_core& n, int B, const float* act1, const float* act2, const float* d2, const float* d3, float* slab2, float* slab3) {
        const bool probe = g_probe.tag != 0 && g_probe.tag == A0_TAG_CONV2_WGRAD && g_probe.used + 2 <= g_probe.ev.size();
        if (probe) A0_HIP_THROW(hipEventRecord(g_probe.ev[g_probe.used], st));
        const int ran = a0_conv23_wgrad_fused_launch(n, B, act1, act2, d2, d3, slab2, slab3, st);
        if (probe && ran) {
            A0_HIP_THROW(hipEventRecord(g_probe.ev[g_probe.used + 1], st));
            g_probe.used += 2;
            g_probe.flops += 2.0 * 64.0 * ((double)n.K2 * n.H2 * n.W2 + (double)n.K3 * n.H3 * n.W3) * B;
        }
        return ran != 0;
    }
    void reduce_slabs(const float* slabs, long long slab_stride, int nslab, float* out, long long count) {
        hipLaunchKernelGGL(a0_reduce_slabs_kernel, dim3((unsigned)((count + 31) / 32)), dim3(256), 0, st, slabs, slab_stride, nslab, out, count);
        A0_HIP_THROW(hipGetLastError());
    }
    void reduce_segments(const a0_reduce_seg* segs, int nseg) {
        if (nseg < 1 || nseg > 8) throw std::runtime_error("reduce_segments: 1..8 segments");
        a0_reduce_multi_args A;
        int blocks = 0;
        for (int k = 0; k < 8; ++k) {
            A.first_block[k] = blocks;
            A.vec[k] = 0;
            if (k < nseg) {
                A.seg[k] = segs[k];
                A.vec[k] = ((segs[k].count | segs[k].slab_stride) % 4 == 0) && ((((uintptr_t)segs[k].slabs) | ((uintptr_t)segs[k].out)) % 16 == 0);
                blocks += A.vec[k] ? (int)((segs[k].count / 4 + 31) / 32) : (int)((segs[k].count + 31) / 32);
            } else {
                A.seg[k] = a0_reduce_seg{nullptr, 0, 0, nullptr, 0};
            }
        }
        A.first_block[8] = blocks;
        for (int k = nseg; k < 8; ++k) A.first_block[k] = 0x7fffffff;        // unused segments are never selected
        hipLaunchKernelGGL(a0_reduce_segments_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A);
        A0_HIP_THROW(hipGetLastError());
    }
    void reduce_bias_act(const float* slabs, long long slab_stride, int nslab, const float* bias, float* out, int rows, int N, int relu) {
        hipLaunchKernelGGL(a0_reduce_bias_act_kernel, dim3(a0_grid_for((long long)rows * N)), dim3(256), 0, st, slabs, slab_stride, nslab, bias, out, rows, N, relu);
        A0_HIP_THROW(hipGetLastError());
    }
};

// ------------------------------------------------------------------------------------------------ net object
struct a0_net {
    a0_net_core core;
    std::vector<void*> owned;
    ~a0_net() { for (void* p : owned) (void)hipFree(p); }
    template <class T> const T* upload(const std::vector<T>& v) {
        void* d = nullptr;
        A0_HIP_THROW(hipMalloc(&d, v.size() * sizeof(T)));
        owned.push_back(d);
        A0_HIP_THROW(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return (const T*)d;
    }
};

extern "C" int a0_net_create(const a0_net_desc* d, a0_net** out) {
    A0_TRY
    if (!d || !out) return a0_fail(A0_EINVAL, "a0_net_create: null argument");
    a0_net* n = new a0_net();
    if (!a0_net_core_init(n->core, d->C, d->H, d->W)) { delete n; return a0_fail(A0_EINVAL, "a0_net_create: observation too small for the 8/4, 4/2, 3/1 conv stack"); }
    a0_net_tables t;
    a0_net_build_tables(n->core, t);
    try {
        n->core.ktab1 = n->upload(t.ktab1); n->core.ktab2 = n->upload(t.ktab2); n->core.ktab3 = n->upload(t.ktab3);
        n->core.ktab_d3 = n->upload(t.ktab_d3); n->core.ktab_d2 = n->upload(t.ktab_d2);
        n->core.wtab_d3 = n->upload(t.wtab_d3);
        for (int i = 0; i < 4; ++i) n->core.wtab_d2[i] = n->upload(t.wtab_d2[i]);
    } catch (...) { delete n; throw; }
    *out = n;
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_net_destroy(a0_net* n) { delete n; return A0_OK; }

extern "C" int a0_net_geometry(const a0_net* n, int* out8) {
    if (!n || !out8) return a0_fail(A0_EINVAL, "a0_net_geometry: null");
    const a0_net_core& c = n->core;
    out8[0] = c.H1; out8[1] = c.W1; out8[2] = c.H2; out8[3] = c.W2; out8[4] = c.H3; out8[5] = c.W3; out8[6] = c.feat; out8[7] = c.K1;
    return A0_OK;
}

extern "C" int a0_net_encoder_fwd(const a0_net* n, const a0_encoder_weights* w, const a0_frames_arg* f, int B,
                                  float* act1, float* act2, float* act3, void* stream) {
    A0_TRY
    if (!n || !w || !f || !f->frames || !act1 || !act2 || !act3 || B < 1) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd: bad argument");
    a0_hip_backend bk{(hipStream_t)stream};
    a0_encoder_fwd_impl(bk, n->core, *w, *f, B, act1, act2, act3);
    return A0_OK;
    A0_CATCH
}

extern "C" long long a0_dense_fwd_scratch(int R, int N, int K) { return a0_dense_fwd_scratch_impl(R, N, K); }

extern "C" int a0_dense_fwd(const float* X, int ldx, const float* W, const float* b, float* Y, int R, int N, int K, int relu,
                            float* scratch, void* stream) {
    A0_TRY
    if (!X || !W || !b || !Y || R < 1 || N < 4 || (N & 3) || (K & 3) || (ldx & 3)) return a0_fail(A0_EINVAL, "a0_dense_fwd: bad shape (N, K, ldx must be multiples of 4)");
    if (a0_dense_fwd_scratch_impl(R, N, K) > 0 && !scratch) return a0_fail(A0_EINVAL, "a0_dense_fwd: split-K needs scratch");
    a0_hip_backend bk{(hipStream_t)stream};
    a0_dense_fwd_impl(bk, X, ldx, W, b, Y, R, N, K, relu, scratch);
    return A0_OK;
    A0_CATCH
}

// ---- weights as term planes (round 6): W [N][K] fp32 -> the three bf16 term planes in OpPlanesKC's layout (igemm_x9.h), once per change of W; a0_dense_fwd_wplanes
// is a0_dense_fwd for unsplit shapes (a0_dense_fwd_scratch == 0) with whole k tiles (K % 32 == 0) reading them: the same exact terms, so the same result bit for bit.
__global__ void a0_split_planes_kernel(const a0_f4* __restrict__ W, uint32_t* __restrict__ planes, long long groups) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups) return;
    a0_x9_piece pc;
    pc.v = W[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) pc.split(e);
    a0_u32x2g hi, mid, lo;
    pc.pack(hi, mid, lo);
    a0_u32x2g* o = (a0_u32x2g*)(planes + i * 6);
    o[0] = hi; o[1] = mid; o[2] = lo;
}
extern "C" long long a0_weight_planes_words(int N, int K) { return (long long)N * (K / 4) * 6; }
extern "C" int a0_split_planes(const float* W, unsigned int* planes, int N, int K, void* stream) {
    if (!W || !planes || N < 1 || K < 4 || (K & 3)) return a0_fail(A0_EINVAL, "a0_split_planes: bad argument (K a multiple of 4)");
    const long long groups = (long long)N * (K / 4);
    hipLaunchKernelGGL(a0_split_planes_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const a0_f4*)W, planes, groups);
    return a0_fail_hip((int)hipGetLastError(), "a0_split_planes");
}
extern "C" int a0_dense_fwd_wplanes_ok(int R, int N, int K) {
    return (R >= 2048 && N >= 128 && !(N & 3) && !(K & 31) && a0_fwd_splits((R + 127) / 128, (N + 63) / 64, K) == 1 && g_gemm_x9 != 0) ? 1 : 0;
}
extern "C" int a0_dense_fwd_wplanes(const float* X, int ldx, const unsigned int* Wplanes, const float* b, float* Y, int R, int N, int K, int relu, void* stream) {
    A0_TRY
    if (!X || !Wplanes || !b || !Y || (ldx & 3) || !a0_dense_fwd_wplanes_ok(R, N, K)) return a0_fail(A0_EINVAL, "a0_dense_fwd_wplanes: shapes a0_dense_fwd_wplanes_ok accepts");
    hipStream_t st = (hipStream_t)stream;
    const a0_mat_src a{X, ldx};
    const a0_planes_src bw{Wplanes, K / 4};
    const EpiBiasAct::Params e{Y, b, N, relu};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool probe = a0_probe_events(A0_TAG_DENSE_FWD, &e0, &e1);
    // the tile choice of a0_hip_backend::igemm for unsplit matrix operands
    const long long huge = (long long)((R + 255) / 256) * ((N + 127) / 128), big = (long long)((R + 127) / 128) * ((N + 127) / 128);
    const double fill_huge = (double)huge / (256.0 * (double)((huge + 255) / 256)), fill_big = (double)big / (256.0 * (double)((big + 255) / 256));
    if (R >= 256 && huge >= g_x9_huge_min && fill_huge >= 0.93 * fill_big) A0_HIP_THROW((a0_igemm_x9_launch<OpMatKC, OpPlanesKC, EpiBiasAct, 4, 2, 2, 2, 1>(st, a, bw, e, R, N, K, 1, e0, e1)));
    else A0_HIP_THROW((a0_igemm_x9_launch<OpMatKC, OpPlanesKC, EpiBiasAct, 4, 2, 1, 2>(st, a, bw, e, R, N, K, 1, e0, e1)));
    if (probe) a0_probe_commit(2.0 * (double)R * (double)N * (double)K);
    return A0_OK;
    A0_CATCH
}

// Y[r][:] = act(X[r] W^T + b) * M[r / group][:] — the quantile networks' embedding times the state features in the GEMM's epilogue
// (reference model.py:244-247: relu(cosine_emb(cos(pi i tau))) * features), for passes that are not differentiated: the embedding never
// goes to HBM and the Hadamard pass disappears.  Only for shapes whose forward GEMM is not split (a0_dense_fwd_scratch == 0).
extern "C" int a0_dense_fwd_mul(const float* X, int ldx, const float* W, const float* b, const float* M, int group, float* Y, int R, int N, int K, int relu, void* stream) {
    A0_TRY
    if (!X || !W || !b || !M || !Y || R < 1 || group < 1 || (N & 3) || (K & 3) || (ldx & 3)) return a0_fail(A0_EINVAL, "a0_dense_fwd_mul: bad shape (N, K, ldx must be multiples of 4)");
    if (a0_fwd_splits((R + 127) / 128, (N + 63) / 64, K) != 1) return a0_fail(A0_EINVAL, "a0_dense_fwd_mul: this shape runs as a split GEMM; use a0_dense_fwd + a0_hadamard_fwd");
    a0_hip_backend bk{(hipStream_t)stream};
    bk.tag = A0_TAG_DENSE_FWD;
    if (a0_short_k_shape(R, N, K, ldx)) { bk.short_k_fwd(X, ldx, W, b, M, group, Y, nullptr, R, N, relu); return A0_OK; }
    a0_mat_src a{X, ldx};
    a0_mat_src bw{W, K};
    EpiBiasActMul::Params e{Y, b, N, relu, M, group, a0_udiv_magic((unsigned)group)};
    bk.tag = A0_TAG_DENSE_FWD;
    if (N <= 32) bk.template igemm<OpMatKC, OpMatKC, EpiBiasActMul, 4, 1, 1, 1>(a, bw, e, R, N, K, 1);
    else bk.template igemm<OpMatKC, OpMatKC, EpiBiasActMul, 4, 1, 1, 2>(a, bw, e, R, N, K, 1);
    return A0_OK;
    A0_CATCH
}

// The same for a pass that IS differentiated: E = act(X W^T + b) is kept for the backward pass (a0_hadamard_bwd) and Y = E * M[r / group] is
// written beside it in the same launch (a0_dense_fwd + a0_hadamard_fwd without re-reading E).  Only for the shapes of the short-reduction
// kernel: a0_dense_fwd_mul_keep_ok tells.
extern "C" int a0_dense_fwd_mul_keep_ok(int R, int N, int K, int ldx) { return a0_short_k_shape(R, N, K, ldx) ? 1 : 0; }

extern "C" int a0_dense_fwd_mul_keep(const float* X, int ldx, const float* W, const float* b, const float* M, int group, float* E, float* Y, int R, int N, int K, int relu, void* stream) {
    A0_TRY
    if (!X || !W || !b || !M || !E || !Y || R < 1 || group < 1 || (N & 3)) return a0_fail(A0_EINVAL, "a0_dense_fwd_mul_keep: bad shape");
    if (!a0_short_k_shape(R, N, K, ldx)) return a0_fail(A0_EINVAL, "a0_dense_fwd_mul_keep: not a short-reduction shape (a0_dense_fwd_mul_keep_ok); use a0_dense_fwd + a0_hadamard_fwd");
    a0_hip_backend bk{(hipStream_t)stream};
    bk.tag = A0_TAG_DENSE_FWD;
    bk.short_k_fwd(X, ldx, W, b, M, group, E, Y, R, N, relu);
    return A0_OK;
    A0_CATCH
}

// The split-K GEMM of a0_dense_fwd WITHOUT its reduction: slab z = X W^T over the z-th k range, [R][N] each at stride R*N; the caller's
// next kernel sums the slabs (a0_dqn_head_loss_slabs, a0_actor_qhead).  Returns the slab count.
// Tile of the split-K fc1 GEMM whose slabs a consumer kernel finishes (actor tail, DQN head + loss).  A0_FC1_VARIANT (tuning aid):
// 0 = 128 x 64, waves 4 x 1, 1 = 128 x 64, waves 2 x 2, 2 = 64 x 64, 3 = 64 x 128, 4 = 128 x 32, 5 / 6 = 128 x 64 / 64 x 128 on EIGHT waves (two per
// SIMD: one wave's staging work in the shadow of the other's MFMAs); unset: 64 x 64 up to 256 rows (the actor's
// batch: 8 slabs instead of 16 for the tail kernel to sum, GEMM + tail 19.9 vs 22.3 us at 256 rows; the eight-wave tiles need 16 slabs
// there and lose), 128 x 64 on eight waves above (512 rows: GEMM + tail 26.5 vs 27.8 us on four waves, whole B = 512 update 380 vs 385 us) — tools/ubench_actor_tail.py, profiles/r02_encoder_experiments.md.
static inline int a0_fc1_variant(int R) {
    static const int v = getenv("A0_FC1_VARIANT") ? atoi(getenv("A0_FC1_VARIANT")) : -1;
    return v >= 0 ? v : (R <= 256 ? 2 : 5);
}
static inline int a0_fc1_splits(int R, int N, int K) {
    const int v = a0_fc1_variant(R);
    const int bx = (v == 2 || v == 3 || v == 6) ? 64 : 128, by = (v == 3 || v == 6) ? 128 : (v == 4 ? 32 : 64);
    static const int wg = getenv("A0_FC1_WGS") ? atoi(getenv("A0_FC1_WGS")) : 256;      // target workgroup count
    const int blocks = ((R + bx - 1) / bx) * ((N + by - 1) / by);
    if (v == 0 && wg == 256) return a0_fwd_splits((R + 127) / 128, (N + 63) / 64, K);
    int splits = blocks < wg ? wg / blocks : 1;
    const int maxs = (K / 32) / 2;
    if (splits > maxs) splits = maxs;
    if (splits > 64) splits = 64;
    return splits < 1 ? 1 : splits;
}
template <class BK>
static void a0_fc1_partial_launch(BK& bk, const a0_mat_src& a, const a0_mat_src& bw, const EpiSlab::Params& ep, int R, int N, int K, int splits) {
    switch (a0_fc1_variant(R)) {
        case 1: bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 2, 2, 2, 1>(a, bw, ep, R, N, K, splits); break;
        case 4: bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 4, 1, 1, 1>(a, bw, ep, R, N, K, splits); break;
        case 2: bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 2, 2, 1, 1>(a, bw, ep, R, N, K, splits); break;
        case 3: bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 2, 2, 1, 2>(a, bw, ep, R, N, K, splits); break;
        case 5: bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 4, 2, 1, 1>(a, bw, ep, R, N, K, splits); break;      // 128 x 64, eight waves
        case 6: bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 2, 4, 1, 1>(a, bw, ep, R, N, K, splits); break;      // 64 x 128, eight waves
        default: bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 4, 1, 1, 2>(a, bw, ep, R, N, K, splits);
    }
}

extern "C" int a0_reduce_bias_act_multi(int n, const float* const* slabs, const long long* slab_stride, const int* nslab, const float* const* bias, float* const* out,
                                       const int* rows, int N, int relu, void* stream) {
    if (n < 1 || n > 4 || !slabs || !slab_stride || !nslab || !bias || !out || !rows || N < 4 || (N & 3)) return a0_fail(A0_EINVAL, "a0_reduce_bias_act_multi: bad argument (1..4 layers, N a multiple of 4)");
    a0_rba_multi_args A;
    A.n = n; A.N = N; A.relu = relu;
    int maxrows = 0;
    for (int i = 0; i < n; ++i) {
        if (!slabs[i] || !bias[i] || !out[i] || rows[i] < 1 || nslab[i] < 1 || (slab_stride[i] & 3) || slab_stride[i] < (long long)rows[i] * N ||
            ((((uintptr_t)slabs[i]) | ((uintptr_t)bias[i]) | ((uintptr_t)out[i])) & 15))
            return a0_fail(A0_EINVAL, "a0_reduce_bias_act_multi: bad layer (16-byte aligned buffers, slab stride a multiple of 4 floats)");
        A.seg[i] = a0_rba_seg{slabs[i], slab_stride[i], nslab[i], bias[i], out[i], rows[i]};
        if (rows[i] > maxrows) maxrows = rows[i];
    }
    for (int i = n; i < 4; ++i) A.seg[i] = A.seg[0];
    hipLaunchKernelGGL(a0_reduce_bias_act_multi_kernel, dim3(a0_grid_for((long long)maxrows * (N >> 2)), n), dim3(256), 0, (hipStream_t)stream, A);
    return a0_fail_hip((int)hipGetLastError(), "a0_reduce_bias_act_multi");
}

extern "C" int a0_dense_fwd_partial_slabs(int R, int N, int K) {
    const int s = N <= 32 ? a0_fwd_splits((R + 127) / 128, (N + 63) / 64, K) : a0_fc1_splits(R, N, K);
    return s < 1 ? 1 : s;
}

extern "C" int a0_dense_fwd_partial(const float* X, int ldx, const float* W, int R, int N, int K, float* slabs, void* stream) {
    A0_TRY
    if (!X || !W || !slabs || R < 1 || (N & 3) || (K & 3) || (ldx & 3)) return a0_fail(A0_EINVAL, "a0_dense_fwd_partial: bad shape (N, K, ldx must be multiples of 4)");
    a0_hip_backend bk{(hipStream_t)stream};
    const int splits = a0_dense_fwd_partial_slabs(R, N, K);
    a0_mat_src a{X, ldx};
    a0_mat_src bw{W, K};
    EpiSlab::Params ep{slabs, (long long)R * N, N};
    bk.tag = A0_TAG_DENSE_FWD;
    if (N <= 32) bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 4, 1, 1, 1>(a, bw, ep, R, N, K, splits);
    else a0_fc1_partial_launch(bk, a, bw, ep, R, N, K, splits);
    return A0_OK;
    A0_CATCH
}

// n = 2 or 3 passes of one dense layer shape — X[i] [R][ldx] x W[i]^T [N][K] -> slabs[i] [splits][R][N] — as ONE launch (a0_igemm_x9_group_kernel): the chip is
// filled with a half / a third of the splits a single pass needs, so every workgroup's k loop is that much longer and the fixed cost of a launch is paid once.  The
// slabs differ from a0_dense_fwd_partial's (fewer, deeper partial sums: another association of the same fp32 additions); a0_dense_fwd_partial_multi_slabs tells the
// consumer how many there are.  Split-operand GEMM mode only.
static int a0_fwd_multi_splits(int n, int R, int N, int K) {
    const int blocks = n * ((R + 127) / 128) * ((N + 63) / 64);
    int splits = blocks < 256 ? 256 / blocks : 1;
    const int maxs = (K / 32) / 2;
    if (splits > maxs) splits = maxs;
    return splits < 1 ? 1 : splits;
}
extern "C" int a0_dense_fwd_partial_multi_ok(int n, int R, int N, int K) {
    static const bool off = getenv("A0_NO_FWD_MULTI") != nullptr;       // tuning aid
    return (!off && g_gemm_x9 != 0 && (n == 2 || n == 3) && R >= 257 && R <= 4096 && N > 32 && !(N & 3) && !(K & 3)) ? 1 : 0;
}
extern "C" int a0_dense_fwd_partial_multi_slabs(int n, int R, int N, int K) { return a0_fwd_multi_splits(n, R, N, K); }

extern "C" int a0_dense_fwd_partial_multi(int n, const float* const* X, int ldx, const float* const* W, int R, int N, int K, float* const* slabs, const long long* slab_stride,
                                          void* stream) {
    A0_TRY
    if (!X || !W || !slabs || (ldx & 3) || !a0_dense_fwd_partial_multi_ok(n, R, N, K)) return a0_fail(A0_EINVAL, "a0_dense_fwd_partial_multi: shapes a0_dense_fwd_partial_multi_ok accepts");
    a0_x9_group<OpMatKC, OpMatKC, EpiSlab> grp;
    for (int i = 0; i < 3; ++i) {
        const int j = i < n ? i : 0;
        if (!X[j] || !W[j] || !slabs[j]) return a0_fail(A0_EINVAL, "a0_dense_fwd_partial_multi: null operand");
        grp.pa[i] = a0_mat_src{X[j], ldx};
        grp.pb[i] = a0_mat_src{W[j], K};
        // slab_stride (optional): floats between a pass's consecutive slabs — two passes may interleave their rows in one buffer [splits][2R][N] (stride 2 R N, bases R N apart)
        const long long stride = slab_stride ? slab_stride[j] : (long long)R * N;
        if (stride < (long long)R * N || (stride & 3)) return a0_fail(A0_EINVAL, "a0_dense_fwd_partial_multi: slab stride");
        grp.pe[i] = EpiSlab::Params{slabs[j], stride, N};
    }
    const int splits = a0_fwd_multi_splits(n, R, N, K);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool probe = a0_probe_events(A0_TAG_DENSE_FWD, &e0, &e1);
    A0_HIP_THROW((a0_igemm_x9_group_launch<OpMatKC, OpMatKC, EpiSlab, 4, 2, 1, 1>((hipStream_t)stream, n, grp, R, N, K, splits, e0, e1)));      // 128 x 64 tiles, eight waves
    if (probe) a0_probe_commit(2.0 * n * (double)R * (double)N * (double)K);
    return A0_OK;
    A0_CATCH
}

// ------------------------------------------------------------------------------------------------ fused actor tail (dqn / mdqn)
// Everything between the conv features and the chosen action of Actor.act (reference agent.py:25-39 with model.py:108-131 behind
// it), for scalar-valued heads: fc1 runs as the usual split-K implicit GEMM, but its slabs are consumed directly by ONE kernel that
// finishes fc1 (slab sum + bias + ReLU), evaluates the q head (A or A+1 rows of 512), applies the dueling combine, takes the first
// maximum and makes the epsilon-greedy draw from the actor's Philox streams.  One wave per environment; replaces six launches
// (reduce, head GEMM, reduce, dueling, select, egreedy) on the actor's critical path.
#include "actor_tail.h"
__global__ __launch_bounds__(256) void a0_actor_qhead_kernel(const float* __restrict__ slabs, long long slab_stride, int nslab, const float* __restrict__ b1,
                                                             const float* __restrict__ W2, const float* __restrict__ b2, int A, int dueling, int E,
                                                             unsigned long long seed, uint32_t stream_a, uint32_t stream_u, unsigned long long off_a,
                                                             unsigned long long off_u, float eps, const long long* __restrict__ ctrl,
                                                             const float* __restrict__ eps_ptr, int* __restrict__ action, float* __restrict__ qmax) {
    __shared__ float raw[4][64];
    extern __shared__ float w2s[];                       // the head's A(+1) rows of 512, staged once per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + wave;
    const int NQ = A + (dueling ? 1 : 0);
    for (int i = threadIdx.x; i < NQ * 128; i += 256) ((a0_f4*)w2s)[i] = ((const a0_f4*)W2)[i];
    __syncthreads();
    if (e >= E) return;
    if (ctrl) { off_a += (unsigned long long)ctrl[A0_CTRL_RNG_ACTION]; off_u += (unsigned long long)ctrl[A0_CTRL_RNG_UNIFORM]; }
    if (eps_ptr) eps = eps_ptr[0];
    int act = 0; float best = 0.f;
    a0_qhead_wave(slabs, slab_stride, nslab, b1, w2s, b2, A, dueling, e, lane, raw[wave], seed, stream_a, stream_u, off_a, off_u, eps, act, best);
    if (lane == 0) { action[e] = act; qmax[e] = best; }
}

__global__ __launch_bounds__(512) void a0_actor_qhead_env_kernel(a0_qenv_args P) {
    __shared__ float raw[64];
    __shared__ int s_chase_cell;
    extern __shared__ float w2s[];
    a0_actor_qhead_env_body(P, raw, w2s, &s_chase_cell);
}

extern "C" long long a0_actor_qhead_scratch(int E, int K) {
    const int splits = a0_fc1_splits(E, 512, K);
    return (long long)splits * E * 512;
}

extern "C" int a0_actor_qhead(const float* feat, int E, int K, const float* W1, const float* b1, const float* W2, const float* b2, int A, int dueling,
                              float* scratch, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                              unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax, void* stream) {
    A0_TRY
    if (!feat || !W1 || !b1 || !W2 || !b2 || !scratch || !action || !qmax || E < 1 || K < 4 || (K & 3) || A < 1 || A + (dueling ? 1 : 0) > 24)
        return a0_fail(A0_EINVAL, "a0_actor_qhead: bad argument (A + dueling <= 24: the head rows are staged in 48 KB of LDS)");
    a0_hip_backend bk{(hipStream_t)stream};
    const int splits = a0_fc1_splits(E, 512, K);
    a0_mat_src a{feat, K};
    a0_mat_src bw{W1, K};
    EpiSlab::Params ep{scratch, (long long)E * 512, 512};
    bk.tag = A0_TAG_DENSE_FWD;
    a0_fc1_partial_launch(bk, a, bw, ep, E, 512, K, splits);
    hipLaunchKernelGGL(a0_actor_qhead_kernel, dim3((E + 3) / 4), dim3(256), (size_t)(A + (dueling ? 1 : 0)) * 512 * sizeof(float), (hipStream_t)stream, scratch, (long long)E * 512, splits, b1, W2, b2, A, dueling, E,
                       seed, stream_a, stream_u, off_a, off_u, eps, ctrl, eps_ptr, action, qmax);
    A0_HIP_THROW(hipGetLastError());
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_actor_qhead_env_step(const float* feat, int E, int K, const float* W1, const float* b1, const float* W2, const float* b2, int A, int dueling,
                                       float* scratch, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                       unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                       unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                       float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                       const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task, void* stream) {
    A0_TRY
    if (!feat || !W1 || !b1 || !W2 || !b2 || !scratch || !action || !qmax || E < 1 || K < 4 || (K & 3) || A < 1 || A + (dueling ? 1 : 0) > 24)
        return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step: bad argument (A + dueling <= 24: the head rows are staged in 48 KB of LDS)");
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !ring_act || !ring_rew || !ring_done || !obs0 || !frames || !r_act ||
        !r_rew || !r_done || n < 1 || steps < 0 || cap < E || start_slot < 0 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE || (task == A0_ENV_TASK_CHASE && A < 4))
        return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step: bad env argument (the chase task needs at least four actions)");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step: buffers must be 16-byte aligned");
    a0_hip_backend bk{(hipStream_t)stream};
    const int splits = a0_fc1_splits(E, 512, K);
    a0_mat_src a{feat, K};
    a0_mat_src bw{W1, K};
    EpiSlab::Params ep{scratch, (long long)E * 512, 512};
    bk.tag = A0_TAG_DENSE_FWD;
    a0_fc1_partial_launch(bk, a, bw, ep, E, 512, K, splits);
    a0_qenv_args P;
    P.slabs = scratch; P.slab_stride = (long long)E * 512; P.nslab = splits; P.b1 = b1; P.W2 = W2; P.b2 = b2; P.A = A; P.dueling = dueling; P.E = E;
    P.rng_seed = seed; P.stream_a = stream_a; P.stream_u = stream_u; P.off_a = off_a; P.off_u = off_u; P.eps = eps; P.ctrl = ctrl; P.eps_ptr = eps_ptr;
    P.action = action; P.qmax = qmax;
    P.env_seed = env_seed; P.rank = rank; P.g = g; P.obs_in = obs_in; P.obs_out = obs_out; P.ep_ret = ep_ret; P.final_mask = final_mask; P.final_ret = final_ret;
    P.n = n; P.steps = steps; P.gamma = gamma; P.ring_act = ring_act; P.ring_rew = ring_rew; P.ring_done = ring_done; P.obs0 = obs0; P.frames = frames;
    P.cap = cap; P.start = start_slot % cap; P.r_act = r_act; P.r_rew = r_rew; P.r_done = r_done; P.task = task;
    hipLaunchKernelGGL(a0_actor_qhead_env_kernel, dim3(E), dim3(512), (size_t)(A + (dueling ? 1 : 0)) * 512 * sizeof(float), (hipStream_t)stream, P);
    A0_HIP_THROW(hipGetLastError());
    return A0_OK;
    A0_CATCH
}

// a0_actor_qhead_env_step whose tail kernel goes on to encode the env's NEW observation (round 5, a0_actor_step_enc_kernel in encoder_fused.hip): fc1 GEMM over
// `feat` (this step's features), then one launch for tail + env step + the NEXT step's features into `act3_next` (which may be `feat` itself: the GEMM has read it).
extern "C" int a0_actor_qhead_env_step_enc(const float* feat, int E, int K, const float* W1, const float* b1, const float* W2, const float* b2, int A, int dueling,
                                           float* scratch, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                           unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                           unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                           float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                           const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task,
                                           const float* wt, const a0_encoder_weights* w, float* act3_next, void* stream) {
    A0_TRY
    if (!feat || !W1 || !b1 || !W2 || !b2 || !scratch || !action || !qmax || E < 1 || K != 3136 || A < 1 || A + (dueling ? 1 : 0) > 24)
        return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step_enc: bad argument (4 x 84 x 84 observations, A + dueling <= 24)");
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !ring_act || !ring_rew || !ring_done || !obs0 || !frames || !r_act ||
        !r_rew || !r_done || n < 1 || steps < 0 || cap < E || start_slot < 0 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE || (task == A0_ENV_TASK_CHASE && A < 4))
        return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step_enc: bad env argument");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step_enc: buffers must be 16-byte aligned");
    a0_hip_backend bk{(hipStream_t)stream};
    const int splits = a0_fc1_splits(E, 512, K);
    a0_mat_src a{feat, K};
    a0_mat_src bw{W1, K};
    EpiSlab::Params ep{scratch, (long long)E * 512, 512};
    bk.tag = A0_TAG_DENSE_FWD;
    a0_fc1_partial_launch(bk, a, bw, ep, E, 512, K, splits);
    a0_qenv_args P;
    P.slabs = scratch; P.slab_stride = (long long)E * 512; P.nslab = splits; P.b1 = b1; P.W2 = W2; P.b2 = b2; P.A = A; P.dueling = dueling; P.E = E;
    P.rng_seed = seed; P.stream_a = stream_a; P.stream_u = stream_u; P.off_a = off_a; P.off_u = off_u; P.eps = eps; P.ctrl = ctrl; P.eps_ptr = eps_ptr;
    P.action = action; P.qmax = qmax;
    P.env_seed = env_seed; P.rank = rank; P.g = g; P.obs_in = obs_in; P.obs_out = obs_out; P.ep_ret = ep_ret; P.final_mask = final_mask; P.final_ret = final_ret;
    P.n = n; P.steps = steps; P.gamma = gamma; P.ring_act = ring_act; P.ring_rew = ring_rew; P.ring_done = ring_done; P.obs0 = obs0; P.frames = frames;
    P.cap = cap; P.start = start_slot % cap; P.r_act = r_act; P.r_rew = r_rew; P.r_done = r_done; P.task = task;
    return a0_actor_step_enc_launch(P, wt, w, act3_next, (hipStream_t)stream);
    A0_CATCH
}

// out[t] = mean_e x[t][e]: the per-step mean max-Q of a rollout (agent.py:38,88), all T steps in one launch
__global__ __launch_bounds__(256) void a0_mean_rows_kernel(const float* __restrict__ x, int E, float* __restrict__ out) {
    __shared__ float red[256];
    const float* p = x + (long long)blockIdx.x * E;
    float s = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) s += p[e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] / (float)E;
}

extern "C" int a0_mean_rows(const float* x, int T, int E, float* out, void* stream) {
    if (!x || !out || T < 1 || E < 1) return a0_fail(A0_EINVAL, "a0_mean_rows: bad argument");
    hipLaunchKernelGGL(a0_mean_rows_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, x, E, out);
    return a0_fail_hip((int)hipGetLastError(), "a0_mean_rows");
}

extern "C" int a0_dense_dgrad(const float* dY, const float* W, const float* act_mask, float* dX, int R, int N, int K, void* stream) {
    A0_TRY
    if (!dY || !W || !dX || R < 1 || (N & 3) || (K & 3)) return a0_fail(A0_EINVAL, "a0_dense_dgrad: bad shape");
    a0_hip_backend bk{(hipStream_t)stream};
    a0_dense_dgrad_impl(bk, dY, W, act_mask, dX, R, N, K);
    return A0_OK;
    A0_CATCH
}

// a0_dense_dgrad(dY, W, no mask) + a0_hadamard_bwd in ONE launch (round 6): dx = dY W [R][K] is consumed in the GEMM's epilogue (EpiHadamard) and never written — R = B * n rows,
// n = 32 or 64 fractions per sample, R a multiple of 256 (whole 256 x 128 tiles along the rows: a wave's 64 rows are whole samples), N (the reduction, fc1's 512) a multiple of 16.
// demb is bit-identical to the two calls' (the same accumulators times the same features); d3 sums the sample's rows in the tile's register order instead of row by row
// (fp32 rounding: 1e-7 of the accumulated magnitude).
extern "C" int a0_dense_dgrad_hadamard_ok(int R, int N, int K, int n) {
    static const bool off = getenv("A0_NO_DGRAD_HADAMARD") != nullptr;      // tuning aid
    return (!off && g_gemm_x9 != 0 && (n == 32 || n == 64) && R >= 4096 && !(R & 255) && !(R % n) && !(N & 15) && N >= 64 && !(K & 3) && K >= 128) ? 1 : 0;
}
extern "C" int a0_dense_dgrad_hadamard(const float* dY, const float* W, const float* emb, const float* feat, float* demb, float* d3, int R, int N, int K, int n, void* stream) {
    A0_TRY
    if (!dY || !W || !emb || !feat || !demb || !d3 || !a0_dense_dgrad_hadamard_ok(R, N, K, n)) return a0_fail(A0_EINVAL, "a0_dense_dgrad_hadamard: shapes a0_dense_dgrad_hadamard_ok accepts");
    const a0_mat_src a{dY, N};
    const a0_mat_src bw{W, K};
    const EpiHadamard::Params e{emb, feat, demb, d3, K, n};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool probe = a0_probe_events(A0_TAG_DENSE_DGRAD, &e0, &e1);
    A0_HIP_THROW((a0_igemm_x9_launch<OpMatKC, OpMatXC, EpiHadamard, 4, 2, 2, 2, 1>((hipStream_t)stream, a, bw, e, R, K, N, 1, e0, e1)));
    if (probe) a0_probe_commit(2.0 * (double)R * (double)N * (double)K);
    return A0_OK;
    A0_CATCH
}

// a0_dense_dgrad (with the ReLU mask) and the unsplit a0_dense_wgrad of ONE layer — same dY, R x N x K with as many 64 x 64 tiles in the data gradient (R x K) as in the
// weight gradient (N x K): N == R — as one launch (a0_igemm_x9_pair_kernel).  fc1 of a 512-row batch: dX = (dY W) * (X > 0) into dX, dW = dY^T X (+ bias row sums) into grad.
// Bit-identical to the two calls.  Shapes: a0_dense_dgrad_wgrad_ok.
extern "C" int a0_dense_dgrad_wgrad_ok(int R, int N, int K) {
    static const bool off = getenv("A0_NO_DGRAD_WGRAD_PAIR") != nullptr || getenv("A0_DGRAD_VARIANT") != nullptr || getenv("A0_WGRAD_VARIANT") != nullptr;      // tuning aids
    const long long b128 = (long long)((R + 127) / 128) * ((K + 63) / 64), b64 = (long long)((N + 63) / 64) * ((K + 63) / 64);
    return (!off && g_gemm_x9 != 0 && g_probe.tag == 0 && R == N && R >= 1 && R <= 1024 && !(N & 3) && !(K & 3) && b128 < 256 && b64 >= 256 && N >= 512 &&
            (long long)((R + 127) / 128) * ((K + 127) / 128) < g_x9_big_min) ? 1 : 0;
}
// (round 5) a second class of shapes: SMALL layers whose two gradients together fit one round of 64 x 64 tiles — the head of a distributional learner at a 512-row batch
// (c51: 64 + 32 tiles, qr: 64 + 104).  Alone, the data gradient occupies a quarter of the chip for its whole k loop and the weight gradient follows it; side by side
// the launch lasts as long as the longer of the two.  The weight gradient is then an UNSPLIT sum on the bf16 pipe (a0_dense_wgrad: slabs on the fp32 pipe).
static int a0_dense_dgrad_wgrad_small_ok(int R, int N, int K) {
    static const bool off = getenv("A0_NO_HEAD_PAIR") != nullptr;      // tuning aid
    const long long t1 = (long long)((R + 63) / 64) * ((K + 63) / 64), t2 = (long long)((N + 63) / 64) * ((K + 63) / 64);
    return (!off && g_gemm_x9 != 0 && g_probe.tag == 0 && R >= 64 && R <= 1024 && N >= 64 && !(N & 3) && !(K & 3) && t1 + t2 <= 256 && t2 >= 16) ? 1 : 0;
}
extern "C" int a0_dense_dgrad_wgrad_ok2(int R, int N, int K) { return (a0_dense_dgrad_wgrad_ok(R, N, K) || a0_dense_dgrad_wgrad_small_ok(R, N, K)) ? 1 : 0; }
extern "C" int a0_dense_dgrad_wgrad(const float* dY, const float* W, const float* X, int ldx, float* dX, float* grad, int R, int N, int K, void* stream) {
    A0_TRY
    if (!dY || !W || !X || !dX || !grad || ldx < K || (ldx & 3) || !a0_dense_dgrad_wgrad_ok2(R, N, K)) return a0_fail(A0_EINVAL, "a0_dense_dgrad_wgrad: shapes a0_dense_dgrad_wgrad_ok2 accepts");
    a0_mat_src a1{dY, N}, b1{W, K};                          // data gradient: rows r, reduction over n
    EpiMaskMat::Params e1{dX, X, K};
    a0_mat_src a2{dY, N}, b2{X, ldx};                        // weight gradient: rows n, reduction over r
    EpiWgradSlab::Params e2{grad, 0, K, (long long)N * K};
    if (!a0_dense_dgrad_wgrad_ok(R, N, K)) {                 // the small class: 64 x 64 tiles, four waves, unequal tile counts
        A0_HIP_THROW((a0_igemm_x9_pair_launch<OpMatKC, OpMatXC, EpiMaskMat, OpMatXC, OpMatXC, EpiWgradSlab, 2, 2, 1, 1>((hipStream_t)stream, a1, b1, e1, R, K, N, a2, b2, e2, N, K, R)));
        return A0_OK;
    }
    // 128 x 64 tiles on eight waves (196 + 196 workgroups, one per CU at a time): the same sums as the 64 x 64 tiles of the separate calls (every output element's k loop is the
    // same sequence of MFMAs), equal on the `main` schedule and ~2 % faster under `launch`, where the rollout's kernels share the chip (A0_PAIR_TILE=0: 64 x 64, tuning aid)
    static const int pv = getenv("A0_PAIR_TILE") ? atoi(getenv("A0_PAIR_TILE")) : 1;
    if (pv == 1) A0_HIP_THROW((a0_igemm_x9_pair_launch<OpMatKC, OpMatXC, EpiMaskMat, OpMatXC, OpMatXC, EpiWgradSlab, 4, 2, 1, 1>((hipStream_t)stream, a1, b1, e1, R, K, N, a2, b2, e2, N, K, R)));
    else A0_HIP_THROW((a0_igemm_x9_pair_launch<OpMatKC, OpMatXC, EpiMaskMat, OpMatXC, OpMatXC, EpiWgradSlab, 2, 2, 1, 1>((hipStream_t)stream, a1, b1, e1, R, K, N, a2, b2, e2, N, K, R)));
    return A0_OK;
    A0_CATCH
}

// a0_dense_dgrad_wgrad plus the NEXT layer's weight gradient in the same launch (round 5): with dY = d loss / d fc1's output and dY2 = d loss / d head's output both
// final (the head / loss kernel writes them), fc1's data gradient, fc1's weight gradient and the head's weight gradient dW2 = dY2^T X2 (+ bias row sums) into grad2
// [N2 x K2 | N2] are independent; the head's few tiles ride in the pair's launch (a0_igemm_x9_trio_kernel).  grad2 gets an UNSPLIT sum over the R rows on the bf16 pipe
// (a0_dense_wgrad splits it into slabs on the fp32 pipe: the same sum up to the association order).  Shapes: a0_dense_dgrad_wgrad_ok(R, N, K), X2 rows of R, ldx2 >= K2.
// where it pays: heads of at most one row of 128 x 64 tiles (scalar heads: 8 workgroups).  Measured with wider heads in the launch (same box, alternating): c51's 16 tiles
// 14.47 - 14.50 -> 14.57 ms, qr's 56 tiles 12.13 -> 12.16 ms — there the fp32-pipe split launch of its own stays; dqn 9.97 -> 9.85 ms, mdqn 11.69 -> 11.60.
extern "C" int a0_dense_dgrad_wgrad2_ok(int R, int N, int K, int N2, int K2) {
    static const bool off = getenv("A0_NO_TRIO") != nullptr;      // tuning aid (the head's weight gradient as a launch of its own: the same sum in another order)
    static const int max_tiles = getenv("A0_TRIO_MAX_TILES") ? atoi(getenv("A0_TRIO_MAX_TILES")) : 8;
    return (!off && a0_dense_dgrad_wgrad_ok(R, N, K) && N2 >= 4 && !(N2 & 3) && K2 >= 4 && !(K2 & 3) &&
            (long long)((N2 + 127) / 128) * ((K2 + 63) / 64) <= max_tiles) ? 1 : 0;
}
extern "C" int a0_dense_dgrad_wgrad2(const float* dY, const float* W, const float* X, int ldx, float* dX, float* grad, int R, int N, int K,
                                     const float* dY2, const float* X2, int ldx2, float* grad2, int N2, int K2, void* stream) {
    A0_TRY
    if (!dY || !W || !X || !dX || !grad || ldx < K || (ldx & 3) || !a0_dense_dgrad_wgrad_ok(R, N, K)) return a0_fail(A0_EINVAL, "a0_dense_dgrad_wgrad2: shapes a0_dense_dgrad_wgrad_ok accepts");
    if (!dY2 || !X2 || !grad2 || N2 < 4 || (N2 & 3) || K2 < 4 || (K2 & 3) || ldx2 < K2 || (ldx2 & 3) ||
        (long long)((N2 + 127) / 128) * ((K2 + 63) / 64) > (long long)((R + 127) / 128) * ((K + 63) / 64))
        return a0_fail(A0_EINVAL, "a0_dense_dgrad_wgrad2: the second layer's weight gradient must have no more 128 x 64 tiles than the first layer's data gradient");
    a0_mat_src a1{dY, N}, b1{W, K};
    EpiMaskMat::Params e1{dX, X, K};
    a0_mat_src a2{dY, N}, b2{X, ldx};
    EpiWgradSlab::Params e2{grad, 0, K, (long long)N * K};
    a0_mat_src a3{dY2, N2}, b3{X2, ldx2};
    EpiWgradSlab::Params e3{grad2, 0, K2, (long long)N2 * K2};
    A0_HIP_THROW((a0_igemm_x9_trio_launch<OpMatKC, OpMatXC, EpiMaskMat, OpMatXC, OpMatXC, EpiWgradSlab, 4, 2, 1, 1>((hipStream_t)stream, a1, b1, e1, R, K, N, a2, b2, e2, N, K, R,
                                                                                                                  a3, b3, e3, N2, K2, R)));
    return A0_OK;
    A0_CATCH
}

extern "C" long long a0_dense_wgrad_scratch(int R, int N, int K) { return a0_dense_wgrad_scratch_impl(R, N, K); }

extern "C" int a0_dense_wgrad(const float* dY, const float* X, int ldx, float* grad, int R, int N, int K, float* slabs, void* stream) {
    A0_TRY
    if (!dY || !X || !grad || R < 1 || (N & 3) || (K & 3) || (ldx & 3)) return a0_fail(A0_EINVAL, "a0_dense_wgrad: bad shape");
    if (a0_dense_wgrad_scratch_impl(R, N, K) > 0 && !slabs) return a0_fail(A0_EINVAL, "a0_dense_wgrad: needs slab scratch");
    a0_hip_backend bk{(hipStream_t)stream};
    a0_dense_wgrad_impl(bk, dY, X, ldx, grad, R, N, K, slabs);
    return A0_OK;
    A0_CATCH
}

// Up to four dense weight gradients whose slab reductions share ONE launch (head + fc1 [+ cosine embedding]): layer i uses
// slabs + slab_off[i] (a0_dense_wgrad_scratch(R[i], N[i], K[i]) floats each).  Same results as four a0_dense_wgrad calls.
extern "C" int a0_dense_wgrad_multi(int n, const float* const* dY, const float* const* X, const int* ldx, float* const* grad, const int* R, const int* N, const int* K,
                                    float* slabs, const long long* slab_off, a0_pending_reduce* pend, void* stream) {
    A0_TRY
    if (n < 1 || n > 4 || !dY || !X || !ldx || !grad || !R || !N || !K || !slab_off) return a0_fail(A0_EINVAL, "a0_dense_wgrad_multi: 1..4 layers");
    a0_hip_backend bk{(hipStream_t)stream};
    a0_reduce_seg segs[4];
    int nseg = 0;
    for (int i = 0; i < n; ++i) {
        if (!dY[i] || !X[i] || !grad[i] || R[i] < 1 || (N[i] & 3) || (K[i] & 3) || (ldx[i] & 3)) return a0_fail(A0_EINVAL, "a0_dense_wgrad_multi: bad shape");
        if (a0_dense_wgrad_scratch_impl(R[i], N[i], K[i]) > 0 && !slabs) return a0_fail(A0_EINVAL, "a0_dense_wgrad_multi: needs slab scratch");
        a0_reduce_seg seg;
        a0_dense_wgrad_impl(bk, dY[i], X[i], ldx[i], grad[i], R[i], N[i], K[i], slabs ? slabs + slab_off[i] : nullptr, &seg);
        if (seg.nslab > 0) segs[nseg++] = seg;
    }
    if (pend) {        // left to the next a0_net_encoder_wgrad(..., pend, ...)
        if (pend->n < 0 || pend->n + nseg > 4) return a0_fail(A0_EINVAL, "a0_dense_wgrad_multi: more than four pending reductions");
        for (int k = 0; k < nseg; ++k) pend->seg[pend->n++] = segs[k];
    } else if (nseg > 0) bk.reduce_segments(segs, nseg);
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_net_encoder_wgrad(const a0_net* n, const a0_encoder_weights* w, const a0_frames_arg* f, int B, const float* act1, const float* act2,
                                    const float* d3, const float* d2, const float* d1, float* g1, float* g2, float* g3, float* slabs, const a0_pending_reduce* pend,
                                    void* stream) {
    A0_TRY
    if (!n || !w || !f || !f->frames || !act1 || !act2 || !d3 || !d2 || !d1 || !g1 || !g2 || !g3 || B < 1) return a0_fail(A0_EINVAL, "a0_net_encoder_wgrad: null argument");
    if (a0_encoder_bwd_scratch_impl(n->core, B) > 0 && !slabs) return a0_fail(A0_EINVAL, "a0_net_encoder_wgrad: needs slab scratch");
    if (pend && (pend->n < 0 || pend->n > 4)) return a0_fail(A0_EINVAL, "a0_net_encoder_wgrad: bad pending reductions");
    a0_hip_backend bk{(hipStream_t)stream};
    a0_encoder_bwd_impl(bk, n->core, *w, *f, B, act1, act2, d3, const_cast<float*>(d2), const_cast<float*>(d1), g1, g2, g3, slabs, false, pend);
    return A0_OK;
    A0_CATCH
}

extern "C" long long a0_net_encoder_bwd_scratch(const a0_net* n, int B) { return n ? a0_encoder_bwd_scratch_impl(n->core, B) : 0; }

extern "C" int a0_net_encoder_bwd(const a0_net* n, const a0_encoder_weights* w, const a0_frames_arg* f, int B,
                                  const float* act1, const float* act2, const float* d3, float* d2, float* d1,
                                  float* g1, float* g2, float* g3, float* slabs, void* stream) {
    A0_TRY
    if (!n || !w || !f || !f->frames || !act1 || !act2 || !d3 || !d2 || !d1 || !g1 || !g2 || !g3 || B < 1) return a0_fail(A0_EINVAL, "a0_net_encoder_bwd: null argument");
    if (a0_encoder_bwd_scratch_impl(n->core, B) > 0 && !slabs) return a0_fail(A0_EINVAL, "a0_net_encoder_bwd: needs slab scratch");
    a0_hip_backend bk{(hipStream_t)stream};
    a0_encoder_bwd_impl(bk, n->core, *w, *f, B, act1, act2, d3, d2, d1, g1, g2, g3, slabs);
    return A0_OK;
    A0_CATCH
}
