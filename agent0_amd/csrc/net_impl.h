// Layer orchestration of the Nature CNN, written once against a "backend" that knows how to run one implicit GEMM
// and three small reductions.  net.hip instantiates it with the HIP backend (real kernels); tests/host_emul.cpp
// instantiates it with a CPU backend that evaluates the very same operand/epilogue policies in plain loops, so every
// table, stride and geometry decision below is exercised by the CPU test-suite before it ever reaches a GPU.
//
// Backend concept:
//   template <class OA, class OB, class EP, int WM, int WN, int MT, int NT>
//   void igemm(const OA::Params&, const OB::Params&, const EP::Params&, int X, int Y, int K, int splits);
//   void reduce_slabs(const float* slabs, long long slab_stride, int nslab, float* out, long long count);
//   void reduce_segments(const a0_reduce_seg* segs, int nseg);     several reduce_slabs in ONE launch
//   void reduce_bias_act(const float* slabs, long long slab_stride, int nslab, const float* bias, float* out, int rows, int N, int relu);
#pragma once
#include <cstdlib>
#include "net_tables.h"
#include "operands.h"
#include "../../include/agent0_hip.h"

// layer tags: set on the backend before each GEMM so a profiler probe can single out one kernel (bench.py roofline)
enum { A0_TAG_NONE = 0, A0_TAG_CONV1_FWD = 1, A0_TAG_CONV2_FWD = 2, A0_TAG_CONV3_FWD = 3, A0_TAG_DENSE_FWD = 4, A0_TAG_DENSE_DGRAD = 5,
       A0_TAG_DENSE_WGRAD = 6, A0_TAG_CONV3_WGRAD = 7, A0_TAG_CONV3_DGRAD = 8, A0_TAG_CONV2_WGRAD = 9, A0_TAG_CONV2_DGRAD = 10, A0_TAG_CONV1_WGRAD = 11 };

// one slab reduction: out[i] = sum_z slabs[z * slab_stride + i], i < count
// a0_reduce_seg, a0_pending_reduce: include/agent0_hip.h

// weight-gradient epilogue: slab z of the layer's [W | b] block.  ROWSUM_A: the GEMM's A operand is dY^T, so the sums of its rows
// over the k range of the split ARE the bias gradient; the kernel produces them as a by-product (from the LDS tiles it stages
// anyway) and stores them at out[z*slab_stride + bias_off + x] — no separate column-sum pass over dY.
struct EpiWgradSlab {
    static constexpr bool ROWSUM_A = true;
    struct Params { float* out; long long slab_stride; int ld; long long bias_off; };
    A0_HD static void store_rowsum(const Params& P, int x, float v, int z) { P.out[(long long)z * P.slab_stride + P.bias_off + x] = v; }
    A0_HD static void store(const Params& P, int x, int y, float v, int z) {
        P.out[(long long)z * P.slab_stride + (long long)x * P.ld + y] = v;
    }
};

static inline a0_frames_src a0_frames(const a0_net_core& n, const a0_frames_arg& f) {
    a0_frames_src s;
    s.frames = f.frames; s.slot = f.slot;
    s.g.Hin = n.H; s.g.Win = n.W; s.g.C = n.C;
    s.g.Hout = n.H1; s.g.Wout = n.W1; s.g.stride = 4; s.g.pad = 0; s.g.HWout = n.H1 * n.W1;
    s.g.hw_magic = a0_udiv_magic((unsigned)s.g.HWout); s.g.w_magic = a0_udiv_magic((unsigned)s.g.Wout);
    s.g.sample_stride = f.sample_stride;
    s.chan_off = f.chan_off;
    s.ktab = n.ktab1;
    s.aligned4 = (n.W % 4 == 0) && (f.sample_stride % 4 == 0) && (f.chan_off % 4 == 0) && (((uintptr_t)f.frames) % 4 == 0);
    return s;
}

static inline a0_act_src a0_act(const float* x, int Hin, int Win, int C, int Hout, int Wout, int stride, int pad, const a0_i4* ktab) {
    a0_act_src s;
    s.x = x; s.ktab = ktab;
    s.g.Hin = Hin; s.g.Win = Win; s.g.C = C; s.g.Hout = Hout; s.g.Wout = Wout; s.g.stride = stride; s.g.pad = pad;
    s.g.HWout = Hout * Wout; s.g.sample_stride = (long long)Hin * Win * C;
    s.g.hw_magic = a0_udiv_magic((unsigned)s.g.HWout); s.g.w_magic = a0_udiv_magic((unsigned)s.g.Wout);
    return s;
}

// ---- split heuristics (pure functions of the shapes, shared by the *_scratch queries)
static inline int a0_fwd_splits(int gx, int gy, int K) {
    static const int forced = getenv("A0_FWD_SPLITS") ? atoi(getenv("A0_FWD_SPLITS")) : 0;     // tuning aid
    if (forced > 0) return forced;
    int blocks = gx * gy, splits = 1;
    if (blocks < 256) {
        splits = 256 / blocks;          // one workgroup per CU: measured best for fc1 at 256 / 512 rows (tools/ubench_dense.py)
        int maxs = (K / 32) / 2;
        if (splits > maxs) splits = maxs;
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
    }
    return splits;
}

static inline int a0_wgrad_splits(int gx, int gy, int R) {
    static const int target = getenv("A0_WGRAD_TARGET") ? atoi(getenv("A0_WGRAD_TARGET")) : 512;     // tuning aid
    int blocks = gx * gy, splits = 1;
    if (blocks < 256) {
        splits = (target + blocks - 1) / blocks;
        int maxs = (R + 63) / 64;
        if (splits > maxs) splits = maxs;
        if (splits > 256) splits = 256;
        if (splits < 1) splits = 1;
    }
    return splits;
}

// Deep weight gradients (the quantile networks' fc1 over B*N = 16 384 ... 32 768 rows): the GEMM runs on 128 x 128 eight-wave tiles, ONE
// workgroup per CU, so its duration is (rounds of 256 workgroups) x (rows per split).  fc1 has 4 x 25 = 100 such tiles; the 64 x 128-tile
// heuristic above gave 3 splits = 300 workgroups = TWO rounds of 10 923 rows (measured 1116 us at 32 768 rows, 94 TFLOP/s, while the forward
// and data-gradient GEMMs of the same size run at 137-139).  Here the split count minimises rounds x rows-per-split plus the price of the
// slabs (one slab = N*K*4 bytes written and read back: ~1 round-row unit per 4 MB at the measured rates): 5 splits = 500 workgroups = two
// rounds of 6 554 rows.  A0_WGRAD_DEEP=0 restores the old choice (tuning aid).
static inline int a0_wgrad_splits_deep(int N, int K, int R) {
    static const int on = getenv("A0_WGRAD_DEEP") ? atoi(getenv("A0_WGRAD_DEEP")) : 1;
    const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
    if (!on || R < 4096 || N < 128 || K < 128) return 0;                       // not the deep, large case: the caller keeps a0_wgrad_splits
    const double slab_rows = ((double)N * K * 4.0 / 4.0e6) * 20.0;             // one slab ~ 20 rows' worth of tile time per 4 MB (write + reduce read)
    int best = 1; double best_cost = 1e300;
    for (int sp = 1; sp <= 16; ++sp) {
        const long long rows = (((R + 31) / 32 + sp - 1) / sp) * 32;
        if (rows < 512) break;                                                 // keep every split deep (the split-operand kernel's condition)
        const long long rounds = (tiles * sp + 255) / 256;
        const double cost = (double)rounds * (double)rows + (sp > 1 ? sp * slab_rows : 0.0);
        if (cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}

// conv2 / conv3 weight gradients: 64 x 64 output tiles and a split count that gives two workgroups per CU, i.e. 400-650 reduction rows
// and 25 MB of slabs instead of 200-330 rows and 40 MB with 64 x 128 tiles (106 vs 110 us for the three layers + reduction at B = 512,
// tools/ubench_convwgrad.py).  A0_CONV_WGRAD_T64 = target workgroup count (tuning aid; 0 = the 64 x 128 tiles of a0_wgrad_splits).
static inline int a0_conv_wgrad_t64() {
    static const int target = getenv("A0_CONV_WGRAD_T64") ? atoi(getenv("A0_CONV_WGRAD_T64")) : 512;
    return target;
}
static inline int a0_conv_wgrad_splits(int K, int M) {
    const int t = a0_conv_wgrad_t64();
    if (t <= 0) return a0_wgrad_splits(1, (K + 127) / 128, M);
    int s = t / ((K + 63) / 64);
    const int maxs = (M + 255) / 256;
    if (s > maxs) s = maxs;
    return s < 1 ? 1 : s;
}

static inline int a0_chunk_rows(int R, int splits) { return (((R + 31) / 32 + splits - 1) / splits) * 32; }

static inline long long a0_dense_fwd_scratch_impl(int R, int N, int K) {
    int s = a0_fwd_splits((R + 127) / 128, (N + 63) / 64, K);
    return s > 1 ? (long long)s * R * N : 0;
}

// A deep weight gradient with a NARROW second dimension (the cosine embedding, [3136][64] over B*N rows): 64 x 64 tiles instead of 64 x 128
// (whose second half would be empty) on four waves — two workgroups per CU — and as many splits as fill those 512 slots once
// (measured at 32 768 rows: 352 us with 64 x 128 tiles and 11 splits = 37 TFLOP/s).
static inline bool a0_wgrad_narrow_deep(int R, int N, int K) {
    static const int on = getenv("A0_WGRAD_DEEP") ? atoi(getenv("A0_WGRAD_DEEP")) : 1;
    return on && K <= 64 && N >= 256 && R >= 4096;
}
static inline int a0_dense_wgrad_splits(int R, int N, int K) {
    if (a0_wgrad_narrow_deep(R, N, K)) {
        const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
        int sp = 512 / tiles;
        const int maxs = R / 512;                          // every split stays deep
        if (sp > maxs) sp = maxs;
        if (sp > 32) sp = 32;
        return sp < 1 ? 1 : sp;
    }
    const int deep = a0_wgrad_splits_deep(N, K, R);
    return deep > 0 ? deep : a0_wgrad_splits((N + 63) / 64, (K + 127) / 128, R);
}

static inline long long a0_dense_wgrad_scratch_impl(int R, int N, int K) {
    int s = a0_dense_wgrad_splits(R, N, K);
    return s > 1 ? (long long)s * ((long long)N * K + N) : 0;
}

// conv23_wgrad.hip (per-observation conv2 / conv3 weight gradients, 84 x 84 geometry): groups of observations per layer.  conv2 is cut into
// 4 parts, conv3 into 3, every (part, group) is one workgroup of equal cost per observation, and two workgroups share a CU: 7 G <= 2 x CUs;
// a group's workgroups all run on XCD g % 8 (its L2 then serves the parts' common operands), so G is a multiple of 8 when it can be: 72.
static inline bool a0_c23w_plan(const a0_net_core& n, int B, int* G2, int* G3) {
    static const bool off = getenv("A0_NO_CONV23_WGRAD_FUSED") != nullptr;
    static const int gmax = getenv("A0_C23W_GROUPS") ? atoi(getenv("A0_C23W_GROUPS")) : ((2 * 256) / 7) / 8 * 8;      // tuning aid
    if (off || B < 1 || n.H1 != 20 || n.W1 != 20 || n.H2 != 9 || n.W2 != 9 || n.H3 != 7 || n.W3 != 7) return false;
    const int g = B < gmax ? B : gmax;
    *G2 = g; *G3 = g;
    return g >= 1;
}

// Slab regions of the three convolution weight gradients inside the caller's scratch buffer.  They are disjoint, so that the three
// GEMMs can leave their partial sums behind and ONE reduction launch finishes all of them (a0_encoder_bwd_impl).  conv1 is sized for
// the per-observation kernel (one slab per workgroup, at most 256) or the GEMM's splits, whichever is larger.
struct a0_enc_slab_plan { long long off[3], total; int splits[3]; };     // index 0 = conv1, 1 = conv2, 2 = conv3

static inline a0_enc_slab_plan a0_encoder_slab_plan(const a0_net_core& n, int B) {
    a0_enc_slab_plan p;
    const int M[3] = {B * n.H1 * n.W1, B * n.H2 * n.W2, B * n.H3 * n.W3};
    const int N[3] = {32, 64, 64};
    const int K[3] = {n.K1, n.K2, n.K3};
    long long off = 0;
    int g23[3] = {0, 0, 0};
    if (!a0_c23w_plan(n, B, &g23[1], &g23[2])) g23[1] = g23[2] = 0;
    for (int l = 2; l >= 0; --l) {
        p.splits[l] = l == 0 ? a0_wgrad_splits(1, (K[l] + 127) / 128, M[l]) : a0_conv_wgrad_splits(K[l], M[l]);
        int slabs = p.splits[l] > 1 ? p.splits[l] : 0;
        if (l == 0) { const int fused = B < 256 ? B : 256; if (fused > slabs) slabs = fused; }
        else if (g23[l] > slabs) slabs = g23[l];             // the per-observation kernel's groups
        p.off[l] = off;
        off += (long long)slabs * ((long long)N[l] * K[l] + N[l]);
    }
    p.total = off;
    return p;
}

static inline long long a0_encoder_bwd_scratch_impl(const a0_net_core& n, int B) { return a0_encoder_slab_plan(n, B).total; }

// ------------------------------------------------------------------------------------------------ encoder forward
template <class BK>
static void a0_encoder_fwd_impl(BK& bk, const a0_net_core& n, const a0_encoder_weights& w, const a0_frames_arg& f, int B,
                                float* act1, float* act2, float* act3) {
    {
        a0_frames_src a = a0_frames(n, f);
        a0_mat_src b{w.w1, n.K1};
        EpiBiasAct::Params e{act1, w.b1, 32, 1};
        bk.tag = A0_TAG_CONV1_FWD;
        bk.template igemm<OpFramesKC, OpMatKC, EpiBiasAct, 4, 1, 1, 1>(a, b, e, B * n.H1 * n.W1, 32, n.K1, 1);
    }
    {
        a0_act_src a = a0_act(act1, n.H1, n.W1, 32, n.H2, n.W2, 2, 0, n.ktab2);
        a0_mat_src b{w.w2, n.K2};
        EpiBiasAct::Params e{act2, w.b2, 64, 1};
        bk.tag = A0_TAG_CONV2_FWD;
        bk.template igemm<OpActKC, OpMatKC, EpiBiasAct, 4, 1, 1, 2>(a, b, e, B * n.H2 * n.W2, 64, n.K2, 1);
    }
    {
        a0_act_src a = a0_act(act2, n.H2, n.W2, 64, n.H3, n.W3, 1, 0, n.ktab3);
        a0_mat_src b{w.w3, n.K3};
        EpiBiasAct::Params e{act3, w.b3, 64, 1};
        const int M = B * n.H3 * n.W3;
        bk.tag = A0_TAG_CONV3_FWD;
        if (M <= 128 * 160)   // small batches (actor): 64-row tiles keep more CUs busy
            bk.template igemm<OpActKC, OpMatKC, EpiBiasAct, 2, 2, 1, 1>(a, b, e, M, 64, n.K3, 1);
        else
            bk.template igemm<OpActKC, OpMatKC, EpiBiasAct, 4, 1, 1, 2>(a, b, e, M, 64, n.K3, 1);
    }
}

// ------------------------------------------------------------------------------------------------ dense layers
// Shapes of the short-reduction forward kernel (short_k_fwd.h): K = 64 exactly (the quantile networks' cosine embedding), rows of X
// 16-byte aligned, enough rows and columns that the general kernel's tiles would be all prologue and epilogue.
static inline bool a0_short_k_shape(int R, int N, int K, int ldx) {
    static const bool off = getenv("A0_NO_SHORT_K") != nullptr;       // tuning aid
    return !off && K == 64 && (ldx & 3) == 0 && ldx >= 64 && N >= 64 && R >= 256;
}

template <class BK>
static void a0_dense_fwd_impl(BK& bk, const float* X, int ldx, const float* W, const float* b, float* Y, int R, int N, int K,
                              int relu, float* scratch) {
    bk.tag = A0_TAG_DENSE_FWD;
    if (a0_short_k_shape(R, N, K, ldx)) { bk.short_k_fwd(X, ldx, W, b, nullptr, 1, Y, nullptr, R, N, relu); return; }
    a0_mat_src a{X, ldx};
    a0_mat_src bw{W, K};
    const int splits = a0_fwd_splits((R + 127) / 128, (N + 63) / 64, K);
    const bool narrow = (N <= 32);
    bk.tag = A0_TAG_DENSE_FWD;
    if (splits == 1) {
        EpiBiasAct::Params e{Y, b, N, relu};
        if (narrow) bk.template igemm<OpMatKC, OpMatKC, EpiBiasAct, 4, 1, 1, 1>(a, bw, e, R, N, K, 1);
        else bk.template igemm<OpMatKC, OpMatKC, EpiBiasAct, 4, 1, 1, 2>(a, bw, e, R, N, K, 1);
    } else {
        EpiSlab::Params e{scratch, (long long)R * N, N};
        if (narrow) bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 4, 1, 1, 1>(a, bw, e, R, N, K, splits);
        else bk.template igemm<OpMatKC, OpMatKC, EpiSlab, 4, 1, 1, 2>(a, bw, e, R, N, K, splits);
        bk.reduce_bias_act(scratch, (long long)R * N, splits, b, Y, R, N, relu);
    }
}

template <class BK>
static void a0_dense_dgrad_impl(BK& bk, const float* dY, const float* W, const float* act_mask, float* dX, int R, int N, int K) {
    a0_mat_src a{dY, N};
    a0_mat_src bw{W, K};
    bk.tag = A0_TAG_DENSE_DGRAD;
    if (act_mask) {
        EpiMaskMat::Params e{dX, act_mask, K};
        static const int var = getenv("A0_DGRAD_VARIANT") ? atoi(getenv("A0_DGRAD_VARIANT")) : 0;      // tuning aid
        const long long blocks = (long long)((R + 127) / 128) * ((K + 63) / 64);
        if (var == 1 || (var == 0 && blocks < 256)) bk.template igemm<OpMatKC, OpMatXC, EpiMaskMat, 2, 2, 1, 1>(a, bw, e, R, K, N, 1);   // 64x64 tiles: twice the workgroups
        else if (var == 2) bk.template igemm<OpMatKC, OpMatXC, EpiMaskMat, 4, 1, 1, 1>(a, bw, e, R, K, N, 1);
        else bk.template igemm<OpMatKC, OpMatXC, EpiMaskMat, 4, 1, 1, 2>(a, bw, e, R, K, N, 1);
    } else {
        EpiSlab::Params e{dX, 0, K};
        bk.template igemm<OpMatKC, OpMatXC, EpiSlab, 4, 1, 1, 2>(a, bw, e, R, K, N, 1);
    }
}

// shared tail of every weight gradient: the slab reduction (weights and the bias row sums the GEMM left behind them)
template <class BK>
static void a0_finish_wgrad(BK& bk, int N, long long wcount, float* grad, float* slabs, int splits) {
    if (splits > 1) bk.reduce_slabs(slabs, wcount + N, splits, grad, wcount + N);
}

// defer != nullptr: leave the partial sums in `slabs` and describe the pending reduction in *defer (nslab = 0: nothing pending, the
// GEMM wrote the gradient itself); the caller finishes several layers with one bk.reduce_segments launch.
template <class BK>
static void a0_dense_wgrad_impl(BK& bk, const float* dY, const float* X, int ldx, float* grad, int R, int N, int K, float* slabs, a0_reduce_seg* defer = nullptr) {
    const long long wcount = (long long)N * K;
    a0_mat_src a{dY, N};
    a0_mat_src b{X, ldx};
    bk.tag = A0_TAG_DENSE_WGRAD;
    if (defer) *defer = a0_reduce_seg{nullptr, 0, 0, nullptr, 0};
    static const int var = getenv("A0_WGRAD_VARIANT") ? atoi(getenv("A0_WGRAD_VARIANT")) : -1;      // tuning aid: 0 = always split, 1 = never for >= 256 tiles
    const long long blocks64 = (long long)((N + 63) / 64) * ((K + 63) / 64);
    if (blocks64 >= 256 && (var == 1 || (var < 0 && R <= 1024))) {
        // 64x64 output tiles without a reduction split when the output alone has >= 256 tiles and the batch is short (fc1 of a 512-row batch:
        // 392 tiles).  As a lone kernel this is slower than 64x128 tiles + 3 slabs, but no slabs means 19 MB less to write and to reduce: the
        // whole B = 512 update measures 419 vs 432 us (tools/ubench_update.py, profiles/r02_encoder_experiments.md).  The quantile networks'
        // 32 768-row reductions keep the split kernel.
        EpiWgradSlab::Params e{grad, 0, K, wcount};
        bk.template igemm<OpMatXC, OpMatXC, EpiWgradSlab, 2, 2, 1, 1>(a, b, e, N, K, R, 1);
        return;
    }
    const int splits = a0_dense_wgrad_splits(R, N, K);
    EpiWgradSlab::Params e{splits > 1 ? slabs : grad, splits > 1 ? wcount + N : 0, K, wcount};
    if (a0_wgrad_narrow_deep(R, N, K)) bk.template igemm<OpMatXC, OpMatXC, EpiWgradSlab, 2, 2, 1, 1>(a, b, e, N, K, R, splits);
    else bk.template igemm<OpMatXC, OpMatXC, EpiWgradSlab, 2, 2, 1, 2>(a, b, e, N, K, R, splits);
    if (defer && splits > 1) *defer = a0_reduce_seg{slabs, wcount + N, splits, grad, wcount + N};
    else a0_finish_wgrad(bk, N, wcount, grad, slabs, splits);
}

// ------------------------------------------------------------------------------------------------ encoder backward
// d3 = dL/d(conv3 pre-activation) [B][H3][W3][64] (ReLU mask already applied by the fc1 data gradient).
template <class BK>
static void a0_encoder_bwd_impl(BK& bk, const a0_net_core& n, const a0_encoder_weights& w, const a0_frames_arg& f, int B,
                                const float* act1, const float* act2, const float* d3, float* d2, float* d1,
                                float* g1, float* g2, float* g3, float* slabs, bool with_dgrad = true, const a0_pending_reduce* pend = nullptr) {
    // with_dgrad == false: d2 / d1 already hold the data gradients (a0_net_encoder_dgrad_fused); only the weight gradients run
    const int M3 = B * n.H3 * n.W3, M2 = B * n.H2 * n.W2, M1 = B * n.H1 * n.W1;
    const a0_enc_slab_plan plan = a0_encoder_slab_plan(n, B);
    a0_reduce_seg segs[8];
    int nseg = 0;
    if (pend) for (int k = 0; k < pend->n && k < 4; ++k) segs[nseg++] = pend->seg[k];      // reductions other launches left to this one (a0_dense_wgrad_multi)
    // d2 already known (the learner's path: a0_net_encoder_dgrad_fused ran first): both weight gradients in ONE per-observation launch on the bf16 pipe
    int G2 = 0, G3 = 0;
    const bool fused23 = !with_dgrad && a0_c23w_plan(n, B, &G2, &G3) &&
                         bk.conv23_wgrad_fused(n, B, act1, act2, d2, d3, slabs + plan.off[1], slabs + plan.off[2]);
    if (fused23) {
        segs[nseg++] = a0_reduce_seg{slabs + plan.off[2], 64LL * n.K3 + 64, G3, g3, 64LL * n.K3 + 64};
        segs[nseg++] = a0_reduce_seg{slabs + plan.off[1], 64LL * n.K2 + 64, G2, g2, 64LL * n.K2 + 64};
    }
    if (!fused23) {   // conv3 weight gradient: dW3[64][K3] = sum_m d3[m][:]^T im2col(act2)[m][:]
        const int splits = plan.splits[2];
        const long long wc = 64LL * n.K3;
        float* sl = slabs + plan.off[2];
        a0_mat_src a{d3, 64};
        a0_act_src b = a0_act(act2, n.H2, n.W2, 64, n.H3, n.W3, 1, 0, n.ktab3);
        EpiWgradSlab::Params e{splits > 1 ? sl : g3, splits > 1 ? wc + 64 : 0, n.K3, wc};
        bk.tag = A0_TAG_CONV3_WGRAD;
        if (a0_conv_wgrad_t64() > 0) bk.template igemm<OpMatXC, OpActXC, EpiWgradSlab, 2, 2, 1, 1>(a, b, e, 64, n.K3, M3, splits);
        else bk.template igemm<OpMatXC, OpActXC, EpiWgradSlab, 2, 2, 1, 2>(a, b, e, 64, n.K3, M3, splits);
        if (splits > 1) segs[nseg++] = a0_reduce_seg{sl, wc + 64, splits, g3, wc + 64};
    }
    if (with_dgrad) {   // conv3 data gradient -> d2 (masked by act2 > 0): gather form, 3x3 taps over d3 with pad 2
        a0_act_src a = a0_act(d3, n.H3, n.W3, 64, n.H2, n.W2, 1, 2, n.ktab_d3);
        a0_wtab_src b{w.w3, n.wtab_d3};
        EpiDgrad::Params e{d2, act2, n.H2 * n.W2, n.W2, n.H2, n.W2, 64, 1, 0, 0, (long long)n.H2 * n.W2 * 64};
        bk.tag = A0_TAG_CONV3_DGRAD;
        bk.template igemm<OpActKC, OpWtabXC, EpiDgrad, 4, 1, 1, 2>(a, b, e, M2, 64, 9 * 64, 1);
    }
    if (!fused23) {   // conv2 weight gradient
        const int splits = plan.splits[1];
        const long long wc = 64LL * n.K2;
        float* sl = slabs + plan.off[1];
        a0_mat_src a{d2, 64};
        a0_act_src b = a0_act(act1, n.H1, n.W1, 32, n.H2, n.W2, 2, 0, n.ktab2);
        EpiWgradSlab::Params e{splits > 1 ? sl : g2, splits > 1 ? wc + 64 : 0, n.K2, wc};
        bk.tag = A0_TAG_CONV2_WGRAD;
        if (a0_conv_wgrad_t64() > 0) bk.template igemm<OpMatXC, OpActXC, EpiWgradSlab, 2, 2, 1, 1>(a, b, e, 64, n.K2, M2, splits);
        else bk.template igemm<OpMatXC, OpActXC, EpiWgradSlab, 2, 2, 1, 2>(a, b, e, 64, n.K2, M2, splits);
        if (splits > 1) segs[nseg++] = a0_reduce_seg{sl, wc + 64, splits, g2, wc + 64};
    }
    // conv2 data gradient -> d1 (masked by act1 > 0): four stride phases, 2x2 taps over d2 with pad 1
    for (int ph = 0; ph < 2 && with_dgrad; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            const int Hv = (n.H1 - ph + 1) / 2, Wv = (n.W1 - pw + 1) / 2;
            if (Hv < 1 || Wv < 1) continue;
            a0_act_src a = a0_act(d2, n.H2, n.W2, 64, Hv, Wv, 1, 1, n.ktab_d2);
            a0_wtab_src b{w.w2, n.wtab_d2[ph * 2 + pw]};
            EpiDgrad::Params e{d1, act1, Hv * Wv, Wv, n.H1, n.W1, 32, 2, ph, pw, (long long)n.H1 * n.W1 * 32};
            bk.tag = A0_TAG_CONV2_DGRAD;
            bk.template igemm<OpActKC, OpWtabXC, EpiDgrad, 4, 1, 1, 1>(a, b, e, B * Hv * Wv, 32, 4 * 64, 1);
        }
    {   // conv1 weight gradient (the input is data: no data gradient)
        const long long wc = 32LL * n.K1;
        float* sl = slabs + plan.off[0];
        bk.tag = A0_TAG_CONV1_WGRAD;
        const int fused_slabs = bk.conv1_wgrad_fused(n, f, B, d1, sl);       // 84x84x4: per-observation kernel on the bf16 pipe
        if (fused_slabs > 0) {
            segs[nseg++] = a0_reduce_seg{sl, wc + 32, fused_slabs, g1, wc + 32};
        } else {
            const int splits = plan.splits[0];
            a0_mat_src a{d1, 32};
            a0_frames_src b = a0_frames(n, f);
            EpiWgradSlab::Params e{splits > 1 ? sl : g1, splits > 1 ? wc + 32 : 0, n.K1, wc};
            bk.template igemm<OpMatXC, OpFramesXC, EpiWgradSlab, 1, 4, 1, 1>(a, b, e, 32, n.K1, M1, splits);
            if (splits > 1) segs[nseg++] = a0_reduce_seg{sl, wc + 32, splits, g1, wc + 32};
        }
    }
    // the three layers' slab reductions (weights and the bias row sums behind them) in one launch
    if (nseg > 0) bk.reduce_segments(segs, nseg);
}
