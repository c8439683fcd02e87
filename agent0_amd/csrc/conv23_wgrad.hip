// conv2 and conv3 weight gradients on the bf16 matrix pipe with the operands resident in LDS per observation (84 x 84 geometry).
//
//   dW3[co][(kh,kw,ci)] = sum_{b,oh,ow} d3[b][oh][ow][co] * act2[b][oh+kh][ow+kw][ci]              (49 positions per observation)
//   dW2[co][(kh,kw,ci)] = sum_{b,oh,ow} d2[b][oh][ow][co] * act1[b][2oh+kh][2ow+kw][ci]            (81 positions per observation)
// — autograd's backward-weight of the second and third conv of ConvEncoder (reference agent0/deepq/model.py:93-105, called from
// agent.py:153-155); db = the row sums of d.  As implicit GEMMs (igemm.h, <OpMatXC, OpActXC, EpiWgradSlab>) these are M = 64, N = 512 / 576,
// K = B*81 / B*49: short, heavily split reductions whose tiles re-stage the same activations for every 64-column block and every split —
// 2 x 32 us at B = 512, 0.45 of the fp32 MFMA peak (profiles/r02).  Here a workgroup keeps ONE observation's operands in LDS as three exact
// bf16 term planes each (x = hi + mid + lo, 8 + 8 + 8 significand bits: igemm_x9.h), every element is fetched and split once per workgroup
// that needs it, and all nine cross products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation — the same real number as the fp32 fmaf
// chain up to the association order of the sum.  The reduction index of a weight gradient is the POSITION, while both operands are stored
// position-major ([position][channel], as the forward / data-gradient kernels leave them): the k-major MFMA fragments are produced by the
// LDS itself with ds_read_b64_tr_b16, whose per-lane row addresses also do the im2col gather of the activation operand.
//
// Work split.  The accumulators stay in registers for the whole launch, so a workgroup's partial sums travel through HBM once, as a slab
// that a0_reduce_segments adds up (deterministic, like every other weight gradient).  Slab bytes = workgroups per layer x layer size, so a
// layer is cut into PARTS and each part runs over a strided subset ("group") of the observations:
//   conv2: 2 parts (32 output channels each: 16 tiles of 32 x 32, two per wave), G2 groups -> slab region g of conv2 gets both halves;
//   conv3: 3 parts (one kernel row kh each: 2 channel blocks x 3 taps x 2 input-channel halves = 12 tiles; per SIMD one wave with two
//          tiles and one with one), G3 groups.
// 2*G2 + 3*G3 <= CUs; (G2, G3) balance the two kinds of workgroup (a0_c23w_plan).  One launch, 512 threads, 93 KB of LDS.
//
// LDS images (bf16, three term planes each; every row is 64 bytes = 32 channels, so the 4 rows x 64 bytes a 32-lane half reads per
// ds_read_b64_tr_b16 tile all 64 banks when the rows are consecutive):
//   conv2  act1: [400 pixels][32 ci], pixel (ih, iw) at index ih*20 + (iw & 1)*10 + (iw >> 1): the stride-2 taps of consecutive output
//                columns are consecutive rows;  d2: [96 rows][32 co of this half], rows 81..95 zero.
//   conv3  act2: [2 ci halves][81 pixels][32];  d3: [2 co halves][64 rows][32], rows 49..63 zero.
// Positions beyond the image (the zero rows of d) read a valid activation row: 0 * finite = 0.
#include "a0_internal.h"
#include "net_impl.h"

#include <cstdlib>

typedef __bf16 a0q_bf16x8 __attribute__((ext_vector_type(8)));
typedef short a0q_s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t a0q_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t a0q_u32x4 __attribute__((ext_vector_type(4)));
typedef float a0_acc16 __attribute__((ext_vector_type(16)));

constexpr int A0Q_THREADS = 512;
constexpr int A0Q_ROW = 64;                                   // bytes per LDS row (32 bf16)

struct a0_c23w_args {
    const float *act1, *act2, *d2, *d3;      // [B][400][32], [B][81][64], [B][81][64], [B][49][64]
    float *slab2, *slab3;                    // [G2][64*512 + 64], [G3][64*576 + 64]
    int B, G2, G3;
};

template <int LAYER> struct a0q_geom;
template <> struct a0q_geom<2> {
    static constexpr int NPOS = 81, STEPS = 6, DROWS = 96, APIX = 400, CIN = 32, K = 512;
    static constexpr int ACT_PLANE = APIX * A0Q_ROW, D_PLANE = DROWS * A0Q_ROW;          // bytes per term plane
    static constexpr int ACT_F4 = 400 * 8, D_F4 = 81 * 8;                                // float4 pieces per observation (d: this half's 32 channels)
    A0_D static int base_row(int p) { return 40 * (p / 9) + (p % 9); }                   // activation row of position p, tap (0, 0)
};
template <> struct a0q_geom<3> {
    static constexpr int NPOS = 49, STEPS = 4, DROWS = 64, APIX = 81, CIN = 64, K = 576;
    static constexpr int ACT_PLANE = 2 * APIX * A0Q_ROW, D_PLANE = 2 * DROWS * A0Q_ROW;
    static constexpr int ACT_F4 = 81 * 16, D_F4 = 49 * 16;
    A0_D static int base_row(int p) { return 9 * (p / 7) + (p % 7); }
};
constexpr int A0Q_LDS_BYTES = 3 * (a0q_geom<2>::ACT_PLANE + a0q_geom<2>::D_PLANE);      // 95 232: the larger of the two images
static_assert(3 * (a0q_geom<3>::ACT_PLANE + a0q_geom<3>::D_PLANE) <= A0Q_LDS_BYTES, "conv3 image fits");

// exact three-term split of a float4, packed as 4 bf16 (8 bytes) per term
A0_D void a0q_split4(const a0_f4& v, a0q_u32x2& hi, a0q_u32x2& mid, a0q_u32x2& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = __float_as_uint(x[e]);
        const float r1 = x[e] - __uint_as_float(h[e] & 0xffff0000u);        // exact: at most 16 significant bits left
        m[e] = __float_as_uint(r1);
        l[e] = __float_as_uint(r1 - __uint_as_float(m[e] & 0xffff0000u));    // exact: at most 8 significant bits left
    }
    hi.x = __builtin_amdgcn_perm(h[1], h[0], 0x07060302u);  hi.y = __builtin_amdgcn_perm(h[3], h[2], 0x07060302u);
    mid.x = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u); mid.y = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
    lo.x = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);  lo.y = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
}

// one k-major fragment (8 consecutive k = positions of this lane's row / column) out of a position-major plane: two transposed reads.
// off0 / off1: byte offsets of the lane's row of the first / second 4-row block (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies
// row q, columns 4p..4p+3, and receives column (lane & 15) of the four rows).  All 64 lanes execute this (EXEC all ones).
A0_D a0q_u32x4 a0q_frag(const unsigned char* plane, int off0, int off1) {
    typedef __attribute__((address_space(3))) a0q_s16x4 lds_s16x4;
    const a0q_s16x4 k03 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + off0));
    const a0q_s16x4 k47 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + off1));
    const a0q_u32x2 a = __builtin_bit_cast(a0q_u32x2, k03), b = __builtin_bit_cast(a0q_u32x2, k47);
    return a0q_u32x4{a.x, a.y, b.x, b.y};
}

// The body of one workgroup: layer LAYER, part `part`, observations g, g + G, ...; NT tiles per wave is a wave-uniform run-time choice
// made by the caller (conv3: 2 or 1), so the MFMA loop is instantiated per NT.
template <int LAYER, int NT>
A0_D void a0q_mma_steps(const unsigned char* act, const unsigned char* dpl, const int (&aoff)[a0q_geom<LAYER>::STEPS][2], const int (&boff)[a0q_geom<LAYER>::STEPS][2],
                        int a_blk, const int (&b_blk)[2], a0_acc16 (&acc)[2]) {
    typedef a0q_geom<LAYER> G;
    // term pairs in the order of increasing magnitude (lo*lo first, hi*hi last), as in igemm_x9.h
    constexpr int TA[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
    constexpr int TB[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
    a0q_u32x4 a[2][3], b[2][NT][3];
    auto fetch = [&](int slot, int s) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            a[slot][t] = a0q_frag(dpl + t * G::D_PLANE + a_blk, aoff[s][0], aoff[s][1]);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[slot][j][t] = a0q_frag(act + t * G::ACT_PLANE + b_blk[j], boff[s][0], boff[s][1]);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < G::STEPS; ++s) {
        if (s + 1 < G::STEPS) fetch((s + 1) & 1, s + 1);          // the next step's fragments are in flight behind this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(a0q_bf16x8, a[s & 1][TA[q]]), __builtin_bit_cast(a0q_bf16x8, b[s & 1][j][TB[q]]), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int LAYER>
A0_D void a0q_body(const a0_c23w_args& P, int part, int g, int ngroups, unsigned char* smem) {
    typedef a0q_geom<LAYER> G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const act = smem;
    unsigned char* const dpl = smem + 3 * G::ACT_PLANE;
    // the d planes' pad rows stay zero for the whole launch (the staging below only writes rows < NPOS)
    for (int i = tid; i < 3 * G::D_PLANE / 16; i += A0Q_THREADS) ((uint4*)dpl)[i] = uint4{0u, 0u, 0u, 0u};

    // ---- this wave's tiles
    int nt, a_blk, b_blk[2], kcol[2], corow;
    if constexpr (LAYER == 2) {
        nt = 2; a_blk = 0; corow = part * 32;                            // this half's 32 output channels
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tap = 2 * wave + j, kh = tap >> 2, kw = tap & 3;
            b_blk[j] = (kh * 20 + (kw & 1) * 10 + (kw >> 1)) * A0Q_ROW;   // row offset of the tap in the parity-split image
            kcol[j] = tap * 32;
        }
    } else {
        const int simd = wave & 3, first = 3 * simd + (wave >> 2) * 2;    // tiles {3s, 3s+1} for wave s, {3s+2} for wave s + 4
        nt = wave < 4 ? 2 : 1;
        const int c = first / 6;
        a_blk = c * G::DROWS * A0Q_ROW; corow = c * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kb = (first + (j < nt ? j : 0)) % 6, kw = kb >> 1, half = kb & 1;
            b_blk[j] = (half * G::APIX + part * 9 + kw) * A0Q_ROW;        // part = kernel row kh
            kcol[j] = (part * 3 + kw) * 64 + half * 32;
        }
    }
    // per-lane row offsets of the two transposed reads of every step (see a0q_frag); + the lane's 8-byte column group
    const int colb = 2 * (16 * ((lane >> 4) & 1) + 4 * (lane & 3));
    int aoff[G::STEPS][2], boff[G::STEPS][2];
#pragma unroll
    for (int s = 0; s < G::STEPS; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = 16 * s + 8 * (lane >> 5) + ((lane & 15) >> 2) + 4 * h;
            aoff[s][h] = p * A0Q_ROW + colb;
            boff[s][h] = G::base_row(p < G::NPOS ? p : G::NPOS - 1) * A0Q_ROW + colb;
        }

    a0_acc16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};        // bias partial: this thread's four channels of d over the positions it stages

    // ---- raw data of one observation in registers, requested for the NEXT observation before the MFMA loop of the current one
    constexpr int RA = (G::ACT_F4 + A0Q_THREADS - 1) / A0Q_THREADS, RD = (G::D_F4 + A0Q_THREADS - 1) / A0Q_THREADS;
    a0_f4 ra[RA], rd[RD];
    auto load_raw = [&](int b) {
        const a0_f4* sa = (const a0_f4*)((LAYER == 2 ? P.act1 : P.act2) + (long long)b * (G::APIX * G::CIN));
#pragma unroll
        for (int j = 0; j < RA; ++j) { const int i = tid + j * A0Q_THREADS; ra[j] = sa[i < G::ACT_F4 ? i : G::ACT_F4 - 1]; }
        const float* sd = (LAYER == 2 ? P.d2 : P.d3) + (long long)b * (G::NPOS * 64);
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            int i = tid + j * A0Q_THREADS;
            i = i < G::D_F4 ? i : G::D_F4 - 1;
            if constexpr (LAYER == 2) rd[j] = *(const a0_f4*)(sd + (i >> 3) * 64 + part * 32 + (i & 7) * 4);
            else rd[j] = *(const a0_f4*)(sd + i * 4);
        }
    };
    int b = g;
    if (b < P.B) load_raw(b);
    __syncthreads();
    for (; b < P.B; b += ngroups) {
        // ---- registers -> three term planes (one float4 = four channels of one pixel / position -> three 8-byte writes)
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int i = tid + j * A0Q_THREADS;
            if (i < G::ACT_F4) {
                int row, cb;
                if constexpr (LAYER == 2) { const int px = i >> 3, ih = px / 20, iw = px - ih * 20; row = ih * 20 + (iw & 1) * 10 + (iw >> 1); cb = (i & 7) * 8; }
                else { const int px = i >> 4, c4 = (i & 15) * 4; row = (c4 >> 5) * G::APIX + px; cb = (c4 & 31) * 2; }
                a0q_u32x2 hi, mid, lo;
                a0q_split4(ra[j], hi, mid, lo);
                unsigned char* d = act + row * A0Q_ROW + cb;
                *(a0q_u32x2*)d = hi; *(a0q_u32x2*)(d + G::ACT_PLANE) = mid; *(a0q_u32x2*)(d + 2 * G::ACT_PLANE) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int i = tid + j * A0Q_THREADS;
            if (i < G::D_F4) {
                int row, cb;
                if constexpr (LAYER == 2) { row = i >> 3; cb = (i & 7) * 8; }
                else { const int c4 = (i & 15) * 4; row = (c4 >> 5) * G::DROWS + (i >> 4); cb = (c4 & 31) * 2; }
                bsum[0] += rd[j].x; bsum[1] += rd[j].y; bsum[2] += rd[j].z; bsum[3] += rd[j].w;
                a0q_u32x2 hi, mid, lo;
                a0q_split4(rd[j], hi, mid, lo);
                unsigned char* d = dpl + row * A0Q_ROW + cb;
                *(a0q_u32x2*)d = hi; *(a0q_u32x2*)(d + G::D_PLANE) = mid; *(a0q_u32x2*)(d + 2 * G::D_PLANE) = lo;
            }
        }
        __syncthreads();
        if (b + ngroups < P.B) load_raw(b + ngroups);
        if (nt == 2) a0q_mma_steps<LAYER, 2>(act, dpl, aoff, boff, a_blk, b_blk, acc);
        else a0q_mma_steps<LAYER, 1>(act, dpl, aoff, boff, a_blk, b_blk, acc);
        __syncthreads();          // the planes are rebuilt for the next observation
    }

    // ---- slab g of this layer: the part's rows / columns of [64][K], then (once per slab) the bias row sums
    float* out = (LAYER == 2 ? P.slab2 : P.slab3) + (long long)g * (64 * G::K + 64);
    // C/D layout of v_mfma_f32_32x32x16_bf16: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(corow + row) * G::K + kcol[j] + (lane & 31)] = acc[j][r];
            }
        }
    }
    // bias: a thread's four channels are the same in all its pieces (the piece index advances by 512, a multiple of the pieces per row)
    const bool has_bias = LAYER == 2 || part == 0;
    if (has_bias) {
        float* red = (float*)smem;                      // the images are dead (the loop ends with a barrier)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid * 4 + e] = bsum[e];
        __syncthreads();
        constexpr int CH = LAYER == 2 ? 32 : 64, TPR = CH / 4;        // threads per row of d pieces: tid % TPR selects the channel group
        if (tid < CH) {
            float s = 0.f;
            for (int t = tid >> 2; t < A0Q_THREADS; t += TPR) s += red[t * 4 + (tid & 3)];       // fixed order
            out[64 * G::K + (LAYER == 2 ? part * 32 : 0) + tid] = s;
        }
    }
}

__global__ __launch_bounds__(A0Q_THREADS) void a0_conv23_wgrad_fused_kernel(a0_c23w_args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char a0q_smem[];
    const int id = blockIdx.x, n2 = 2 * P.G2;
    // the heavier conv2 workgroups first: they are resident from the start, the conv3 ones fill the remaining CUs
    if (id < n2) a0q_body<2>(P, id & 1, id >> 1, P.G2, a0q_smem);
    else a0q_body<3>(P, (id - n2) % 3, (id - n2) / 3, P.G3, a0q_smem);
}

// returns 1 when the kernel ran (slab2: G2 slabs of 64*512 + 64 floats, slab3: G3 slabs of 64*576 + 64; a0_c23w_plan), 0 = shape not
// supported, nothing launched
int a0_conv23_wgrad_fused_launch(const a0_net_core& n, int B, const float* act1, const float* act2, const float* d2, const float* d3, float* slab2, float* slab3,
                                 hipStream_t st) {
    int G2 = 0, G3 = 0;
    if (!a0_c23w_plan(n, B, &G2, &G3) || !slab2 || !slab3) return 0;
    a0_c23w_args P;
    P.act1 = act1; P.act2 = act2; P.d2 = d2; P.d3 = d3; P.slab2 = slab2; P.slab3 = slab3; P.B = B; P.G2 = G2; P.G3 = G3;
    static bool configured = false;
    if (!configured) {
        A0_HIP_THROW(hipFuncSetAttribute((const void*)a0_conv23_wgrad_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A0Q_LDS_BYTES));
        configured = true;
    }
    hipLaunchKernelGGL(a0_conv23_wgrad_fused_kernel, dim3(2 * G2 + 3 * G3), dim3(A0Q_THREADS), A0Q_LDS_BYTES, st, P);
    A0_HIP_THROW(hipGetLastError());
    return 1;
}
