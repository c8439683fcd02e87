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
//   conv2: 4 parts = the parity classes (kh & 1, kw & 1) of the 4 x 4 stride-2 kernel.  A tap of class (a, b) only ever touches input pixels
//          (ih, iw) with ih & 1 = a, iw & 1 = b, so a part needs a QUARTER of act1 — the 10 x 10 sub-image of its class, in which the taps of
//          consecutive output columns are consecutive rows — and all of d2: 2 channel blocks x 4 taps = 8 tiles of 32 x 32, one per wave;
//   conv3: 3 parts (one kernel row kh each: 2 channel blocks x 3 taps x 2 input-channel halves = 12 tiles; per SIMD one wave with two
//          tiles and one with one).
// The two kinds of workgroup cost 2 x 5 resp. 3 x 3 k-steps of nine MFMAs (+ one fp32 MFMA per tile) per observation and SIMD and hold 55 KB of LDS, so TWO are resident per CU:
// while one stages its next observation (global -> registers is prefetched; split + LDS writes between two barriers) the other one's waves
// keep the matrix pipe busy.  G groups per layer, 7 G <= 2 x CUs workgroups (a0_c23w_plan); slab region g of a layer receives all its parts.
//
// LDS images (bf16, three term planes each; every row is 64 bytes = 32 channels, so the 4 rows x 64 bytes a 32-lane half reads per
// ds_read_b64_tr_b16 tile all 64 banks when the rows are consecutive):
//   conv2  act1 class image: [100 pixels][32 ci], pixel (ih, iw) at row (ih >> 1)*10 + (iw >> 1);  d2: [2 co halves][96 rows][32], rows 81..95 zero.
//   conv3  act2: [2 ci halves][81 pixels][32];  d3: [2 co halves][64 rows][32], rows 49..63 zero.
// The k-steps of 16 positions cover 80 / 48 of a layer's 81 / 49 positions; the last one runs through one fp32 MFMA (a0q_mma_steps).
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
    static constexpr int NPOS = 81, STEPS = 5, DROWS = 96, APIX = 100, CIN = 32, K = 512;       // 81 = 5 x 16 + 1
    static constexpr int ACT_PLANE = APIX * A0Q_ROW, D_PLANE = 2 * DROWS * A0Q_ROW;      // bytes per term plane
    static constexpr int ACT_F4 = 100 * 8, D_F4 = 81 * 16;                               // float4 pieces per observation (act: this parity class only)
    A0_D static int base_row(int p) { return 10 * (p / 9) + (p % 9); }                   // class-image row of position p, first tap of the class
};
template <> struct a0q_geom<3> {
    static constexpr int NPOS = 49, STEPS = 3, DROWS = 64, APIX = 81, CIN = 64, K = 576;         // 49 = 3 x 16 + 1
    static constexpr int ACT_PLANE = 2 * APIX * A0Q_ROW, D_PLANE = 2 * DROWS * A0Q_ROW;
    static constexpr int ACT_F4 = 81 * 16, D_F4 = 49 * 16;
    A0_D static int base_row(int p) { return 9 * (p / 7) + (p % 7); }
};
constexpr int A0Q_LDS_BYTES = 3 * (a0q_geom<2>::ACT_PLANE + a0q_geom<2>::D_PLANE);      // 56 064: the larger of the two images
static_assert(3 * (a0q_geom<3>::ACT_PLANE + a0q_geom<3>::D_PLANE) <= A0Q_LDS_BYTES && 2 * A0Q_LDS_BYTES <= 160 * 1024, "conv3 image fits; two workgroups per CU");

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

// One observation's k-steps for NT tiles of one wave (NT is wave-uniform: conv3 has waves with two tiles and with one).  The fragments are
// not double-buffered: four waves per SIMD (two workgroups per CU) cover the LDS latency, and the registers buy the second workgroup.
template <int NPR, int LAYER, int NT>
A0_D void a0q_mma_steps(const unsigned char* act, const unsigned char* dpl, int aoff, const int (&boff)[a0q_geom<LAYER>::STEPS][2],
                        int a_blk, const int (&b_blk)[2], a0_acc16 (&acc)[2]) {
    typedef a0q_geom<LAYER> G;
    const int lane = threadIdx.x & 63;
    // term pairs in the order of increasing magnitude (lo*lo first, hi*hi last), as in igemm_x9.h
    constexpr int TA[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
    constexpr int TB[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int s = 0; s < G::STEPS; ++s) {
        a0q_u32x4 a[3], b[NT][3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            a[t] = a0q_frag(dpl + t * G::D_PLANE + a_blk + s * 16 * A0Q_ROW, aoff, aoff + 4 * A0Q_ROW);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j][t] = a0q_frag(act + t * G::ACT_PLANE + b_blk[j], boff[s][0], boff[s][1]);
        }
#pragma unroll
        for (int q = 9 - NPR; q < 9; ++q)       // six products (a0_x9_products): lo*lo, lo*mid, mid*lo are not formed
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(a0q_bf16x8, a[TA[q]]), __builtin_bit_cast(a0q_bf16x8, b[j][TB[q]]), acc[j], 0, 0, 0);
    }
    // The ONE position the k-steps of 16 leave over (49 = 3 x 16 + 1, 81 = 5 x 16 + 1): a padded k-step would cost nine more MFMAs per tile
    // (a fifth / a quarter of the work) for it.  Instead it goes through one v_mfma_f32_32x32x2_f32 — the exact fp32 chain, same accumulator
    // layout — with the fp32 values put back together from their three bf16 terms (hi + mid + lo is exact): lanes 0-31 carry k = 0, the
    // upper half (k = 1) carries zeros.
    {
        const int lrow = lane & 31;
        const bool k0 = lane < 32;
        auto f32_at = [](const unsigned char* p, int plane_bytes) {
            const float hi = __uint_as_float((uint32_t)(*(const uint16_t*)p) << 16), mid = __uint_as_float((uint32_t)(*(const uint16_t*)(p + plane_bytes)) << 16),
                        lo = __uint_as_float((uint32_t)(*(const uint16_t*)(p + 2 * plane_bytes)) << 16);
            return (hi + mid) + lo;
        };
        const float av = f32_at(dpl + a_blk + (G::NPOS - 1) * A0Q_ROW + 2 * lrow, G::D_PLANE);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float bv = f32_at(act + b_blk[j] + G::base_row(G::NPOS - 1) * A0Q_ROW + 2 * lrow, G::ACT_PLANE);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0 ? av : 0.f, k0 ? bv : 0.f, acc[j], 0, 0, 0);
        }
    }
}

template <int NPR, int LAYER>
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
        // part = parity class (a, b) = (part >> 1, part & 1); wave -> channel block wave & 1, tap (a + 2 th, b + 2 tw) with (th, tw) = wave >> 1
        const int c = wave & 1, th = wave >> 2, tw = (wave >> 1) & 1, kh = (part >> 1) + 2 * th, kw = (part & 1) + 2 * tw;
        nt = 1; a_blk = c * G::DROWS * A0Q_ROW; corow = c * 32;
        b_blk[0] = b_blk[1] = (th * 10 + tw) * A0Q_ROW;                  // row offset of the tap in the class image
        kcol[0] = kcol[1] = (kh * 4 + kw) * 32;
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
    const int aoff = (8 * (lane >> 5) + ((lane & 15) >> 2)) * A0Q_ROW + colb;      // + (16 s + 4 h) rows: compile-time
    int boff[G::STEPS][2];
#pragma unroll
    for (int s = 0; s < G::STEPS; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = 16 * s + 8 * (lane >> 5) + ((lane & 15) >> 2) + 4 * h;
            boff[s][h] = G::base_row(p) * A0Q_ROW + colb;                 // p < 16 * STEPS = NPOS - 1: always a real position
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
        const float* sa = LAYER == 2 ? P.act1 + (long long)b * (400 * 32) : P.act2 + (long long)b * (81 * 64);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            int i = tid + j * A0Q_THREADS;
            i = i < G::ACT_F4 ? i : G::ACT_F4 - 1;
            if constexpr (LAYER == 2) {       // pixel (2r + a, 2c + b) of the class image's row 10 r + c
                const int px = i >> 3, r = px / 10, c = px - r * 10;
                ra[j] = *(const a0_f4*)(sa + ((2 * r + (part >> 1)) * 20 + 2 * c + (part & 1)) * 32 + (i & 7) * 4);
            } else ra[j] = *(const a0_f4*)(sa + i * 4);
        }
        const a0_f4* sd = (const a0_f4*)((LAYER == 2 ? P.d2 : P.d3) + (long long)b * (G::NPOS * 64));
#pragma unroll
        for (int j = 0; j < RD; ++j) { const int i = tid + j * A0Q_THREADS; rd[j] = sd[i < G::D_F4 ? i : G::D_F4 - 1]; }
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
                if constexpr (LAYER == 2) { row = i >> 3; cb = (i & 7) * 8; }
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
                const int c4 = (i & 15) * 4, row = (c4 >> 5) * G::DROWS + (i >> 4), cb = (c4 & 31) * 2;
                bsum[0] += rd[j].x; bsum[1] += rd[j].y; bsum[2] += rd[j].z; bsum[3] += rd[j].w;
                a0q_u32x2 hi, mid, lo;
                a0q_split4(rd[j], hi, mid, lo);
                unsigned char* d = dpl + row * A0Q_ROW + cb;
                *(a0q_u32x2*)d = hi; *(a0q_u32x2*)(d + G::D_PLANE) = mid; *(a0q_u32x2*)(d + 2 * G::D_PLANE) = lo;
            }
        }
        __syncthreads();
        if (b + ngroups < P.B) load_raw(b + ngroups);
        if (nt == 2) a0q_mma_steps<NPR, LAYER, 2>(act, dpl, aoff, boff, a_blk, b_blk, acc);
        else a0q_mma_steps<NPR, LAYER, 1>(act, dpl, aoff, boff, a_blk, b_blk, acc);
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
    const bool has_bias = part == 0;
    if (has_bias) {
        float* red = (float*)smem;                      // the images are dead (the loop ends with a barrier)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid * 4 + e] = bsum[e];
        __syncthreads();
        constexpr int CH = 64, TPR = CH / 4;        // threads per row of d pieces: tid % TPR selects the channel group
        if (tid < CH) {
            float s = 0.f;
            for (int t = tid >> 2; t < A0Q_THREADS; t += TPR) s += red[t * 4 + (tid & 3)];       // fixed order
            out[64 * G::K + tid] = s;
        }
    }
}

template <int NPR>
__global__ __launch_bounds__(A0Q_THREADS, 4) void a0_conv23_wgrad_fused_kernel(a0_c23w_args P) {      // four waves per SIMD: two workgroups per CU
    extern __shared__ __attribute__((aligned(16))) unsigned char a0q_smem[];
    // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.  The seven parts of a group read the
    // same observations (d2 four times, d3 and act2 three times), so all of them are given to ONE XCD: group g lives on XCD g % 8, and
    // inside an XCD the workgroups run group by group, part by part (parts 0-3: conv2 classes, 4-6: conv3 kernel rows).
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, g = xcd + 8 * (k / 7), part = k % 7;
    if (g >= P.G2) return;                                   // G2 == G3 (a0_c23w_plan); the grid is rounded up to whole XCD rows
    if (part < 4) a0q_body<NPR, 2>(P, part, g, P.G2, a0q_smem);
    else a0q_body<NPR, 3>(P, part - 4, g, P.G3, a0q_smem);
}

// returns 1 when the kernel ran (slab2: G2 slabs of 64*512 + 64 floats, slab3: G3 slabs of 64*576 + 64; a0_c23w_plan), 0 = shape not
// supported, nothing launched
int a0_conv23_wgrad_fused_launch(const a0_net_core& n, int B, const float* act1, const float* act2, const float* d2, const float* d3, float* slab2, float* slab3,
                                 hipStream_t st) {
    int G2 = 0, G3 = 0;
    if (!a0_c23w_plan(n, B, &G2, &G3) || !slab2 || !slab3) return 0;
    a0_c23w_args P;
    P.act1 = act1; P.act2 = act2; P.d2 = d2; P.d3 = d3; P.slab2 = slab2; P.slab3 = slab3; P.B = B; P.G2 = G2; P.G3 = G3;
    const int six = a0_x9_products_now() == 6;
    auto kern = six ? a0_conv23_wgrad_fused_kernel<6> : a0_conv23_wgrad_fused_kernel<9>;
    static bool configured[2] = {false, false};
    if (!configured[six]) {
        A0_HIP_THROW(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, A0Q_LDS_BYTES));
        configured[six] = true;
    }
    if (G2 != G3) return 0;
    hipLaunchKernelGGL(kern, dim3(8 * 7 * ((G2 + 7) / 8)), dim3(A0Q_THREADS), A0Q_LDS_BYTES, st, P);
    A0_HIP_THROW(hipGetLastError());
    return 1;
}
