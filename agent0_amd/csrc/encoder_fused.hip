// Fused Nature-CNN encoder forward: one workgroup per observation, all three convolutions back to back with the
// activations resident in LDS (gfx950).
//
// Replaces ConvEncoder.forward (reference agent0/deepq/model.py:93-105) plus the uint8 -> fp32 /255 in front of it
// (agent.py:27, agent.py:129-134) for the shapes the actor-learner loop uses.  Why: as three separate implicit GEMMs the
// actor's 256-observation forward launches 162-800 workgroups per layer and is latency-bound (profiles/r01); here every
// CU owns one observation (28 KB u8 -> LDS once) and reads its im2col A operands straight out of LDS with computed addresses
// (no staging pass, no HBM round trip for act1/act2).  The B operands (the weights, shared by every workgroup and L2-resident)
// never touch LDS: a0_conv_wt_kernel lays them out in MFMA-fragment order so that each lane streams ITS fragment values with
// one 16-byte global load per 16 k, through a register ring several chunks deep.  There is therefore no barrier and no
// s_waitcnt vmcnt(0) anywhere inside a convolution's k loop — the eight waves drift apart and keep the matrix pipe shared —
// only one barrier per layer, when its output lands in LDS.
// conv2 / conv3 use v_mfma_f32_16x16x4_f32 (16-row tiles fit M = 81 / 49 rows with little padding and split evenly over the eight
// waves) and are the same k-ascending fp32 fmaf chains as in igemm.h.
// conv1 multiplies BYTES by fp32 weights, and both factors can be fed to the 16x-faster bf16 matrix pipe without giving up a bit:
// a byte 0..255 is exact in bf16 (8 significant bits), the reference's x/255 is folded into the weight (w' = fl(w/255), same two
// roundings per term as fl(x/255)*w), and w' is split EXACTLY into three bf16 terms w' = hi + mid + lo (8 + 8 + 8 mantissa bits,
// a0_conv_wt_kernel).  Every product byte * term is exact in fp32, so three v_mfma_f32_16x16x32_bf16 per 32 k accumulate the same
// real number as the fp32 chain, in fp32 accumulators (measured against fp64: 6.0e-8 vs 7.9e-8 for the fmaf chain,
// tools/check_bf16x3.hip) at 48 instead of 256 matrix-pipe cycles.  The result agrees with the unfused path (and the reference) to
// fp32 rounding — tests/test_gpu_engine.py::test_fused_encoder_matches_unfused, rtol 2e-6.
#include "a0_internal.h"
#include "net_tables.h"
#include "operands.h"

#include <cstdlib>
#include <type_traits>

// Cross products a_i * b_j of the three-term splits that are formed: those with i + j <= A0_X9_MAXORD.  4 = all nine (every partial product
// of the fp32 fmaf chain, exactly) — the default.  2 = six: a1*b2, a2*b1 and a2*b2 are left out — each is below 2^-24 of a*b (|a1| <= 2^-8 |a|, |b2| <= 2^-16 |b|),
// i.e. below the rounding the fp32 chain itself applies to every partial SUM, which for a K-term dot product is ~sqrt(K) times larger.
// Measured (tools/build_variant.sh x6 -DA0_X9_MAXORD=2; profiles/r02_encoder_experiments.md): a third fewer conv2 / conv3 MFMAs shorten the
// forward kernel by 7-9 % and the data-gradient kernel by 9 % with every parity test unchanged — the stages are not matrix-pipe-bound —
// and the build keeps all nine: exact products are worth more here than 3.6 % of an iteration.
#ifndef A0_X9_MAXORD
#define A0_X9_MAXORD 4
#endif
typedef float a0_acc4 __attribute__((ext_vector_type(4)));

struct a0_fused_args {
    const uint8_t* frames; const int* slot; long long sample_stride; int chan_off;
    const float *wt1, *wt2, *wt3;        // fragment-major weight copies (a0_conv_wt_kernel)
    const float *wx2, *wx3;              // conv2 / conv3 as three exact bf16 terms (split-operand path)
    const float *b1, *b2, *b3;
    float *act1, *act2, *act3;           // act1/act2 optional (needed only when a backward pass follows)
    int B;
    int C, H, W, H1, W1, H2, W2, H3, W3;
    int off_act1, off_act2, off_end;     // float offsets of the LDS regions behind the u8 observation
    int rp1, rp2;                        // LDS row pitches (floats) of act1 / act2, padded so that A-fragment reads are bank-conflict free
};

// Eight waves per workgroup = two per SIMD: while one wave waits for its LDS operands the other keeps the matrix pipe busy.
#ifndef A0_FUSED_WAVES_D
#define A0_FUSED_WAVES_D 8
#endif
constexpr int A0_FUSED_WAVES = A0_FUSED_WAVES_D;
constexpr int A0_WMG1 = A0_FUSED_WAVES / 2;       // conv1: two waves along N, the rest along M
constexpr int A0_FUSED_THREADS = 64 * A0_FUSED_WAVES;
// LDS layout of the activations: [row][pixel][channel] with pixel pitch P and row pitch RP.  An A-fragment read (ds_read_b32) is served
// in two groups of 32 lanes = 16 consecutive output positions m x 2 adjacent k; it is conflict-free when position m lands on bank
// 2m (mod 32).  conv2 reads act1 at stride 2: 2*P1 = 66 = 2 (mod 32) along a row, and the row pitch is padded until
// 2*RP1 = 2*W2 (mod 32), so stepping to the next output row continues the sequence; conv3 (stride 1): P2 = 66, RP2 = 2*W3 (mod 32).
constexpr int A0_P1 = 33;
constexpr int A0_P2 = 66;

// ---- A-operand fetchers: element (row m, k = 16*c + 4*j + q) of the im2col matrix, read directly from LDS.
// address = row(m) + chunk_off(c) + step_off(j) + q: the chunk part is added once per 16-deep chunk, the step part is a
// compile-time constant that folds into the DS instruction's immediate offset (WC > 0: width known at compile time).
struct AF2 {   // conv2 4x4/2 over act1 [H1][RP1]; k = (kh*4 + kw)*32 + c
    const float* act; int RP1, W2;
    A0_D int row(int m) const { const int oh = m / W2, ow = m - oh * W2; return (2 * oh) * RP1 + 2 * ow * A0_P1; }
    A0_D int chunk_off(int c) const { const int cell = c >> 1; return (cell >> 2) * RP1 + (cell & 3) * A0_P1 + 16 * (c & 1); }
    A0_D int step_off(int s) const { return 4 * s; }
    typedef float Raw;
    A0_D Raw load(int addr) const { return act[addr]; }
    static A0_D float value(Raw r) { return r; }
};
struct AF3 {   // conv3 3x3/1 over act2 [H2][RP2]; k = (kh*3 + kw)*64 + c
    const float* act; int RP2, W3;
    A0_D int row(int m) const { const int oh = m / W3, ow = m - oh * W3; return oh * RP2 + ow * A0_P2; }
    A0_D int chunk_off(int c) const { const int cell = c >> 2; return (cell / 3) * RP2 + (cell % 3) * A0_P2 + 16 * (c & 3); }
    A0_D int step_off(int s) const { return 4 * s; }
    typedef float Raw;
    A0_D Raw load(int addr) const { return act[addr]; }
    static A0_D float value(Raw r) { return r; }
};

// ---- B operands.  Packed layout of one layer (a0_conv_wt_kernel): float index ((c*N + n)*4 + q)*4 + j holds W[n][k = 16c + 4j + q],
// i.e. lane (n, q) of the MFMA B fragment finds the values of the four steps j of chunk c in one aligned float4, and a wave's
// sixteen columns x four q of a chunk are 1 KB contiguous.
template <int N, int WN, int R>
struct a0_wring {
    static constexpr int NBW = N / 16 / WN;
    a0_f4 v[R][NBW];
    const float* base;            // this lane's float4 of chunk 0, column block 0
    const float* p;               // next chunk to request: the ring only ever walks forward, one running pointer instead of R addresses
    int nch, left;
    A0_D void init(const float* wp, int K) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int n = (wave % WN) * (NBW * 16) + (lane & 15), q = lane >> 4;
        base = wp + (n * 4 + q) * 4;
        nch = K >> 4;
    }
    A0_D void fill(int slot) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) v[slot][j] = *(const a0_f4*)(p + j * 256);
        const bool more = left > 1;                    // past the end: re-read the last chunk (never consumed), keeps the loop branch-free
        p += more ? N * 16 : 0;
        left -= more ? 1 : 0;
    }
    A0_D void prologue() {
        p = base; left = nch;
#pragma unroll
        for (int u = 0; u < R; ++u) fill(u);
    }
};
A0_D float a0_f4_get(const a0_f4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// ---- epilogue policies.  Values an epilogue needs from global memory (bias, ReLU masks) are requested at the START of the layer:
// a load issued in the epilogue would wait behind the next layer's weight prologue.
template <int OWC>
struct EpiFwd {                 // y = relu(acc + bias[n]) -> LDS image [oh][ow][n] (pitch / row pitch) and / or global [m][n]
    static constexpr bool PER_ELEM = false;
    static constexpr bool TR = false;
    const float* bias; float* lds; int pitch, rp, ow; float* glb; int N;
    A0_D float pre_col(int n) const { return bias[n]; }
    A0_D float pre_elem(int, int) const { return 0.f; }
    A0_D void emit(int m, int n, float acc, float pre) const {
        float v = acc + pre;
        v = (v < 0.f) ? 0.f : v;
        if (lds) {
            const int w_ = OWC > 0 ? OWC : ow, oh = m / w_;
            lds[oh * rp + (m - oh * w_) * pitch + n] = v;
        }
        if (glb) glb[(unsigned)(m * N + n)] = v;      // per-observation base (uniform) + 32-bit offset: no 64-bit address per element
    }
    // rows m0 .. m0+3 (m0 a multiple of 4) of one accumulator: when the image width is a multiple of 4 they sit in one image row
    static constexpr bool ROW4 = OWC > 0 && (OWC % 4) == 0;
    A0_D void emit4(int m0, int n, const a0_acc4& acc, float pre) const {
        constexpr int OWD = OWC > 0 ? OWC : 1;
        const int oh = m0 / OWD, ow0 = m0 - oh * OWD;
        float* l = lds ? lds + oh * rp + ow0 * pitch + n : nullptr;
        const unsigned go = (unsigned)(m0 * N + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[r] + pre;
            v = (v < 0.f) ? 0.f : v;
            if (l) l[r * pitch] = v;
            if (glb) glb[go + (unsigned)(r * N)] = v;
        }
    }
};
template <int OW, int S>
struct EpiBwd {                 // dx = mask > 0 ? acc : 0 at pixel (oh*S + ph, ow*S + pw) of a Wfull-wide NHWC image; optional padded LDS image
    static constexpr bool PER_ELEM = true;
    static constexpr bool TR = false;
    static constexpr bool ROW4 = false;
    A0_D void emit4(int, int, const a0_acc4&, float) const {}
    const float* mask; float* dst; float* lds; int pitch, rp, Wfull, ph, pw, N;
    A0_D unsigned gi(int m, int n) const { const int oh = m / OW, ow = m - oh * OW; return (unsigned)(((oh * S + ph) * Wfull + ow * S + pw) * N + n); }
    A0_D float pre_col(int) const { return 0.f; }
    A0_D float pre_elem(int m, int n) const { return mask[gi(m, n)]; }
    A0_D void emit(int m, int n, float acc, float pre) const {
        const float v = pre > 0.f ? acc : 0.f;
        dst[gi(m, n)] = v;
        if (lds) { const int oh = m / OW; lds[oh * rp + (m - oh * OW) * pitch + n] = v; }
    }
};

// What a layer's epilogue needs from global memory, in the wave's tile layout.  Loaded well ahead of the layer (kernel start, or the
// previous layer's `between` slot) so that its latency is never waited for on its own.
// Epilogues with TR = true receive TRANSPOSED accumulators (the split-operand stages issue their MFMAs with the operands swapped, weights
// as A and activations as B): a lane then holds four consecutive output CHANNELS n0 .. n0+3 (n0 = 4 * (lane >> 4) within the 16-column
// block) of ONE output position m = lane & 15, instead of four positions of one channel — so one bf16 term of its four values is a
// single 8-byte LDS write (was four 2-byte writes), its fp32 values one 16-byte global store, and a ReLU mask one 16-byte load.
template <int N, int WN, int MBW, class EPI>
struct a0_pre {
    static constexpr int NBW = N / 16 / WN, WMG = A0_FUSED_WAVES / WN;
    float pc[NBW][EPI::TR ? 4 : 1], pe[EPI::PER_ELEM ? MBW : 1][NBW][4];
    A0_D void load(const EPI& epi, int M) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wn = wave % WN, wmg = wave / WN, q = lane >> 4, r16 = lane & 15;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            if constexpr (EPI::TR) {
                const int n0 = (wn * NBW + j) * 16 + 4 * q;
#pragma unroll
                for (int r = 0; r < 4; ++r) pc[j][r] = epi.pre_col(n0 + r);
                if constexpr (EPI::PER_ELEM) {
#pragma unroll
                    for (int i = 0; i < MBW; ++i) {
                        const int m = (wmg + i * WMG) * 16 + r16;
                        const a0_f4 v = epi.pre_elem4(m < M ? m : M - 1, n0);
                        pe[i][j][0] = v.x; pe[i][j][1] = v.y; pe[i][j][2] = v.z; pe[i][j][3] = v.w;
                    }
                }
            } else {
                const int n = (wn * NBW + j) * 16 + r16;
                pc[j][0] = epi.pre_col(n);
                if constexpr (EPI::PER_ELEM) {
#pragma unroll
                    for (int i = 0; i < MBW; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = (wmg + i * WMG) * 16 + 4 * q + r;
                            pe[i][j][r] = epi.pre_elem(m < M ? m : M - 1, n);
                        }
                }
            }
        }
    }
};

// One convolution-shaped GEMM: C[M x N] = A[M x K] * B, A read through AF from LDS, B from the register ring.  Waves: WN along
// N, 8/WN along M; a wave owns MBW 16-row blocks (interleaved) x NBW 16-column blocks.  (conv1 2 x 4 waves, conv2 / conv3 4 x 2.)
// The k loop is one software pipeline over all K/4 MFMA steps: the LDS operands of step g + PD are requested before the MFMAs of
// step g are issued (register ring of PD + 1 slots); chunk c + R of the weights is requested as soon as chunk c has been consumed.
// Everything is branch-free (16-row blocks beyond M recompute row 0 and are never stored) and pinned with sched_barriers — left
// alone, the scheduler sinks every prefetch down to its first use.  `between` runs after the last MFMA and before the epilogue:
// the caller issues the next layer's weight prologue there, so its L2 latency hides behind the epilogue and the barrier.
template <int N, int WN, int MBW, int PD, int R, int MBWP, class AF, class EPI, class Between>
A0_D void a0_conv_stage(const AF& af, int M, int K, a0_wring<N, WN, R>& ring, const EPI& epi, const a0_pre<N, WN, MBWP, EPI>& pre, Between&& between) {
    static_assert(MBWP >= MBW, "prefetched epilogue values cover the wave's blocks");
    constexpr int NB = N / 16, NBW = NB / WN, WMG = A0_FUSED_WAVES / WN, RS = PD + 1;
    static_assert(NBW >= 1 && (4 % RS) == 0, "tile shape");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wmg = wave / WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int MB = (M + 15) >> 4, NCH = K >> 4;      // NCH is a multiple of R for every supported shape (checked on the host)

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

    typename AF::Raw a[RS][MBW];
    auto fetch = [&](int slot, int ao, int j) {
#pragma unroll
        for (int i = 0; i < MBW; ++i) a[slot][i] = af.load(rows[i] + ao + af.step_off(j));
    };
    {
        const int ao = af.chunk_off(0);
#pragma unroll
        for (int p = 0; p < PD; ++p) fetch(p, ao, p);
    }
    // Drain vmcnt once, here: the ring's R prologue loads were issued together (one latency) and the previous layer's activation
    // stores may still be in flight.  gfx9 counts loads and stores in the same counter and lets them complete out of order, so
    // with a store pending the compiler must answer every later "is chunk c here?" with vmcnt(0) — which would also wait for the
    // R-1 younger refills.  From an empty counter on, only in-order loads are pending and the waits inside the loop are counted.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll 1      // rolled on purpose: unrolled, every refill address becomes a hoisted 64-bit loop invariant (70 register pairs)
    for (int cb = 0; cb < NCH; cb += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int c = cb + u;
            const int ao_c = af.chunk_off(c), ao_n = af.chunk_off(c + 1 < NCH ? c + 1 : c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j + PD < 4) fetch((j + PD) % RS, ao_c, j + PD);
                else fetch((j + PD) % RS, ao_n, j + PD - 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MBW; ++i)
#pragma unroll
                    for (int jn = 0; jn < NBW; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF::value(a[j % RS][i]), a0_f4_get(ring.v[u][jn], j), acc[i][jn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            ring.fill(u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    between();
    // C/D layout of v_mfma_f32_16x16x4_f32: column = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int mb = wmg + i * WMG;
        if (mb < MB) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const int n = (wn * NBW + j) * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb * 16 + 4 * q + r;
                    if (m < M) epi.emit(m, n, acc[i][j][r], EPI::PER_ELEM ? pre.pe[EPI::PER_ELEM ? i : 0][j][r] : pre.pc[j][0]);
                }
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ conv1 on the bf16 pipe
typedef __bf16 a0_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t a0_u32x4 __attribute__((ext_vector_type(4)));

// conv1 weights: three exact bf16 terms of fl(w/255), fragment-major: uint4 index ((t*32 + n)*4 + q)*3 + s holds the eight k =
// 32t + 8q .. + 7 of output channel n, term s — the B fragment of lane (n, q) for MFMA step t; a lane's three terms are adjacent.
template <int R>
struct a0_wring1 {
    uint4 v[R][3];
    const uint4* base;
    const uint4* p;
    int nst, left;
    A0_D void init(const float* wp, int C) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        base = (const uint4*)wp + (((wave % 2) * 16 + (lane & 15)) * 4 + (lane >> 4)) * 3;
        nst = 2 * C;
    }
    A0_D void fill(int slot) {
#pragma unroll
        for (int s = 0; s < 3; ++s) v[slot][s] = p[s];
        const bool more = left > 1;
        p += more ? 3 * 128 : 0;
        left -= more ? 1 : 0;
    }
    A0_D void prologue() {
        p = base; left = nst;
#pragma unroll
        for (int u = 0; u < R; ++u) fill(u);
    }
};

// conv1 8x8/4 over the bf16 image [C][H][W] in LDS.  One MFMA step = 32 k = kernel rows kh = 4(t&1) + q (q = lane >> 4), all eight
// kw, of channel t >> 1: the A fragment of lane (row m, q) is the 8 consecutive pixels img[c][4oh + kh][4ow .. 4ow + 7] (16 bytes,
// 8-byte aligned: two ds_read_b64).  Waves: 2 along N x 4 along M, MBW 16-row blocks each; same pipeline discipline as a0_conv_stage.
template <int MBW, int R, int WC, int MBWP, class EPI, class Between>
A0_D void a0_conv1_stage(const uint16_t* img, int HW, int Wrt, int W1, int M, a0_wring1<R>& ring, const EPI& epi, const a0_pre<32, 2, MBWP, EPI>& pre,
                         Between&& between) {
    static_assert((R % 2) == 0 && MBWP >= MBW, "ring shape");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % 2, wmg = wave / 2;
    const int q = lane >> 4, r16 = lane & 15;
    const int W = WC > 0 ? WC : Wrt;
    const int MB = (M + 15) >> 4, NST = ring.nst;
    int rows[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        int m = (wmg + i * A0_WMG1) * 16 + r16;
        m = m < M ? m : 0;
        const int oh = m / W1, ow = m - oh * W1;
        rows[i] = (4 * oh + q) * W + 4 * ow;
    }
    a0_acc4 acc[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) acc[i] = a0_acc4{0.f, 0.f, 0.f, 0.f};
    uint2 a[2][MBW][2];
    auto fetch = [&](int slot, int off) {
#pragma unroll
        for (int i = 0; i < MBW; ++i) {
            a[slot][i][0] = *(const uint2*)(img + rows[i] + off);
            a[slot][i][1] = *(const uint2*)(img + rows[i] + off + 4);
        }
    };
    fetch(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see a0_conv_stage
#pragma unroll 1
    for (int tb = 0; tb < NST; tb += R) {
        const int base = (tb >> 1) * HW;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const bool last = tb + u + 1 >= NST;     // past the end: re-read the current step (never consumed)
            const int un = u + 1;
            fetch(un & 1, last ? base + (u >> 1) * HW + (u & 1) * 4 * W : base + (un >> 1) * HW + (un & 1) * 4 * W);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < MBW; ++i) {
                    const a0_u32x4 av = {a[u & 1][i][0].x, a[u & 1][i][0].y, a[u & 1][i][1].x, a[u & 1][i][1].y};
                    const a0_u32x4 bv = {ring.v[u][s].x, ring.v[u][s].y, ring.v[u][s].z, ring.v[u][s].w};
                    if constexpr (EPI::TR) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(a0_bf16x8, bv), __builtin_bit_cast(a0_bf16x8, av), acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(a0_bf16x8, av), __builtin_bit_cast(a0_bf16x8, bv), acc[i], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            ring.fill(u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    between();
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int mb = wmg + i * A0_WMG1;
        if (mb < MB) {
            if constexpr (EPI::TR) {
                const int m = mb * 16 + r16;
                if (m < M) epi.emit_n4(m, wn * 16 + 4 * q, acc[i], pre.pc[0]);
            } else {
                const int n = wn * 16 + r16;
                if (EPI::ROW4 && mb * 16 + 16 <= M) {
                    epi.emit4(mb * 16 + 4 * q, n, acc[i], pre.pc[0][0]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = mb * 16 + 4 * q + r;
                        if (m < M) epi.emit(m, n, acc[i][r], pre.pc[0][0]);
                    }
                }
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ conv2 / conv3 on the bf16 pipe ("x9")
// fp32 x fp32 products without rounding an operand: BOTH factors are split exactly into three bf16 terms (a = a0 + a1 + a2, 8 + 8 + 8
// mantissa bits; the activations by the producing layer's epilogue, the weights by a0_conv_wt_kernel), every cross product ai * bj is
// exact in fp32, and nine v_mfma_f32_16x16x32_bf16 per 32 k (144 matrix-pipe cycles) replace eight v_mfma_f32_16x16x4_f32 (256).
// The sum is accumulated in fp32 like the fmaf chain; against fp64 it is as close (tools/check_bf16x9.hip: 4.8e-7 vs 3.3e-7 of the
// scale at K = 512).  Activations live in LDS as three bf16 planes [term][pixel][channels + 8]; a lane's A fragment is the 8
// consecutive channels 8q .. 8q+7 of one pixel: one aligned 16-byte read per term.
template <int N, int WN, int R, int WK = 1>
struct a0_wring9 {          // uint4 index ((t*N + n)*4 + q)*3 + s: the eight k = 32t + 8q .. +7 of output channel n, term s
    static constexpr int NBW = N / 16 / WN;
    uint4 v[R][NBW][3];
    const uint4* base;
    const uint4* p;
    int nst, left, stride;
    // WK > 1 (a0_conv_stage_x9k): the waves wave / WN = 0 .. WK-1 share the k steps round-robin; this ring walks steps wk, wk + WK, ...
    A0_D void init(const float* wp, int K) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        base = (const uint4*)wp + (((wave % WN) * (NBW * 16) + (lane & 15)) * 4 + (lane >> 4)) * 3 + (WK > 1 ? (wave / WN) * (N * 12) : 0);
        nst = (K >> 5) / WK;
        stride = N * 12 * WK;
    }
    // explicit form: `first` = this lane's fragment of the wave's first own step, `step` = uint4 between two of its own steps
    A0_D void init_at(const uint4* first, int own_steps, int step) { base = first; nst = own_steps; stride = step; }
    A0_D void fill(int slot) {
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int s = 0; s < 3; ++s) v[slot][j][s] = p[j * 192 + s];
        const bool more = left > 1;
        p += more ? stride : 0;
        left -= more ? 1 : 0;
    }
    A0_D void prologue() {
        p = base; left = nst;
#pragma unroll
        for (int u = 0; u < R; ++u) fill(u);
    }
};
template <int TERM>
struct AF2X {   // conv2 4x4/2 over act1 planes [pixel = h*W1 + w][P]; MFMA step = tap (kh, kw), all 32 channels
    static constexpr int term = TERM;       // elements per term plane: compile-time, so the three term reads of a fragment differ only in the DS immediate
    const uint16_t* planes; int RP, W2, P;          // RP: elements per image row (W1 * P + pad)
    A0_D int row(int m) const { const int oh = m / W2, ow = m - oh * W2; return (2 * oh) * RP + 2 * ow * P; }
    A0_D int step_off(int st) const { return (st >> 2) * RP + (st & 3) * P; }
};
template <int TERM>
struct AF3X {   // conv3 3x3/1 over act2 planes [pixel][P]; MFMA step = half (32 channels) of tap st >> 1
    static constexpr int term = TERM;
    const uint16_t* planes; int RP, W3, P;          // RP: elements per image row (W2 * P + pad)
    A0_D int row(int m) const { const int oh = m / W3, ow = m - oh * W3; return oh * RP + ow * P; }
    A0_D int step_off(int st) const { const int tap = st >> 1; return (tap / 3) * RP + (tap % 3) * P + 32 * (st & 1); }
};
// exact three-term bf16 split of four fp32 values, packed: hi / mid / lo as four bf16 (8 bytes) each
A0_D void a0_split4(const float (&v)[4], uint2& hi, uint2& mid, uint2& lo) {
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r] = __float_as_uint(v[r]);
        const float r1 = v[r] - __uint_as_float(h[r] & 0xffff0000u);          // exact: at most 16 significant bits left
        m[r] = __float_as_uint(r1);
        l[r] = __float_as_uint(r1 - __uint_as_float(m[r] & 0xffff0000u));      // exact: at most 8 significant bits left
    }
    hi = uint2{__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};      // v_perm_b32: the upper halves side by side (the truncation itself)
    mid = uint2{__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
    lo = uint2{__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
}
template <int OWC, int P, int RP, int term>
struct EpiFwdX {            // y = relu(acc + bias[n]) -> three bf16 planes in LDS [term][oh][ow][P], image rows RP apart (+ fp32 to global [m][N]); transposed accumulators
    static constexpr bool PER_ELEM = false;
    static constexpr bool ROW4 = false;
    static constexpr bool TR = true;
    static_assert(OWC > 0 && (P % 4) == 0 && (RP % 4) == 0 && (term % 4) == 0, "output width known at compile time; 8-byte aligned plane writes");
    const float* bias; uint16_t* planes; float* glb; int N;          // bias: the workgroup's LDS copy (one 16-byte read per block; kept in registers across the launch the biases cost 12 VGPRs the kernel does not have)
    A0_D float pre_col(int) const { return 0.f; }
    A0_D void emit_n4(int m, int n0, const a0_acc4& acc, const float*) const {       // channels n0 .. n0+3 of output position m
        const a0_f4 bv = *(const a0_f4*)(bias + n0);
        float v[4] = {acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] < 0.f) ? 0.f : v[r];
        if (glb) *(a0_f4*)(glb + (unsigned)(m * N + n0)) = a0_f4{v[0], v[1], v[2], v[3]};
        uint2 hi, mid, lo;
        a0_split4(v, hi, mid, lo);
        const int oh = m / OWC;
        uint16_t* d = planes + oh * RP + (m - oh * OWC) * P + n0;
        *(uint2*)d = hi; *(uint2*)(d + term) = mid; *(uint2*)(d + 2 * term) = lo;
    }
};
struct EpiFwdT {            // y = relu(acc + bias[n]) -> fp32 global [m][N] (conv3 of the split-operand path); transposed accumulators
    static constexpr bool PER_ELEM = false;
    static constexpr bool ROW4 = false;
    static constexpr bool TR = true;
    const float* bias; float* glb; int N;          // bias: LDS copy, as in EpiFwdX
    A0_D float pre_col(int) const { return 0.f; }
    A0_D void emit_n4(int m, int n0, const a0_acc4& acc, const float*) const {
        const a0_f4 bv = *(const a0_f4*)(bias + n0);
        float v[4] = {acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] < 0.f) ? 0.f : v[r];
        *(a0_f4*)(glb + (unsigned)(m * N + n0)) = a0_f4{v[0], v[1], v[2], v[3]};
    }
};

template <int ORD, int N, int WN, int MBW, int R, int MBWP, class AFX, class EPI, class Between>
A0_D void a0_conv_stage_x9(const AFX& af, int M, a0_wring9<N, WN, R>& ring, const EPI& epi, const a0_pre<N, WN, MBWP, EPI>& pre, Between&& between) {
    constexpr int NB = N / 16, NBW = NB / WN, WMG = A0_FUSED_WAVES / WN;
    static_assert(NBW >= 1 && (R % 2) == 0 && MBWP >= MBW, "tile shape");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wmg = wave / WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int MB = (M + 15) >> 4, NST = ring.nst;       // NST is a multiple of R for every supported shape
    int rows[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int m = (wmg + i * WMG) * 16 + r16;
        rows[i] = af.row(m < M ? m : 0) + 8 * q;
    }
    a0_acc4 acc[MBW][NBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = a0_acc4{0.f, 0.f, 0.f, 0.f};
    uint4 a[2][MBW][3];
    auto fetch = [&](int slot, int off) {
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[slot][i][t] = *(const uint4*)(af.planes + rows[i] + off + t * AFX::term);
    };
    fetch(0, af.step_off(0));
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see a0_conv_stage
#pragma unroll 1
    for (int tb = 0; tb < NST; tb += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int nxt = tb + u + 1;
            fetch((u + 1) & 1, af.step_off(nxt < NST ? nxt : NST - 1));          // past the end: re-read the last step (never consumed)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ta = 0; ta < 3; ++ta)
#pragma unroll
                for (int tw = 0; tw < 3; ++tw)
#pragma unroll
                    for (int i = 0; i < MBW; ++i)
#pragma unroll
                        for (int jn = 0; jn < NBW; ++jn) {
                            if (ta + tw > ORD) continue;      // see A0_X9_MAXORD
                            const a0_u32x4 av = {a[u & 1][i][ta].x, a[u & 1][i][ta].y, a[u & 1][i][ta].z, a[u & 1][i][ta].w};
                            const a0_u32x4 bv = {ring.v[u][jn][tw].x, ring.v[u][jn][tw].y, ring.v[u][jn][tw].z, ring.v[u][jn][tw].w};
                            if constexpr (EPI::TR) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(a0_bf16x8, bv), __builtin_bit_cast(a0_bf16x8, av), acc[i][jn], 0, 0, 0);
                            else acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(a0_bf16x8, av), __builtin_bit_cast(a0_bf16x8, bv), acc[i][jn], 0, 0, 0);
                        }
            __builtin_amdgcn_sched_barrier(0);
            ring.fill(u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    between();
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int mb = wmg + i * WMG;
        if (mb < MB) {
            if constexpr (EPI::TR) {
                const int m = mb * 16 + r16;
                if (m < M) {
#pragma unroll
                    for (int j = 0; j < NBW; ++j) epi.emit_n4(m, (wn * NBW + j) * 16 + 4 * q, acc[i][j], EPI::PER_ELEM ? pre.pe[EPI::PER_ELEM ? i : 0][j] : pre.pc[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    const int n = (wn * NBW + j) * 16 + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = mb * 16 + 4 * q + r;
                        if (m < M) epi.emit(m, n, acc[i][j][r], EPI::PER_ELEM ? pre.pe[EPI::PER_ELEM ? i : 0][j][r] : pre.pc[j][0]);
                    }
                }
            }
        }
    }
    __syncthreads();
}
// N-stationary form of a0_conv_stage_x9 for the forward stages (N = 64: four waves along N, the two groups of four along K).  In the form
// above the two groups of waves along M both stream the complete weight set (1 MB per observation and CU through the vector L1), and the
// kernel is more sensitive to that stream than to anything else (profiles/r02_encoder_experiments.md: half the weight loads = -8 %, half the
// LDS fragment reads = -2 %).  Here wave (wn, wk) owns column block wn, ALL 16-row blocks and the k steps wk, wk + 2, ...: every weight
// fragment is loaded by exactly one wave.  A k step is processed in two halves of MBT / 2 row blocks (the A fragments of the next half are in
// flight while this one's MFMAs issue: same register ring as above).  At the end the two waves of a column block exchange partial sums
// through LDS — wave wk finishes row-block half wk, so each sends the other's half — which costs one barrier and 2 x 4 KB x MBT of LDS traffic.
// `xch`: 8 * MBT / 2 KB of LDS nobody reads any more when the first wave leaves its k loop.  Association of the k sum: (even steps) + (odd steps).
template <int ORD, int N, int WN, int MBT, int R, int LR, class AFX, class EPI, class Between>
A0_D void a0_conv_stage_x9k(const AFX& af, int M, a0_wring9<N, WN, R, A0_FUSED_WAVES / WN>& ring, const EPI& epi, float* xch0, float* xch1, Between&& between) {
    constexpr int WK = A0_FUSED_WAVES / WN, HB0 = (MBT + 1) / 2, HB1 = MBT / 2;       // row blocks of the two halves of a k step: [0, HB0) and [HB0, MBT)
    static_assert(WK == 2 && N == 16 * WN && HB1 >= 1 && EPI::TR && LR >= 1 && LR <= 16, "tile shape");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wk = wave / WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int MB = (M + 15) >> 4, NSTW = ring.nst;       // NSTW is a multiple of R for every supported shape
    int rows[MBT];
#pragma unroll
    for (int i = 0; i < MBT; ++i) {
        const int m = i * 16 + r16;
        rows[i] = af.row(m < M ? m : 0) + 8 * q;
    }
    a0_acc4 acc[MBT];
#pragma unroll
    for (int i = 0; i < MBT; ++i) acc[i] = a0_acc4{0.f, 0.f, 0.f, 0.f};
    uint4 a[2][HB0][3];
    auto fetch = [&](int slot, int h, int off) {
#pragma unroll
        for (int i = 0; i < HB0; ++i)
            if (h * HB0 + i < MBT) {
#pragma unroll
                for (int t = 0; t < 3; ++t) a[slot][i][t] = *(const uint4*)(af.planes + rows[h * HB0 + i] + off + t * AFX::term);
            }
    };
    fetch(0, 0, af.step_off(wk));
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see a0_conv_stage
#pragma unroll 1
    for (int tb = 0; tb < NSTW; tb += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int st = wk + WK * (tb + u);
            const int off = af.step_off(st), offn = af.step_off(tb + u + 1 < NSTW ? st + WK : st);      // past the end: re-read the last step (never consumed)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 0) fetch(1, 1, off);
                else fetch(0, 0, offn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ta = 0; ta < 3; ++ta)
#pragma unroll
                    for (int tw = 0; tw < 3; ++tw)
#pragma unroll
                        for (int i = 0; i < HB0; ++i) {
                            if (ta + tw > ORD || h * HB0 + i >= MBT) continue;      // see A0_X9_MAXORD
                            const a0_u32x4 av = {a[h][i][ta].x, a[h][i][ta].y, a[h][i][ta].z, a[h][i][ta].w};
                            const a0_u32x4 bv = {ring.v[u][0][tw].x, ring.v[u][0][tw].y, ring.v[u][0][tw].z, ring.v[u][0][tw].w};
                            acc[h * HB0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(a0_bf16x8, bv), __builtin_bit_cast(a0_bf16x8, av), acc[h * HB0 + i], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            ring.fill(u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    between();
    // xch0: WN * HB0 KB for the partial sums of row-block half 0 (written by the wk = 1 waves), xch1: the same for half 1 (written by
    // wk = 0).  Only the first LR rows of the last row block exist (M = 16 * (MBT - 1) + LR): with transposed accumulators a lane holds
    // row lane & 15, so that block travels as 4 * LR lanes instead of 64 (conv2, LR = 1: 64 bytes per wave instead of 1 KB — which is
    // what lets both regions fit beside the stage's live data, no barrier before the exchange).
    a0_acc4* x0 = (a0_acc4*)xch0;
    a0_acc4* x1 = (a0_acc4*)xch1;
    constexpr int P1 = (HB1 - 1) * 64 + 4 * LR;     // float4 slots per wave in region 1
    const int last = q * LR + r16;                  // slot of this lane's values of the last block (lanes with r16 < LR)
    const int n0 = wn * 16 + 4 * q;
    // H = row-block half this wave finishes: compile-time in the lambdas, so the accumulators stay in registers.  Per-element epilogue
    // inputs (ReLU masks) of that half are requested before the exchange and used after it.
    float pe[EPI::PER_ELEM ? HB0 : 1][4];
    auto send = [&](auto mine) {
        constexpr int H = decltype(mine)::value;
        if constexpr (EPI::PER_ELEM) {
#pragma unroll
            for (int i = 0; i < (H == 0 ? HB0 : HB1); ++i) {
                const int m = (H * HB0 + i) * 16 + r16;
                const a0_f4 v = epi.pre_elem4(m < M ? m : M - 1, n0);
                pe[i][0] = v.x; pe[i][1] = v.y; pe[i][2] = v.z; pe[i][3] = v.w;
            }
        }
        if constexpr (H == 1) {
#pragma unroll
            for (int i = 0; i < HB0; ++i) x0[(wn * HB0 + i) * 64 + lane] = acc[i];
        } else {
#pragma unroll
            for (int i = 0; i < HB1; ++i) {
                if (i < HB1 - 1) x1[wn * P1 + i * 64 + lane] = acc[HB0 + i];
                else if (r16 < LR) x1[wn * P1 + (HB1 - 1) * 64 + last] = acc[HB0 + i];
            }
        }
    };
    auto recv = [&](auto mine) {
        constexpr int H = decltype(mine)::value;
#pragma unroll
        for (int i = 0; i < (H == 0 ? HB0 : HB1); ++i) {
            const int mb = H * HB0 + i;
            const int m = mb * 16 + r16;
            if constexpr (H == 0) acc[mb] += x0[(wn * HB0 + i) * 64 + lane];
            else {
                if (i < HB1 - 1) acc[mb] += x1[wn * P1 + i * 64 + lane];
                else if (r16 < LR) acc[mb] += x1[wn * P1 + (HB1 - 1) * 64 + last];
            }
            if (mb < MB && m < M) epi.emit_n4(m, n0, acc[mb], EPI::PER_ELEM ? pe[EPI::PER_ELEM ? i : 0] : nullptr);
        }
    };
    if (wk == 0) send(std::integral_constant<int, 0>{});
    else send(std::integral_constant<int, 1>{});
    __syncthreads();
    if (wk == 0) recv(std::integral_constant<int, 0>{});
    else recv(std::integral_constant<int, 1>{});
    __syncthreads();
}
constexpr int A0_P1X = 40, A0_P2X = 80;       // pixel pitches (bf16 elements) of the act1 / act2 term planes: channels + 8 / + 16
// Image-row pitches of the term planes (84x84 geometry).  A fragment read is a ds_read_b128 whose 16-lane groups are {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+32); along one image row the pixel pitches above put a group's 16 reads on 16 different 16-byte slots of the
// 256-byte bank row, but a 16-row block spans two or three image rows, and with rows packed back to back the reads behind a row wrap
// fell onto slots already taken: 8.0 (conv2) and 7.0 (conv3) LDS cycles per wave-read instead of 4.  The pads below are the smallest that
// bring the average over all blocks and taps down to 4.7 / 5.0 cycles (tools/lds_bank_model.py enumerates the layouts).
#ifndef A0_PADS
#define A0_PADS 1                             // tuning aid: 0 = image rows packed back to back (the round-1 layout)
#endif
constexpr int A0_RP1X = 20 * A0_P1X + (A0_PADS ? 8 : 0);      // act1 planes: 20 pixels per row + 16 bytes
constexpr int A0_RP2X = 9 * A0_P2X + (A0_PADS ? 96 : 0);      // act2 planes: 9 pixels per row + 192 bytes
#ifndef A0_RX2_D
#define A0_RX2_D 4
#endif
#ifndef A0_RX3_D
#define A0_RX3_D 6
#endif
[[maybe_unused]] constexpr int A0_RX2 = A0_RX2_D, A0_RX3 = A0_RX3_D;         // 32-k steps of split weights in flight (tuning aids: -DA0_RX2_D / -DA0_RX3_D)
#ifndef A0_KSPLIT_D
#define A0_KSPLIT_D 1                         // 1: N-stationary stages in the data-gradient kernel too (two stride phases per stage)
#endif
#ifndef A0_KSPLIT
#define A0_KSPLIT 1                           // 1: N-stationary conv2 / conv3 stages of the forward kernel (a0_conv_stage_x9k); 0: M x N wave tiling (a0_conv_stage_x9)
#endif
#ifndef A0_WNX
#define A0_WNX 4                              // waves along N in the split-operand conv2 / conv3 stages
#endif

// Register-ring depths (16-k chunks in flight per wave): >= 2 us of MFMA work ahead of every weight load.
constexpr int A0_R1 = 4, A0_R2 = 4, A0_R3 = 6;     // conv1: 32-k steps (three 16-byte terms each); conv2 / conv3: 16-k chunks

#ifndef A0_FUSED_MINWAVES
#define A0_FUSED_MINWAVES 1
#endif
template <int MBW1, int MBW2, int MBW3, int WC, bool X9, bool LOOP = false, int ORD = 4>
__global__ __launch_bounds__(A0_FUSED_THREADS, A0_FUSED_MINWAVES) void a0_encoder_fused_kernel(a0_fused_args P);

// Split-operand variant (84 x 84 geometry): all three layers on the bf16 pipe.  LDS: bf16 image [0, 56 448), act1 term planes behind it;
// the act2 term planes reuse the image's bytes (the image is dead once conv1 has finished).
constexpr int A0_X9_BIAS_OFF = 2 * 4 * 84 * 84 + 3 * 20 * A0_RP1X * 2;      // bf16 image + act1 term planes
#ifndef A0_RK2_D
#define A0_RK2_D 2
#endif
#ifndef A0_RK3_D
#define A0_RK3_D 3
#endif
constexpr int A0_RK2 = A0_RK2_D, A0_RK3 = A0_RK3_D;      // weight-ring depths of the N-stationary stages, in own k steps (8 resp. 9 per wave)
constexpr int A0_X9_XCH0 = 3 * 9 * A0_RP2X * 2;             // conv2 exchange region 0: behind the act2 planes, inside the (dead) image
constexpr int A0_X9_XCH1 = A0_X9_BIAS_OFF + 160 * 4;        // region 1: behind the biases (4 waves x (2 KB + 64 B))
static_assert(A0_X9_XCH0 + 4 * 3 * 1024 <= 2 * 4 * 84 * 84, "conv2 exchange region 0 fits into the image");
constexpr int A0_X9_LDS_BYTES = A0_X9_XCH1 + (A0_KSPLIT ? 4 * (2 * 64 + 4) * 16 : 0);
static_assert(A0_X9_LDS_BYTES <= 160 * 1024, "LDS");
// conv2 + conv3 of the split-operand encoder for observation b, behind conv1's epilogue (act1 term planes in LDS): shared by a0_encoder_fused_x9_body and the actor-step
// kernel that overlaps conv1 with the step's tail (a0_actor_step_enc2_kernel).
#if A0_KSPLIT
typedef a0_wring9<64, 4, A0_RK2, 2> a0_ring2_t;      // 8 own steps of conv2's 16
typedef a0_wring9<64, 4, A0_RK3, 2> a0_ring3_t;      // 9 own steps of conv3's 18
#else
typedef a0_wring9<64, A0_WNX, A0_RX2> a0_ring2_t;
typedef a0_wring9<64, A0_WNX, A0_RX3> a0_ring3_t;
#endif
template <int ORD, int MBW2, int MBW3, bool LOOP>
A0_D void a0_x9_conv23(const a0_fused_args& P, int b, unsigned char* smem, float* bias_lds, a0_wring1<A0_R1>& ring1, a0_ring2_t& ring2, a0_ring3_t& ring3) {
    const int obs_bytes = P.C * P.H * P.W;
    const int M2 = P.H2 * P.W2, M3 = P.H3 * P.W3;
    uint16_t* a1p = (uint16_t*)(smem + 2 * obs_bytes);
    constexpr int term1 = 20 * A0_RP1X, term2 = 9 * A0_RP2X;
    uint16_t* a2p = (uint16_t*)smem;
    typedef EpiFwdX<9, A0_P2X, A0_RP2X, term2> E2X;
    constexpr int WNX = A0_WNX, WMGX = A0_FUSED_WAVES / WNX;
    constexpr int MBW2X = (6 + WMGX - 1) / WMGX, MBW3X = (4 + WMGX - 1) / WMGX;
    static_assert(MBW2X <= MBW2 * 2 && MBW3X <= MBW3 * 2, "84x84 geometry");
#if !A0_KSPLIT
    a0_pre<64, WNX, MBW2X, E2X> pre2;
    a0_pre<64, WNX, MBW3X, EpiFwdT> pre3;
#endif
    (void)a1p; (void)term1;
    {
        const AF2X<term1> f2{a1p, A0_RP1X, P.W2, A0_P1X};
        const E2X e2{bias_lds + 32, a2p, P.act2 ? P.act2 + (long long)b * M2 * 64 : nullptr, 64};
#if A0_KSPLIT
        // exchange buffers, all dead while their stage's k loops run: conv2's partial sums go behind the act2 planes-to-be (the top 12 KB of the
        // image region) and into the tail of the LDS allocation, conv3's where act1 was
        const AF3X<term2> f3k{a2p, A0_RP2X, P.W3, A0_P2X};
        const EpiFwdT e3k{bias_lds + 96, P.act3 + (long long)b * M3 * 64, 64};
        a0_conv_stage_x9k<ORD, 64, 4, 6, A0_RK2, 1>(f2, M2, ring2, e2, (float*)(smem + A0_X9_XCH0), (float*)(smem + A0_X9_XCH1), [&] { ring3.prologue(); });
        a0_conv_stage_x9k<ORD, 64, 4, 4, A0_RK3, 16>(f3k, M3, ring3, e3k, (float*)a1p, (float*)a1p + 4 * 2 * 256, [&] { if (LOOP) ring1.prologue(); });
#else
        const int wmgx = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / WNX;
        constexpr bool uneven2 = MBW2X > 1 && MBW2X * WMGX > 6, uneven3 = MBW3X > 1 && MBW3X * WMGX > 4;      // some M groups own one block less
        if (uneven2 && wmgx + (MBW2X - 1) * WMGX >= 6)
            a0_conv_stage_x9<ORD, 64, WNX, (MBW2X > 1 ? MBW2X - 1 : 1), A0_RX2>(f2, M2, ring2, e2, pre2, [&] { ring3.prologue(); });
        else
            a0_conv_stage_x9<ORD, 64, WNX, MBW2X, A0_RX2>(f2, M2, ring2, e2, pre2, [&] { ring3.prologue(); });
        const AF3X<term2> f3{a2p, A0_RP2X, P.W3, A0_P2X};
        const EpiFwdT e3{bias_lds + 96, P.act3 + (long long)b * M3 * 64, 64};
        if (uneven3 && wmgx + (MBW3X - 1) * WMGX >= 4)
            a0_conv_stage_x9<ORD, 64, WNX, (MBW3X > 1 ? MBW3X - 1 : 1), A0_RX3>(f3, M3, ring3, e3, pre3, [&] { if (LOOP) ring1.prologue(); });
        else
            a0_conv_stage_x9<ORD, 64, WNX, MBW3X, A0_RX3>(f3, M3, ring3, e3, pre3, [&] { if (LOOP) ring1.prologue(); });
#endif
    }
}

// LOOP: the workgroup walks over several observations (b += gridDim.x; launches of more observations than CUs) and requests the next
// observation's conv1 weights behind conv3; without it (the actor's launches: one observation per workgroup) that request is not made.
// bid / nblk: this workgroup's index among the workgroups that serve P and their number (blockIdx.x / gridDim.x for a launch of one pass;
// a0_encoder_fused_multi_kernel hands every pass a share of the grid).
template <int ORD, int MBW1, int MBW2, int MBW3, bool LOOP>
A0_D void a0_encoder_fused_x9_body(const a0_fused_args& P, int bid, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* img = (uint16_t*)smem;
    const int obs_bytes = P.C * P.H * P.W;
    const int M1 = P.H1 * P.W1;
    uint16_t* a1p = (uint16_t*)(smem + 2 * obs_bytes);
    constexpr int term1 = 20 * A0_RP1X;
    typedef EpiFwdX<20, A0_P1X, A0_RP1X, term1> E1X;
    // conv2 / conv3 wave tiling: WNX waves along N (each owns 64 / 16 / WNX column blocks), 8 / WNX groups along M.  An A fragment read
    // from LDS feeds 9 * NBW MFMAs, and the waves along N re-read the same fragments: with four waves along N (NBW = 1) the two stages
    // moved 72 / 48 KB of LDS per 32-k step against 864 / 576 matrix-pipe cycles — LDS-bound at 128 B/clk.  Two waves along N halve that.
    // Four M groups over conv2's six 16-row blocks: groups 0, 1 own two blocks, groups 2, 3 one (the MBW - 1 instantiation); a SIMD hosts
    // one wave of each kind (wave w and w + 4), so the matrix pipes stay evenly loaded.  conv3: four blocks, one per group.
    constexpr int WNX = A0_WNX, WMGX = A0_FUSED_WAVES / WNX;
    constexpr int MBW2X = (6 + WMGX - 1) / WMGX, MBW3X = (4 + WMGX - 1) / WMGX;
    static_assert(MBW2X <= MBW2 * 2 && MBW3X <= MBW3 * 2, "84x84 geometry");
    a0_wring1<A0_R1> ring1;
#if A0_KSPLIT
    static_assert(WNX == 4 && A0_FUSED_WAVES == 8, "N-stationary stages: four waves along N, two along K");
#endif
    a0_ring2_t ring2;
    a0_ring3_t ring3;
    ring1.init(P.wt1, P.C);
    ring2.init(P.wx2, 512);
    ring3.init(P.wx3, 576);
    ring1.prologue();
    constexpr int MBW1X = (25 + A0_WMG1 - 1) / A0_WMG1;
    static_assert(A0_FUSED_WAVES != 8 || MBW1X == MBW1, "84x84 geometry");
    a0_pre<32, 2, MBW1X, E1X> pre1;           // (empty: these epilogues take nothing from global memory)
    float* bias_lds = (float*)(smem + A0_X9_BIAS_OFF);      // b1 | b2 | b3 behind the activation planes; visible after the first barrier below
    if (threadIdx.x < 160) bias_lds[threadIdx.x] = threadIdx.x < 32 ? P.b1[threadIdx.x] : threadIdx.x < 96 ? P.b2[threadIdx.x - 32] : P.b3[threadIdx.x - 96];
    // the pad channels (32..39 / 64..71) of the term planes are never read; nothing to initialise
    for (int b = bid; b < P.B; b += nblk) {
        const long long s = P.slot ? (long long)P.slot[b] : (long long)b;
        const uint4* src = (const uint4*)(P.frames + s * P.sample_stride + P.chan_off);
        constexpr int TRIPS = 4;
        const int n16 = obs_bytes >> 4;
        for (int i0 = threadIdx.x; i0 < n16; i0 += TRIPS * A0_FUSED_THREADS) {
            uint4 v[TRIPS];
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                const int i = i0 + j * A0_FUSED_THREADS;
                v[j] = src[i < n16 ? i : n16 - 1];
            }
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                int i = i0 + j * A0_FUSED_THREADS;
                i = i < n16 ? i : n16 - 1;
                const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t f0 = __float_as_uint((float)(w[k] & 0xffu)), f1 = __float_as_uint((float)((w[k] >> 8) & 0xffu));
                    const uint32_t f2 = __float_as_uint((float)((w[k] >> 16) & 0xffu)), f3 = __float_as_uint((float)(w[k] >> 24));
                    o[2 * k] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
                    o[2 * k + 1] = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
                }
                ((uint4*)img)[2 * i] = uint4{o[0], o[1], o[2], o[3]};
                ((uint4*)img)[2 * i + 1] = uint4{o[4], o[5], o[6], o[7]};
            }
        }
        __syncthreads();
        const E1X e1{bias_lds, a1p, P.act1 ? P.act1 + (long long)b * M1 * 32 : nullptr, 32};
        const int wmg1 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / 2;
        if (MBW1X > 1 && wmg1 + (MBW1X - 1) * A0_WMG1 >= ((M1 + 15) >> 4))
            a0_conv1_stage<(MBW1X > 1 ? MBW1X - 1 : 1), A0_R1, 84>(img, P.H * P.W, P.W, P.W1, M1, ring1, e1, pre1, [&] { ring2.prologue(); });
        else
            a0_conv1_stage<MBW1X, A0_R1, 84>(img, P.H * P.W, P.W, P.W1, M1, ring1, e1, pre1, [&] { ring2.prologue(); });
        a0_x9_conv23<ORD, MBW2, MBW3, LOOP>(P, b, smem, bias_lds, ring1, ring2, ring3);
        if constexpr (!LOOP) break;
    }
}

// Round 5: ONE launch per actor step behind the fc1 GEMM instead of two.  A kernel boundary on this part costs 4-9 us of ramp-up and drain (a reduction kernel that
// moves 4 MB takes 4.8 us; the PMC's wave lifetimes put ~17 us of waves inside the one-observation encoder's 25.7 us), and the actor's step crossed three of them:
// encoder | fc1 GEMM | tail + env step.  The tail + env step of step t and the encoder of step t + 1 are both one workgroup of 512 threads PER ENV, and the encoder's
// input is exactly what that workgroup has just written (the env's new observation): so the same workgroup goes on — tail, env step, barrier, encoder — and a step
// is fc1 GEMM | tail + env step + next encoder.  Same device functions as the two kernels it replaces (a0_actor_qhead_env_body, a0_encoder_fused_x9_body): the same
// bytes and the same features (tests/test_gpu_trainer.py); a rollout's first step still needs the encoder on its own, and its last step the tail without one.
#include "actor_tail.h"
// (the encoder's arguments travel as five pointers: the 4 x 84 x 84 geometry is a compile-time constant here, and the two full argument structs together do not fit
// the scalar registers)
struct a0_step_enc_args { const float *wt, *b1, *b2, *b3; float* act3; };
template <int ORD>
__global__ __launch_bounds__(A0_FUSED_THREADS, A0_FUSED_MINWAVES) void a0_actor_step_enc_kernel(a0_qenv_args Q, a0_step_enc_args N) {
    __shared__ float raw[64];
    __shared__ int s_chase_cell;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the encoder's pointers wait in VECTOR registers while the tail runs (the tail's own uniform values fill the scalar file) and come back as scalars behind it
    unsigned long long v_wt = (unsigned long long)N.wt, v_b1 = (unsigned long long)N.b1, v_b2 = (unsigned long long)N.b2, v_b3 = (unsigned long long)N.b3,
                       v_a3 = (unsigned long long)N.act3, v_obs = (unsigned long long)Q.obs_out;
    A0_TO_VGPR(v_wt); A0_TO_VGPR(v_b1); A0_TO_VGPR(v_b2); A0_TO_VGPR(v_b3); A0_TO_VGPR(v_a3); A0_TO_VGPR(v_obs);
    int v_E = Q.E;
    A0_TO_VGPR(v_E);
    a0_actor_qhead_env_body(Q, raw, (float*)smem, &s_chase_cell);
    __syncthreads();      // the frame waves' stores of obs_out[e] have been acknowledged (vmcnt(0) ahead of the barrier) and the tail's LDS is dead
    auto uni = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull));
    };
    const float* wt = (const float*)uni(v_wt);
    a0_fused_args P;
    P.frames = (const uint8_t*)uni(v_obs); P.slot = nullptr; P.sample_stride = 4 * 84 * 84; P.chan_off = 0;
    P.wt1 = wt; P.wt2 = wt + 48LL * 4 * 64; P.wt3 = P.wt2 + 64LL * 512;
    P.wx2 = P.wt3 + 64LL * 576 + 64LL * 576 + 4LL * 32 * 256; P.wx3 = P.wx2 + 96LL * 512;
    P.b1 = (const float*)uni(v_b1); P.b2 = (const float*)uni(v_b2); P.b3 = (const float*)uni(v_b3);
    P.act1 = nullptr; P.act2 = nullptr; P.act3 = (float*)uni(v_a3); P.B = __builtin_amdgcn_readfirstlane(v_E);
    P.C = 4; P.H = 84; P.W = 84; P.H1 = 20; P.W1 = 20; P.H2 = 9; P.W2 = 9; P.H3 = 7; P.W3 = 7;
    P.off_act1 = 0; P.off_act2 = 0; P.off_end = 0; P.rp1 = 0; P.rp2 = 0;      // (the fp32-chain variants' LDS layout: not used by the split-operand body)
    a0_encoder_fused_x9_body<ORD, 7, 3, 2, false>(P, (int)blockIdx.x, (int)gridDim.x);
}

// The distributional heads' step (c51 / qr): a0_actor_dist_tail_env_kernel's body (head slab sum, dueling, expectation or quantile mean, first maximum, epsilon-greedy,
// env step, replay row), then the same encoder phase.
template <int ORD>
__global__ __launch_bounds__(A0_FUSED_THREADS, A0_FUSED_MINWAVES) void a0_actor_dist_step_enc_kernel(a0_dtenv_args Q, a0_step_enc_args N) {
    __shared__ int s_chase_cell;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long v_wt = (unsigned long long)N.wt, v_b1 = (unsigned long long)N.b1, v_b2 = (unsigned long long)N.b2, v_b3 = (unsigned long long)N.b3,
                       v_a3 = (unsigned long long)N.act3, v_obs = (unsigned long long)Q.obs_out;
    A0_TO_VGPR(v_wt); A0_TO_VGPR(v_b1); A0_TO_VGPR(v_b2); A0_TO_VGPR(v_b3); A0_TO_VGPR(v_a3); A0_TO_VGPR(v_obs);
    int v_E = Q.E;
    A0_TO_VGPR(v_E);
    a0_actor_dist_tail_env_body(Q, (float*)smem, &s_chase_cell);
    __syncthreads();
    auto uni = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull));
    };
    const float* wt = (const float*)uni(v_wt);
    a0_fused_args P;
    P.frames = (const uint8_t*)uni(v_obs); P.slot = nullptr; P.sample_stride = 4 * 84 * 84; P.chan_off = 0;
    P.wt1 = wt; P.wt2 = wt + 48LL * 4 * 64; P.wt3 = P.wt2 + 64LL * 512;
    P.wx2 = P.wt3 + 64LL * 576 + 64LL * 576 + 4LL * 32 * 256; P.wx3 = P.wx2 + 96LL * 512;
    P.b1 = (const float*)uni(v_b1); P.b2 = (const float*)uni(v_b2); P.b3 = (const float*)uni(v_b3);
    P.act1 = nullptr; P.act2 = nullptr; P.act3 = (float*)uni(v_a3); P.B = __builtin_amdgcn_readfirstlane(v_E);
    P.C = 4; P.H = 84; P.W = 84; P.H1 = 20; P.W1 = 20; P.H2 = 9; P.W2 = 9; P.H3 = 7; P.W3 = 7;
    P.off_act1 = 0; P.off_act2 = 0; P.off_end = 0; P.rp1 = 0; P.rp2 = 0;
    a0_encoder_fused_x9_body<ORD, 7, 3, 2, false>(P, (int)blockIdx.x, (int)gridDim.x);
}

// ---- Round 5, second form: the step's tail BESIDE conv1 (scalar heads).  In a0_actor_step_enc_kernel seven waves copy frames and then wait for wave 0's latency chain
// (slab sums, head, Philox, n-step bookkeeping) before anyone touches the matrix pipe.  But three of the new observation's four channels are the OLD stack's frames
// 1..3 — known when the kernel starts — and conv1's reduction runs channel by channel (MFMA steps 2c, 2c + 1 = channel c): so the frame waves put those channels
// into the LDS image themselves (from the registers they copy the stack with) and run conv1's steps 0..5 while wave 0 is in the tail; behind the barrier that
// publishes the action the workgroup adds the newest frame's steps 6, 7.  Same MFMAs on the same operands in the same k order per accumulator: bit-identical features.
//   barrier 0: start of the kernel (zeroes the counter below);
//   "barrier 1": the seven frame waves meet through a counter in LDS — a workgroup barrier would hold wave 0's chain until their staging is done;
//   barrier 2: wave 0's tail is done (action, chase cell);  barrier 3 (chase only): the action-dependent new frame is in LDS.
// A terminal step of the chase task (all four channels = the new frame, which needs the action) runs the whole of conv1 behind barrier 3; wave 0 catches up on its own
// row blocks' steps 0..5 behind barrier 2.  The frame waves issue their global stores BEHIND steps 0..5: loads and stores share vmcnt on this part and complete out of
// order, so behind a pending store every wait for a weight fragment is a wait for the stores' acknowledgements.
// Measured (profiles/r05_experiments.md): 32.3 -> 31.1 us per launch by the kernel trace.  Timing-only builds of this kernel: without the 21.7 MB of stores 29.9 us,
// without the head's evaluation 30.7, without both and without wave 0's catch-up 28.7 — the floor of the structure; the stand-alone encoder is 25.7.
A0_D void a0_bytes16_to_img(uint16_t* img, int idx16, const uint4 v) {      // 16 pixels -> 16 bf16 (exact) at element 16 * idx16 of the [4][84 * 84] image
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t f0 = __float_as_uint((float)(w[k] & 0xffu)), f1 = __float_as_uint((float)((w[k] >> 8) & 0xffu));
        const uint32_t f2 = __float_as_uint((float)((w[k] >> 16) & 0xffu)), f3 = __float_as_uint((float)(w[k] >> 24));
        o[2 * k] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
        o[2 * k + 1] = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
    }
    ((uint4*)img)[2 * idx16] = uint4{o[0], o[1], o[2], o[3]};
    ((uint4*)img)[2 * idx16 + 1] = uint4{o[4], o[5], o[6], o[7]};
}

// a0_conv1_stage for the 4 x 84 x 84 image with the k loop cut behind step 5: steps 0..5 (channels 0..2) run before `mid()` on waves with first_before_mid, behind it on
// the others; steps 6, 7 behind it on all.  Same fetch / MFMA / ring-fill sequence per step as a0_conv1_stage.
template <int MBW, class EPI, class Mid, class Between>
A0_D void a0_conv1_stage_split(const uint16_t* img, a0_wring1<A0_R1>& ring, const EPI& epi, bool first_before_mid, Mid&& mid, Between&& between) {
    static_assert(A0_R1 == 4 && EPI::TR, "ring of four steps, transposed accumulators");
    constexpr int HW = 84 * 84, W = 84, W1 = 20, M = 400, MB = 25;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % 2, wmg = wave / 2;
    const int q = lane >> 4, r16 = lane & 15;
    int rows[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        int m = (wmg + i * A0_WMG1) * 16 + r16;
        m = m < M ? m : 0;
        const int oh = m / W1, ow = m - oh * W1;
        rows[i] = (4 * oh + q) * W + 4 * ow;
    }
    a0_acc4 acc[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) acc[i] = a0_acc4{0.f, 0.f, 0.f, 0.f};
    uint2 a[2][MBW][2];
    auto fetch = [&](int slot, int off) {
#pragma unroll
        for (int i = 0; i < MBW; ++i) {
            a[slot][i][0] = *(const uint2*)(img + rows[i] + off);
            a[slot][i][1] = *(const uint2*)(img + rows[i] + off + 4);
        }
    };
    auto off_of = [&](int st) { return (st >> 1) * HW + (st & 1) * 4 * W; };
    auto step = [&](auto UC, int off_next) {      // ring slot u = step % 4, fragments in a[u & 1]; requests the next step's fragments first
        constexpr int u = decltype(UC)::value;
        fetch((u + 1) & 1, off_next);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int i = 0; i < MBW; ++i) {
                const a0_u32x4 av = {a[u & 1][i][0].x, a[u & 1][i][0].y, a[u & 1][i][1].x, a[u & 1][i][1].y};
                const a0_u32x4 bv = {ring.v[u][s3].x, ring.v[u][s3].y, ring.v[u][s3].z, ring.v[u][s3].w};
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(a0_bf16x8, bv), __builtin_bit_cast(a0_bf16x8, av), acc[i], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        ring.fill(u);
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> U0; typedef std::integral_constant<int, 1> U1; typedef std::integral_constant<int, 2> U2; typedef std::integral_constant<int, 3> U3;
    auto steps05 = [&] {
        fetch(0, off_of(0));
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see a0_conv_stage (here also the frame waves' stores)
        step(U0{}, off_of(1)); step(U1{}, off_of(2)); step(U2{}, off_of(3)); step(U3{}, off_of(4));
        step(U0{}, off_of(5)); step(U1{}, off_of(5));      // (the last request re-reads step 5: channel 3 may not be there yet)
    };
    if (first_before_mid) steps05();
    mid();
    if (!first_before_mid) steps05();
    fetch(0, off_of(6));
    __builtin_amdgcn_s_waitcnt(0x0F70);
    step(U2{}, off_of(7)); step(U3{}, off_of(7));
    between();
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int mb = wmg + i * A0_WMG1;
        if (mb < MB) {
            const int m = mb * 16 + r16;
            if (m < M) epi.emit_n4(m, wn * 16 + 4 * q, acc[i], nullptr);
        }
    }
    __syncthreads();
}

struct TailScalar {
    typedef a0_qenv_args Args;
    static A0_D void run(const a0_qenv_args& Q, uint32_t e, uint32_t g, long long steps, unsigned long long off_a, unsigned long long off_u, float eps, long long slot, const a0_u4& x,
                         bool chase, int lane, float* w2s /* wave-private LDS: where conv1's epilogue will write act1 (behind barrier 2) */, float* raw, int* s_chase_cell_p) {
        const int NQ = Q.A + (Q.dueling ? 1 : 0);
        a0_env_pre Z;
        a0_env_commit_prefetch(Z, e, Q.E, Q.n, steps, Q.ep_ret, Q.ring_act, Q.ring_rew, Q.ring_done);
        a0_env_pre_to_vgpr(Z);
        const a0_env_out O = a0_env_out_vgpr(Q.ep_ret, Q.final_mask, Q.final_ret, Q.ring_act, Q.ring_rew, Q.ring_done, Q.r_act, Q.r_rew, Q.r_done);
        A0_TO_VGPR(steps); A0_TO_VGPR(off_a); A0_TO_VGPR(off_u); A0_TO_VGPR(eps);
        int n_v = Q.n, E_v = Q.E, task_v = Q.task; double gamma_v = Q.gamma; int* action_v = Q.action; float* qmax_v = Q.qmax;
        A0_TO_VGPR(n_v); A0_TO_VGPR(E_v); A0_TO_VGPR(task_v); A0_TO_VGPR(gamma_v); A0_TO_VGPR(action_v); A0_TO_VGPR(qmax_v);
#pragma unroll 4
        for (int i = lane; i < NQ * 128; i += 64) ((a0_f4*)w2s)[i] = ((const a0_f4*)Q.W2)[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int act = 0; float best = 0.f;
        a0_qhead_wave(Q.slabs, Q.slab_stride, Q.nslab, Q.b1, w2s, Q.b2, Q.A, Q.dueling, (int)e, lane, raw, Q.rng_seed, Q.stream_a, Q.stream_u, off_a, off_u, eps, act, best);
        if (lane == 0) {
            action_v[e] = act; qmax_v[e] = best;
            float r_chase = 0.f;
            if (chase) *s_chase_cell_p = a0_chase_step(a0_chase_cell(Q.obs_in + ((size_t)e * 4 + 3) * A0_ENV_PIX, e), act, x.w, r_chase);
            a0_env_commit_finish(Z, x, e, g, task_v, Q.A, E_v, n_v, steps, gamma_v, act, O.ep_ret, O.final_mask, O.final_ret, O.ring_act, O.ring_rew, O.ring_done, O.r_act, O.r_rew,
                                 O.r_done, slot, r_chase);
        }
    }
};
// TAIL: the step's wave-0 part — TailScalar (a0_qenv_args: fc1 slabs -> head -> action, a0_actor_qhead_env_body); a TailDist over a0_actor_dist_tail_wave0 (a0_dtenv_args names
// the env's fields alike) was measured and not kept, see a0_actor_dist_step_enc_launch.
template <int ORD, class TAIL>
A0_D void a0_actor_step_enc2_body(const typename TAIL::Args& Q, const a0_step_enc_args& N) {
    __shared__ float raw[64];
    __shared__ int s_chase_cell;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HW = 84 * 84, G16 = HW / 16;      // 441 sixteen-byte groups per frame
    uint16_t* img = (uint16_t*)smem;
    uint16_t* a1p = (uint16_t*)(smem + 2 * 4 * HW);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t e = blockIdx.x;
    // ---- the encoder's set-up: weight rings, biases (a0_encoder_fused_x9_body)
    a0_fused_args P;
    P.frames = nullptr; P.slot = nullptr; P.sample_stride = 4 * HW; P.chan_off = 0;
    P.wt1 = N.wt; P.wt2 = N.wt + 48LL * 4 * 64; P.wt3 = P.wt2 + 64LL * 512;
    P.wx2 = P.wt3 + 64LL * 576 + 64LL * 576 + 4LL * 32 * 256; P.wx3 = P.wx2 + 96LL * 512;
    P.b1 = N.b1; P.b2 = N.b2; P.b3 = N.b3;
    P.act1 = nullptr; P.act2 = nullptr; P.act3 = N.act3; P.B = Q.E;
    P.C = 4; P.H = 84; P.W = 84; P.H1 = 20; P.W1 = 20; P.H2 = 9; P.W2 = 9; P.H3 = 7; P.W3 = 7;
    P.off_act1 = 0; P.off_act2 = 0; P.off_end = 0; P.rp1 = 0; P.rp2 = 0;
    a0_wring1<A0_R1> ring1;
    a0_ring2_t ring2;
    a0_ring3_t ring3;
    ring1.init(P.wt1, 4);
    ring1.prologue();
    float* bias_lds = (float*)(smem + A0_X9_BIAS_OFF);
    if (tid < 160) bias_lds[tid] = tid < 32 ? P.b1[tid] : tid < 96 ? P.b2[tid - 32] : P.b3[tid - 96];
    // (conv2's / conv3's rings are set up behind the step: their pointers wait in vector registers, the scalar file is the tail's until then)
    unsigned long long v_wx2 = (unsigned long long)P.wx2, v_wx3 = (unsigned long long)P.wx3;
    A0_TO_VGPR(v_wx2); A0_TO_VGPR(v_wx3);
    // ---- the step (a0_actor_qhead_env_body)
    uint32_t g = Q.g; long long steps = Q.steps, start = Q.start; unsigned long long off_a = Q.off_a, off_u = Q.off_u; float eps = Q.eps;
    if (Q.ctrl) {
        g += (uint32_t)Q.ctrl[A0_CTRL_ENV_STEP]; steps += Q.ctrl[A0_CTRL_ACTOR_STEPS]; start += Q.ctrl[A0_CTRL_REPLAY_SLOT];
        off_a += (unsigned long long)Q.ctrl[A0_CTRL_RNG_ACTION]; off_u += (unsigned long long)Q.ctrl[A0_CTRL_RNG_UNIFORM];
    }
    if (Q.eps_ptr) eps = Q.eps_ptr[0];
    const long long slot = (start + e) % Q.cap;
    uint32_t e_v = e;
    asm volatile("" : "+v"(e_v));
    const a0_u4 x = a0_philox4x32_10(e_v, g, 0u, 0x454E56u, (uint32_t)Q.env_seed, (uint32_t)(Q.env_seed >> 32) ^ Q.rank);
    const bool term = __builtin_amdgcn_readfirstlane((int)((x.y % 500u) == 0u)) != 0;
    const bool chase = Q.task == A0_ENV_TASK_CHASE;
    const bool early = !(chase && term);
    // frame work of waves 1..7: lane j < 441 owns sixteen-byte group j of every frame
    const int j = tid - 64;
    // (recomputed at each use: kept, the lane mask would sit in a scalar register pair across the tail — one more than the file has)
    auto has_group = [&] { int jj = j; asm volatile("" : "+v"(jj)); return wave != 0 && jj < G16; };
    const uint32_t pbase = (uint32_t)Q.env_seed ^ a0_env_mix32(e * 0x9E3779B1u + g);
    // (per-lane pointers: vector registers — the scalar file holds the tail's and the encoder's uniform values at the same time here)
    uint4* out16 = (uint4*)(Q.obs_out + (size_t)e * 4 * HW) + j;
    uint4* row16 = (uint4*)(Q.frames + slot * (8LL * HW)) + j;
    A0_TO_VGPR(out16); A0_TO_VGPR(row16);
    float* act3_v = N.act3;
    A0_TO_VGPR(act3_v);
    auto new_group = [&](uint32_t by, uint32_t bx) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = 16u * (uint32_t)j + 4u * (uint32_t)k;
            w[k] = (uint32_t)a0_env_pixel(pbase, by, bx, p, chase) | ((uint32_t)a0_env_pixel(pbase, by, bx, p + 1, chase) << 8) |
                   ((uint32_t)a0_env_pixel(pbase, by, bx, p + 2, chase) << 16) | ((uint32_t)a0_env_pixel(pbase, by, bx, p + 3, chase) << 24);
        }
        return uint4{w[0], w[1], w[2], w[3]};
    };
    auto put = [&](int c, const uint4 v) {      // channel c of the new observation: global stack, replay row's st_next half, LDS image
        out16[c * G16] = v; row16[(4 + c) * G16] = v;
        a0_bytes16_to_img(img, c * G16 + j, v);
    };
    // the frame waves meet each other WITHOUT wave 0 (a workgroup barrier would hold wave 0's latency chain until their staging is done): a counter in LDS
    __shared__ int s_staged;
    if (tid == 0) s_staged = 0;
    __syncthreads();      // barrier 0: at the kernel's start, nobody waits
    uint4 keep[9];        // the frame waves' global stores are issued BEHIND conv1's first steps (gfx9 counts stores and loads together: pending stores would make the
                          // stage's first wait for its weights a wait for their acknowledgements)
    if (wave == 0) {
        TAIL::run(Q, e, g, steps, off_a, off_u, eps, slot, x, chase, lane, (float*)a1p, raw, &s_chase_cell);
    } else {
        if (has_group()) {
            const uint4* in16 = (const uint4*)(Q.obs_in + (size_t)e * 4 * HW) + j;
            A0_TO_VGPR(in16);
            const uint4 i0 = in16[0], i1 = in16[G16], i2 = in16[2 * G16], i3 = in16[3 * G16];
            if (Q.obs0 == Q.obs_in) { keep[0] = i0; keep[1] = i1; keep[2] = i2; keep[3] = i3; }
            else {
                const uint4* o016 = (const uint4*)(Q.obs0 + (size_t)e * 4 * HW) + j;
                A0_TO_VGPR(o016);
                keep[0] = o016[0]; keep[1] = o016[G16]; keep[2] = o016[2 * G16]; keep[3] = o016[3 * G16];
            }
            if (!chase) {
                const uint4 nw = new_group((3u * g + 11u * e) % 77u, (5u * g + 7u * e) % 77u);
                if (term) { keep[4] = nw; keep[5] = nw; keep[6] = nw; } else { keep[4] = i1; keep[5] = i2; keep[6] = i3; }
                keep[7] = nw;
                for (int c = 0; c < 4; ++c) a0_bytes16_to_img(img, c * G16 + j, keep[4 + c]);
            } else if (!term) {
                keep[4] = i1; keep[5] = i2; keep[6] = i3;
                for (int c = 0; c < 3; ++c) a0_bytes16_to_img(img, c * G16 + j, keep[4 + c]);
            }
        }
        // "barrier 1" among the seven frame waves: their parts of the image are in LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&s_staged, 1);
        while (__builtin_amdgcn_readfirstlane(*(volatile int*)&s_staged) < A0_FUSED_WAVES - 1) __builtin_amdgcn_s_sleep(1);
    }
    auto early_stores = [&] {      // everything of the step's frame traffic that does not wait for the action
        if (has_group()) {
#pragma unroll
            for (int c = 0; c < 4; ++c) row16[c * G16] = keep[c];
            const int nc = !chase ? 4 : (!term ? 3 : 0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nc) { out16[c * G16] = keep[4 + c]; row16[(4 + c) * G16] = keep[4 + c]; }
        }
    };
    auto mid = [&] {
        if (wave != 0) early_stores();
        __syncthreads();      // barrier 2
        if (chase) {
            if (has_group()) {
                uint32_t by, bx;
                a0_chase_pos(s_chase_cell, by, bx);
                const uint4 nw = new_group(by, bx);
                if (term) { put(0, nw); put(1, nw); put(2, nw); }
                put(3, nw);
            }
            __syncthreads();      // barrier 3
        }
    };
    auto uni = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull));
    };
    ring2.init((const float*)uni(v_wx2), 512);
    ring3.init((const float*)uni(v_wx3), 576);
    typedef EpiFwdX<20, A0_P1X, A0_RP1X, 20 * A0_RP1X> E1X;
    const E1X e1{bias_lds, a1p, nullptr, 32};
    constexpr int MBW1X = (25 + A0_WMG1 - 1) / A0_WMG1;
    const bool before = early && wave != 0;
    if (MBW1X > 1 && wave / 2 + (MBW1X - 1) * A0_WMG1 >= 25)
        a0_conv1_stage_split<(MBW1X > 1 ? MBW1X - 1 : 1)>(img, ring1, e1, before, mid, [&] { ring2.prologue(); });
    else
        a0_conv1_stage_split<MBW1X>(img, ring1, e1, before, mid, [&] { ring2.prologue(); });
    P.act3 = (float*)uni((unsigned long long)act3_v);
    a0_x9_conv23<ORD, 3, 2, false>(P, (int)e, smem, bias_lds, ring1, ring2, ring3);
}

template <int ORD>
__global__ __launch_bounds__(A0_FUSED_THREADS, A0_FUSED_MINWAVES) void a0_actor_step_enc2_kernel(a0_qenv_args Q, a0_step_enc_args N) { a0_actor_step_enc2_body<ORD, TailScalar>(Q, N); }

static bool a0_fused_layout(int C, int H, int W, a0_fused_args& P, size_t& lds_bytes);
int a0_actor_dist_step_enc_launch(const a0_dtenv_args& Q, size_t tail_lds, const float* wt, const a0_encoder_weights* w, float* act3, hipStream_t st) {
    a0_fused_args P;
    size_t lds = 0;
    if (!wt || !w || !w->b1 || !w->b2 || !w->b3 || !act3 || !a0_fused_layout(4, 84, 84, P, lds)) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: bad encoder argument");
    if (getenv("A0_NO_X9") != nullptr) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: the split-operand encoder only");
    if (P.H1 != 20 || P.W1 != 20 || P.H2 != 9 || P.W2 != 9 || P.H3 != 7 || P.W3 != 7) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: geometry");
    const a0_step_enc_args N{wt, w->b1, w->b2, w->b3, act3};
    lds = A0_X9_LDS_BYTES > tail_lds ? (size_t)A0_X9_LDS_BYTES : tail_lds;
    if (lds > 160 * 1024 - 64) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: head too wide for LDS");
    const int six = a0_x9_products_now() == 6;
    auto kern = six ? a0_actor_dist_step_enc_kernel<2> : a0_actor_dist_step_enc_kernel<4>;
    static size_t configured[2] = {0, 0};
    if (lds > configured[six]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: LDS");
        configured[six] = lds;
    }
    // (the second form — conv1's channels 0..2 beside the tail, a0_actor_step_enc2_body<TailDist> — was built for these heads too: bit-identical, and c51 14.21 - 14.24 ->
    // 14.32 - 14.34 ms, qr 11.75 - 11.86 -> 11.74 - 11.77 with the same kernel durations: not kept, profiles/r05_experiments.md)
    A0_LAUNCH_PROBED(A0_TAG_ACTOR_STEP_ENC, 2.0 * (400.0 * 32 * 256 + 81.0 * 64 * 512 + 49.0 * 64 * 576) * Q.E, kern, dim3(Q.E), dim3(A0_FUSED_THREADS), lds, st, Q, N);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_dist_tail_env_step_enc");
}
int a0_actor_step_enc_launch(const a0_qenv_args& Q, const float* wt, const a0_encoder_weights* w, float* act3, hipStream_t st) {
    a0_fused_args P;
    size_t lds = 0;
    if (!wt || !w || !w->b1 || !w->b2 || !w->b3 || !act3 || !a0_fused_layout(4, 84, 84, P, lds)) return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step_enc: bad encoder argument");
    if (getenv("A0_NO_X9") != nullptr) return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step_enc: the split-operand encoder only");
    if (P.H1 != 20 || P.W1 != 20 || P.H2 != 9 || P.W2 != 9 || P.H3 != 7 || P.W3 != 7) return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step_enc: geometry");
    const a0_step_enc_args N{wt, w->b1, w->b2, w->b3, act3};
    const size_t tail_lds = (size_t)(Q.A + (Q.dueling ? 1 : 0)) * 512 * sizeof(float);
    lds = A0_X9_LDS_BYTES > tail_lds ? (size_t)A0_X9_LDS_BYTES : tail_lds;
    static const int form = getenv("A0_STEP_ENC") ? atoi(getenv("A0_STEP_ENC")) : 2;      // tuning aid: 1 = tail, barrier, whole encoder (the first form); 2 = conv1's channels 0..2 beside the tail
    const int six = a0_x9_products_now() == 6;
    auto kern = form != 1 ? (six ? a0_actor_step_enc2_kernel<2> : a0_actor_step_enc2_kernel<4>) : (six ? a0_actor_step_enc_kernel<2> : a0_actor_step_enc_kernel<4>);
    static size_t configured[2] = {0, 0};
    if (lds > configured[six]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_actor_qhead_env_step_enc: LDS");
        configured[six] = lds;
    }
    A0_LAUNCH_PROBED(A0_TAG_ACTOR_STEP_ENC, 2.0 * (400.0 * 32 * 256 + 81.0 * 64 * 512 + 49.0 * 64 * 576) * Q.E, kern, dim3(Q.E), dim3(A0_FUSED_THREADS), lds, st, Q, N);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_qhead_env_step_enc");
}

// Up to three forward passes of the split-operand encoder in ONE launch (round 4): the learner's passes over a batch — target network on s', online network on s'
// (double-Q), online network on s — are independent of one another, and a launch of 256 looping workgroups is 9 % cheaper per observation over 1024 observations
// than over 512 (ring set-up and the first weights per workgroup, the tail of the last wave of workgroups: tools/ubench_encoder_fwd.py).  Every pass gets a share
// of the grid proportional to its observations; its workgroups run the unchanged per-pass body with their own weights, frames and outputs.
struct a0_fused_multi_args { a0_fused_args p[3]; int first[4]; };
template <int ORD>
__global__ __launch_bounds__(A0_FUSED_THREADS, A0_FUSED_MINWAVES) void a0_encoder_fused_multi_kernel(a0_fused_multi_args M) {
    const int bid = (int)blockIdx.x;
    const int k = bid >= M.first[2] ? 2 : (bid >= M.first[1] ? 1 : 0);
    a0_encoder_fused_x9_body<ORD, 7, 3, 2, true>(M.p[k], bid - M.first[k], M.first[k + 1] - M.first[k]);
}

template <int MBW1, int MBW2, int MBW3, int WC, bool X9, bool LOOP, int ORD>
__global__ __launch_bounds__(A0_FUSED_THREADS, A0_FUSED_MINWAVES) void a0_encoder_fused_kernel(a0_fused_args P) {
    if constexpr (X9) {
        a0_encoder_fused_x9_body<ORD, MBW1, MBW2, MBW3, LOOP>(P, (int)blockIdx.x, (int)gridDim.x);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* img = (uint16_t*)smem;            // the observation as bf16 (exact for bytes), [C][H][W]
    float* fl = (float*)smem;
    float* act1 = fl + P.off_act1;
    float* act2 = fl + P.off_act2;
    const int obs_bytes = P.C * P.H * P.W;
    const int M1 = P.H1 * P.W1, M2 = P.H2 * P.W2, M3 = P.H3 * P.W3;
    a0_wring1<A0_R1> ring1;
    a0_wring<64, 4, A0_R2> ring2;
    a0_wring<64, 4, A0_R3> ring3;
    ring1.init(P.wt1, P.C);
    ring2.init(P.wt2, 512);
    ring3.init(P.wt3, 576);
    ring1.prologue();
    constexpr int OW1 = WC == 84 ? 20 : 0, OW2 = WC == 84 ? 9 : 0;      // output widths of conv1 / conv2 when the input is 84 wide
    a0_pre<32, 2, MBW1, EpiFwd<OW1>> pre1;
    a0_pre<64, 4, MBW2, EpiFwd<OW2>> pre2;
    a0_pre<64, 4, MBW3, EpiFwd<0>> pre3;
    pre1.load(EpiFwd<OW1>{P.b1, nullptr, 0, 0, 1, nullptr, 32}, M1);   // biases: once per launch, ahead of everything
    pre2.load(EpiFwd<OW2>{P.b2, nullptr, 0, 0, 1, nullptr, 64}, M2);
    pre3.load(EpiFwd<0>{P.b3, nullptr, 0, 0, 1, nullptr, 64}, M3);
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        // ---- observation -> LDS as bf16: 16 bytes in, 32 bytes out per lane and trip; (float)byte is exact and its low 16 bits are zero
        const long long s = P.slot ? (long long)P.slot[b] : (long long)b;
        const uint4* src = (const uint4*)(P.frames + s * P.sample_stride + P.chan_off);
        // (all of a lane's loads are requested before the first is converted: one HBM latency per observation, not one per trip)
        constexpr int TRIPS = 4;
        const int n16 = obs_bytes >> 4;
        for (int i0 = threadIdx.x; i0 < n16; i0 += TRIPS * A0_FUSED_THREADS) {
            uint4 v[TRIPS];
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                const int i = i0 + j * A0_FUSED_THREADS;
                v[j] = src[i < n16 ? i : n16 - 1];
            }
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                int i = i0 + j * A0_FUSED_THREADS;
                i = i < n16 ? i : n16 - 1;                 // lanes past the end redo the last group (same bytes, same address): no branch
                const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t f0 = __float_as_uint((float)(w[k] & 0xffu)), f1 = __float_as_uint((float)((w[k] >> 8) & 0xffu));
                    const uint32_t f2 = __float_as_uint((float)((w[k] >> 16) & 0xffu)), f3 = __float_as_uint((float)(w[k] >> 24));
                    o[2 * k] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);          // high halves of f0 (low) and f1 (high)
                    o[2 * k + 1] = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
                }
                ((uint4*)img)[2 * i] = uint4{o[0], o[1], o[2], o[3]};
                ((uint4*)img)[2 * i + 1] = uint4{o[4], o[5], o[6], o[7]};
            }
        }
        __syncthreads();
        const EpiFwd<OW1> e1{P.b1, act1, A0_P1, P.rp1, P.W1, P.act1 ? P.act1 + (long long)b * M1 * 32 : nullptr, 32};
        // conv1's 16-row blocks do not divide evenly over the four M groups (25 blocks: 7 + 6 + 6 + 6): groups that own one block less
        // run the MBW1 - 1 instantiation instead of recomputing a dummy block (same barrier count on both paths)
        const int wmg1 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / 2;
        if (MBW1 > 1 && wmg1 + (MBW1 - 1) * 4 >= ((M1 + 15) >> 4))
            a0_conv1_stage<(MBW1 > 1 ? MBW1 - 1 : 1), A0_R1, WC>(img, P.H * P.W, P.W, P.W1, M1, ring1, e1, pre1, [&] { ring2.prologue(); });
        else
            a0_conv1_stage<MBW1, A0_R1, WC>(img, P.H * P.W, P.W, P.W1, M1, ring1, e1, pre1, [&] { ring2.prologue(); });
        AF2 f2{act1, P.rp1, P.W2};
        const EpiFwd<OW2> e2{P.b2, act2, A0_P2, P.rp2, P.W2, P.act2 ? P.act2 + (long long)b * M2 * 64 : nullptr, 64};
        a0_conv_stage<64, 4, MBW2, 3, A0_R2>(f2, M2, 512, ring2, e2, pre2, [&] { ring3.prologue(); });
        AF3 f3{act2, P.rp2, P.W3};
        const EpiFwd<0> e3{P.b3, nullptr, 0, 0, 1, P.act3 + (long long)b * M3 * 64, 64};
        a0_conv_stage<64, 4, MBW3, 3, A0_R3>(f3, M3, 576, ring3, e3, pre3, [&] { ring1.prologue(); });
    }
}

// ------------------------------------------------------------------------------------------------ fused data gradients
// conv3 and conv2 data gradients of one observation back to back (84x84 geometry: d3 7x7x64 -> d2 9x9x64 -> d1 20x20x32), the
// counterpart of autograd's conv backward-data for ConvEncoder (reference model.py:93-105, called from agent.py:136-141).  Both are
// written as stride-1 "gather" convolutions over ZERO-PADDED LDS images, so a0_conv_stage runs them unchanged:
//   d2[h][w][ci] = sum_{kh',kw',co} d3pad[h + kh'][w + kw'][co] * W3[co][ci][2 - kh'][2 - kw']                 (pad 2, 3x3 taps)
//   d1[2h2+ph][2w2+pw][ci] = sum_{a',b',co} d2pad[h2 + a'][w2 + b'][co] * W2[co][ci][ph + 2(1-a')][pw + 2(1-b')]   (pad 1, 2x2 taps)
// the second once per stride phase (ph, pw).  The ReLU masks (act2 > 0, act1 > 0) come from the forward activations in HBM; d2 and d1
// go to HBM for the weight-gradient GEMMs, d2 also stays in LDS for the four phases.  66 KB of LDS: two workgroups per CU.
struct a0_dgrad_args {
    const float *d3, *act1, *act2;
    float *d2, *d1;
    const float *wd3, *wd2;          // fragment-major copies (a0_conv_wt_kernel): [576][64] and 4 x [256][32]
    int B, rpa, rpb;                 // row pitches of the two padded 11 x 11 x 64 LDS images
};
struct AFD3 {   // 3x3 taps over d3pad [11][rpa]; k = (kh'*3 + kw')*64 + co; output 9 wide
    const float* img; int RP;
    A0_D int row(int m) const { const int oh = m / 9, ow = m - oh * 9; return oh * RP + ow * A0_P2; }
    A0_D int chunk_off(int c) const { const int cell = c >> 2; return (cell / 3) * RP + (cell % 3) * A0_P2 + 16 * (c & 3); }
    A0_D int step_off(int j) const { return 4 * j; }
    typedef float Raw;
    A0_D Raw load(int addr) const { return img[addr]; }
    static A0_D float value(Raw r) { return r; }
};
struct AFD2 {   // 2x2 taps over d2pad [11][rpb]; k = (a'*2 + b')*64 + co; output 10 wide
    const float* img; int RP;
    A0_D int row(int m) const { const int oh = m / 10, ow = m - oh * 10; return oh * RP + ow * A0_P2; }
    A0_D int chunk_off(int c) const { const int cell = c >> 2; return (cell >> 1) * RP + (cell & 1) * A0_P2 + 16 * (c & 3); }
    A0_D int step_off(int j) const { return 4 * j; }
    typedef float Raw;
    A0_D Raw load(int addr) const { return img[addr]; }
    static A0_D float value(Raw r) { return r; }
};
constexpr int A0_RD3 = 6, A0_RD2 = 4;

__global__ __launch_bounds__(A0_FUSED_THREADS) void a0_encoder_dgrad_fused_kernel(a0_dgrad_args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* imgA = (float*)smem;                 // d3, padded by 2
    float* imgB = imgA + 11 * P.rpa;            // d2, padded by 1 (last row / column unused)
    for (int i = threadIdx.x; i < 11 * (P.rpa + P.rpb); i += A0_FUSED_THREADS) imgA[i] = 0.f;     // borders stay zero for the whole launch
    a0_wring<64, 4, A0_RD3> ring3;
    a0_wring<32, 2, A0_RD2> ringp[2];
    ring3.init(P.wd3, 576);
    ring3.prologue();
    typedef EpiBwd<9, 1> E3;
    typedef EpiBwd<10, 2> E2;
    a0_pre<64, 4, 3, E3> pre3;
    a0_pre<32, 2, 2, E2> prep[2];
    auto epi3 = [&](int b) { return E3{P.act2 + (long long)b * 81 * 64, P.d2 + (long long)b * 81 * 64, imgB + P.rpb + A0_P2, A0_P2, P.rpb, 9, 0, 0, 64}; };
    auto epi2 = [&](int b, int ph, int pw) { return E2{P.act1 + (long long)b * 400 * 32, P.d1 + (long long)b * 400 * 32, nullptr, 0, 0, 20, ph, pw, 32}; };
    if ((int)blockIdx.x < P.B) pre3.load(epi3(blockIdx.x), 81);
    __syncthreads();
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        const a0_f4* src = (const a0_f4*)(P.d3 + (long long)b * 49 * 64);
        for (int i = threadIdx.x; i < 49 * 16; i += A0_FUSED_THREADS) {
            const a0_f4 v = src[i];
            const int pos = i >> 4, c4 = (i & 15) * 4, h = pos / 7, w = pos - h * 7;
            float* d = imgA + (h + 2) * P.rpa + (w + 2) * A0_P2 + c4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        AFD3 f3{imgA, P.rpa};
        AFD2 f2{imgB, P.rpb};
        const int bn = b + gridDim.x < P.B ? b + gridDim.x : b;       // next observation of this workgroup (its masks are prefetched in the last slot)
        // each `between` slot requests the NEXT layer's weights and ReLU masks before this layer's epilogue runs
        a0_conv_stage<64, 4, 3, 3, A0_RD3>(f3, 81, 576, ring3, epi3(b), pre3, [&] { ringp[0].init(P.wd2, 256); ringp[0].prologue(); prep[0].load(epi2(b, 0, 0), 100); });
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[0], epi2(b, 0, 0), prep[0], [&] { ringp[1].init(P.wd2 + 1 * 8192, 256); ringp[1].prologue(); prep[1].load(epi2(b, 0, 1), 100); });
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[1], epi2(b, 0, 1), prep[1], [&] { ringp[0].init(P.wd2 + 2 * 8192, 256); ringp[0].prologue(); prep[0].load(epi2(b, 1, 0), 100); });
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[0], epi2(b, 1, 0), prep[0], [&] { ringp[1].init(P.wd2 + 3 * 8192, 256); ringp[1].prologue(); prep[1].load(epi2(b, 1, 1), 100); });
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[1], epi2(b, 1, 1), prep[1], [&] { ring3.prologue(); pre3.load(epi3(bn), 81); });
    }
}

// ---- the same two data gradients on the bf16 pipe with both operands split exactly into three bf16 terms (see a0_conv_stage_x9):
// d3 is split when it is loaded, d2 by conv3's epilogue; both live in LDS as three zero-padded 11 x 11 term-plane images (116 KB:
// one workgroup per CU, eight waves), the flipped / phase-split weights come pre-split from a0_conv_wt_kernel (segments 8, 9).
// Nine v_mfma_f32_16x16x32_bf16 per 32 k replace eight v_mfma_f32_16x16x4_f32 of twice the pipe time each.
// padded images: 11 x 11 pixels, A0_P2X bf16 per pixel (64 channels + 16); image rows padded like the forward planes, per image (3x3 taps,
// 9-wide output: +192 bytes -> 4.7 LDS cycles per fragment read instead of 8.0; 2x2 taps, 10-wide output: +96 bytes -> 4.6 instead of 7.4)
constexpr int A0_RPDA = 11 * A0_P2X + (A0_PADS ? 96 : 0), A0_RPDB = 11 * A0_P2X + (A0_PADS ? 48 : 0);
constexpr int A0_DTERMA = 11 * A0_RPDA, A0_DTERMB = 11 * A0_RPDB;      // elements per term plane
struct AFD3X {   // 3x3 taps over the d3pad planes; MFMA step = half (32 channels) of tap st >> 1; output 9 wide
    static constexpr int term = A0_DTERMA;
    const uint16_t* planes;
    A0_D int row(int m) const { const int oh = m / 9, ow = m - oh * 9; return oh * A0_RPDA + ow * A0_P2X; }
    A0_D int step_off(int st) const { const int tap = st >> 1; return (tap / 3) * A0_RPDA + (tap % 3) * A0_P2X + 32 * (st & 1); }
};
struct AFD2X {   // 2x2 taps over the d2pad planes; output 10 wide
    static constexpr int term = A0_DTERMB;
    const uint16_t* planes;
    A0_D int row(int m) const { const int oh = m / 10, ow = m - oh * 10; return oh * A0_RPDB + ow * A0_P2X; }
    A0_D int step_off(int st) const { const int cell = st >> 1; return (cell >> 1) * A0_RPDB + (cell & 1) * A0_P2X + 32 * (st & 1); }
};
struct EpiBwd3X {               // d2 = act2 > 0 ? acc : 0 -> global [81][64] and, split into three bf16 terms, the interior of the d2pad planes; transposed accumulators
    static constexpr bool PER_ELEM = true;
    static constexpr bool ROW4 = false;
    static constexpr bool TR = true;
    const float* mask; float* dst; uint16_t* planes;
    A0_D float pre_col(int) const { return 0.f; }
    A0_D a0_f4 pre_elem4(int m, int n0) const { return *(const a0_f4*)(mask + (unsigned)(m * 64 + n0)); }
    A0_D void emit_n4(int m, int n0, const a0_acc4& acc, const float* pre) const {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = pre[r] > 0.f ? acc[r] : 0.f;
        *(a0_f4*)(dst + (unsigned)(m * 64 + n0)) = a0_f4{v[0], v[1], v[2], v[3]};
        uint2 hi, mid, lo;
        a0_split4(v, hi, mid, lo);
        const int oh = m / 9, ow = m - oh * 9;
        uint16_t* d = planes + (oh + 1) * A0_RPDB + (ow + 1) * A0_P2X + n0;
        *(uint2*)d = hi; *(uint2*)(d + A0_DTERMB) = mid; *(uint2*)(d + 2 * A0_DTERMB) = lo;
    }
};
static_assert((A0_RPDB % 4) == 0 && (A0_DTERMB % 4) == 0 && (A0_P2X % 4) == 0, "8-byte aligned plane writes");
template <int OW, int S>
struct EpiBwdT {                // dx = mask > 0 ? acc : 0 at pixel (oh*S + ph, ow*S + pw) of a Wfull-wide NHWC image in global memory; transposed accumulators
    static constexpr bool PER_ELEM = true;
    static constexpr bool ROW4 = false;
    static constexpr bool TR = true;
    const float* mask; float* dst; int Wfull, ph, pw, N;
    A0_D unsigned gi(int m, int n) const { const int oh = m / OW, ow = m - oh * OW; return (unsigned)(((oh * S + ph) * Wfull + ow * S + pw) * N + n); }
    A0_D float pre_col(int) const { return 0.f; }
    A0_D a0_f4 pre_elem4(int m, int n0) const { return *(const a0_f4*)(mask + gi(m, n0)); }
    A0_D void emit_n4(int m, int n0, const a0_acc4& acc, const float* pre) const {
        *(a0_f4*)(dst + gi(m, n0)) = a0_f4{pre[0] > 0.f ? acc[0] : 0.f, pre[1] > 0.f ? acc[1] : 0.f, pre[2] > 0.f ? acc[2] : 0.f, pre[3] > 0.f ? acc[3] : 0.f};
    }
};
struct EpiBwdPair {             // two stride phases of conv2's data gradient side by side: virtual column n0 -> phase p0 + (n0 >> 5), channel n0 & 31; transposed accumulators
    static constexpr bool PER_ELEM = true;
    static constexpr bool ROW4 = false;
    static constexpr bool TR = true;
    const float* mask; float* dst; int p0;         // act1 / d1 of this observation, [20][20][32]
    A0_D unsigned gi(int m, int n0) const {
        const int p = p0 + (n0 >> 5), oh = m / 10, ow = m - oh * 10;
        return (unsigned)(((oh * 2 + (p >> 1)) * 20 + ow * 2 + (p & 1)) * 32 + (n0 & 31));
    }
    A0_D float pre_col(int) const { return 0.f; }
    A0_D a0_f4 pre_elem4(int m, int n0) const { return *(const a0_f4*)(mask + gi(m, n0)); }
    A0_D void emit_n4(int m, int n0, const a0_acc4& acc, const float* pre) const {
        *(a0_f4*)(dst + gi(m, n0)) = a0_f4{pre[0] > 0.f ? acc[0] : 0.f, pre[1] > 0.f ? acc[1] : 0.f, pre[2] > 0.f ? acc[2] : 0.f, pre[3] > 0.f ? acc[3] : 0.f};
    }
};
constexpr int A0_RXD3 = 6, A0_RXD2 = 4;               // 32-k steps of split weights in flight (18 and 8 steps per stage)

template <int ORD>
__global__ __launch_bounds__(A0_FUSED_THREADS) void a0_encoder_dgrad_fused_x9_kernel(a0_dgrad_args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* plA = (uint16_t*)smem;                  // d3, padded by 2: three term planes
    uint16_t* plB = plA + 3 * A0_DTERMA;              // d2, padded by 1 (last row / column unused)
    for (int i = threadIdx.x; i < 3 * (A0_DTERMA + A0_DTERMB) / 2; i += A0_FUSED_THREADS) ((uint32_t*)smem)[i] = 0u;     // both images; borders stay zero for the whole launch
#if A0_KSPLIT_D
    // N-stationary stages (a0_conv_stage_x9k): conv3's data gradient as in the forward kernel; conv2's four stride phases share their A operand
    // (the d2pad taps), so two phases at a time form ONE stage of 64 virtual columns — column block wn = phase (wn >> 1) of the pair, channel
    // block wn & 1 — and every weight fragment of the kernel is loaded once per workgroup (was twice / four times), every d2pad fragment
    // read for two phases at once.  Partial sums are exchanged through the 38 KB the two padded images leave free.
    float* const xch0 = (float*)(smem + 3 * (A0_DTERMA + A0_DTERMB) * 2);
    float* const xch1 = xch0 + 4 * 4 * 256;                // region 0: 4 column blocks x up to 4 row blocks x 1 KB
    {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wn = wave & 3, wk = wave >> 2;
        a0_wring9<64, 4, 3, 2> ring3;
        a0_wring9<64, 4, 2, 2> ringq[2];
        ring3.init(P.wd3, 576);
        ring3.prologue();
        // phase matrices: 4 x [256 k][32 n] in the N = 32 ring layout (3072 uint4 each, 384 per k step)
        const uint4* const wq = (const uint4*)P.wd2 + (((wn & 1) * 16 + (lane & 15)) * 4 + (lane >> 4)) * 3 + (wn >> 1) * 3072 + wk * 384;
        __syncthreads();
        for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
            const a0_f4* src = (const a0_f4*)(P.d3 + (long long)b * 49 * 64);
            for (int i = threadIdx.x; i < 49 * 16; i += A0_FUSED_THREADS) {
                const a0_f4 v = src[i];
                const int pos = i >> 4, c4 = (i & 15) * 4, h = pos / 7, w = pos - h * 7;
                const float x[4] = {v.x, v.y, v.z, v.w};
                uint2 hi, mid, lo;
                a0_split4(x, hi, mid, lo);
                uint16_t* d = plA + (h + 2) * A0_RPDA + (w + 2) * A0_P2X + c4;
                *(uint2*)(d) = hi; *(uint2*)(d + A0_DTERMA) = mid; *(uint2*)(d + 2 * A0_DTERMA) = lo;
            }
            __syncthreads();
            const AFD3X f3{plA};
            const AFD2X f2{plB};
            const EpiBwd3X e3{P.act2 + (long long)b * 81 * 64, P.d2 + (long long)b * 81 * 64, plB};
            const EpiBwdPair e2a{P.act1 + (long long)b * 400 * 32, P.d1 + (long long)b * 400 * 32, 0};
            const EpiBwdPair e2b{P.act1 + (long long)b * 400 * 32, P.d1 + (long long)b * 400 * 32, 2};
            a0_conv_stage_x9k<ORD, 64, 4, 6, 3, 1>(f3, 81, ring3, e3, xch0, xch1, [&] { ringq[0].init_at(wq, 4, 768); ringq[0].prologue(); });
            a0_conv_stage_x9k<ORD, 64, 4, 7, 2, 4>(f2, 100, ringq[0], e2a, xch0, xch1, [&] { ringq[1].init_at(wq + 2 * 3072, 4, 768); ringq[1].prologue(); });
            a0_conv_stage_x9k<ORD, 64, 4, 7, 2, 4>(f2, 100, ringq[1], e2b, xch0, xch1, [&] { ring3.prologue(); });
        }
        return;
    }
#endif
    a0_wring9<64, 4, A0_RXD3> ring3;
    a0_wring9<32, 2, A0_RXD2> ringp[2];
    ring3.init(P.wd3, 576);
    ring3.prologue();
    typedef EpiBwd3X E3;
    typedef EpiBwdT<10, 2> E2;
    a0_pre<64, 4, 3, E3> pre3;
    a0_pre<32, 2, 2, E2> prep[2];
    auto epi3 = [&](int b) { return E3{P.act2 + (long long)b * 81 * 64, P.d2 + (long long)b * 81 * 64, plB}; };
    auto epi2 = [&](int b, int ph, int pw) { return E2{P.act1 + (long long)b * 400 * 32, P.d1 + (long long)b * 400 * 32, 20, ph, pw, 32}; };
    if ((int)blockIdx.x < P.B) pre3.load(epi3(blockIdx.x), 81);
    __syncthreads();
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        const a0_f4* src = (const a0_f4*)(P.d3 + (long long)b * 49 * 64);
        for (int i = threadIdx.x; i < 49 * 16; i += A0_FUSED_THREADS) {
            const a0_f4 v = src[i];
            const int pos = i >> 4, c4 = (i & 15) * 4, h = pos / 7, w = pos - h * 7;
            const float x[4] = {v.x, v.y, v.z, v.w};
            uint32_t hh[4], mm[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hh[e] = __float_as_uint(x[e]);
                const float r1 = x[e] - __uint_as_float(hh[e] & 0xffff0000u);
                mm[e] = __float_as_uint(r1);
                ll[e] = __float_as_uint(r1 - __uint_as_float(mm[e] & 0xffff0000u));
            }
            uint16_t* d = plA + (h + 2) * A0_RPDA + (w + 2) * A0_P2X + c4;
            *(uint2*)(d) = uint2{__builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u), __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302u)};
            *(uint2*)(d + A0_DTERMA) = uint2{__builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u), __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302u)};
            *(uint2*)(d + 2 * A0_DTERMA) = uint2{__builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u), __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302u)};
        }
        __syncthreads();
        const AFD3X f3{plA};
        const AFD2X f2{plB};
        const int bn = b + gridDim.x < P.B ? b + gridDim.x : b;       // next observation of this workgroup (its masks are prefetched in the last slot)
        // each `between` slot requests the NEXT stage's weights and ReLU masks before this stage's epilogue runs
        a0_conv_stage_x9<ORD, 64, 4, 3, A0_RXD3>(f3, 81, ring3, epi3(b), pre3, [&] { ringp[0].init(P.wd2, 256); ringp[0].prologue(); prep[0].load(epi2(b, 0, 0), 100); });
        a0_conv_stage_x9<ORD, 32, 2, 2, A0_RXD2>(f2, 100, ringp[0], epi2(b, 0, 0), prep[0], [&] { ringp[1].init(P.wd2 + 1 * 12288, 256); ringp[1].prologue(); prep[1].load(epi2(b, 0, 1), 100); });
        a0_conv_stage_x9<ORD, 32, 2, 2, A0_RXD2>(f2, 100, ringp[1], epi2(b, 0, 1), prep[1], [&] { ringp[0].init(P.wd2 + 2 * 12288, 256); ringp[0].prologue(); prep[0].load(epi2(b, 1, 0), 100); });
        a0_conv_stage_x9<ORD, 32, 2, 2, A0_RXD2>(f2, 100, ringp[0], epi2(b, 1, 0), prep[0], [&] { ringp[1].init(P.wd2 + 3 * 12288, 256); ringp[1].prologue(); prep[1].load(epi2(b, 1, 1), 100); });
        a0_conv_stage_x9<ORD, 32, 2, 2, A0_RXD2>(f2, 100, ringp[1], epi2(b, 1, 1), prep[1], [&] { ring3.prologue(); pre3.load(epi3(bn), 81); });
    }
}

// ---- weight copies for the fused kernels, from the packed [N][K] blocks (layouts: a0_wring1 / a0_wring):
//   seg 1  conv1: fl(w/255) split exactly into three bf16 terms, 16-byte fragments ((t*32 + n)*4 + q)*3 + s   (12 C KB)
//   seg 2,3 conv2, conv3 fragment-major fp32;  seg 4,5 the flipped / phase-split matrices of the data gradients (wd3 [576][64], wd2 4 x [256][32])
//   seg 6,7 conv2, conv3 as three exact bf16 terms (a0_wring9 layout) for the split-operand forward path
//   seg 8,9 the data-gradient matrices of seg 4,5 as three exact bf16 terms (a0_wring9 layout, N = 64 / 4 x N = 32)
A0_HD uint32_t a0_bf16_trunc(float f) { return __float_as_uint(f) >> 16; }
A0_HD float a0_bf16_up(uint32_t h) { return __uint_as_float(h << 16); }
// wt2 / state: optional second destination (the target network's copies), written only when state[4] ("sync now", optim.hip) is set.
// commit (a0_adam_step_sync_wt): the folded Adam kernel left the new step count in state[5]; this kernel, the next on the stream, moves it
// to state[1] and clears the NaN flag state[0] — words the Adam kernel's workgroups were still reading.
__global__ void a0_conv_wt_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3, float* __restrict__ wt, int K1,
                                  float* __restrict__ wt2, const int* __restrict__ state, int* __restrict__ commit) {
    const bool mirror = wt2 != nullptr && state[4] != 0;
    if (commit != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { commit[1] = commit[5]; commit[0] = 0; }
    const int n1 = 48 * K1, n2 = 64 * 512, n3 = 64 * 576, n4 = 64 * 576, n5 = 4 * 32 * 256;      // n1: 32 channels x K1 x 3 terms x 2 bytes, in floats
    const int n6 = 96 * 512, n7 = 96 * 576;                                                       // conv2 / conv3 as three bf16 terms: 64 x K x 3 x 2 bytes
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float* dst = wt + i;
    if (i < n1) {      // one dword = two consecutive k of one (t, s, n, q) fragment
        const int pair = i & 3, f = i >> 2, s = f % 3, q = (f / 3) & 3, n = (f / 12) & 31, t = f / 384;
        uint32_t out = 0;
        for (int h = 0; h < 2; ++h) {
            const int k = 32 * t + 8 * q + 2 * pair + h;
            const float w = w1[n * K1 + k] / 255.0f;          // conv1 multiplies raw bytes: the reference's /255 is folded in here
            const uint32_t hi = a0_bf16_trunc(w);
            const float r1 = w - a0_bf16_up(hi);              // exact
            const uint32_t mid = a0_bf16_trunc(r1);
            const float r2 = r1 - a0_bf16_up(mid);            // exact, at most 8 significant bits left
            const uint32_t lo = a0_bf16_trunc(r2);
            out |= (s == 0 ? hi : s == 1 ? mid : lo) << (16 * h);
        }
        { *(uint32_t*)dst = out; if (mirror) *(uint32_t*)(wt2 + (dst - wt)) = out; }
        return;
    }
    i -= n1;
    {   // seg 6 / 7: exact three-term bf16 split of conv2 / conv3, same fragment layout as seg 1 with 64 output channels
        const int j6 = i - (n2 + n3 + n4 + n5);
        if (j6 >= 0) {
            if (j6 >= n6 + n7) {       // seg 8 / 9: the data-gradient matrices (seg 4 / 5) as three exact bf16 terms, a0_wring9 layout
                const int j8 = j6 - (n6 + n7), n8 = 96 * 576, n9p = 48 * 256;
                if (j8 >= n8 + 4 * n9p) return;
                const bool d2 = j8 >= n8;
                const int phase = d2 ? (j8 - n8) / n9p : 0;
                const int jj = d2 ? (j8 - n8) - phase * n9p : j8, N = d2 ? 32 : 64;
                const int pair = jj & 3, f = jj >> 2, s = f % 3, q = (f / 3) & 3, n = (f / 12) % N, t = f / (12 * N);
                uint32_t out = 0;
                for (int h = 0; h < 2; ++h) {
                    const int k = 32 * t + 8 * q + 2 * pair + h, cell = k >> 6, co = k & 63;
                    float w;
                    if (!d2) { const int kh = 2 - cell / 3, kw = 2 - cell % 3; w = w3[co * 576 + (kh * 3 + kw) * 64 + n]; }
                    else { const int kh = (phase >> 1) + 2 * (1 - (cell >> 1)), kw = (phase & 1) + 2 * (1 - (cell & 1)); w = w2[co * 512 + (kh * 4 + kw) * 32 + n]; }
                    const uint32_t hi = a0_bf16_trunc(w);
                    const float r1 = w - a0_bf16_up(hi);
                    const uint32_t mid = a0_bf16_trunc(r1);
                    const uint32_t lo = a0_bf16_trunc(r1 - a0_bf16_up(mid));
                    out |= (s == 0 ? hi : s == 1 ? mid : lo) << (16 * h);
                }
                { *(uint32_t*)dst = out; if (mirror) *(uint32_t*)(wt2 + (dst - wt)) = out; }
                return;
            }
            const bool c3 = j6 >= n6;
            const int jj = c3 ? j6 - n6 : j6, K = c3 ? 576 : 512;
            const float* wsrc = c3 ? w3 : w2;
            const int pair = jj & 3, f = jj >> 2, s = f % 3, q = (f / 3) & 3, n = (f / 12) & 63, t = f / 768;
            uint32_t out = 0;
            for (int h = 0; h < 2; ++h) {
                const float w = wsrc[n * K + 32 * t + 8 * q + 2 * pair + h];
                const uint32_t hi = a0_bf16_trunc(w);
                const float r1 = w - a0_bf16_up(hi);
                const uint32_t mid = a0_bf16_trunc(r1);
                const uint32_t lo = a0_bf16_trunc(r1 - a0_bf16_up(mid));
                out |= (s == 0 ? hi : s == 1 ? mid : lo) << (16 * h);
            }
            { *(uint32_t*)dst = out; if (mirror) *(uint32_t*)(wt2 + (dst - wt)) = out; }
            return;
        }
    }
    int N, seg;
    if (i < n2) { seg = 2; N = 64; }
    else if ((i -= n2) < n3) { seg = 3; N = 64; }
    else if ((i -= n3) < n4) { seg = 4; N = 64; }
    else if ((i -= n4) < n5) { seg = 5; N = 32; }
    else return;
    const int phase = seg == 5 ? i / 8192 : 0;
    if (seg == 5) i -= phase * 8192;
    const int j = i & 3, q = (i >> 2) & 3, n = (i >> 4) % N, c = (i >> 4) / N;
    const int k = 16 * c + 4 * j + q;
    float v;
    if (seg == 2) v = w2[n * 512 + k];
    else if (seg == 3) v = w3[n * 576 + k];
    else if (seg == 4) {                               // k = (kh'*3 + kw')*64 + co, n = ci: W3[co][2-kh'][2-kw'][ci]
        const int cell = k >> 6, co = k & 63, kh = 2 - cell / 3, kw = 2 - cell % 3;
        v = w3[co * 576 + (kh * 3 + kw) * 64 + n];
    } else {                                           // k = (a'*2 + b')*64 + co, n = ci: W2[co][ph + 2(1-a')][pw + 2(1-b')][ci]
        const int cell = k >> 6, co = k & 63, kh = (phase >> 1) + 2 * (1 - (cell >> 1)), kw = (phase & 1) + 2 * (1 - (cell & 1));
        v = w2[co * 512 + (kh * 4 + kw) * 32 + n];
    }
    *dst = v;
    if (mirror) wt2[dst - wt] = v;
}

extern "C" long long a0_net_conv_wt_floats(int C) {
    return 48LL * C * 64 + 64LL * 512 + 64LL * 576 + 64LL * 576 + 4LL * 32 * 256 + 96LL * 512 + 96LL * 576 + 96LL * 576 + 4LL * 48 * 256;
}

extern "C" int a0_net_conv_wt_refresh(const a0_encoder_weights* w, int C, float* wt, void* stream) {
    if (!w || !w->w1 || !w->w2 || !w->w3 || !wt || C < 1) return a0_fail(A0_EINVAL, "a0_net_conv_wt_refresh: bad argument");
    const long long n = a0_net_conv_wt_floats(C);
    hipLaunchKernelGGL(a0_conv_wt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w->w1, w->w2, w->w3, wt, C * 64, (float*)nullptr,
                       (const int*)nullptr, (int*)nullptr);
    return a0_fail_hip((int)hipGetLastError(), "a0_net_conv_wt_refresh");
}

// The online network's copies after an optimizer step, mirrored into the target network's when that step triggered a target sync
// (state from a0_adam_step_sync): one launch per update instead of a refresh per network.
extern "C" int a0_net_conv_wt_refresh_sync(const a0_encoder_weights* w, int C, float* wt, float* wt_target, const int* state, void* stream) {
    if (!w || !w->w1 || !w->w2 || !w->w3 || !wt || !wt_target || !state || C < 1) return a0_fail(A0_EINVAL, "a0_net_conv_wt_refresh_sync: bad argument");
    const long long n = a0_net_conv_wt_floats(C);
    hipLaunchKernelGGL(a0_conv_wt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w->w1, w->w2, w->w3, wt, C * 64, wt_target, state, (int*)nullptr);
    return a0_fail_hip((int)hipGetLastError(), "a0_net_conv_wt_refresh_sync");
}

// ... and with the step-counter commit of a0_adam_step_sync_wt (optim.hip)
int a0_conv_wt_refresh_commit(const a0_encoder_weights* w, int C, float* wt, float* wt_target, int* state, hipStream_t st) {
    if (!w || !w->w1 || !w->w2 || !w->w3) return a0_fail(A0_EINVAL, "a0_adam_step_sync_wt: encoder weights missing");
    const long long n = a0_net_conv_wt_floats(C);
    hipLaunchKernelGGL(a0_conv_wt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w->w1, w->w2, w->w3, wt, C * 64, wt_target, (const int*)state, state);
    return a0_fail_hip((int)hipGetLastError(), "a0_adam_step_sync_wt");
}

static bool a0_fused_layout(int C, int H, int W, a0_fused_args& P, size_t& lds_bytes) {
    a0_net_core n;
    if (!a0_net_core_init(n, C, H, W)) return false;
    const int M1 = n.H1 * n.W1, M2 = n.H2 * n.W2, M3 = n.H3 * n.W3;
    if (M1 > 7 * 4 * 16 || M2 > 4 * 2 * 16 || M3 > 2 * 2 * 16) return false;        // tile capacity of the three stages
    if ((C * 2) % A0_R1 || (W & 3)) return false;                                       // conv1 steps (K1/32 = 2C) fill whole ring turns; 8-byte aligned bf16 rows
    const int obs_bytes = C * H * W;
    if (obs_bytes % 16) return false;
    P.C = C; P.H = H; P.W = W; P.H1 = n.H1; P.W1 = n.W1; P.H2 = n.H2; P.W2 = n.W2; P.H3 = n.H3; P.W3 = n.W3;
    P.rp1 = n.W1 * A0_P1;
    while ((P.rp1 - n.W2) & 15) ++P.rp1;                   // 2*RP1 = 2*W2 (mod 32)
    P.rp2 = n.W2 * A0_P2;
    while ((P.rp2 - 2 * n.W3) & 31) ++P.rp2;               // RP2 = 2*W3 (mod 32)
    P.off_act1 = obs_bytes / 2;                            // the bf16 image takes 2 bytes per pixel
    P.off_act2 = P.off_act1 + ((n.H1 * P.rp1 + 3) & ~3);
    P.off_end = P.off_act2 + ((n.H2 * P.rp2 + 3) & ~3);
    lds_bytes = (size_t)P.off_end * 4;
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
    P.wt1 = wt; P.wt2 = wt + 48LL * C * 64; P.wt3 = P.wt2 + 64LL * 512;
    P.wx2 = P.wt3 + 64LL * 576 + 64LL * 576 + 4LL * 32 * 256; P.wx3 = P.wx2 + 96LL * 512;      // behind the data-gradient copies
    P.b1 = w->b1; P.b2 = w->b2; P.b3 = w->b3;
    P.act1 = act1; P.act2 = act2; P.act3 = act3; P.B = B;
    // 16-row blocks per wave: conv1 ceil(MB1/4), conv2 ceil(MB2/2), conv3 ceil(MB3/2); exact for 84x84, generous otherwise
    const int mb1 = (P.H1 * P.W1 + 15) / 16, mb2 = (P.H2 * P.W2 + 15) / 16, mb3 = (P.H3 * P.W3 + 15) / 16;
    const bool standard = ((mb1 + 3) / 4 == 7) && ((mb2 + 1) / 2 == 3) && ((mb3 + 1) / 2 == 2) && W == 84;
    // instantiations: split-operand (all layers on the bf16 pipe; 84x84, C = 4; one observation per workgroup, or looping), fp32 conv2/conv3 for 84-wide inputs, generic
    static const bool no_x9 = getenv("A0_NO_X9") != nullptr;
    const bool x9 = standard && H == 84 && C == 4 && !no_x9;
    int which = x9 ? 2 : (standard ? 1 : 0);
    if (x9) lds = A0_X9_LDS_BYTES;            // image + act1 term planes (act2 planes reuse the image) + the biases
    static_assert(3 * 9 * A0_RP2X * 2 <= 2 * 4 * 84 * 84, "the act2 term planes fit into the dead image");
    static const int grid_cap = getenv("A0_ENC_GRID") ? atoi(getenv("A0_ENC_GRID")) : 256;
    const int gridx = (which == 2 && grid_cap > 0 && B > grid_cap) ? grid_cap : B;
    if (which == 2 && gridx < B) which = 3;                  // the looping instantiation of the split-operand kernel
    static size_t configured[6] = {0, 0, 0, 0, 0, 0};
    const int six = a0_x9_products_now() == 6;
    if (six && which >= 2) which += 2;       // 4 / 5: the six-product forms of 2 / 3
    typedef void (*a0_enc_kern)(a0_fused_args);
    const a0_enc_kern fn = which == 5 ? a0_encoder_fused_kernel<7, 3, 2, 84, true, true, 2> : which == 4 ? a0_encoder_fused_kernel<7, 3, 2, 84, true, false, 2>
                         : which == 3 ? a0_encoder_fused_kernel<7, 3, 2, 84, true, true> : which == 2 ? a0_encoder_fused_kernel<7, 3, 2, 84, true>
                         : which == 1 ? a0_encoder_fused_kernel<7, 3, 2, 84, false> : a0_encoder_fused_kernel<7, 4, 2, 0, false>;
    if (lds > configured[which]) {
        A0_HIP_THROW(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[which] = lds;
    }
    // split-operand kernel: at most one workgroup per CU (152 KB of LDS each), looping over its observations (b += gridDim.x) when there are more
    // observations than that: ring set-up is paid once and the next observation's conv1 weights are requested behind conv3 (-4 % per 512
    // observations, tools/ubench_encoder_fwd.py).  A0_ENC_GRID: tuning aid (0 = one workgroup per observation)
    // algorithmic FLOP of the three convolutions: 2 * (M1*32*K1 + M2*64*512 + M3*64*576) per observation
    const double per_obs = 2.0 * ((double)P.H1 * P.W1 * 32 * (P.C * 64) + (double)P.H2 * P.W2 * 64 * 512 + (double)P.H3 * P.W3 * 64 * 576);
    A0_LAUNCH_PROBED(A0_TAG_ENCODER_FUSED, per_obs * B, fn, dim3((which == 3 || which == 5) ? gridx : B), dim3(A0_FUSED_THREADS), lds, (hipStream_t)stream, P);
    A0_HIP_THROW(hipGetLastError());
    return A0_OK;
    A0_CATCH
}

// n <= 3 passes of a0_net_encoder_fwd_fused in one launch (84 x 84 x 4 observations: the split-operand kernel); same outputs, bit for bit.
extern "C" int a0_net_encoder_fwd_fused_multi(int C, int H, int W, int n, const a0_encoder_pass* pass, void* stream) {
    A0_TRY
    if (n < 1 || n > 3 || !pass) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused_multi: 1 to 3 passes");
    static const bool no_x9 = getenv("A0_NO_X9") != nullptr;
    if (!(C == 4 && H == 84 && W == 84) || no_x9) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused_multi: 4 x 84 x 84 observations on the split-operand kernel only");
    a0_fused_multi_args M;
    size_t lds = 0;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        const a0_encoder_pass& q = pass[i];
        if (!q.wt || !q.w || !q.w->b1 || !q.w->b2 || !q.w->b3 || !q.f || !q.f->frames || !q.act3 || q.B < 1) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused_multi: bad pass");
        if ((q.f->sample_stride % 16) || (q.f->chan_off % 16) || (((uintptr_t)q.f->frames) % 16)) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused_multi: frames must be 16-byte aligned");
        a0_fused_args& P = M.p[i];
        if (!a0_fused_layout(C, H, W, P, lds)) return a0_fail(A0_EINVAL, "a0_net_encoder_fwd_fused_multi: observation shape");
        P.frames = q.f->frames; P.slot = q.f->slot; P.sample_stride = q.f->sample_stride; P.chan_off = q.f->chan_off;
        P.wt1 = q.wt; P.wt2 = q.wt + 48LL * C * 64; P.wt3 = P.wt2 + 64LL * 512;
        P.wx2 = P.wt3 + 64LL * 576 + 64LL * 576 + 4LL * 32 * 256; P.wx3 = P.wx2 + 96LL * 512;
        P.b1 = q.w->b1; P.b2 = q.w->b2; P.b3 = q.w->b3;
        P.act1 = q.act1; P.act2 = q.act2; P.act3 = q.act3; P.B = q.B;
        total += q.B;
    }
    for (int i = n; i < 3; ++i) M.p[i] = M.p[0];
    // at most one workgroup per CU; every pass at least one, shares proportional to the observations
    static const int grid_cap = getenv("A0_ENC_GRID") ? atoi(getenv("A0_ENC_GRID")) : 256;
    const int cap = grid_cap > 0 ? grid_cap : 256;
    int grid = (int)(total < cap ? total : cap), used = 0;
    if (grid < n) grid = n;
    for (int i = 0; i < n; ++i) {
        M.first[i] = used;
        int share = (i == n - 1) ? grid - used : (int)((long long)grid * pass[i].B / total);
        if (share < 1) share = 1;
        if (share > pass[i].B) share = pass[i].B;
        used += share;
    }
    for (int i = n; i <= 3; ++i) M.first[i] = 0x7fffffff;
    M.first[n] = used;
    lds = A0_X9_LDS_BYTES;
    const int six = a0_x9_products_now() == 6;
    auto kern = six ? a0_encoder_fused_multi_kernel<2> : a0_encoder_fused_multi_kernel<4>;
    static bool configured[2] = {false, false};
    if (!configured[six]) {
        A0_HIP_THROW(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[six] = true;
    }
    const a0_fused_args& P0 = M.p[0];
    const double per_obs = 2.0 * ((double)P0.H1 * P0.W1 * 32 * (P0.C * 64) + (double)P0.H2 * P0.W2 * 64 * 512 + (double)P0.H3 * P0.W3 * 64 * 576);
    A0_LAUNCH_PROBED(A0_TAG_ENCODER_FUSED, per_obs * (double)total, kern, dim3(used), dim3(A0_FUSED_THREADS), lds, (hipStream_t)stream, M);
    A0_HIP_THROW(hipGetLastError());
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_net_encoder_dgrad_fused_supported(int C, int H, int W) { return (C >= 1 && H == 84 && W == 84) ? 1 : 0; }

// d3 [B][7][7][64] (ReLU-masked) -> d2 [B][9][9][64], d1 [B][20][20][32], masked by act2 / act1 > 0; wt from a0_net_conv_wt_refresh.
extern "C" int a0_net_encoder_dgrad_fused(int C, int H, int W, const float* wt, const float* d3, const float* act1, const float* act2, int B, float* d2,
                                          float* d1, void* stream) {
    A0_TRY
    if (!wt || !d3 || !act1 || !act2 || !d2 || !d1 || B < 1) return a0_fail(A0_EINVAL, "a0_net_encoder_dgrad_fused: bad argument");
    if (!a0_net_encoder_dgrad_fused_supported(C, H, W)) return a0_fail(A0_EINVAL, "a0_net_encoder_dgrad_fused: 84x84 observations only");
    a0_dgrad_args P;
    P.d3 = d3; P.act1 = act1; P.act2 = act2; P.d2 = d2; P.d1 = d1; P.B = B;
    P.wd3 = wt + 48LL * C * 64 + 64LL * 512 + 64LL * 576;
    P.wd2 = P.wd3 + 64LL * 576;
    P.rpa = 11 * A0_P2; while ((P.rpa - 2 * 9) & 31) ++P.rpa;      // conflict-free A reads: RP = 2 * (output width) (mod 32)
    P.rpb = 11 * A0_P2; while ((P.rpb - 2 * 10) & 31) ++P.rpb;
    static const bool no_x9 = getenv("A0_NO_X9") != nullptr;       // fp32-chain variant (as for the forward kernel)
    const bool x9 = !no_x9;
    if (x9) {      // split weights behind the forward path's (segments 8, 9)
        P.wd3 = P.wd2 + 4LL * 32 * 256 + 96LL * 512 + 96LL * 576;
        P.wd2 = P.wd3 + 96LL * 576;
    }
    const size_t lds = x9 ? (size_t)3 * (A0_DTERMA + A0_DTERMB) * 2 + (A0_KSPLIT_D ? (4 * 4 + 4 * 3) * 1024 : 0) : (size_t)11 * (P.rpa + P.rpb) * 4;      // + the exchange regions of the N-stationary stages
    static bool configured[3] = {false, false, false};
    const int kv = x9 ? (a0_x9_products_now() == 6 ? 2 : 1) : 0;
    typedef void (*a0_dg_kern)(a0_dgrad_args);
    const a0_dg_kern fn = kv == 2 ? a0_encoder_dgrad_fused_x9_kernel<2> : kv == 1 ? a0_encoder_dgrad_fused_x9_kernel<4> : a0_encoder_dgrad_fused_kernel;
    if (!configured[kv]) {
        A0_HIP_THROW(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[kv] = true;
    }
    // at most one workgroup per CU, each looping over its observations (b += gridDim.x): the next observation's masks and the first
    // stage's weights are prefetched in the last stage's `between` slot instead of at workgroup start (-5 us per 512 observations)
    static const int grid_cap = getenv("A0_DGRAD_GRID") ? atoi(getenv("A0_DGRAD_GRID")) : 256;
    const int gridx = (grid_cap > 0 && B > grid_cap) ? grid_cap : B;
    A0_LAUNCH_PROBED(A0_TAG_ENCODER_DGRAD_FUSED, 2.0 * (81.0 * 64 * 576 + 400.0 * 32 * 256) * B, fn, dim3(x9 ? gridx : B), dim3(A0_FUSED_THREADS), lds, (hipStream_t)stream, P);
    A0_HIP_THROW(hipGetLastError());
    return A0_OK;
    A0_CATCH
}
