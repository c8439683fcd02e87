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
// The MFMA is v_mfma_f32_16x16x4_f32: 16-row tiles fit M = 400 / 81 / 49 rows with little padding and split evenly over the
// eight waves.  conv2 / conv3 are the same k-ascending fp32 fmaf chains as in igemm.h.  conv1 feeds the RAW byte values to the
// MFMA and folds the reference's x/255 into the weight copy (w/255): its inner loop is instruction-issue bound and the exact
// three-instruction division per operand element cost more than the MFMA itself; sum_k x_k*fl(w_k/255) and sum_k fl(x_k/255)*w_k
// carry the same two roundings per term, so the result agrees with the unfused path (and the reference) to fp32 rounding —
// tests/test_gpu_engine.py::test_fused_encoder_matches_unfused, rtol 2e-6.
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
    int off_act1, off_act2, off_end;     // float offsets of the LDS regions behind the u8 observation
    int rp1, rp2;                        // LDS row pitches (floats) of act1 / act2, padded so that A-fragment reads are bank-conflict free
};

// Eight waves per workgroup = two per SIMD: while one wave waits for its LDS operands the other keeps the matrix pipe busy.
constexpr int A0_FUSED_WAVES = 8;
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
template <int WC> struct AF1 {   // conv1 8x8/4 over the u8 observation [C][H][W]; k = c*64 + kh*8 + kw
    const uint8_t* obs; int HW, W, W1;
    A0_D int width() const { return WC > 0 ? WC : W; }
    A0_D int row(int m) const { const int oh = m / W1, ow = m - oh * W1; return (4 * oh) * width() + 4 * ow; }
    A0_D int chunk_off(int c) const { return (c >> 2) * HW + 2 * (c & 3) * width(); }
    A0_D int step_off(int j) const { return (j >> 1) * width() + 4 * (j & 1); }
    // raw byte value 0..255: the 1/255 of the reference's normalisation (agent.py:27,132) is folded into the k-major weight copy.
    // The ring keeps the byte as loaded; the conversion happens next to the MFMA that consumes it (a convert placed behind the
    // read would wait for it on the spot and undo the prefetch).
    typedef uint32_t Raw;
    A0_D Raw load(int addr) const { return obs[addr]; }
    static A0_D float value(Raw r) { return (float)r; }
};
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
    int nch;
    A0_D void init(const float* wp, int K) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int n = (wave % WN) * (NBW * 16) + (lane & 15), q = lane >> 4;
        base = wp + (n * 4 + q) * 4;
        nch = K >> 4;
    }
    A0_D void fill(int slot, int c) {
        c = c < nch ? c : nch - 1;                     // past the end: re-read the last chunk (never consumed), keeps the loop branch-free
#pragma unroll
        for (int j = 0; j < NBW; ++j) v[slot][j] = *(const a0_f4*)(base + (long long)c * (N * 16) + j * 256);
    }
    A0_D void prologue() {
#pragma unroll
        for (int u = 0; u < R; ++u) fill(u, u);
    }
};
A0_D float a0_f4_get(const a0_f4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// ---- epilogue policies.  Values an epilogue needs from global memory (bias, ReLU masks) are requested at the START of the layer:
// a load issued in the epilogue would wait behind the next layer's weight prologue.
template <int OWC>
struct EpiFwd {                 // y = relu(acc + bias[n]) -> LDS image [oh][ow][n] (pitch / row pitch) and / or global [m][n]
    static constexpr bool PER_ELEM = false;
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
        if (glb) glb[(long long)m * N + n] = v;
    }
};
template <int OW, int S>
struct EpiBwd {                 // dx = mask > 0 ? acc : 0 at pixel (oh*S + ph, ow*S + pw) of a Wfull-wide NHWC image; optional padded LDS image
    static constexpr bool PER_ELEM = true;
    const float* mask; float* dst; float* lds; int pitch, rp, Wfull, ph, pw, N;
    A0_D long long gi(int m, int n) const { const int oh = m / OW, ow = m - oh * OW; return (long long)((oh * S + ph) * Wfull + ow * S + pw) * N + n; }
    A0_D float pre_col(int) const { return 0.f; }
    A0_D float pre_elem(int m, int n) const { return mask[gi(m, n)]; }
    A0_D void emit(int m, int n, float acc, float pre) const {
        const float v = pre > 0.f ? acc : 0.f;
        dst[gi(m, n)] = v;
        if (lds) { const int oh = m / OW; lds[oh * rp + (m - oh * OW) * pitch + n] = v; }
    }
};

// One convolution-shaped GEMM: C[M x N] = A[M x K] * B, A read through AF from LDS, B from the register ring.  Waves: WN along
// N, 8/WN along M; a wave owns MBW 16-row blocks (interleaved) x NBW 16-column blocks.  (conv1 2 x 4 waves, conv2 / conv3 4 x 2.)
// The k loop is one software pipeline over all K/4 MFMA steps: the LDS operands of step g + PD are requested before the MFMAs of
// step g are issued (register ring of PD + 1 slots); chunk c + R of the weights is requested as soon as chunk c has been consumed.
// Everything is branch-free (16-row blocks beyond M recompute row 0 and are never stored) and pinned with sched_barriers — left
// alone, the scheduler sinks every prefetch down to its first use.  `between` runs after the last MFMA and before the epilogue:
// the caller issues the next layer's weight prologue there, so its L2 latency hides behind the epilogue and the barrier.
template <int N, int WN, int MBW, int PD, int R, class AF, class EPI, class Between>
A0_D void a0_conv_stage(const AF& af, int M, int K, a0_wring<N, WN, R>& ring, const EPI& epi, Between&& between) {
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

    float pc[NBW], pe[EPI::PER_ELEM ? MBW : 1][NBW][4];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int n = (wn * NBW + j) * 16 + r16;
        pc[j] = epi.pre_col(n);
        if (EPI::PER_ELEM) {
#pragma unroll
            for (int i = 0; i < MBW; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = (wmg + i * WMG) * 16 + 4 * q + r;
                    pe[EPI::PER_ELEM ? i : 0][j][r] = epi.pre_elem(m < M ? m : M - 1, n);
                }
        }
    }
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
            ring.fill(u, c + R);
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
                    if (m < M) epi.emit(m, n, acc[i][j][r], EPI::PER_ELEM ? pe[EPI::PER_ELEM ? i : 0][j][r] : pc[j]);
                }
            }
        }
    }
    __syncthreads();
}

// Register-ring depths (16-k chunks in flight per wave): >= 2 us of MFMA work ahead of every weight load.
constexpr int A0_R1 = 4, A0_R2 = 8, A0_R3 = 12;

template <int MBW1, int MBW2, int MBW3, int WC>
__global__ __launch_bounds__(A0_FUSED_THREADS) void a0_encoder_fused_kernel(a0_fused_args P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* obs = smem;
    float* fl = (float*)smem;
    float* act1 = fl + P.off_act1;
    float* act2 = fl + P.off_act2;
    const int obs_bytes = P.C * P.H * P.W;
    const int M1 = P.H1 * P.W1, M2 = P.H2 * P.W2, M3 = P.H3 * P.W3;
    a0_wring<32, 2, A0_R1> ring1;
    a0_wring<64, 4, A0_R2> ring2;
    a0_wring<64, 4, A0_R3> ring3;
    ring1.init(P.wt1, P.C * 64);
    ring2.init(P.wt2, 512);
    ring3.init(P.wt3, 576);
    ring1.prologue();
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        // ---- observation -> LDS (16 B per lane)
        const long long s = P.slot ? (long long)P.slot[b] : (long long)b;
        const uint4* src = (const uint4*)(P.frames + s * P.sample_stride + P.chan_off);
        for (int i = threadIdx.x; i < (obs_bytes >> 4); i += A0_FUSED_THREADS) ((uint4*)obs)[i] = src[i];
        __syncthreads();
        AF1<WC> f1{obs, P.H * P.W, P.W, P.W1};
        constexpr int OW1 = WC == 84 ? 20 : 0, OW2 = WC == 84 ? 9 : 0;      // output widths of conv1 / conv2 when the input is 84 wide
        const EpiFwd<OW1> e1{P.b1, act1, A0_P1, P.rp1, P.W1, P.act1 ? P.act1 + (long long)b * M1 * 32 : nullptr, 32};
        // conv1's 16-row blocks do not divide evenly over the four M groups (25 blocks: 7 + 6 + 6 + 6): groups that own one block less
        // run the MBW1 - 1 instantiation instead of recomputing a dummy block (same barrier count on both paths)
        const int wmg1 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) / 2;
        if (MBW1 > 1 && wmg1 + (MBW1 - 1) * 4 >= ((M1 + 15) >> 4))
            a0_conv_stage<32, 2, (MBW1 > 1 ? MBW1 - 1 : 1), 1, A0_R1>(f1, M1, P.C * 64, ring1, e1, [&] { ring2.prologue(); });
        else
            a0_conv_stage<32, 2, MBW1, 1, A0_R1>(f1, M1, P.C * 64, ring1, e1, [&] { ring2.prologue(); });
        AF2 f2{act1, P.rp1, P.W2};
        const EpiFwd<OW2> e2{P.b2, act2, A0_P2, P.rp2, P.W2, P.act2 ? P.act2 + (long long)b * M2 * 64 : nullptr, 64};
        a0_conv_stage<64, 4, MBW2, 3, A0_R2>(f2, M2, 512, ring2, e2, [&] { ring3.prologue(); });
        AF3 f3{act2, P.rp2, P.W3};
        const EpiFwd<0> e3{P.b3, nullptr, 0, 0, 1, P.act3 + (long long)b * M3 * 64, 64};
        a0_conv_stage<64, 4, MBW3, 3, A0_R3>(f3, M3, 576, ring3, e3, [&] { ring1.prologue(); });
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
        const EpiBwd<9, 1> e3{P.act2 + (long long)b * 81 * 64, P.d2 + (long long)b * 81 * 64, imgB + P.rpb + A0_P2, A0_P2, P.rpb, 9, 0, 0, 64};
        a0_conv_stage<64, 4, 3, 3, A0_RD3>(f3, 81, 576, ring3, e3, [&] { ringp[0].init(P.wd2, 256); ringp[0].prologue(); });
        AFD2 f2{imgB, P.rpb};
        const float* m1 = P.act1 + (long long)b * 400 * 32;
        float* o1 = P.d1 + (long long)b * 400 * 32;
        const EpiBwd<10, 2> p00{m1, o1, nullptr, 0, 0, 20, 0, 0, 32}, p01{m1, o1, nullptr, 0, 0, 20, 0, 1, 32}, p10{m1, o1, nullptr, 0, 0, 20, 1, 0, 32},
            p11{m1, o1, nullptr, 0, 0, 20, 1, 1, 32};
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[0], p00, [&] { ringp[1].init(P.wd2 + 1 * 8192, 256); ringp[1].prologue(); });
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[1], p01, [&] { ringp[0].init(P.wd2 + 2 * 8192, 256); ringp[0].prologue(); });
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[0], p10, [&] { ringp[1].init(P.wd2 + 3 * 8192, 256); ringp[1].prologue(); });
        a0_conv_stage<32, 2, 2, 3, A0_RD2>(f2, 100, 256, ringp[1], p11, [&] { ring3.prologue(); });
    }
}

// ---- fragment-major weight copies (layout: see a0_wring) from the packed [N][K] blocks: conv1 (pre-divided by 255), conv2, conv3 for
// the forward pass, then the flipped / phase-split matrices of the data gradients: wd3 [576][64], wd2 4 x [256][32]
__global__ void a0_conv_wt_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3, float* __restrict__ wt, int K1) {
    const int n1 = 32 * K1, n2 = 64 * 512, n3 = 64 * 576, n4 = 64 * 576, n5 = 4 * 32 * 256;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float* dst = wt + i;
    int N, seg;
    if (i < n1) { seg = 1; N = 32; }
    else if ((i -= n1) < n2) { seg = 2; N = 64; }
    else if ((i -= n2) < n3) { seg = 3; N = 64; }
    else if ((i -= n3) < n4) { seg = 4; N = 64; }
    else if ((i -= n4) < n5) { seg = 5; N = 32; }
    else return;
    const int phase = seg == 5 ? i / 8192 : 0;
    if (seg == 5) i -= phase * 8192;
    const int j = i & 3, q = (i >> 2) & 3, n = (i >> 4) % N, c = (i >> 4) / N;
    const int k = 16 * c + 4 * j + q;
    float v;
    if (seg == 1) v = w1[n * K1 + k] / 255.0f;        // conv1 reads raw bytes: the /255 of the reference's normalisation is folded in here
    else if (seg == 2) v = w2[n * 512 + k];
    else if (seg == 3) v = w3[n * 576 + k];
    else if (seg == 4) {                               // k = (kh'*3 + kw')*64 + co, n = ci: W3[co][2-kh'][2-kw'][ci]
        const int cell = k >> 6, co = k & 63, kh = 2 - cell / 3, kw = 2 - cell % 3;
        v = w3[co * 576 + (kh * 3 + kw) * 64 + n];
    } else {                                           // k = (a'*2 + b')*64 + co, n = ci: W2[co][ph + 2(1-a')][pw + 2(1-b')][ci]
        const int cell = k >> 6, co = k & 63, kh = (phase >> 1) + 2 * (1 - (cell >> 1)), kw = (phase & 1) + 2 * (1 - (cell & 1));
        v = w2[co * 512 + (kh * 4 + kw) * 32 + n];
    }
    *dst = v;
}

extern "C" long long a0_net_conv_wt_floats(int C) { return 32LL * C * 64 + 64LL * 512 + 64LL * 576 + 64LL * 576 + 4LL * 32 * 256; }

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
    if ((C * 4) % A0_R1) return false;                                                  // conv1 chunks (K1/16 = 4C) must fill whole ring turns
    const int obs_bytes = C * H * W;
    if (obs_bytes % 16) return false;
    P.C = C; P.H = H; P.W = W; P.H1 = n.H1; P.W1 = n.W1; P.H2 = n.H2; P.W2 = n.W2; P.H3 = n.H3; P.W3 = n.W3;
    P.rp1 = n.W1 * A0_P1;
    while ((P.rp1 - n.W2) & 15) ++P.rp1;                   // 2*RP1 = 2*W2 (mod 32)
    P.rp2 = n.W2 * A0_P2;
    while ((P.rp2 - 2 * n.W3) & 31) ++P.rp2;               // RP2 = 2*W3 (mod 32)
    P.off_act1 = obs_bytes / 4;
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
    P.wt1 = wt; P.wt2 = wt + 32LL * C * 64; P.wt3 = P.wt2 + 64LL * 512;
    P.b1 = w->b1; P.b2 = w->b2; P.b3 = w->b3;
    P.act1 = act1; P.act2 = act2; P.act3 = act3; P.B = B;
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

extern "C" int a0_net_encoder_dgrad_fused_supported(int C, int H, int W) { return (C >= 1 && H == 84 && W == 84) ? 1 : 0; }

// d3 [B][7][7][64] (ReLU-masked) -> d2 [B][9][9][64], d1 [B][20][20][32], masked by act2 / act1 > 0; wt from a0_net_conv_wt_refresh.
extern "C" int a0_net_encoder_dgrad_fused(int C, int H, int W, const float* wt, const float* d3, const float* act1, const float* act2, int B, float* d2,
                                          float* d1, void* stream) {
    A0_TRY
    if (!wt || !d3 || !act1 || !act2 || !d2 || !d1 || B < 1) return a0_fail(A0_EINVAL, "a0_net_encoder_dgrad_fused: bad argument");
    if (!a0_net_encoder_dgrad_fused_supported(C, H, W)) return a0_fail(A0_EINVAL, "a0_net_encoder_dgrad_fused: 84x84 observations only");
    a0_dgrad_args P;
    P.d3 = d3; P.act1 = act1; P.act2 = act2; P.d2 = d2; P.d1 = d1; P.B = B;
    P.wd3 = wt + 32LL * C * 64 + 64LL * 512 + 64LL * 576;
    P.wd2 = P.wd3 + 64LL * 576;
    P.rpa = 11 * A0_P2; while ((P.rpa - 2 * 9) & 31) ++P.rpa;      // conflict-free A reads: RP = 2 * (output width) (mod 32)
    P.rpb = 11 * A0_P2; while ((P.rpb - 2 * 10) & 31) ++P.rpb;
    const size_t lds = (size_t)11 * (P.rpa + P.rpb) * 4;
    static bool configured = false;
    if (!configured) {
        A0_HIP_THROW(hipFuncSetAttribute((const void*)a0_encoder_dgrad_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const bool probed = a0_probe_start(A0_TAG_ENCODER_DGRAD_FUSED, (hipStream_t)stream);
    hipLaunchKernelGGL(a0_encoder_dgrad_fused_kernel, dim3(B), dim3(A0_FUSED_THREADS), lds, (hipStream_t)stream, P);
    if (probed) a0_probe_stop((hipStream_t)stream, 2.0 * (81.0 * 64 * 576 + 400.0 * 32 * 256) * B);
    A0_HIP_THROW(hipGetLastError());
    return A0_OK;
    A0_CATCH
}
