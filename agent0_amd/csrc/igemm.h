// Implicit-GEMM kernel on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Why fp32 MFMA: the reference network runs in fp32 (agent0/deepq/model.py, no autocast) and the north star asks
// for Q-values / TD losses within a tight tolerance; gfx950's f32-input MFMA is an exact k-ordered fmaf chain at
// the fp32 vector peak (157 TFLOP/s), so parity is a few ulp while the convs still run on the matrix pipe.
//
// Tiling: 256 threads = 4 wave64s laid out WM x WN; each wave owns MT x NT accumulator blocks of 32x32, so the
// block tile is BX = WM*MT*32 rows by BY = WN*NT*32 columns, BK = 32 deep.  Operands are gathered by the policies
// in operands.h into k-major LDS tiles As[k][x], Bs[k][y]; a fragment read As[2s + (lane>>5)][x0 + (lane&31)] is
// one conflict-free ds_read_b32 per 32x32x2 MFMA operand (the MFMA takes 64 cycles, so LDS is never the limit).
// Two tiles of global loads are in flight in registers; the LDS tiles are double-buffered (one barrier per 32 k).
#pragma once
#include "operands.h"

#if defined(__HIPCC__)

typedef float a0_acc16 __attribute__((ext_vector_type(16)));

template <class OP, int BX, int MODE = OP::MODE> struct a0_stager;

template <class OP, int BX> struct a0_stager<OP, BX, A0_KC> {
    static constexpr int R = BX / 32;
    static constexpr int LD = BX + 1;
    typename OP::Row rows[R];
    typename OP::KInfo ki;        // gather-table entry of the tile about to be fetched (loaded one tile ahead)
    struct Slot { typename OP::Raw raw[R]; unsigned okmask; };
    A0_D void init(const typename OP::Params& P, int x0, int X, int kb, int ke, int tid) {
#pragma unroll
        for (int j = 0; j < R; ++j) rows[j] = OP::row(P, x0 + (tid >> 3) + 32 * j, X);
        ki = OP::kinfo(P, kb + 4 * (tid & 7), ke);
    }
    A0_D void fetch(const typename OP::Params& P, Slot& s, int k0, int ke, int tid) {
        s.okmask = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) { bool ok; s.raw[j] = OP::load(P, rows[j], ki, ok); s.okmask |= ok ? (1u << j) : 0u; }
        ki = OP::kinfo(P, k0 + 32 + 4 * (tid & 7), ke);     // for the NEXT fetch; its latency hides behind this tile's MFMAs
    }
    A0_D void commit(const Slot& s, float* lds, int tid) const {
        const int kk = 4 * (tid & 7);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = (tid >> 3) + 32 * j;
            const a0_f4 v = OP::finish(s.raw[j], (s.okmask >> j) & 1u);
            lds[(kk + 0) * LD + r] = v.x;
            lds[(kk + 1) * LD + r] = v.y;
            lds[(kk + 2) * LD + r] = v.z;
            lds[(kk + 3) * LD + r] = v.w;
        }
    }
};

template <class OP, int BX> struct a0_stager<OP, BX, A0_XC> {
    static constexpr int R = BX / 32;
    static constexpr int LD = BX + 4;
    static constexpr int Q = BX / 4;   // 16-byte groups per k row
    typename OP::XInfo xi[R];
    struct Slot { typename OP::Raw raw[R]; unsigned okmask; };
    A0_D void init(const typename OP::Params& P, int x0, int X, int, int, int tid) {
#pragma unroll
        for (int j = 0; j < R; ++j) xi[j] = OP::xinfo(P, x0 + 4 * ((tid + 256 * j) % Q), X);
    }
    A0_D void fetch(const typename OP::Params& P, Slot& s, int k0, int ke, int tid) {
        s.okmask = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) { bool ok; s.raw[j] = OP::load(P, k0 + (tid + 256 * j) / Q, ke, xi[j], ok); s.okmask |= ok ? (1u << j) : 0u; }
    }
    A0_D void commit(const Slot& s, float* lds, int tid) const {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int f = tid + 256 * j;
            *(a0_f4*)&lds[(f / Q) * LD + 4 * (f % Q)] = OP::finish(s.raw[j], (s.okmask >> j) & 1u);
        }
    }
};

template <class OA, class OB, class EP, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) void a0_igemm_kernel(typename OA::Params pa, typename OB::Params pb,
                                                        typename EP::Params pe, int X, int Y, int K, int kchunk) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BK = 32;
    constexpr int BX = WM * MT * 32;
    constexpr int BY = WN * NT * 32;
    typedef a0_stager<OA, BX> SA;
    typedef a0_stager<OB, BY> SB;
    constexpr int LDA = SA::LD, LDB = SB::LD;
    // two LDS buffers per operand: tile t+1 is written while tile t is being multiplied, one barrier per tile
    __shared__ __attribute__((aligned(16))) float As[2 * BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY;
    const int kb = blockIdx.z * kchunk;
    const int ke = (K < kb + kchunk) ? K : (kb + kchunk);

    SA sa; SB sb;
    sa.init(pa, x0, X, kb, ke, tid);
    sb.init(pb, y0, Y, kb, ke, tid);

    a0_acc16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two tiles of global loads stay in flight: tile t+2 is requested right after tile t has been committed to LDS, so
    // its latency is covered by the MFMAs of tiles t and t+1 (one tile of cover is not enough at 1-3 waves per SIMD).
    typename SA::Slot sa0, sa1;
    typename SB::Slot sb0, sb1;
    if (kb < ke) { sa.fetch(pa, sa0, kb, ke, tid); sb.fetch(pb, sb0, kb, ke, tid); }
    if (kb + BK < ke) { sa.fetch(pa, sa1, kb + BK, ke, tid); sb.fetch(pb, sb1, kb + BK, ke, tid); }

    // EP::ROWSUM_A (weight gradients): row sums of the A tiles, accumulated from LDS by the blockIdx.y == 0 column of workgroups.
    // Thread (g, xr) adds rows kk = g, g + G, ... of column xr of every tile; the G partial sums meet in LDS after the k loop.
    constexpr int RS_G = 256 / BX;
    float rowsum = 0.f;
    auto rowsum_tile = [&](int buf) {
        if constexpr (EP::ROWSUM_A) {
            if (blockIdx.y == 0) {
                const int xr = tid % BX, g = tid / BX;
#pragma unroll
                for (int kk = 0; kk < BK / RS_G; ++kk) rowsum += As[buf * BK * LDA + (g + kk * RS_G) * LDA + xr];
            }
        }
    };

    const float* ap0 = As + (lane >> 5) * LDA + wm * (MT * 32) + (lane & 31);
    const float* bp0 = Bs + (lane >> 5) * LDB + wn * (NT * 32) + (lane & 31);

    auto mma_tile = [&](int buf) {
        const float* ap = ap0 + buf * BK * LDA;
        const float* bp = bp0 + buf * BK * LDB;
        // fragment reads run one k-step ahead of the MFMAs that consume them
        float a[2][MT], b[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[0][i] = ap[i * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[0][j] = bp[j * 32];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) a[nxt][i] = ap[2 * (s + 1) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) b[nxt][j] = bp[2 * (s + 1) * LDB + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the reads of step s+1 ahead of the MFMAs of step s (the scheduler sinks them otherwise)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Pipeline: registers hold tiles t+1 and t+2 (global loads in flight), LDS buffer t&1 holds tile t.  In iteration t the waves
    // first move tile t+1 from registers into the OTHER buffer (its last readers finished before the barrier that ended iteration
    // t-1), request tile t+3 into the freed registers, then multiply tile t; one barrier closes the iteration.
    if (kb < ke) {
        sa.commit(sa0, As, tid);
        sb.commit(sb0, Bs, tid);
        if (kb + 2 * BK < ke) { sa.fetch(pa, sa0, kb + 2 * BK, ke, tid); sb.fetch(pb, sb0, kb + 2 * BK, ke, tid); }
    }
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += 2 * BK) {
        if (k0 + BK < ke) {
            sa.commit(sa1, As + BK * LDA, tid);
            sb.commit(sb1, Bs + BK * LDB, tid);
            if (k0 + 3 * BK < ke) { sa.fetch(pa, sa1, k0 + 3 * BK, ke, tid); sb.fetch(pb, sb1, k0 + 3 * BK, ke, tid); }
        }
        rowsum_tile(0);
        mma_tile(0);
        __syncthreads();
        if (k0 + BK >= ke) break;
        if (k0 + 2 * BK < ke) {
            sa.commit(sa0, As, tid);
            sb.commit(sb0, Bs, tid);
            if (k0 + 4 * BK < ke) { sa.fetch(pa, sa0, k0 + 4 * BK, ke, tid); sb.fetch(pb, sb0, k0 + 4 * BK, ke, tid); }
        }
        rowsum_tile(1);
        mma_tile(1);
        __syncthreads();
    }
    if constexpr (EP::ROWSUM_A) {
        if (blockIdx.y == 0) {               // As is free: the loop ends with a barrier
            As[tid] = rowsum;
            __syncthreads();
            if (tid < BX && x0 + tid < X) {
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < RS_G; ++g) t += As[g * BX + tid];
                EP::store_rowsum(pe, x0 + tid, t, blockIdx.z);
            }
        }
    }

    // C/D layout of v_mfma_f32_32x32x2_f32: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int z = blockIdx.z;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int y = y0 + wn * (NT * 32) + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (x < X && y < Y) EP::store(pe, x, y, acc[i][j][r], z);
            }
        }
}

template <class OA, class OB, class EP, int WM, int WN, int MT, int NT>
static inline hipError_t a0_igemm_launch(hipStream_t st, const typename OA::Params& pa, const typename OB::Params& pb,
                                         const typename EP::Params& pe, int X, int Y, int K, int splits) {
    constexpr int BX = WM * MT * 32, BY = WN * NT * 32;
    if (splits < 1) splits = 1;
    const int ktiles = (K + 31) / 32;
    const int kchunk = ((ktiles + splits - 1) / splits) * 32;
    dim3 grid((X + BX - 1) / BX, (Y + BY - 1) / BY, splits);
    hipLaunchKernelGGL((a0_igemm_kernel<OA, OB, EP, WM, WN, MT, NT>), grid, dim3(256), 0, st, pa, pb, pe, X, Y, K, kchunk);
    return hipGetLastError();
}

#endif  // __HIPCC__
