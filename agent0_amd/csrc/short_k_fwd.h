// Dense forward layers with a SHORT reduction (K = 64): Y = act(X W^T + b), optionally times M[row / group] in the same pass.
//
// The quantile networks' cosine embedding (reference model.py:219-247: relu(Linear(64 -> 3136)(cos(pi i tau))) * features) is a GEMM of
// 8 192 (actor) to 32 768 (learner) rows with only 64 k: 3.3 - 13 GFLOP against 103 - 411 MB of output.  As a tile of the general
// kernel (igemm_x9.h: 128 x 128 outputs per workgroup, one workgroup per CU, two k tiles) it was all prologue and epilogue: 81 / 286 us
// with the product in the epilogue, 106 / 410 us as GEMM + Hadamard pass when the embedding is kept (1.4 - 1.7 TB/s of stores).
// Here the roles are turned round: a WAVE keeps the three bf16 term planes of a 64-column strip of W in registers as MFMA fragments
// (96 VGPRs) and streams blocks of 32 rows through them; X goes from global memory straight into fragment registers (k contiguous: the
// eight k of a lane are two 16-byte loads), is split exactly as in igemm_x9.h, and each block is 72 v_mfma_f32_32x32x16_bf16 + its
// epilogue.  No LDS, no barriers: waves are independent, two per SIMD, one wave's loads, splits and stores beside the other's MFMAs.
// Measured (tools/ubench_short_k.py): 38 / 144 us with the product, 49 / 187 us with both outputs.  The floor is the output stream:
// with the loads and eight of nine MFMAs taken out the launch still takes 22 / 85 us = 4.8 TB/s of stores (tools/exp_sk).
//
// Work is cut into units (strip, row block); XCD x (workgroup id mod 8) takes the x-th eighth of the rows, and within it the units go
// strip-major to the waves as equal contiguous ranges, so a wave reloads its strip of W at most once or twice per launch.
#pragma once
#include "igemm_x9.h"

#if defined(__HIPCC__)

struct a0_short_k_args {
    const float* X; const float* W; const float* bias; const float* M;
    float* Y; float* Y2;
    int ldx, R, N, relu, group; unsigned group_magic;
    int rblocks;           // ceil(R / 32)
    int strips;            // ceil(N / 64)
    int units;             // strips * rblocks
};

// eight consecutive k of one row: two float4 -> the three bf16 term fragments of one k-step
A0_D void a0_short_k_split8(const a0_f4& v0, const a0_f4& v1, a0_u32x4g& hi, a0_u32x4g& mid, a0_u32x4g& lo) {
    a0_x9_piece p;
    a0_u32x2g h, m, l;
    p.v = v0;
#pragma unroll
    for (int e = 0; e < 4; ++e) p.split(e);
    p.pack(h, m, l);
    hi.x = h.x; hi.y = h.y; mid.x = m.x; mid.y = m.y; lo.x = l.x; lo.y = l.y;
    p.v = v1;
#pragma unroll
    for (int e = 0; e < 4; ++e) p.split(e);
    p.pack(h, m, l);
    hi.z = h.x; hi.w = h.y; mid.z = m.x; mid.w = m.y; lo.z = l.x; lo.w = l.y;
}

// MODE 0: Y = act(.)   1: Y = act(.) * M   2: Y = act(.), Y2 = act(.) * M (the differentiated pass keeps the embedding for its backward)
// GUARD = false: R % 32 == 0, N % 64 == 0 and groups of at least 31 rows (a 32-row block then touches at most two groups: both M rows are fetched ahead and
// selected per row), checked by the launcher; the loop then holds nothing but
// straight-line loads and stores.  That matters because loads and stores return through ONE in-order counter (vmcnt): a unit's rows can
// only be waited for together with every store issued before them.  With guarded stores anywhere in the loop the compiler cannot count the
// stores that follow the loads and waits for ALL of them to be acknowledged before every unit (46 instead of 38 us at 8 192 rows); the same
// happens for any load still pending at the loop's entry (the strip's bias) and for loads issued after the previous unit's stores (the M
// row of a unit is therefore requested one unit ahead, with the rows).
template <int MODE, bool GUARD, int NPR = 9>
__global__ __launch_bounds__(256, 2) void a0_short_k_fwd_kernel(a0_short_k_args P) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware split.  Hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own 4 MB L2.  XCD x takes the
    // x-th EIGHTH of the rows (for every strip): its slice of X (1 MB of the learner's 8 MB) then stays in its L2 beside the 800 KB of W
    // while 13 - 51 MB of output stream through it.  (With every wave walking all of X the L2 missed on half of its requests, X came from HBM
    // 16 times per launch and the waves spent 64 % of their cycles waiting for their rows.)
    const int xcd = blockIdx.x & 7, nxcd = gridDim.x >= 8 ? 8 : 1;
    const int rb_lo = (int)((long long)P.rblocks * (nxcd == 8 ? xcd : 0) / nxcd), rb_hi = (int)((long long)P.rblocks * ((nxcd == 8 ? xcd : 0) + 1) / nxcd);
    const int nrb = rb_hi - rb_lo;                               // row blocks of this XCD's slice
    const unsigned long long gw = (unsigned long long)(nxcd == 8 ? blockIdx.x >> 3 : blockIdx.x) * 4 + wave, nw = (unsigned long long)(gridDim.x / nxcd) * 4;
    const unsigned long long lunits = (unsigned long long)P.strips * nrb;     // units of the slice, strip-major
    const int u0 = (int)(lunits * gw / nw), u1 = (int)(lunits * (gw + 1) / nw);
    if (gw >= nw || u0 >= u1) return;
    const int half = lane >> 5, l31 = lane & 31;
    const int last_group = MODE != 0 ? a0_udiv(P.R - 1, P.group, P.group_magic) : 0;
    const float floor_v = P.relu ? 0.f : -__builtin_inff();      // (v < floor) ? floor : v keeps NaN like torch.relu

    a0_u32x4g bf[2][4][3];
    float bias_v[2];
    int col[2];
    int strip = u0 / nrb, rb = rb_lo + (u0 - strip * nrb);         // wave-uniform

    auto xrow = [&](int rblock) -> const float* {
        if (rblock >= rb_hi) rblock = rb_lo;                      // requests past the slice wrap to its first rows (the next strip's first unit)
        int r = rblock * 32 + l31;
        r = r < P.R ? r : P.R - 1;
        return P.X + (long long)r * P.ldx + 8 * half;
    };
    // the M rows of a block of 32 rows: with groups of at least 31 rows (checked by the launcher) a block touches at most TWO groups, g0 = group of its
    // first row and g0 + 1; both rows are fetched with the block's X rows, one unit ahead, for the same reason (GUARD: per-row lookups in the epilogue instead)
    auto request_m = [&](float (&mv)[2][2], int rblock) {
        if (MODE != 0 && !GUARD) {
            if (rblock >= rb_hi) rblock = rb_lo;
            const int g0 = a0_udiv(rblock * 32, P.group, P.group_magic), g1 = g0 < last_group ? g0 + 1 : g0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = col[j] < P.N ? col[j] : P.N - 1;
                mv[0][j] = P.M[(long long)g0 * P.N + c];
                mv[1][j] = P.M[(long long)g1 * P.N + c];
            }
        }
    };
    auto request = [&](a0_f4 (&raw)[4][2], int rblock) {
        const float* p = xrow(rblock);
#pragma unroll
        for (int s = 0; s < 4; ++s) { raw[s][0] = *(const a0_f4*)(p + 16 * s); raw[s][1] = *(const a0_f4*)(p + 16 * s + 4); }
    };
    // one unit: rows of block rblock (in raw) times the resident strip; raw is refilled with the rows of block `refill` as its k-steps are consumed
    auto unit = [&](a0_f4 (&raw)[4][2], float (&mv)[2][2], int rblock, int refill) {
        const float mval[2][2] = {{mv[0][0], mv[0][1]}, {mv[1][0], mv[1][1]}};
        // first in-block row of the block's second group (32 or more: the whole block lies in one group)
        const int bnd = MODE != 0 ? (a0_udiv(rblock * 32, P.group, P.group_magic) + 1) * P.group - rblock * 32 : 32;
        const float* pn = xrow(refill);
        a0_acc16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // all four k-steps' fragments first, then the next rows in ONE burst of eight loads: a lane's 16-byte pieces use a quarter of each
        // 128-byte line they touch, and only loads issued together find the line still in the 32 KB L1 (spread over the unit they re-fetched
        // every line from L2 four times: 30 us of a 35 us launch went to that)
        a0_u32x4g a[4][3];
#pragma unroll
        for (int s = 0; s < 4; ++s) a0_short_k_split8(raw[s][0], raw[s][1], a[s][0], a[s][1], a[s][2]);
#pragma unroll
        for (int s = 0; s < 4; ++s) { raw[s][0] = *(const a0_f4*)(pn + 16 * s); raw[s][1] = *(const a0_f4*)(pn + 16 * s + 4); }
        request_m(mv, refill);
        constexpr int TA[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};       // term pairs in the order of increasing magnitude (lo*lo first, hi*hi last), as in igemm_x9.h
        constexpr int TB[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 9 - NPR; q < 9; ++q)      // six products (a0_x9_products): lo*lo, lo*mid, mid*lo are not formed
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(a0_bf16x8g, a[s][TA[q]]),
                                                                      __builtin_bit_cast(a0_bf16x8g, bf[j][s][TB[q]]), acc[j], 0, 0, 0);
        // C/D layout: column = lane & 31 (W's row), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        if constexpr (!GUARD) {
            // straight-line stores off one wave-uniform 64-bit base + a 32-bit byte offset per lane (the store's scalar-base form: one add per store)
            const long long blk = (long long)rblock * 32 * P.N;
            char* const y = (char*)(P.Y + blk);
            char* const y2 = MODE == 2 ? (char*)(P.Y2 + blk) : nullptr;
            const unsigned nb = (unsigned)P.N * 4u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned oj = (unsigned)(4 * half * P.N + col[j]) * 4u;
                asm volatile("" : "+v"(oj));          // keeps the 32 store addresses from being hoisted out of the loop as 64 registers
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned o = oj + (unsigned)((r & 3) + 8 * (r >> 2)) * nb;
                    float v = acc[j][r] + bias_v[j];
                    v = (v < floor_v) ? floor_v : v;
                    const float m = ((r & 3) + 8 * (r >> 2) + 4 * half < bnd) ? mval[0][j] : mval[1][j];
                    if (MODE == 0) *(float*)(y + o) = v;
                    else if (MODE == 1) *(float*)(y + o) = v * m;
                    else { *(float*)(y + o) = v; *(float*)(y2 + o) = v * m; }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (col[j] >= P.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rblock * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row >= P.R) continue;
                    float v = acc[j][r] + bias_v[j];
                    v = (v < floor_v) ? floor_v : v;
                    const long long o = (long long)row * P.N + col[j];
                    if (MODE == 0) P.Y[o] = v;
                    else {
                        const float m = P.M[(long long)a0_udiv(row, P.group, P.group_magic) * P.N + col[j]];
                        if (MODE == 1) P.Y[o] = v * m;
                        else { P.Y[o] = v; P.Y2[o] = v * m; }
                    }
                }
            }
        }
    };

    a0_f4 raw_a[4][2];
    float m_a[2][2] = {{1.f, 1.f}, {1.f, 1.f}};
    request(raw_a, rb);
    for (int u = u0; u < u1; ++strip, rb = rb_lo) {
        // this strip of W: 64 rows x 64 k -> fragments of the three term planes
        col[0] = strip * 64 + l31; col[1] = col[0] + 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = col[j] < P.N ? col[j] : P.N - 1;
            const float* w = P.W + (long long)c * 64 + 8 * half;
            bias_v[j] = P.bias[c];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a0_short_k_split8(*(const a0_f4*)(w + 16 * s), *(const a0_f4*)(w + 16 * s + 4), bf[j][s][0], bf[j][s][1], bf[j][s][2]);
        }
        // the strip's loads are consumed HERE, once: a load still pending at the loop's entry would be waited for in every trip of the loop
        // (and with it every store issued before that point of the trip)
        // (the M rows are per column: the first units of a strip fetch theirs here)
        request_m(m_a, rb);
        asm volatile("" :: "v"(bias_v[0]), "v"(bias_v[1]), "v"(m_a[0][0]), "v"(m_a[0][1]), "v"(m_a[1][0]), "v"(m_a[1][1]));
        const int rb_end = (u1 - u < rb_hi - rb) ? rb + (u1 - u) : rb_hi;      // this wave's row blocks of the strip: [rb, rb_end)
        for (; rb < rb_end; ++rb, ++u) unit(raw_a, m_a, rb, rb + 1);
    }
}

static inline hipError_t a0_short_k_fwd_launch(hipStream_t st, const float* X, int ldx, const float* W, const float* bias, const float* M, int group,
                                               float* Y, float* Y2, int R, int N, int relu) {
    a0_short_k_args P;
    P.X = X; P.W = W; P.bias = bias; P.M = M; P.Y = Y; P.Y2 = Y2;
    P.ldx = ldx; P.R = R; P.N = N; P.relu = relu; P.group = group > 0 ? group : 1; P.group_magic = a0_udiv_magic((unsigned)P.group);
    P.rblocks = (R + 31) / 32;
    const int strips = (N + 63) / 64;
    P.strips = strips;
    P.units = strips * P.rblocks;
    // two workgroups of four waves per CU: 2048 waves; fewer when there are not two units for each
    int wg = (P.units / 2 + 3) / 4;
    if (wg > 512) wg = 512;
    if (wg >= 8) wg &= ~7;          // the kernel's XCD split wants whole rounds of eight workgroups
    if (wg < 1) wg = 1;
    const int mode = !M ? 0 : (Y2 ? 2 : 1);
    const bool guard = (R & 31) != 0 || (N & 63) != 0 || (mode != 0 && P.group < 31);       // groups of >= 31 rows: a 32-row block touches at most two of them
    const dim3 grid((unsigned)wg), block(256);
    typedef void (*a0_sk_kern)(a0_short_k_args);
    const bool six = a0_x9_products_now() == 6;
    const a0_sk_kern kern = guard ? (mode == 0 ? (six ? a0_short_k_fwd_kernel<0, true, 6> : a0_short_k_fwd_kernel<0, true, 9>)
                                  : mode == 1 ? (six ? a0_short_k_fwd_kernel<1, true, 6> : a0_short_k_fwd_kernel<1, true, 9>)
                                              : (six ? a0_short_k_fwd_kernel<2, true, 6> : a0_short_k_fwd_kernel<2, true, 9>))
                                  : (mode == 0 ? (six ? a0_short_k_fwd_kernel<0, false, 6> : a0_short_k_fwd_kernel<0, false, 9>)
                                  : mode == 1 ? (six ? a0_short_k_fwd_kernel<1, false, 6> : a0_short_k_fwd_kernel<1, false, 9>)
                                              : (six ? a0_short_k_fwd_kernel<2, false, 6> : a0_short_k_fwd_kernel<2, false, 9>));
    hipLaunchKernelGGL(kern, grid, block, 0, st, P);
    return hipGetLastError();
}

#endif  // __HIPCC__
