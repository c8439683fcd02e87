// Operand and epilogue policies of the implicit-GEMM kernel (igemm.h).
//
// Every layer of the Nature CNN (reference agent0/deepq/model.py:90-134) — forward, data gradient and weight
// gradient — is one GEMM  C[X x Y] = sum_k A(x,k) * B(k,y)  whose operands are *gathered* on the fly:
//   forward   x = (b,oh,ow) im2col rows   k = patch element       y = out channel      (model.py:93-101,112-114)
//   dgrad     x = (b,h,w) input pixels    k = (tap, out channel)  y = in channel       gather form, deterministic
//   wgrad     x = out channel             k = (b,oh,ow) rows      y = patch element    reduction split over blocks
// A policy tells the kernel how to fetch four consecutive elements of its operand:
//   A0_KC  "k-contiguous":  fetch(row, k..k+3)   -> transposed store into the k-major LDS tile
//   A0_XC  "x-contiguous":  fetch(k, x..x+3)     -> straight 16-byte store
// All policies are plain C++ so tests/host_emul.cpp can run the same index math on the CPU.
#pragma once
#include "a0_defs.h"

enum { A0_KC = 0, A0_XC = 1 };

// ------------------------------------------------------------------------------------------------ u8 frames, conv1
// Frames are stored as the actor packs them (reference agent.py:78-81): [slot][8][H][W] u8 = st || st_next.
// The uint8 -> fp32 /255 normalisation of agent.py:27 / agent.py:129-134 is fused into this load (true division).
struct a0_frames_src {
    const uint8_t* frames;
    const int* slot;          // optional gather index per batch row (replay sample), nullptr = identity
    a0_geom g;                // C = stacked channels used (4), sample_stride in bytes (8*H*W)
    int chan_off;             // byte offset of the first channel: 0 = obs, 4*H*W = next_obs
    const a0_i4* ktab;        // [K/4]: {byte offset c*H*W + kh*W + kw0, 0, 0, 0}
    int aligned4;             // 1 if every 4-byte group is 4-byte aligned (W % 4 == 0, stride % 4 == 0)
};

A0_HD a0_f4 a0_u8x4_to_f4(const uint8_t* p, int aligned4) {
    a0_f4 v;
    if (aligned4) {
        uint32_t w = *(const uint32_t*)p;
        v.x = (float)(w & 255u) / 255.0f;
        v.y = (float)((w >> 8) & 255u) / 255.0f;
        v.z = (float)((w >> 16) & 255u) / 255.0f;
        v.w = (float)(w >> 24) / 255.0f;
    } else {
        v.x = (float)p[0] / 255.0f; v.y = (float)p[1] / 255.0f; v.z = (float)p[2] / 255.0f; v.w = (float)p[3] / 255.0f;
    }
    return v;
}

A0_HD const uint8_t* a0_frames_row(const a0_frames_src& P, int m) {
    int b = m / P.g.HWout;
    int rem = m - b * P.g.HWout;
    int oh = rem / P.g.Wout;
    int ow = rem - oh * P.g.Wout;
    long long s = P.slot ? (long long)P.slot[b] : (long long)b;
    return P.frames + s * P.g.sample_stride + P.chan_off + (long long)(oh * P.g.stride) * P.g.Win + ow * P.g.stride;
}

struct OpFramesKC {
    static constexpr int MODE = A0_KC;
    typedef a0_frames_src Params;
    struct Row { const uint8_t* p; };
    A0_HD static Row row(const Params& P, int m, int M) {
        Row r; r.p = (m < M) ? a0_frames_row(P, m) : nullptr; return r;
    }
    A0_HD static a0_f4 load(const Params& P, const Row& r, int k, int Kend) {
        if (!r.p || k >= Kend) return a0_zero4();
        return a0_u8x4_to_f4(r.p + P.ktab[k >> 2].x, P.aligned4);
    }
};

struct OpFramesXC {   // wgrad of conv1: reduction index = im2col row m, x = patch element
    static constexpr int MODE = A0_XC;
    typedef a0_frames_src Params;
    A0_HD static a0_f4 load(const Params& P, int m, int x, int Mend, int X) {
        if (m >= Mend || x >= X) return a0_zero4();
        return a0_u8x4_to_f4(a0_frames_row(P, m) + P.ktab[x >> 2].x, P.aligned4);
    }
};

// ------------------------------------------------------------------------------------------------ fp32 NHWC activations
struct a0_act_src {
    const float* x;
    a0_geom g;
    const a0_i4* ktab;        // [K/4]: {element offset (kh*Win + kw)*C + c, kh, kw, 0}
};

struct a0_act_row { const float* p; int h0, w0; };

A0_HD a0_act_row a0_act_rowdesc(const a0_act_src& P, int m) {
    a0_act_row r;
    int b = m / P.g.HWout;
    int rem = m - b * P.g.HWout;
    int oh = rem / P.g.Wout;
    int ow = rem - oh * P.g.Wout;
    r.h0 = oh * P.g.stride - P.g.pad;
    r.w0 = ow * P.g.stride - P.g.pad;
    r.p = P.x + (long long)b * P.g.sample_stride + ((long long)r.h0 * P.g.Win + r.w0) * P.g.C;
    return r;
}

A0_HD a0_f4 a0_act_fetch(const a0_act_src& P, const a0_act_row& r, int k) {
    a0_i4 t = P.ktab[k >> 2];
    int h = r.h0 + t.y, w = r.w0 + t.z;
    if ((unsigned)h >= (unsigned)P.g.Hin || (unsigned)w >= (unsigned)P.g.Win) return a0_zero4();
    return *(const a0_f4*)(r.p + t.x);
}

struct OpActKC {
    static constexpr int MODE = A0_KC;
    typedef a0_act_src Params;
    typedef a0_act_row Row;
    A0_HD static Row row(const Params& P, int m, int M) {
        if (m >= M) { Row r; r.p = nullptr; r.h0 = 0; r.w0 = 0; return r; }
        return a0_act_rowdesc(P, m);
    }
    A0_HD static a0_f4 load(const Params& P, const Row& r, int k, int Kend) {
        if (!r.p || k >= Kend) return a0_zero4();
        return a0_act_fetch(P, r, k);
    }
};

struct OpActXC {
    static constexpr int MODE = A0_XC;
    typedef a0_act_src Params;
    A0_HD static a0_f4 load(const Params& P, int m, int x, int Mend, int X) {
        if (m >= Mend || x >= X) return a0_zero4();
        return a0_act_fetch(P, a0_act_rowdesc(P, m), x);
    }
};

// ------------------------------------------------------------------------------------------------ dense row-major
struct a0_mat_src { const float* x; int ld; };

struct OpMatKC {      // X[row][k], k contiguous
    static constexpr int MODE = A0_KC;
    typedef a0_mat_src Params;
    struct Row { const float* p; };
    A0_HD static Row row(const Params& P, int r, int R) { Row o; o.p = (r < R) ? P.x + (long long)r * P.ld : nullptr; return o; }
    A0_HD static a0_f4 load(const Params&, const Row& r, int k, int Kend) {
        if (!r.p || k >= Kend) return a0_zero4();
        return *(const a0_f4*)(r.p + k);
    }
};

struct OpMatXC {      // X[k][x], x contiguous
    static constexpr int MODE = A0_XC;
    typedef a0_mat_src Params;
    A0_HD static a0_f4 load(const Params& P, int k, int x, int Kend, int X) {
        if (k >= Kend || x >= X) return a0_zero4();
        return *(const a0_f4*)(P.x + (long long)k * P.ld + x);
    }
};

// weights seen by a data-gradient GEMM: B(k = (tap, oc), y = c) = W[oc][kh(tap)][kw(tap)][c]
struct a0_wtab_src { const float* w; const int* wtab; };   // wtab[k] = element offset of W[oc][kh][kw][0]

struct OpWtabXC {
    static constexpr int MODE = A0_XC;
    typedef a0_wtab_src Params;
    A0_HD static a0_f4 load(const Params& P, int k, int x, int Kend, int X) {
        if (k >= Kend || x >= X) return a0_zero4();
        return *(const a0_f4*)(P.w + P.wtab[k] + x);
    }
};

// ------------------------------------------------------------------------------------------------ epilogues
// store(params, x, y, value, z): x = GEMM row, y = GEMM column, z = blockIdx.z (split / slab index)
struct EpiBiasAct {            // y = act(v + bias[y]);  relu keeps NaN like torch.relu
    struct Params { float* out; const float* bias; int ld; int relu; };
    A0_HD static void store(const Params& P, int x, int y, float v, int) {
        v += P.bias[y];
        if (P.relu) v = (v < 0.f) ? 0.f : v;
        P.out[(long long)x * P.ld + y] = v;
    }
};

struct EpiSlab {               // raw partial sums, one slab per z (split-K forward, weight gradients)
    struct Params { float* out; long long slab_stride; int ld; };
    A0_HD static void store(const Params& P, int x, int y, float v, int z) {
        P.out[(long long)z * P.slab_stride + (long long)x * P.ld + y] = v;
    }
};

struct EpiDgrad {              // scatter-free data gradient: row x = (b, h2, w2) of one stride phase
    struct Params {
        float* dx; const float* act;   // act = forward output of the layer below (ReLU mask), same layout as dx
        int HWv, Wv;                   // rows per sample / width of this phase's virtual output
        int Hfull, Wfull, C;           // full input-gradient geometry (NHWC)
        int s, ph, pw;                 // phase: h = h2*s + ph, w = w2*s + pw
        long long sample_stride;
    };
    A0_HD static void store(const Params& P, int x, int y, float v, int) {
        int b = x / P.HWv;
        int rem = x - b * P.HWv;
        int h2 = rem / P.Wv;
        int w2 = rem - h2 * P.Wv;
        int h = h2 * P.s + P.ph, w = w2 * P.s + P.pw;
        if (h >= P.Hfull || w >= P.Wfull) return;
        long long i = (long long)b * P.sample_stride + ((long long)h * P.Wfull + w) * P.C + y;
        P.dx[i] = (P.act[i] > 0.f) ? v : 0.f;
    }
};

struct EpiMaskMat {            // dX[x][y] = act[x][y] > 0 ? v : 0   (dense layers)
    struct Params { float* dx; const float* act; int ld; };
    A0_HD static void store(const Params& P, int x, int y, float v, int) {
        long long i = (long long)x * P.ld + y;
        P.dx[i] = (P.act[i] > 0.f) ? v : 0.f;
    }
};
