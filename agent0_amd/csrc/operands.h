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

// Policy interface (all loads are BRANCH-FREE: an invalid element reads a safe in-bounds address and is then zeroed, so
// the compiler can issue every load of a tile back to back behind one wait instead of a wait per conditional load):
//   KC:  Row   row(P, x, X)            per-row state, computed once per workgroup (the row set of a thread never changes)
//        KInfo kinfo(P, k, Kend)       per-k state (gather-table entry), fetched one tile AHEAD of its use
//        Raw   load(P, row, kinfo, ok) raw bits + validity; nothing may CONSUME the bits here, or the wave would wait for the
//        f4    finish(raw, ok)         load before the tile's MFMAs instead of after them: zero-fill / u8->fp32 happen at commit
//   XC:  XInfo xinfo(P, x, X)          per-column state, computed once (a thread's columns never change)
//        Raw   load(P, k, Kend, xinfo, ok);  f4 finish(raw, ok)
//
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

A0_HD uint32_t a0_u8x4_load(const uint8_t* p, int aligned4) {
    if (aligned4) return *(const uint32_t*)p;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// x / 255.0f for an integer 0 <= x <= 255, correctly rounded (== the fp32 division of the reference, checked exhaustively
// in tests/test_oracle_core.py) in three instructions: one Newton correction of x * fl(1/255)
A0_HD float a0_div255(float x) {
    const float r = 1.0f / 255.0f;
    const float q0 = x * r;
#if defined(__HIPCC__)
    const float e = __builtin_fmaf(-q0, 255.0f, x);
    return __builtin_fmaf(e, r, q0);
#else
    const float e = __builtin_fmaf(-q0, 255.0f, x);
    return __builtin_fmaf(e, r, q0);
#endif
}

A0_HD a0_f4 a0_u8x4_to_f4(uint32_t w, bool valid) {
    if (!valid) w = 0;
    a0_f4 v;
    v.x = a0_div255((float)(w & 255u));
    v.y = a0_div255((float)((w >> 8) & 255u));
    v.z = a0_div255((float)((w >> 16) & 255u));
    v.w = a0_div255((float)(w >> 24));
    return v;
}

A0_HD const uint8_t* a0_frames_row(const a0_frames_src& P, int m) {
    int b = a0_udiv(m, P.g.HWout, P.g.hw_magic);
    int rem = m - b * P.g.HWout;
    int oh = a0_udiv(rem, P.g.Wout, P.g.w_magic);
    int ow = rem - oh * P.g.Wout;
    long long s = P.slot ? (long long)P.slot[b] : (long long)b;
    return P.frames + s * P.g.sample_stride + P.chan_off + (long long)(oh * P.g.stride) * P.g.Win + ow * P.g.stride;
}

struct OpFramesKC {
    static constexpr int MODE = A0_KC;
    typedef a0_frames_src Params;
    struct Row { const uint8_t* p; bool ok; };
    struct KInfo { int off; bool ok; };
    A0_HD static Row row(const Params& P, int m, int M) {
        Row r; r.ok = m < M; r.p = a0_frames_row(P, r.ok ? m : 0); return r;
    }
    A0_HD static KInfo kinfo(const Params& P, int k, int Kend) {
        KInfo ki; ki.ok = k < Kend; ki.off = P.ktab[ki.ok ? (k >> 2) : 0].x; return ki;
    }
    typedef uint32_t Raw;
    A0_HD static Raw load(const Params& P, const Row& r, const KInfo& ki, bool& ok) {
        ok = r.ok && ki.ok;
        return a0_u8x4_load(r.p + ki.off, P.aligned4);
    }
    A0_HD static a0_f4 finish(Raw w, bool ok) { return a0_u8x4_to_f4(w, ok); }
};

struct OpFramesXC {   // wgrad of conv1: reduction index = im2col row m, x = patch element
    static constexpr int MODE = A0_XC;
    typedef a0_frames_src Params;
    struct XInfo { int off; bool ok; };
    A0_HD static XInfo xinfo(const Params& P, int x, int X) {
        XInfo xi; xi.ok = x < X; xi.off = P.ktab[xi.ok ? (x >> 2) : 0].x; return xi;
    }
    typedef uint32_t Raw;
    A0_HD static Raw load(const Params& P, int m, int Mend, const XInfo& xi, bool& ok) {
        ok = xi.ok && m < Mend;
        return a0_u8x4_load(a0_frames_row(P, ok ? m : 0) + xi.off, P.aligned4);
    }
    A0_HD static a0_f4 finish(Raw w, bool ok) { return a0_u8x4_to_f4(w, ok); }
};

// ------------------------------------------------------------------------------------------------ fp32 NHWC activations
struct a0_act_src {
    const float* x;
    a0_geom g;
    const a0_i4* ktab;        // [K/4]: {element offset (kh*Win + kw)*C + c, kh, kw, 0}
};

struct a0_act_row { long long base; int h0, w0; bool ok; };

A0_HD a0_act_row a0_act_rowdesc(const a0_act_src& P, int m, bool ok) {
    a0_act_row r;
    if (!ok) m = 0;
    int b = a0_udiv(m, P.g.HWout, P.g.hw_magic);         // the weight-gradient GEMMs evaluate this per fetched k row: no integer division
    int rem = m - b * P.g.HWout;
    int oh = a0_udiv(rem, P.g.Wout, P.g.w_magic);
    int ow = rem - oh * P.g.Wout;
    r.h0 = oh * P.g.stride - P.g.pad;
    r.w0 = ow * P.g.stride - P.g.pad;
    r.base = (long long)b * P.g.sample_stride + ((long long)r.h0 * P.g.Win + r.w0) * P.g.C;
    r.ok = ok;
    return r;
}

A0_HD a0_f4 a0_act_fetch(const a0_act_src& P, const a0_act_row& r, const a0_i4& t, bool kok, bool& ok) {
    const int h = r.h0 + t.y, w = r.w0 + t.z;
    ok = r.ok && kok && (unsigned)h < (unsigned)P.g.Hin && (unsigned)w < (unsigned)P.g.Win;
    return *(const a0_f4*)(P.x + (ok ? r.base + t.x : 0));
}

A0_HD a0_f4 a0_f4_select(a0_f4 v, bool ok) { if (!ok) v = a0_zero4(); return v; }

struct OpActKC {
    static constexpr int MODE = A0_KC;
    typedef a0_act_src Params;
    typedef a0_act_row Row;
    struct KInfo { a0_i4 t; bool ok; };
    A0_HD static Row row(const Params& P, int m, int M) { return a0_act_rowdesc(P, m, m < M); }
    A0_HD static KInfo kinfo(const Params& P, int k, int Kend) {
        KInfo ki; ki.ok = k < Kend; ki.t = P.ktab[ki.ok ? (k >> 2) : 0]; return ki;
    }
    typedef a0_f4 Raw;
    A0_HD static Raw load(const Params& P, const Row& r, const KInfo& ki, bool& ok) { return a0_act_fetch(P, r, ki.t, ki.ok, ok); }
    A0_HD static a0_f4 finish(Raw v, bool ok) { return a0_f4_select(v, ok); }
};

struct OpActXC {
    static constexpr int MODE = A0_XC;
    typedef a0_act_src Params;
    struct XInfo { a0_i4 t; bool ok; };
    A0_HD static XInfo xinfo(const Params& P, int x, int X) {
        XInfo xi; xi.ok = x < X; xi.t = P.ktab[xi.ok ? (x >> 2) : 0]; return xi;
    }
    typedef a0_f4 Raw;
    A0_HD static Raw load(const Params& P, int m, int Mend, const XInfo& xi, bool& ok) {
        return a0_act_fetch(P, a0_act_rowdesc(P, m, m < Mend), xi.t, xi.ok, ok);
    }
    A0_HD static a0_f4 finish(Raw v, bool ok) { return a0_f4_select(v, ok); }
};

// ------------------------------------------------------------------------------------------------ dense row-major
struct a0_mat_src { const float* x; int ld; };

struct OpMatKC {      // X[row][k], k contiguous
    static constexpr int MODE = A0_KC;
    typedef a0_mat_src Params;
    struct Row { const float* p; bool ok; };
    struct KInfo { int k; bool ok; };
    A0_HD static Row row(const Params& P, int r, int R) { Row o; o.ok = r < R; o.p = P.x + (o.ok ? (long long)r * P.ld : 0); return o; }
    A0_HD static KInfo kinfo(const Params&, int k, int Kend) { KInfo ki; ki.ok = k < Kend; ki.k = ki.ok ? k : 0; return ki; }
    typedef a0_f4 Raw;
    A0_HD static Raw load(const Params&, const Row& r, const KInfo& ki, bool& ok) {
        ok = r.ok && ki.ok;
        return *(const a0_f4*)(r.p + ki.k);
    }
    A0_HD static a0_f4 finish(Raw v, bool ok) { return a0_f4_select(v, ok); }
};

struct OpMatXC {      // X[k][x], x contiguous
    static constexpr int MODE = A0_XC;
    typedef a0_mat_src Params;
    struct XInfo { int x; bool ok; };
    A0_HD static XInfo xinfo(const Params&, int x, int X) { XInfo xi; xi.ok = x < X; xi.x = xi.ok ? x : 0; return xi; }
    typedef a0_f4 Raw;
    A0_HD static Raw load(const Params& P, int k, int Kend, const XInfo& xi, bool& ok) {
        ok = xi.ok && k < Kend;
        return *(const a0_f4*)(P.x + (ok ? (long long)k * P.ld : 0) + xi.x);
    }
    A0_HD static a0_f4 finish(Raw v, bool ok) { return a0_f4_select(v, ok); }
};

// weights seen by a data-gradient GEMM: B(k = (tap, oc), y = c) = W[oc][kh(tap)][kw(tap)][c]
struct a0_wtab_src { const float* w; const int* wtab; };   // wtab[k] = element offset of W[oc][kh][kw][0]

struct OpWtabXC {
    static constexpr int MODE = A0_XC;
    typedef a0_wtab_src Params;
    struct XInfo { int x; bool ok; };
    A0_HD static XInfo xinfo(const Params&, int x, int X) { XInfo xi; xi.ok = x < X; xi.x = xi.ok ? x : 0; return xi; }
    typedef a0_f4 Raw;
    A0_HD static Raw load(const Params& P, int k, int Kend, const XInfo& xi, bool& ok) {
        ok = xi.ok && k < Kend;
        return *(const a0_f4*)(P.w + P.wtab[ok ? k : 0] + xi.x);
    }
    A0_HD static a0_f4 finish(Raw v, bool ok) { return a0_f4_select(v, ok); }
};

// ------------------------------------------------------------------------------------------------ epilogues
// store(params, x, y, value, z): x = GEMM row, y = GEMM column, z = blockIdx.z (split / slab index)
struct EpiBiasAct {            // y = act(v + bias[y]);  relu keeps NaN like torch.relu
    static constexpr bool ROWSUM_A = false;
    struct Params { float* out; const float* bias; int ld; int relu; };
    A0_HD static void store(const Params& P, int x, int y, float v, int) {
        v += P.bias[y];
        if (P.relu) v = (v < 0.f) ? 0.f : v;
        P.out[(long long)x * P.ld + y] = v;
    }
};

struct EpiBiasActMul {         // y = act(v + bias[y]) * mul[(x / group)][y]: the IQN / FQF embedding relu(cos W + b) times the state features (model.py:244-247)
    static constexpr bool ROWSUM_A = false;
    struct Params { float* out; const float* bias; int ld; int relu; const float* mul; int group; unsigned group_magic; };
    A0_HD static void store(const Params& P, int x, int y, float v, int) {
        v += P.bias[y];
        if (P.relu) v = (v < 0.f) ? 0.f : v;
        const int g = a0_udiv(x, P.group, P.group_magic);
        P.out[(long long)x * P.ld + y] = v * P.mul[(long long)g * P.ld + y];
    }
};

struct EpiSlab {               // raw partial sums, one slab per z (split-K forward, weight gradients)
    static constexpr bool ROWSUM_A = false;
    struct Params { float* out; long long slab_stride; int ld; };
    A0_HD static void store(const Params& P, int x, int y, float v, int z) {
        P.out[(long long)z * P.slab_stride + (long long)x * P.ld + y] = v;
    }
};

struct EpiDgrad {              // scatter-free data gradient: row x = (b, h2, w2) of one stride phase
    static constexpr bool ROWSUM_A = false;
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

// The quantile networks' dx = dh W_fc1 [B * n][feat] consumed where it is produced (round 6; == a0_dense_dgrad + a0_hadamard_bwd, model.py:244-247 backwards):
//   demb[x][y] = emb[x][y] > 0 ? dx * feat[x / n][y] : 0        (gradient w.r.t. the cosine layer's pre-activation)
//   d3[b][y]   = feat[b][y] > 0 ? sum over the sample's n rows of dx * emb : 0
// dx never goes to HBM (411 MB written and read back at B * n = 32 768).  Not a per-element store: the GEMM body reduces over the n rows of a sample inside the wave
// that holds them (igemm_x9.h, a0_is_hadamard).
struct EpiHadamard {
    static constexpr bool ROWSUM_A = false;
    struct Params { const float* emb; const float* feat; float* demb; float* d3; int ld; int n; };
    A0_HD static void store(const Params&, int, int, float, int) {}
};

struct EpiMaskMat {            // dX[x][y] = act[x][y] > 0 ? v : 0   (dense layers)
    static constexpr bool ROWSUM_A = false;
    struct Params { float* dx; const float* act; int ld; };
    A0_HD static void store(const Params& P, int x, int y, float v, int) {
        long long i = (long long)x * P.ld + y;
        P.dx[i] = (P.act[i] > 0.f) ? v : 0.f;
    }
};
