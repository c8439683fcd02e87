// CPU emulation of the implicit-GEMM layer stack (TEST INFRASTRUCTURE, built with g++ by tests/test_host_emul.py).
//
// It instantiates agent0_amd/csrc/net_impl.h — the exact orchestration code the HIP library runs — with a backend
// that evaluates the operand / epilogue policies of operands.h in plain loops (fp32 fmaf chain in k order, which is
// also what v_mfma_f32_32x32x2_f32 computes).  Every gather table, stride, phase decomposition, split heuristic and
// slab layout is therefore checked against the oracle on the CPU; only the MFMA tile mechanics of igemm.h remain
// for the GPU tests.  Nothing here is shipped or used by the product path.
#include <cmath>
#include <cstring>
#include <vector>

#include "../agent0_amd/csrc/net_impl.h"

template <class OP, int MODE = OP::MODE> struct fetcher;
template <class OP> struct fetcher<OP, A0_KC> {   // element (x, k) for k in [kb, ke)
    static void fill(const typename OP::Params& P, int X, int kb, int ke, std::vector<float>& out /* [X][ke-kb] */) {
        const int kc = ke - kb;
        out.assign((size_t)X * kc, 0.f);
        for (int x = 0; x < X; ++x) {
            typename OP::Row r = OP::row(P, x, X);
            for (int k = kb; k < ke; k += 4) {
                bool ok; typename OP::Raw raw = OP::load(P, r, OP::kinfo(P, k, ke), ok); a0_f4 v = OP::finish(raw, ok);
                float* o = &out[(size_t)x * kc + (k - kb)];
                o[0] = v.x; if (k + 1 < ke) o[1] = v.y; if (k + 2 < ke) o[2] = v.z; if (k + 3 < ke) o[3] = v.w;
            }
        }
    }
};
template <class OP> struct fetcher<OP, A0_XC> {
    static void fill(const typename OP::Params& P, int X, int kb, int ke, std::vector<float>& out) {
        const int kc = ke - kb;
        out.assign((size_t)X * kc, 0.f);
        for (int x = 0; x < X; x += 4) {
            typename OP::XInfo xi = OP::xinfo(P, x, X);
            for (int k = kb; k < ke; ++k) {
                bool ok; typename OP::Raw raw = OP::load(P, k, ke, xi, ok); a0_f4 v = OP::finish(raw, ok);
                out[(size_t)x * kc + (k - kb)] = v.x;
                if (x + 1 < X) out[(size_t)(x + 1) * kc + (k - kb)] = v.y;
                if (x + 2 < X) out[(size_t)(x + 2) * kc + (k - kb)] = v.z;
                if (x + 3 < X) out[(size_t)(x + 3) * kc + (k - kb)] = v.w;
            }
        }
    }
};

struct host_backend {
    int tag = 0;
    template <class OA, class OB, class EP, int WM, int WN, int MT, int NT>
    void igemm(const typename OA::Params& pa, const typename OB::Params& pb, const typename EP::Params& pe, int X, int Y, int K, int splits) {
        if (splits < 1) splits = 1;
        const int ktiles = (K + 31) / 32;
        const int kchunk = ((ktiles + splits - 1) / splits) * 32;
        std::vector<float> A, Bm;
        for (int z = 0; z < splits; ++z) {
            const int kb = z * kchunk;
            const int ke = (K < kb + kchunk) ? K : kb + kchunk;
            const int kc = ke > kb ? ke - kb : 0;
            if (kc > 0) { fetcher<OA>::fill(pa, X, kb, ke, A); fetcher<OB>::fill(pb, Y, kb, ke, Bm); }
            for (int x = 0; x < X; ++x)
                for (int y = 0; y < Y; ++y) {
                    float acc = 0.f;
                    for (int k = 0; k < kc; ++k) acc = std::fmaf(A[(size_t)x * kc + k], Bm[(size_t)y * kc + k], acc);
                    EP::store(pe, x, y, acc, z);
                }
            if constexpr (EP::ROWSUM_A)
                for (int x = 0; x < X; ++x) {
                    float t = 0.f;
                    for (int k = 0; k < kc; ++k) t += A[(size_t)x * kc + k];
                    EP::store_rowsum(pe, x, t, z);
                }
        }
    }
    void short_k_fwd(const float* X, int ldx, const float* W, const float* b, const float* M, int group, float* Y, float* Y2, int R, int N, int relu) {
        for (int r = 0; r < R; ++r)
            for (int c = 0; c < N; ++c) {
                float v = 0.f;
                for (int k = 0; k < 64; ++k) v = std::fmaf(X[(size_t)r * ldx + k], W[(size_t)c * 64 + k], v);
                v += b[c];
                if (relu) v = (v < 0.f) ? 0.f : v;
                const size_t o = (size_t)r * N + c;
                if (!M) Y[o] = v;
                else {
                    const float m = M[(size_t)(r / group) * N + c];
                    if (Y2) { Y[o] = v; Y2[o] = v * m; } else Y[o] = v * m;
                }
            }
    }
    int conv1_wgrad_fused(const a0_net_core&, const a0_frames_arg&, int, const float*, float*) { return 0; }   // GPU-only fast path
    bool conv23_wgrad_fused(const a0_net_core&, int, const float*, const float*, const float*, const float*, float*, float*) { return false; }   // GPU-only fast path
    void reduce_slabs(const float* slabs, long long slab_stride, int nslab, float* out, long long count) {
        for (long long i = 0; i < count; ++i) {
            float s = 0.f;
            for (int z = 0; z < nslab; ++z) s += slabs[(long long)z * slab_stride + i];
            out[i] = s;
        }
    }
    void reduce_segments(const a0_reduce_seg* segs, int nseg) {
        for (int k = 0; k < nseg; ++k) reduce_slabs(segs[k].slabs, segs[k].slab_stride, segs[k].nslab, segs[k].out, segs[k].count);
    }
    void reduce_bias_act(const float* slabs, long long slab_stride, int nslab, const float* bias, float* out, int rows, int N, int relu) {
        for (long long i = 0; i < (long long)rows * N; ++i) {
            float s = 0.f;
            for (int z = 0; z < nslab; ++z) s += slabs[(long long)z * slab_stride + i];
            s += bias[i % N];
            if (relu) s = (s < 0.f) ? 0.f : s;
            out[i] = s;
        }
    }
};

struct host_net {
    a0_net_core core;
    a0_net_tables t;
};

extern "C" {

host_net* emul_net_create(int C, int H, int W) {
    host_net* n = new host_net();
    if (!a0_net_core_init(n->core, C, H, W)) { delete n; return nullptr; }
    a0_net_build_tables(n->core, n->t);
    n->core.ktab1 = n->t.ktab1.data(); n->core.ktab2 = n->t.ktab2.data(); n->core.ktab3 = n->t.ktab3.data();
    n->core.ktab_d3 = n->t.ktab_d3.data(); n->core.ktab_d2 = n->t.ktab_d2.data();
    n->core.wtab_d3 = n->t.wtab_d3.data();
    for (int i = 0; i < 4; ++i) n->core.wtab_d2[i] = n->t.wtab_d2[i].data();
    return n;
}
void emul_net_destroy(host_net* n) { delete n; }
void emul_net_geometry(const host_net* n, int* o) {
    const a0_net_core& c = n->core;
    o[0] = c.H1; o[1] = c.W1; o[2] = c.H2; o[3] = c.W2; o[4] = c.H3; o[5] = c.W3; o[6] = c.feat; o[7] = c.K1;
}
void emul_encoder_fwd(const host_net* n, const a0_encoder_weights* w, const a0_frames_arg* f, int B, float* a1, float* a2, float* a3) {
    host_backend bk;
    a0_encoder_fwd_impl(bk, n->core, *w, *f, B, a1, a2, a3);
}
long long emul_encoder_bwd_scratch(const host_net* n, int B) { return a0_encoder_bwd_scratch_impl(n->core, B); }
void emul_encoder_bwd(const host_net* n, const a0_encoder_weights* w, const a0_frames_arg* f, int B, const float* a1, const float* a2,
                      const float* d3, float* d2, float* d1, float* g1, float* g2, float* g3, float* slabs) {
    host_backend bk;
    a0_encoder_bwd_impl(bk, n->core, *w, *f, B, a1, a2, d3, d2, d1, g1, g2, g3, slabs);
}
int emul_short_k_shape(int R, int N, int K, int ldx) { return a0_short_k_shape(R, N, K, ldx) ? 1 : 0; }
long long emul_dense_fwd_scratch(int R, int N, int K) { return a0_dense_fwd_scratch_impl(R, N, K); }
void emul_dense_fwd(const float* X, int ldx, const float* W, const float* b, float* Y, int R, int N, int K, int relu, float* scratch) {
    host_backend bk;
    a0_dense_fwd_impl(bk, X, ldx, W, b, Y, R, N, K, relu, scratch);
}
void emul_dense_dgrad(const float* dY, const float* W, const float* mask, float* dX, int R, int N, int K) {
    host_backend bk;
    a0_dense_dgrad_impl(bk, dY, W, mask, dX, R, N, K);
}
long long emul_dense_wgrad_scratch(int R, int N, int K) { return a0_dense_wgrad_scratch_impl(R, N, K); }
void emul_dense_wgrad(const float* dY, const float* X, int ldx, float* grad, int R, int N, int K, float* slabs) {
    host_backend bk;
    a0_dense_wgrad_impl(bk, dY, X, ldx, grad, R, N, K, slabs);
}

}  // extern "C"

// a0_udiv (a0_defs.h): number of m in [lo, hi) with step `step` where the multiply-high quotient differs from m / d
extern "C" long long emul_udiv_mismatches(unsigned d, unsigned lo, unsigned hi, unsigned step) {
    const unsigned magic = a0_udiv_magic(d);
    long long bad = 0;
    for (unsigned long long m = lo; m < hi; m += step)
        if (a0_udiv((int)(unsigned)m, (int)d, magic) != (int)((unsigned)m / d)) ++bad;
    return bad;
}
