// State of the a0_learner handle (learner.hip) — shared with runtime.hip, whose actor handle acts with the learner's online network.
#pragma once
#include "a0_internal.h"

#include <cmath>
#include <vector>

struct Blk { long long off; int N, K; long long w() const { return off; } long long b() const { return off + (long long)N * K; } long long size() const { return (long long)N * K + N; } };

static inline long long ceil_to(long long x, long long m) { return (x + m - 1) / m * m; }

// agent0_amd/common/utils.py DeviceRng: per-stream running offsets, every reservation rounded up to a multiple of four draws
struct a0_host_rng {
    unsigned long long seed = 0;
    unsigned long long off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void init(unsigned long long s, unsigned rank) { seed = (s & 0xFFFFFFFFull) | ((unsigned long long)(rank & 0xFFFFu) << 32); }
    unsigned long long reserve(int stream, long long n) { const unsigned long long o = off[stream]; off[stream] += (unsigned long long)((n + 3) / 4 * 4); return o; }
    unsigned next_seed32(int stream) { const unsigned long long o = reserve(stream, 4); return (unsigned)((seed * 0x9E3779B1ull + o * 0x85EBCA77ull + (unsigned long long)stream) & 0xFFFFFFFFull); }
};

struct a0_noise_mod { int block; int r0, r1, in_f; long long off_in, off_w, off_b; };      // block: 0 = fc1, 1 = head; offsets into one network's noise buffer

struct a0_learner {
    a0_learner_desc d;
    a0_net* net = nullptr;
    int C = 4, H = 84, W = 84, H1 = 20, W1 = 20, H2 = 9, W2 = 9, feat = 3136, Npad = 32, NQ = 0;
    Blk conv1, conv2, conv3, fc1, head;      // noisy: fc1 / head are the mu blocks
    // ---- distributional / NoisyNet extension (A0_ALGO_C51)
    int T = 1, Nq = 0, V = 0;
    Blk fc1_sigma, head_sigma;               // NoisyNet: the sigma blocks (layout.py: fc1.mu | fc1.sigma | head.mu | head.sigma)
    Blk eff_fc1, eff_head;                   // offsets into one network's composed-weight scratch
    long long n_eff = 0, noise_len = 0;
    int n_mods = 0;
    a0_noise_mod mods[3];
    a0_host_rng rng;
    float *eff_on = nullptr, *eff_tg = nullptr, *noise = nullptr;       // noise: [online | target], noise_len floats each
    float *atoms = nullptr, *m_proj = nullptr;
    int* a_star = nullptr;
    float *act3_on = nullptr, *fc1_on = nullptr, *fc1_tg = nullptr, *h_on = nullptr, *h_tg = nullptr, *hs_on = nullptr, *hs_tg = nullptr;
    int R_on = 0, ns_on = 1, nh_on = 1, nh_tg = 1;
    // ---- implicit quantile networks (A0_ALGO_IQN)
    Blk cos;                                                 // cosine embedding [feat][64]
    struct QWs { int n_tau = 0; long long R = 0; float *h = nullptr, *raw = nullptr, *q = nullptr, *cosx = nullptr, *emb = nullptr, *x = nullptr, *act3 = nullptr;
                 float *dq = nullptr, *dx = nullptr, *demb = nullptr; } qo, qt, qs;      // online on s (differentiated), target on s', online on s' (double-Q)
    float *t_sel = nullptr, *t_tgt = nullptr, *t_on = nullptr, *y = nullptr, *fwd_scratch = nullptr;
    // ---- fully parameterised quantile function (A0_ALGO_FQF): fraction net block behind the Adam range, its RMSprop state, per-pass fraction buffers
    Blk frac;
    int F = 0;
    struct FWs { float *logits = nullptr, *tau_all = nullptr, *tau_hat = nullptr; } fo, ft, fs;
    QWs qf;                                                   // q at the interior fractions taus[1:-1] (F - 1 per sample) on the online features
    float *inner_taus = nullptr, *rms_sq = nullptr, *frac_loss = nullptr, *dfrac = nullptr, *clip = nullptr;
    // ---- dense heads evaluated layer by layer (A0_ALGO_QR, A0_ALGO_MDQN): fc1 output, raw head output, combined head output per pass; dq of the differentiated pass
    struct DWs { float *act3 = nullptr, *h = nullptr, *raw = nullptr, *q = nullptr; } go, gt, gs, gm;      // online on s, target on s', online on s' (double-Q), target on s (mdqn)
    float *g_dq = nullptr, *qr_taus = nullptr;
    bool qr_fused = false, mdqn_fused = false;      // round 5: qr from the head GEMMs' slabs (a0_qr_head_loss_slabs), mdqn from the fc1 slabs (a0_mdqn_head_loss_slabs)
    float* q_cur = nullptr;                         // mdqn: target(obs) [B][A]
    long long slab_off3[3] = {0, 0, 0};
    // effective (W, b) of a dense layer of the online / target network
    const float* Wf(bool tg) const { return d.noisy ? (tg ? eff_tg : eff_on) + eff_fc1.w() : (tg ? target : online) + fc1.w(); }
    const float* bf(bool tg) const { return d.noisy ? (tg ? eff_tg : eff_on) + eff_fc1.b() : (tg ? target : online) + fc1.b(); }
    const float* Wh(bool tg) const { return d.noisy ? (tg ? eff_tg : eff_on) + eff_head.w() : (tg ? target : online) + head.w(); }
    const float* bh(bool tg) const { return d.noisy ? (tg ? eff_tg : eff_on) + eff_head.b() : (tg ? target : online) + head.b(); }
    long long n_adam = 0, n_pad = 0, wt_floats = 0;
    float gamma_n = 0.f;
    int ns_fc1 = 1;
    long long slab_off[2] = {0, 0}, enc_slab_off = 0;
    int loss_ring_cap = 1024;
    // library-owned HBM
    float *online = nullptr, *target = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr, *scalars = nullptr, *loss_ring = nullptr;
    float *wt_on = nullptr, *wt_tg = nullptr;
    int* state = nullptr;
    float *act1 = nullptr, *act2 = nullptr, *act3_o = nullptr, *act3_t = nullptr, *act3_s = nullptr;
    float *fc1_slabs[3] = {nullptr, nullptr, nullptr};
    float *h = nullptr, *q_o = nullptr, *q_t = nullptr, *draw = nullptr, *dh = nullptr, *d3 = nullptr, *d2 = nullptr, *d1 = nullptr, *loss = nullptr, *slabs = nullptr;
    std::vector<void*> owned;
    // ---- data parallelism: a communicator of a0_dp_init; the dense bucket's all-reduce runs on `dp_side` beside the encoder backward
    long long dp_comm = 0;
    hipStream_t dp_side = nullptr;
    bool dp_one_rank = false;         // the communicator has one rank: the exchange runs on the update's own stream (a0_learner_set_exchange)
    hipEvent_t dp_ev[3] = {nullptr, nullptr, nullptr};

    template <class T> T* alloc(long long n, bool zero = false) {
        void* p = nullptr;
        A0_HIP_THROW(hipMalloc(&p, (size_t)(n > 0 ? n : 1) * sizeof(T)));
        owned.push_back(p);
        if (zero) A0_HIP_THROW(hipMemset(p, 0, (size_t)(n > 0 ? n : 1) * sizeof(T)));
        return (T*)p;
    }
    ~a0_learner() {
        for (void* p : owned) (void)hipFree(p);
        if (net) a0_net_destroy(net);
        for (hipEvent_t e : dp_ev) if (e) (void)hipEventDestroy(e);
        if (dp_side) (void)hipStreamDestroy(dp_side);
    }
    a0_encoder_weights enc(const float* flat) const { return a0_encoder_weights{flat + conv1.w(), flat + conv1.b(), flat + conv2.w(), flat + conv2.b(), flat + conv3.w(), flat + conv3.b()}; }
};

#define A0_CHECK(call) do { int a0_rc_ = (call); if (a0_rc_ != A0_OK) return a0_rc_; } while (0)
