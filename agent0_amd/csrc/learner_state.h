// State of the a0_learner handle (learner.hip) — shared with runtime.hip, whose actor handle acts with the learner's online network.
#pragma once
#include "a0_internal.h"

#include <cmath>
#include <vector>

struct Blk { long long off; int N, K; long long w() const { return off; } long long b() const { return off + (long long)N * K; } long long size() const { return (long long)N * K + N; } };

static inline long long ceil_to(long long x, long long m) { return (x + m - 1) / m * m; }

struct a0_learner {
    a0_learner_desc d;
    a0_net* net = nullptr;
    int C = 4, H = 84, W = 84, H1 = 20, W1 = 20, H2 = 9, W2 = 9, feat = 3136, Npad = 32, NQ = 0;
    Blk conv1, conv2, conv3, fc1, head;
    long long n_adam = 0, n_pad = 0, wt_floats = 0;
    float gamma_n = 0.f;
    int ns_fc1 = 1;
    long long slab_off[2] = {0, 0}, enc_slab_off = 0;
    // library-owned HBM
    float *online = nullptr, *target = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr, *scalars = nullptr, *loss_ring = nullptr;
    float *wt_on = nullptr, *wt_tg = nullptr;
    int* state = nullptr;
    float *act1 = nullptr, *act2 = nullptr, *act3_o = nullptr, *act3_t = nullptr, *act3_s = nullptr;
    float *fc1_slabs[3] = {nullptr, nullptr, nullptr};
    float *h = nullptr, *q_o = nullptr, *q_t = nullptr, *draw = nullptr, *dh = nullptr, *d3 = nullptr, *d2 = nullptr, *d1 = nullptr, *loss = nullptr, *slabs = nullptr;
    std::vector<void*> owned;

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
    }
    a0_encoder_weights enc(const float* flat) const { return a0_encoder_weights{flat + conv1.w(), flat + conv1.b(), flat + conv2.w(), flat + conv2.b(), flat + conv3.w(), flat + conv3.b()}; }
};

#define A0_CHECK(call) do { int a0_rc_ = (call); if (a0_rc_ != A0_OK) return a0_rc_; } while (0)
