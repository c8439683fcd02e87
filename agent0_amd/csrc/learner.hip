// a0_learner: one DQN-family learner — online + target parameters, gradients, Adam moments and every workspace — as ONE opaque handle whose HBM the
// library owns, and BaseLearner.train (reference agent0/deepq/agent.py:124-169 with DQNLearner.train_step 173-190 behind it) as ONE C call per update.
//
// Everything a0_learner_update does is a sequence of the entry points declared above it in include/agent0_hip.h, in exactly the order
// agent0_amd/deepq/engine.py issues them for the same configuration (so the two are bit-identical: tests/test_gpu_engine.py::test_native_learner_...);
// what this file adds is the part a non-Python host would otherwise have to re-implement: the packed parameter layout (deepq/layout.py), buffer
// sizes, split-K slab bookkeeping and the call order.  SURVEY.md §8(b) "ownership": opaque handle, library-owned HBM, caller passes borrowed device
// pointers valid for the call, no allocation and no host synchronisation after a0_learner_create.
//
// Scope: on 4 x 84 x 84 observations, the scalar-head learners (dqn; dueling; double-Q; n-step through discount^n) — BASELINE configs[1], the bench line — and
// the categorical one (c51, also with NoisyLinear layers: BASELINE configs[2], rainbow-lite), the implicit quantile network (configs[3]) and the fully
// parameterised quantile function (configs[4]), plus qr and mdqn: the reference's six learners.
#include "learner_state.h"

extern "C" int a0_learner_create(const a0_learner_desc* d, a0_learner** out) { return a0_learner_create_on(d, nullptr, out); }

extern "C" int a0_learner_set_rng(a0_learner* L, int stream_id, unsigned long long offset) {
    if (!L || stream_id < 0 || stream_id > 7 || (offset & 3)) return a0_fail(A0_EINVAL, "a0_learner_set_rng: stream 0..7, offset a multiple of four");
    L->rng.off[stream_id] = offset;
    return A0_OK;
}

extern "C" int a0_learner_create_on(const a0_learner_desc* d, const a0_learner_buffers* bufs, a0_learner** out) {
    A0_TRY
    if (!d || !out) return a0_fail(A0_EINVAL, "a0_learner_create: null argument");
    const a0_learner_buffers none{};
    const a0_learner_buffers& U = bufs ? *bufs : none;
    if (U.loss_ring && U.loss_ring_cap < 1) return a0_fail(A0_EINVAL, "a0_learner_create_on: loss_ring_cap");
    if (d->A < 1 || d->B < 1 || d->n_step < 1 || !(d->discount > 0.0) || !(d->lr >= 0.0) || d->target_update_freq < 1 ||
        (d->algo != A0_ALGO_DQN && d->algo != A0_ALGO_C51 && d->algo != A0_ALGO_IQN && d->algo != A0_ALGO_FQF && d->algo != A0_ALGO_QR && d->algo != A0_ALGO_MDQN))
        return a0_fail(A0_EINVAL, "a0_learner_create: bad description");
    if (d->algo == A0_ALGO_QR && (d->num_atoms < 1 || d->num_atoms > 1024)) return a0_fail(A0_EINVAL, "a0_learner_create: qr needs 1 <= num_atoms <= 1024");
    if (d->algo == A0_ALGO_MDQN && !(d->mdqn_tau > 0.0)) return a0_fail(A0_EINVAL, "a0_learner_create: mdqn needs tau > 0");
    if (d->algo == A0_ALGO_FQF && (d->fqf_F < 2 || d->fqf_F > 32 || d->A + (d->dueling ? 1 : 0) > 32))
        return a0_fail(A0_EINVAL, "a0_learner_create: fqf needs 2 <= F <= 32, A + dueling <= 32");
    if (d->algo == A0_ALGO_IQN && (d->iqn_K < 1 || d->iqn_N < 1 || d->iqn_N_dash < 1 || d->A + (d->dueling ? 1 : 0) > 32))
        return a0_fail(A0_EINVAL, "a0_learner_create: iqn needs K, N, N' >= 1, A + dueling <= 32");
    if (d->algo == A0_ALGO_DQN && d->A + (d->dueling ? 1 : 0) > 24)
        return a0_fail(A0_EINVAL, "a0_learner_create: the dqn handle covers scalar heads with A + dueling <= 24 actions");
    if (d->algo == A0_ALGO_C51 && (d->num_atoms < 2 || d->num_atoms > 64 || !(d->vmax > d->vmin)))
        return a0_fail(A0_EINVAL, "a0_learner_create: c51 needs 2 <= num_atoms <= 64 and vmin < vmax");
    a0_learner* L = new a0_learner();
    try {
        L->d = *d;
        a0_net_desc nd{4, 84, 84};
        if (a0_net_create(&nd, &L->net) != A0_OK) { delete L; return A0_EINVAL; }
        const bool c51 = d->algo == A0_ALGO_C51;
        L->T = (c51 || d->algo == A0_ALGO_QR) ? d->num_atoms : 1;
        L->Nq = d->A * L->T;
        L->V = d->dueling ? L->T : 0;
        L->NQ = d->A + (d->dueling ? 1 : 0);
        L->Npad = (int)ceil_to(L->Nq + L->V, 32);
        // round 5: qr runs c51's layer structure with a0_qr_head_loss_slabs behind it when a sample's staged head outputs fit in LDS (engine.py::_qr_fused_ok), mdqn
        // runs dqn's with the Munchausen target (a0_mdqn_head_loss_slabs) when the scalar-head kernel covers the action set
        {
            const int Ron = d->double_q ? 2 * d->B : d->B;
            L->qr_fused = d->algo == A0_ALGO_QR && d->A <= 32 && (3LL * L->Npad + ceil_to(L->T, 4)) * 4 <= 150 * 1024 &&
                          std::max(a0_dense_fwd_partial_slabs(Ron, L->Npad, 512), a0_dense_fwd_partial_slabs(d->B, L->Npad, 512)) <= 8;
            L->mdqn_fused = d->algo == A0_ALGO_MDQN && L->NQ <= 24;
        }
        const bool dist = c51 || L->qr_fused;                                            // distributional heads from the head GEMMs' slabs (engine.py::_dist_heads_to_slabs)
        const bool generic = (d->algo == A0_ALGO_QR && !L->qr_fused) || (d->algo == A0_ALGO_MDQN && !L->mdqn_fused);          // dense heads evaluated layer by layer (engine.py's unfused path)
        long long off = 0;
        auto add = [&](Blk& b, int N, int K) { b = Blk{off, N, K}; off += b.size(); };
        add(L->conv1, 32, L->C * 64); add(L->conv2, 64, 512); add(L->conv3, 64, 576);                                                              // deepq/layout.py
        if (d->noisy) { add(L->fc1, 512, L->feat); add(L->fc1_sigma, 512, L->feat); add(L->head, L->Npad, 512); add(L->head_sigma, L->Npad, 512); }
        else { add(L->fc1, 512, L->feat); add(L->head, L->Npad, 512); }
        const bool fqf = d->algo == A0_ALGO_FQF;
        const bool iqn = d->algo == A0_ALGO_IQN || fqf;           // the quantile family: cosine-embedding heads over B * n_tau rows
        if (iqn) add(L->cos, L->feat, 64);
        L->n_adam = off;
        if (fqf) { add(L->frac, 32, L->feat); L->F = d->fqf_F; }  // behind the Adam range: its own RMSprop step (agent.py:139-148)
        if (d->noisy) {
            // composed weights [fc1 | head] per network; noise vectors per NoisyLinear module in the reference's module order (first_dense, q_head, value_head), each
            // (noise_in, noise_out_weight, noise_out_bias) padded to four floats — the layout one Philox fill of the whole buffer reproduces (engine.py DeviceNet)
            L->eff_fc1 = Blk{0, 512, L->feat};
            L->eff_head = Blk{L->eff_fc1.size(), L->Npad, 512};
            L->n_eff = L->eff_fc1.size() + L->eff_head.size();
            long long no = 0;
            auto mod = [&](int block, int r0, int r1, int in_f) {
                a0_noise_mod m{block, r0, r1, in_f, 0, 0, 0};
                m.off_in = no; no += ceil_to(in_f, 4);
                m.off_w = no; no += ceil_to(r1 - r0, 4);
                m.off_b = no; no += ceil_to(r1 - r0, 4);
                L->mods[L->n_mods++] = m;
            };
            mod(0, 0, 512, L->feat);
            mod(1, 0, L->Nq, 512);
            if (d->dueling) mod(1, L->Nq, L->Nq + L->V, 512);
            L->noise_len = no;
            L->rng.init(d->seed, 0);
        }
        L->n_pad = ceil_to(off, 4);
        L->wt_floats = a0_net_conv_wt_floats(L->C);
        L->gamma_n = (float)std::pow(d->discount, (double)d->n_step);
        const int B = d->B;
        L->online = U.online ? U.online : L->alloc<float>(L->n_pad, true); L->target = U.target ? U.target : L->alloc<float>(L->n_pad, true);
        L->grads = U.grads ? U.grads : L->alloc<float>(L->n_pad + 4, true);
        L->m = U.adam_m ? U.adam_m : L->alloc<float>(L->n_pad, true); L->v = U.adam_v ? U.adam_v : L->alloc<float>(L->n_pad, true);
        L->state = U.state ? U.state : L->alloc<int>(8, true); L->scalars = U.scalars ? U.scalars : L->alloc<float>(4, true);
        L->loss_ring = U.loss_ring ? U.loss_ring : L->alloc<float>(1024, true); L->loss_ring_cap = U.loss_ring ? U.loss_ring_cap : 1024;
        L->wt_on = U.wt_online ? U.wt_online : L->alloc<float>(L->wt_floats, true); L->wt_tg = U.wt_target ? U.wt_target : L->alloc<float>(L->wt_floats, true);
        L->act1 = L->alloc<float>((long long)B * L->H1 * L->W1 * 32); L->act2 = L->alloc<float>((long long)B * L->H2 * L->W2 * 64);
        L->act3_t = L->alloc<float>((long long)B * L->feat);
        L->ns_fc1 = a0_dense_fwd_partial_slabs(B, 512, L->feat);
        if (!dist) {
            L->act3_o = L->alloc<float>((long long)B * L->feat);
            if (d->double_q || L->mdqn_fused) L->act3_s = L->alloc<float>((long long)B * L->feat);      // mdqn: the target network's features of the CURRENT observation
            for (int i = 0; i < ((d->double_q || L->mdqn_fused) ? 3 : 2); ++i) L->fc1_slabs[i] = L->alloc<float>((long long)L->ns_fc1 * B * 512);
            L->h = L->alloc<float>((long long)B * 512);
            if (L->mdqn_fused) L->q_cur = L->alloc<float>((long long)B * d->A);
        }
        L->q_o = L->alloc<float>((long long)B * d->A * L->T); L->q_t = L->alloc<float>((long long)B * d->A * L->T);
        const long long Rg = fqf ? (long long)B * d->fqf_F : (iqn ? (long long)B * d->iqn_N : (long long)B);           // rows of the differentiated pass
        L->draw = L->alloc<float>(Rg * L->Npad); L->dh = L->alloc<float>(Rg * 512); L->d3 = L->alloc<float>((long long)B * L->feat);
        L->d2 = L->alloc<float>((long long)B * L->H2 * L->W2 * 64); L->d1 = L->alloc<float>((long long)B * L->H1 * L->W1 * 32); L->loss = L->alloc<float>(B);
        // one slab scratch for the dense weight gradients (disjoint regions, one reduction launch) and, after them, the encoder's
        const long long s_head = ceil_to(a0_dense_wgrad_scratch((int)Rg, L->Npad, 512), 4), s_fc1 = ceil_to(a0_dense_wgrad_scratch((int)Rg, 512, L->feat), 4);
        const long long s_cos = iqn ? ceil_to(a0_dense_wgrad_scratch((int)Rg, L->feat, 64), 4) : 0;
        L->slab_off[0] = 0; L->slab_off[1] = s_head;
        L->slab_off3[0] = 0; L->slab_off3[1] = s_head; L->slab_off3[2] = s_head + s_fc1;
        // (the dense layers' slab reductions ride in the encoder's reduction launch, a0_pending_reduce: the encoder's slabs start behind theirs)
        L->enc_slab_off = s_head + s_fc1 + s_cos;
        long long n_slab = L->enc_slab_off + a0_net_encoder_bwd_scratch(L->net, B);
        if (fqf && a0_dense_wgrad_scratch(B, 32, L->feat) > n_slab) n_slab = a0_dense_wgrad_scratch(B, 32, L->feat);
        L->slabs = L->alloc<float>(n_slab > 4 ? n_slab : 4);
        if (d->noisy) {
            L->eff_on = U.eff_online ? U.eff_online : L->alloc<float>(L->n_eff, true); L->eff_tg = U.eff_target ? U.eff_target : L->alloc<float>(L->n_eff, true);
            if (U.noise) L->noise = U.noise;      // the caller's vectors as they are; its stream position comes through a0_learner_set_rng
            else {
            L->noise = L->alloc<float>(2 * L->noise_len, true);
            // BaseLearner.__init__ builds the online and the target network, and a NoisyLinear draws its first noise when it is built (model.py:44-52): the two
            // networks' first draws come off the learner's stream here, so that the updates' draws continue where the Python classes' do
            const unsigned long long o = L->rng.reserve(4, L->noise_len);
            (void)L->rng.reserve(4, L->noise_len);
            if (a0_rng_normal(L->rng.seed, 4, o, 0.1f, L->noise, 2 * L->noise_len, nullptr) != A0_OK) { delete L; return A0_EINVAL; }
            A0_HIP_THROW(hipDeviceSynchronize());
            }
        }
        if (dist) {
            const int dq = d->double_q ? 1 : 0;
            L->R_on = dq ? 2 * B : B;
            L->ns_on = a0_dense_fwd_partial_slabs(L->R_on, 512, L->feat);
            L->nh_on = a0_dense_fwd_partial_slabs(L->R_on, L->Npad, 512);
            L->nh_tg = a0_dense_fwd_partial_slabs(B, L->Npad, 512);
            L->act3_on = L->alloc<float>((long long)L->R_on * L->feat);
            // the online network's features of s and (double-Q) of s' back to back: fc1 and the head run over both as ONE GEMM each (same weights)
            L->act3_o = L->act3_on;
            if (dq) L->act3_s = L->act3_on + (long long)B * L->feat;
            L->fc1_on = L->alloc<float>((long long)L->ns_on * L->R_on * 512); L->fc1_tg = L->alloc<float>((long long)L->ns_fc1 * B * 512);
            L->h_on = L->alloc<float>((long long)L->R_on * 512); L->h_tg = L->alloc<float>((long long)B * 512);
            L->h = L->h_on;                                  // h(s) of the online network: what the backward pass reads
            L->hs_on = L->alloc<float>((long long)L->nh_on * L->R_on * L->Npad); L->hs_tg = L->alloc<float>((long long)L->nh_tg * B * L->Npad);
            L->a_star = L->alloc<int>(B, true);
        }
        if (L->qr_fused) {
            L->qr_taus = L->alloc<float>(L->T);
            std::vector<float> t((size_t)L->T);
            for (int i = 0; i < L->T; ++i) t[(size_t)i] = (2.0f * (float)i + 1.0f) / (2.0f * (float)L->T);          // agent.py:274: the quantile midpoints
            A0_HIP_THROW(hipMemcpy(L->qr_taus, t.data(), (size_t)L->T * 4, hipMemcpyHostToDevice));
        }
        if (c51) {
            L->atoms = L->alloc<float>(L->T); L->m_proj = L->alloc<float>((long long)B * L->T);
            // torch.linspace(vmin, vmax, T) in fp32 as ATen's vectorised CPU kernel computes it (RangeFactories: step = (end - start) / (steps - 1); fma(step, i, start)
            // below the middle, fma(-step, steps - 1 - i, end) above).  A host whose torch build rounds differently hands its own values to a0_learner_set_support.
            std::vector<float> at((size_t)L->T);
            const float lo = (float)d->vmin, hi = (float)d->vmax, step = (hi - lo) / (float)(L->T - 1);
            for (int i = 0; i < L->T; ++i) at[(size_t)i] = i < L->T / 2 ? std::fmaf(step, (float)i, lo) : std::fmaf(-step, (float)(L->T - 1 - i), hi);
            A0_HIP_THROW(hipMemcpy(L->atoms, at.data(), (size_t)L->T * 4, hipMemcpyHostToDevice));
        }
        if (generic) {
            const long long nq = (long long)B * d->A * L->T;
            auto gw = [&](a0_learner::DWs& w, float* act3, float* h, float* q) { w.act3 = act3; w.h = h ? h : L->alloc<float>((long long)B * 512); w.raw = L->alloc<float>((long long)B * L->Npad); w.q = q ? q : L->alloc<float>(nq); };
            gw(L->go, L->act3_o, L->h, L->q_o);
            gw(L->gt, L->act3_t, nullptr, L->q_t);
            if (d->double_q && d->algo == A0_ALGO_QR) gw(L->gs, L->act3_s ? L->act3_s : (L->act3_s = L->alloc<float>((long long)B * L->feat)), nullptr, nullptr);
            if (d->algo == A0_ALGO_MDQN) gw(L->gm, L->alloc<float>((long long)B * L->feat), nullptr, nullptr);
            L->g_dq = L->alloc<float>(nq, true);
            L->a_star = L->alloc<int>(B, true);
            const long long s1 = a0_dense_fwd_scratch(B, 512, L->feat), s2 = a0_dense_fwd_scratch(B, L->Npad, 512);
            L->fwd_scratch = L->alloc<float>(std::max(4LL, std::max(s1, s2)));
            if (d->algo == A0_ALGO_QR) {
                L->y = L->alloc<float>((long long)B * L->T);
                L->qr_taus = L->alloc<float>(L->T);
                std::vector<float> t((size_t)L->T);
                for (int i = 0; i < L->T; ++i) t[(size_t)i] = (2.0f * (float)i + 1.0f) / (2.0f * (float)L->T);          // agent.py:274: the quantile midpoints
                A0_HIP_THROW(hipMemcpy(L->qr_taus, t.data(), (size_t)L->T * 4, hipMemcpyHostToDevice));
            }
        }
        if (iqn) {
            const int K = fqf ? d->fqf_F : d->iqn_K, N = fqf ? d->fqf_F : d->iqn_N, Nd = fqf ? d->fqf_F : d->iqn_N_dash;
            L->rng.init(d->seed, 0);
            L->a_star = L->alloc<int>(B, true);
            long long sc = 4;
            auto ws = [&](a0_learner::QWs& w, int n_tau, float* act3, bool grads) {
                w.n_tau = n_tau; w.R = (long long)B * n_tau; w.act3 = act3;
                w.h = L->alloc<float>(w.R * 512); w.raw = L->alloc<float>(w.R * L->Npad); w.q = L->alloc<float>(w.R * d->A);
                w.cosx = L->alloc<float>(w.R * 64); w.emb = L->alloc<float>(w.R * L->feat); w.x = L->alloc<float>(w.R * L->feat);
                if (grads) { w.dq = L->alloc<float>(w.R * d->A, true); w.dx = L->alloc<float>(w.R * L->feat); w.demb = L->alloc<float>(w.R * L->feat); }
                for (int r : {B * K, B * Nd, B * N, B * (N - 1)}) {
                    if (r > w.R || r < 1) continue;
                    const long long a = a0_dense_fwd_scratch(r, L->feat, 64), b2 = a0_dense_fwd_scratch(r, 512, L->feat), c = a0_dense_fwd_scratch(r, L->Npad, 512);
                    sc = std::max(sc, std::max(a, std::max(b2, c)));
                }
            };
            ws(L->qo, N, L->act3_o, true);
            ws(L->qt, Nd > K ? Nd : K, L->act3_t, false);
            if (d->double_q) ws(L->qs, K, L->act3_s, false);
            if (fqf) sc = std::max(sc, (long long)a0_dense_fwd_scratch(B, 32, L->feat));
            L->fwd_scratch = L->alloc<float>(sc);
            L->t_sel = L->alloc<float>(ceil_to((long long)B * K, 4)); L->t_tgt = L->alloc<float>(ceil_to((long long)B * Nd, 4)); L->t_on = L->alloc<float>(ceil_to((long long)B * N, 4));
            L->y = L->alloc<float>((long long)B * Nd);
            if (fqf) {
                const int F = d->fqf_F;
                auto fw = [&](a0_learner::FWs& w) { w.logits = L->alloc<float>((long long)B * 32, true); w.tau_all = L->alloc<float>((long long)B * (F + 1)); w.tau_hat = L->alloc<float>((long long)B * F); };
                fw(L->fo); fw(L->ft);
                if (d->double_q) fw(L->fs);
                ws(L->qf, F, L->act3_o, false);
                L->inner_taus = L->alloc<float>((long long)B * F);
                L->rms_sq = U.rms_sq ? U.rms_sq : L->alloc<float>(L->frac.size(), true);
                L->frac_loss = L->alloc<float>(B, true); L->dfrac = L->alloc<float>((long long)B * 32, true); L->clip = L->alloc<float>(4, true);
            }
        }
    } catch (...) { delete L; throw; }
    *out = L;
    return A0_OK;
    A0_CATCH
}

// DeepQHead.forward (model.py:108-131; the distributional variants 137-177): relu(first_dense(x)), the q / value heads, the dueling combine — DeviceNet.head, dense branch
static int a0_dense_head(a0_learner* L, bool target, a0_learner::DWs& w, void* stream) {
    const int B = L->d.B;
    A0_CHECK(a0_dense_fwd(w.act3, L->feat, L->Wf(target), L->bf(target), w.h, B, 512, L->feat, 1, L->fwd_scratch, stream));
    A0_CHECK(a0_dense_fwd(w.h, 512, L->Wh(target), L->bh(target), w.raw, B, L->Npad, 512, 0, L->fwd_scratch, stream));
    return a0_dueling_fwd(w.raw, L->Npad, w.q, B, L->d.A, L->T, L->d.dueling ? 1 : 0, stream);
}

// IQNHead.forward (model.py:235-251) for `n_tau` fractions per sample: cosine features, the embedding times the state features (in the embedding GEMM's epilogue where the
// shape allows; kept for the backward pass when the pass is differentiated), fc1, the head, the dueling combine — DeviceNet.head of agent0_amd/deepq/engine.py
static int a0_iqn_head(a0_learner* L, bool target, a0_learner::QWs& w, const float* taus, int n_tau, bool grads, void* stream) {
    const int B = L->d.B, A = L->d.A;
    const int R = B * n_tau;
    const float* flat = target ? L->target : L->online;
    // (fc1 and the heads are NoisyLinear layers under NoisyNet — their composed weights; the cosine embedding is a plain Linear: model.py:204-217)
    const float *Wc = flat + L->cos.w(), *bc = flat + L->cos.b(), *Wf = L->Wf(target), *bf = L->bf(target), *Wh = L->Wh(target), *bh = L->bh(target);
    A0_CHECK(a0_cos_features(taus, w.cosx, R, 64, stream));
    if (!grads && a0_dense_fwd_scratch(R, L->feat, 64) == 0) A0_CHECK(a0_dense_fwd_mul(w.cosx, 64, Wc, bc, w.act3, n_tau, w.x, R, L->feat, 64, 1, stream));
    else if (grads && a0_dense_fwd_mul_keep_ok(R, L->feat, 64, 64)) A0_CHECK(a0_dense_fwd_mul_keep(w.cosx, 64, Wc, bc, w.act3, n_tau, w.emb, w.x, R, L->feat, 64, 1, stream));
    else {
        A0_CHECK(a0_dense_fwd(w.cosx, 64, Wc, bc, w.emb, R, L->feat, 64, 1, L->fwd_scratch, stream));
        A0_CHECK(a0_hadamard_fwd(w.emb, w.act3, w.x, B, n_tau, L->feat, stream));
    }
    A0_CHECK(a0_dense_fwd(w.x, L->feat, Wf, bf, w.h, R, 512, L->feat, 1, L->fwd_scratch, stream));
    A0_CHECK(a0_dense_fwd(w.h, 512, Wh, bh, w.raw, R, L->Npad, 512, 0, L->fwd_scratch, stream));
    return a0_dueling_fwd(w.raw, L->Npad, w.q, R, A, 1, L->d.dueling ? 1 : 0, stream);
}

extern "C" int a0_learner_destroy(a0_learner* L) { delete L; return A0_OK; }

extern "C" long long a0_learner_param_floats(const a0_learner* L) { return L ? L->n_pad : 0; }

// Data parallelism through the handle (the reference has no multi-GPU learner: launch.py's actors are its only parallelism): `comm` from
// a0_dp_init makes every a0_learner_update SUM its gradients over the ranks between backward and Adam — the two buckets, the side stream and the order of
// agent0_amd/deepq/dist.py::RcclGradAllReduce, issued eagerly.  comm = 0 switches the exchange off again.  The caller owns the communicator.
extern "C" int a0_learner_set_exchange(a0_learner* L, long long comm) {
    A0_TRY
    if (!L) return a0_fail(A0_EINVAL, "a0_learner_set_exchange: null handle");
    if (comm && !L->dp_side) {
        A0_HIP_THROW(hipStreamCreateWithFlags(&L->dp_side, hipStreamNonBlocking));
        for (hipEvent_t& e : L->dp_ev) A0_HIP_THROW(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    L->dp_comm = comm;
    L->dp_one_rank = false;
    if (comm) {       // a one-rank group (a rehearsal, or a job of one GPU): nothing to overlap, so both all-reduces go on the update's own stream
        int info[3] = {0, 0, 0};
        if (a0_dp_info(comm, info) == A0_OK) L->dp_one_rank = info[0] == 1;
    }
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_learner_set_params(a0_learner* L, const float* online_packed, const float* target_packed, void* stream) {
    A0_TRY
    if (!L || !online_packed) return a0_fail(A0_EINVAL, "a0_learner_set_params: null argument");
    hipStream_t st = (hipStream_t)stream;
    A0_HIP_THROW(hipMemcpyAsync(L->online, online_packed, (size_t)L->n_pad * 4, hipMemcpyDeviceToDevice, st));
    A0_HIP_THROW(hipMemcpyAsync(L->target, target_packed ? target_packed : online_packed, (size_t)L->n_pad * 4, hipMemcpyDeviceToDevice, st));      // target = deepcopy(model), agent.py:100
    a0_encoder_weights wo = L->enc(L->online), wtg = L->enc(L->target);
    A0_CHECK(a0_net_conv_wt_refresh(&wo, L->C, L->wt_on, stream));
    A0_CHECK(a0_net_conv_wt_refresh(&wtg, L->C, L->wt_tg, stream));
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_learner_loss_buffer(const a0_learner* L, float** loss_dev) {
    if (!L || !loss_dev) return a0_fail(A0_EINVAL, "a0_learner_loss_buffer: null argument");
    *loss_dev = L->loss;
    return A0_OK;
}

// Borrowed views of the handle's own HBM for inspection (parity harnesses walking the handle path link by link against the CPU oracle need the gradients and the
// differentiated pass's activations — the ReLU decisions — that a0_learner_get does not copy out).  Valid until the next call on the handle / a0_learner_destroy.
extern "C" int a0_learner_peek(const a0_learner* L, int what, float** dev_ptr, long long* count) {
    if (!L || !dev_ptr || !count) return a0_fail(A0_EINVAL, "a0_learner_peek: null argument");
    const long long B = L->d.B;
    switch (what) {
        case A0_PEEK_GRADS: *dev_ptr = L->grads; *count = L->n_pad; break;
        case A0_PEEK_ACT1: *dev_ptr = L->act1; *count = B * L->H1 * L->W1 * 32; break;
        case A0_PEEK_ACT2: *dev_ptr = L->act2; *count = B * L->H2 * L->W2 * 64; break;
        case A0_PEEK_ACT3: *dev_ptr = L->act3_o; *count = B * L->feat; break;
        case A0_PEEK_FC1: *dev_ptr = (L->d.algo == A0_ALGO_IQN || L->d.algo == A0_ALGO_FQF) ? L->qo.h : L->h; *count = ((L->d.algo == A0_ALGO_IQN || L->d.algo == A0_ALGO_FQF) ? L->qo.R : B) * 512; break;
        case A0_PEEK_LOSS: *dev_ptr = L->loss; *count = B; break;
        default: return a0_fail(A0_EINVAL, "a0_learner_peek: unknown buffer");
    }
    if (!*dev_ptr) return a0_fail(A0_ESTATE, "a0_learner_peek: this learner does not keep that buffer");
    return A0_OK;
}

extern "C" int a0_learner_get_frac_loss(const a0_learner* L, float* out_dev, void* stream) {
    A0_TRY
    if (!L || !out_dev || !L->frac_loss) return a0_fail(A0_EINVAL, "a0_learner_get_frac_loss: an fqf handle and an output buffer");
    A0_HIP_THROW(hipMemcpyAsync(out_dev, L->frac_loss, (size_t)L->d.B * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_learner_set_support(a0_learner* L, const float* atoms_host) {
    A0_TRY
    if (!L || !atoms_host || !L->atoms) return a0_fail(A0_EINVAL, "a0_learner_set_support: a c51 handle and a host array of num_atoms floats");
    A0_HIP_THROW(hipMemcpy(L->atoms, atoms_host, (size_t)L->T * 4, hipMemcpyHostToDevice));
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_learner_get(const a0_learner* L, float* online_out, float* target_out, float* adam_m_out, float* adam_v_out, int* state_out8, void* stream) {
    A0_TRY
    if (!L) return a0_fail(A0_EINVAL, "a0_learner_get: null handle");
    hipStream_t st = (hipStream_t)stream;
    if (online_out) A0_HIP_THROW(hipMemcpyAsync(online_out, L->online, (size_t)L->n_pad * 4, hipMemcpyDeviceToDevice, st));
    if (target_out) A0_HIP_THROW(hipMemcpyAsync(target_out, L->target, (size_t)L->n_pad * 4, hipMemcpyDeviceToDevice, st));
    if (adam_m_out) A0_HIP_THROW(hipMemcpyAsync(adam_m_out, L->m, (size_t)L->n_pad * 4, hipMemcpyDeviceToDevice, st));
    if (adam_v_out) A0_HIP_THROW(hipMemcpyAsync(adam_v_out, L->v, (size_t)L->n_pad * 4, hipMemcpyDeviceToDevice, st));
    if (state_out8) A0_HIP_THROW(hipMemcpyAsync(state_out8, L->state, 8 * sizeof(int), hipMemcpyDeviceToDevice, st));
    return A0_OK;
    A0_CATCH
}

// BaseLearner.train (agent.py:124-169): frames = u8 replay rows st || st_next of `row_bytes` bytes each, addressed through `slot` (ring slots of the sampled
// batch; NULL = dense batch), act / rew / done / wgt [B] on the device.  loss_out (optional) receives the per-sample losses [B] (what the caller feeds to
// update_priority, trainer.py:103-104).  NaN skip, step counter and target sync are decided on the device (state words as in a0_adam_step_sync_wt).
extern "C" int a0_learner_update(a0_learner* L, const uint8_t* frames, const int* slot, long long row_bytes, const int* act, const float* rew, const float* done,
                                 const float* wgt, float* loss_out, void* stream) {
    A0_TRY
    a0_trace_scope range("update");
    if (!L || !frames || !act || !rew || !done || !wgt) return a0_fail(A0_EINVAL, "a0_learner_update: null argument");
    const int B = L->d.B, A = L->d.A, dq = L->d.double_q ? 1 : 0, obs = L->C * L->H * L->W;
    if (row_bytes < 2LL * obs) return a0_fail(A0_EINVAL, "a0_learner_update: a replay row holds st || st_next (2 x C x H x W bytes)");
    float* on = L->online; float* tg = L->target;
    a0_encoder_weights w_on = L->enc(on), w_tg = L->enc(tg);
    if (L->d.noisy) {
        // BaseLearner.train (agent.py:125-127): reset_noise of the online, then of the target network — ONE Philox fill of the joint buffer (the same draws as two
        // fills: every vector is padded to the offsets' stride of four) — and both networks' effective weights in one launch
        const long long nn = 2 * L->noise_len;
        A0_CHECK(a0_rng_normal(L->rng.seed, 4 /* STREAM_NOISE */, L->rng.reserve(4, nn), 0.1f, L->noise, nn, stream));
        const float *mu[6], *sg[6], *nin[6], *nw[6], *nb[6];
        float* eff[6];
        int N[6], K[6], r0[6], r1[6], nm = 0;
        for (int net = 0; net < 2; ++net) {
            const float* flat = net ? tg : on;
            float* e = net ? L->eff_tg : L->eff_on;
            const float* nz = L->noise + (net ? L->noise_len : 0);
            for (int k = 0; k < L->n_mods; ++k, ++nm) {
                const a0_noise_mod& m = L->mods[k];
                const Blk &bm = m.block ? L->head : L->fc1, &bs = m.block ? L->head_sigma : L->fc1_sigma, &be = m.block ? L->eff_head : L->eff_fc1;
                mu[nm] = flat + bm.off; sg[nm] = flat + bs.off; eff[nm] = e + be.off; N[nm] = bm.N; K[nm] = bm.K; r0[nm] = m.r0; r1[nm] = m.r1;
                nin[nm] = nz + m.off_in; nw[nm] = nz + m.off_w; nb[nm] = nz + m.off_b;
            }
        }
        A0_CHECK(a0_noisy_multi(0, nm, mu, sg, eff, N, K, r0, r1, nin, nw, nb, stream));
    }
    a0_pending_reduce pend;
    pend.n = 0;
    bool head_wgrad_done = false;      // a distributional head's weight gradient rode in its data gradient's launch (a0_dense_dgrad_wgrad, small class)
    // data parallelism: the dense range is exchanged right after the dense backward, so its slab reductions cannot wait for the encoder's launch (engine.py::_backward_dense)
    const bool dp = L->dp_comm != 0;
    // A0_DP_ONE_STREAM=1 (the same on every rank; a tuning aid): both all-reduces on the caller's stream — no overlap with the encoder backward, but none of the three
    // cross-stream hand-offs either, which cost an eager host ~10 us each (profiles/r04_experiments.md)
    // Default (round 6): the side stream when the group has more than one rank (the dense bucket, 95 % of the bytes, travels beside the encoder backward — what the
    // captured form does), the caller's stream in a one-rank group (+0.1 ms per iteration instead of +0.65, profiles/r04_native_dp_one_rank_ab.txt); A0_DP_ONE_STREAM=0 / 1 forces either.
    static const int dp_force = getenv("A0_DP_ONE_STREAM") != nullptr ? (atoi(getenv("A0_DP_ONE_STREAM")) != 0 ? 1 : 0) : -1;
    const bool dp_inline = dp_force >= 0 ? dp_force == 1 : L->dp_one_rank;
    a0_pending_reduce* const pp = dp ? nullptr : &pend;
    a0_frames_arg f_next{frames, slot, row_bytes, obs}, f_obs{frames, slot, row_bytes, 0};
    if ((L->d.algo == A0_ALGO_QR && !L->qr_fused) || (L->d.algo == A0_ALGO_MDQN && !L->mdqn_fused)) {
        // ---- engine.py's layer-by-layer path: the passes' encoders in one launch, then per pass fc1, head, dueling combine
        const bool mdqn = L->d.algo == A0_ALGO_MDQN;
        const int T = L->T;
        a0_encoder_pass passes[3];
        int np = 0;
        passes[np++] = a0_encoder_pass{L->wt_tg, &w_tg, &f_next, B, nullptr, nullptr, L->gt.act3};
        if (mdqn) passes[np++] = a0_encoder_pass{L->wt_tg, &w_tg, &f_obs, B, nullptr, nullptr, L->gm.act3};            // the target network on the CURRENT observation (agent.py:202-204)
        else if (dq) passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_next, B, nullptr, nullptr, L->gs.act3};
        passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_obs, B, L->act1, L->act2, L->go.act3};
        A0_CHECK(a0_net_encoder_fwd_fused_multi(L->C, L->H, L->W, np, passes, stream));
        A0_CHECK(a0_dense_head(L, true, L->gt, stream));
        if (mdqn) {
            A0_CHECK(a0_dense_head(L, true, L->gm, stream));
            A0_CHECK(a0_dense_head(L, false, L->go, stream));
            A0_CHECK(a0_loss_mdqn(L->go.q, L->gt.q, L->gm.q, A, act, rew, done, wgt, L->gamma_n, (float)L->d.mdqn_tau, (float)L->d.mdqn_lo, B, L->loss, L->g_dq, L->state, stream));
        } else {
            if (dq) {
                A0_CHECK(a0_dense_head(L, false, L->gs, stream));
                A0_CHECK(a0_select_action(L->gs.q, (long long)A * T, T, 1, B, A, T, 1, nullptr, L->a_star, nullptr, nullptr, stream));
            } else {
                A0_CHECK(a0_select_action(L->gt.q, (long long)A * T, T, 1, B, A, T, 1, nullptr, L->a_star, nullptr, nullptr, stream));
            }
            A0_CHECK(a0_dense_head(L, false, L->go, stream));
            A0_CHECK(a0_quantile_target(L->gt.q, (long long)A * T, 1, T, L->a_star, rew, done, L->gamma_n, B, T, L->y, stream));
            A0_HIP_THROW(hipMemsetAsync(L->g_dq, 0, (size_t)B * A * T * 4, (hipStream_t)stream));
            A0_CHECK(a0_loss_quantile_huber(L->go.q, (long long)A * T, 1, T, L->y, L->qr_taus, 0, act, wgt, B, T, T, L->loss, L->g_dq, L->state, stream));
        }
        // DeviceLearner._backward_dense: the dueling combine's and the head's backward-data passes; fc1's and the weight gradients follow in the common block
        A0_CHECK(a0_dueling_bwd(L->g_dq, L->draw, L->Npad, B, A, T, L->d.dueling ? 1 : 0, stream));
        if (a0_dense_dgrad_wgrad_ok2(B, L->Npad, 512)) {      // ... side by side with the head's weight gradient (DeviceLearner._backward_dense)
            A0_CHECK(a0_dense_dgrad_wgrad(L->draw, L->Wh(false), L->h, 512, L->dh, L->grads + L->head.off, B, L->Npad, 512, stream));
            head_wgrad_done = true;
        } else
        A0_CHECK(a0_dense_dgrad(L->draw, L->Wh(false), L->h, L->dh, B, L->Npad, 512, stream));
    } else if (L->d.algo == A0_ALGO_FQF) {
        // ---- FQFLearner.train_step (agent.py:334-388) in the order of agent0_amd/deepq/engine.py's fqf path
        const int F = L->F;
        a0_encoder_pass passes[3];
        int np = 0;
        passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_obs, B, L->act1, L->act2, L->act3_o};
        passes[np++] = a0_encoder_pass{L->wt_tg, &w_tg, &f_next, B, nullptr, nullptr, L->act3_t};
        if (dq) passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_next, B, nullptr, nullptr, L->act3_s};
        A0_CHECK(a0_net_encoder_fwd_fused_multi(L->C, L->H, L->W, np, passes, stream));
        // FQFHead.prop_taus (model.py:268-278): the fraction net on the (detached) features -> taus, tau_hats
        auto taus = [&](const float* flat, const float* act3, a0_learner::FWs& w) -> int {
            A0_CHECK(a0_dense_fwd(act3, L->feat, flat + L->frac.w(), flat + L->frac.b(), w.logits, B, 32, L->feat, 0, L->fwd_scratch, stream));
            return a0_fqf_taus(w.logits, 32, w.tau_all, w.tau_hat, B, F, stream);
        };
        A0_CHECK(taus(on, L->act3_o, L->fo));
        A0_CHECK(a0_iqn_head(L, false, L->qo, L->fo.tau_hat, F, true, stream));
        if (dq) {
            A0_CHECK(taus(on, L->act3_s, L->fs));
            A0_CHECK(a0_iqn_head(L, false, L->qs, L->fs.tau_hat, F, false, stream));
            A0_CHECK(a0_select_action(L->qs.q, (long long)F * A, 1, A, B, A, F, 3, L->fs.tau_all, L->a_star, nullptr, nullptr, stream));
        } else {
            A0_CHECK(taus(tg, L->act3_t, L->ft));
            A0_CHECK(a0_iqn_head(L, true, L->qt, L->ft.tau_hat, F, false, stream));
            A0_CHECK(a0_select_action(L->qt.q, (long long)F * A, 1, A, B, A, F, 3, L->ft.tau_all, L->a_star, nullptr, nullptr, stream));
        }
        A0_CHECK(a0_iqn_head(L, true, L->qt, L->fo.tau_hat, F, false, stream));                      // quirk Q16: the target evaluated at the ONLINE tau-hats
        A0_CHECK(a0_quantile_target(L->qt.q, (long long)F * A, A, 1, L->a_star, rew, done, L->gamma_n, B, F, L->y, stream));
        A0_HIP_THROW(hipMemsetAsync(L->qo.dq, 0, (size_t)L->qo.R * A * 4, (hipStream_t)stream));
        A0_CHECK(a0_loss_quantile_huber(L->qo.q, (long long)F * A, A, 1, L->y, L->fo.tau_hat, F, act, wgt, B, F, F, L->loss, L->qo.dq, L->state, stream));
        // fraction loss: q at the interior taus (no grad), its gradient w.r.t. the fraction logits; the fraction net's RMSprop step follows the backward pass
        A0_CHECK(a0_fqf_inner_taus(L->fo.tau_all, L->inner_taus, B, F, stream));
        A0_CHECK(a0_iqn_head(L, false, L->qf, L->inner_taus, F - 1, false, stream));
        A0_CHECK(a0_fqf_fraction_loss(L->qf.q, L->qo.q, L->fo.tau_all, act, wgt, B, F, A, 32, L->frac_loss, L->dfrac, L->fo.logits, stream));
        A0_CHECK(a0_dense_wgrad(L->dfrac, L->act3_o, L->feat, L->grads + L->frac.off, B, 32, L->feat, L->slabs, stream));
    } else if (L->d.algo == A0_ALGO_IQN) {
        // ---- IQNLearner.train_step (agent.py:296-331) in the order of agent0_amd/deepq/engine.py's iqn path.  The update's three tau draws first (BaseLearner.train_batch:
        // K, N', N fractions per sample, Philox stream 3)
        const int K = L->d.iqn_K, N = L->d.iqn_N, Nd = L->d.iqn_N_dash;
        A0_CHECK(a0_rng_uniform(L->rng.seed, 3, L->rng.reserve(3, (long long)B * K), L->t_sel, (long long)B * K, stream));
        A0_CHECK(a0_rng_uniform(L->rng.seed, 3, L->rng.reserve(3, (long long)B * Nd), L->t_tgt, (long long)B * Nd, stream));
        A0_CHECK(a0_rng_uniform(L->rng.seed, 3, L->rng.reserve(3, (long long)B * N), L->t_on, (long long)B * N, stream));
        a0_encoder_pass passes[3];
        int np = 0;
        passes[np++] = a0_encoder_pass{L->wt_tg, &w_tg, &f_next, B, nullptr, nullptr, L->act3_t};
        if (dq) passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_next, B, nullptr, nullptr, L->act3_s};
        passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_obs, B, L->act1, L->act2, L->act3_o};
        A0_CHECK(a0_net_encoder_fwd_fused_multi(L->C, L->H, L->W, np, passes, stream));
        if (dq) {        // greedy next action from the online network's K-sample mean (agent.py:303-306)
            A0_CHECK(a0_iqn_head(L, false, L->qs, L->t_sel, K, false, stream));
            A0_CHECK(a0_select_action(L->qs.q, (long long)K * A, 1, A, B, A, K, 1, nullptr, L->a_star, nullptr, nullptr, stream));
        } else {
            A0_CHECK(a0_iqn_head(L, true, L->qt, L->t_sel, K, false, stream));
            A0_CHECK(a0_select_action(L->qt.q, (long long)K * A, 1, A, B, A, K, 1, nullptr, L->a_star, nullptr, nullptr, stream));
        }
        A0_CHECK(a0_iqn_head(L, true, L->qt, L->t_tgt, Nd, false, stream));
        A0_CHECK(a0_quantile_target(L->qt.q, (long long)Nd * A, A, 1, L->a_star, rew, done, L->gamma_n, B, Nd, L->y, stream));
        A0_CHECK(a0_iqn_head(L, false, L->qo, L->t_on, N, true, stream));
        A0_HIP_THROW(hipMemsetAsync(L->qo.dq, 0, (size_t)L->qo.R * A * 4, (hipStream_t)stream));
        A0_CHECK(a0_loss_quantile_huber(L->qo.q, (long long)N * A, A, 1, L->y, L->t_on, N, act, wgt, B, N, Nd, L->loss, L->qo.dq, L->state, stream));
    } else if (L->d.algo == A0_ALGO_C51 || L->d.algo == A0_ALGO_QR) {
        // ---- C51Learner.train_step (agent.py:218-268) / QRLearner.train_step (agent.py:272-293), in the order of agent0_amd/deepq/engine.py's c51 / qr path: the three encoder passes in one launch; the online
        // fc1 over [s ; s'] rows as ONE GEMM and the target's, their slabs finished by one reduction launch; the two head GEMMs; and one launch for everything behind
        // them (slab sums, dueling, greedy next action, projection, cross entropy, head gradient)
        a0_encoder_pass passes[3];
        int np = 0;
        passes[np++] = a0_encoder_pass{L->wt_tg, &w_tg, &f_next, B, nullptr, nullptr, L->act3_t};
        if (dq) passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_next, B, nullptr, nullptr, L->act3_s};
        passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_obs, B, L->act1, L->act2, L->act3_o};
        A0_CHECK(a0_net_encoder_fwd_fused_multi(L->C, L->H, L->W, np, passes, stream));
        // the passes (online on s, online on s' under double-Q, target on s') are GEMMs of one shape each for fc1 and for the head: one grouped launch per layer, the
        // online passes' rows interleaved into the [splits][R_on][N] buffers the reduction and the loss kernel read
        const int npass = dq ? 3 : 2;
        int ns_on = L->ns_on, ns_tg = L->ns_fc1, nh_on = L->nh_on, nh_tg = L->nh_tg;
        const bool grouped = a0_dense_fwd_partial_multi_ok(npass, B, 512, L->feat) && a0_dense_fwd_partial_multi_ok(npass, B, L->Npad, 512);
        if (grouped) {
            const float* Xs[3] = {L->act3_on, dq ? L->act3_s : L->act3_t, L->act3_t};
            const float* Ws[3] = {L->Wf(false), dq ? L->Wf(false) : L->Wf(true), L->Wf(true)};
            float* sl[3] = {L->fc1_on, dq ? L->fc1_on + (long long)B * 512 : L->fc1_tg, L->fc1_tg};
            const long long st[3] = {(long long)L->R_on * 512, dq ? (long long)L->R_on * 512 : (long long)B * 512, (long long)B * 512};
            A0_CHECK(a0_dense_fwd_partial_multi(npass, Xs, L->feat, Ws, B, 512, L->feat, sl, st, stream));
            ns_on = ns_tg = a0_dense_fwd_partial_multi_slabs(npass, B, 512, L->feat);
        } else {
            A0_CHECK(a0_dense_fwd_partial(L->act3_t, L->feat, L->Wf(true), B, 512, L->feat, L->fc1_tg, stream));
            A0_CHECK(a0_dense_fwd_partial(L->act3_on, L->feat, L->Wf(false), L->R_on, 512, L->feat, L->fc1_on, stream));
        }
        {
            const float* sl[2] = {L->fc1_on, L->fc1_tg};
            const long long st[2] = {(long long)L->R_on * 512, (long long)B * 512};
            const int ns[2] = {ns_on, ns_tg}, rows[2] = {L->R_on, B};
            const float* bias[2] = {L->bf(false), L->bf(true)};
            float* out[2] = {L->h_on, L->h_tg};
            A0_CHECK(a0_reduce_bias_act_multi(2, sl, st, ns, bias, out, rows, 512, 1, stream));
        }
        if (grouped) {
            const float* Xs[3] = {L->h_on, dq ? L->h_on + (long long)B * 512 : L->h_tg, L->h_tg};
            const float* Ws[3] = {L->Wh(false), dq ? L->Wh(false) : L->Wh(true), L->Wh(true)};
            float* sl[3] = {L->hs_on, dq ? L->hs_on + (long long)B * L->Npad : L->hs_tg, L->hs_tg};
            const long long st[3] = {(long long)L->R_on * L->Npad, dq ? (long long)L->R_on * L->Npad : (long long)B * L->Npad, (long long)B * L->Npad};
            A0_CHECK(a0_dense_fwd_partial_multi(npass, Xs, 512, Ws, B, L->Npad, 512, sl, st, stream));
            nh_on = nh_tg = a0_dense_fwd_partial_multi_slabs(npass, B, L->Npad, 512);
        } else {
            A0_CHECK(a0_dense_fwd_partial(L->h_on, 512, L->Wh(false), L->R_on, L->Npad, 512, L->hs_on, stream));
            A0_CHECK(a0_dense_fwd_partial(L->h_tg, 512, L->Wh(true), B, L->Npad, 512, L->hs_tg, stream));
        }
        if (L->d.algo == A0_ALGO_QR)
            A0_CHECK(a0_qr_head_loss_slabs(L->hs_on, (long long)L->R_on * L->Npad, nh_on, L->R_on, L->hs_tg, (long long)B * L->Npad, nh_tg, dq ? B : -1, L->bh(false), L->bh(true),
                                           L->Npad, A, L->T, L->d.dueling ? 1 : 0, act, rew, done, wgt, L->qr_taus, L->gamma_n, B, L->loss, L->draw, L->q_o, L->q_t, L->a_star,
                                           L->state, stream));
        else
        A0_CHECK(a0_c51_head_loss_slabs(L->hs_on, (long long)L->R_on * L->Npad, nh_on, L->R_on, L->hs_tg, (long long)B * L->Npad, nh_tg, dq ? B : -1, L->bh(false), L->bh(true),
                                        L->Npad, A, L->T, L->d.dueling ? 1 : 0, act, rew, done, wgt, L->atoms, L->gamma_n, (float)L->d.vmin, (float)L->d.vmax, B, L->loss, L->draw,
                                        L->q_o, L->q_t, L->m_proj, L->a_star, L->state, stream));
        // the head's data gradient (the scalar-head kernel above folds it in; the distributional one does not)
        if (a0_dense_dgrad_wgrad_ok2(B, L->Npad, 512)) {      // ... side by side with the head's weight gradient (DeviceLearner._backward_dense)
            A0_CHECK(a0_dense_dgrad_wgrad(L->draw, L->Wh(false), L->h, 512, L->dh, L->grads + L->head.off, B, L->Npad, 512, stream));
            head_wgrad_done = true;
        } else
        A0_CHECK(a0_dense_dgrad(L->draw, L->Wh(false), L->h, L->dh, B, L->Npad, 512, stream));
    } else {
    // ---- forward: the target pass on s', the online pass on s' (double-Q) and the online pass on s as ONE encoder launch, then their fc1 GEMMs (split-K slabs)
    // (mdqn, round 5: the third pass is the TARGET network on the CURRENT observation, agent.py:202-204, in the order engine.py's mdqn path issues the passes)
    const bool mdqn = L->d.algo == A0_ALGO_MDQN;
    a0_encoder_pass passes[3];
    int np = 0;
    passes[np++] = a0_encoder_pass{L->wt_tg, &w_tg, &f_next, B, nullptr, nullptr, L->act3_t};
    if (mdqn) passes[np++] = a0_encoder_pass{L->wt_tg, &w_tg, &f_obs, B, nullptr, nullptr, L->act3_s};
    else if (dq) passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_next, B, nullptr, nullptr, L->act3_s};
    passes[np++] = a0_encoder_pass{L->wt_on, &w_on, &f_obs, B, L->act1, L->act2, L->act3_o};
    A0_CHECK(a0_net_encoder_fwd_fused_multi(L->C, L->H, L->W, np, passes, stream));
    int ns = L->ns_fc1;
    const bool third = mdqn || dq;
    const int n_fc1 = third ? 3 : 2;
    const float* W3 = mdqn ? L->Wf(true) : L->Wf(false);
    if (a0_dense_fwd_partial_multi_ok(n_fc1, B, 512, L->feat)) {      // the passes' fc1 GEMMs as one launch (fewer, deeper splits each)
        const float* Xs[3] = {L->act3_o, L->act3_t, L->act3_s};
        const float* Ws[3] = {L->Wf(false), L->Wf(true), W3};
        A0_CHECK(a0_dense_fwd_partial_multi(n_fc1, Xs, L->feat, Ws, B, 512, L->feat, L->fc1_slabs, nullptr, stream));
        ns = a0_dense_fwd_partial_multi_slabs(n_fc1, B, 512, L->feat);
    } else {
        A0_CHECK(a0_dense_fwd_partial(L->act3_t, L->feat, L->Wf(true), B, 512, L->feat, L->fc1_slabs[1], stream));
        if (third) A0_CHECK(a0_dense_fwd_partial(L->act3_s, L->feat, W3, B, 512, L->feat, L->fc1_slabs[2], stream));
        A0_CHECK(a0_dense_fwd_partial(L->act3_o, L->feat, L->Wf(false), B, 512, L->feat, L->fc1_slabs[0], stream));
    }
    // ---- heads of both networks, dueling, argmax, Huber loss, head gradient and the head's backward-data pass in one launch (agent.py:173-190; mdqn 193-215)
    if (mdqn)
        A0_CHECK(a0_mdqn_head_loss_slabs(L->fc1_slabs[0], L->fc1_slabs[1], L->fc1_slabs[2], (long long)B * 512, ns, L->bf(false), L->bf(true), L->h, L->Wh(false), L->bh(false),
                                         L->Wh(true), L->bh(true), A, L->d.dueling ? 1 : 0, L->Npad, act, rew, done, wgt, L->gamma_n, (float)L->d.mdqn_tau, (float)L->d.mdqn_lo, B,
                                         L->loss, L->q_o, L->q_t, L->q_cur, L->draw, L->state, L->dh, stream));
    else
    A0_CHECK(a0_dqn_head_loss_slabs(L->fc1_slabs[0], L->fc1_slabs[1], dq ? L->fc1_slabs[2] : nullptr, (long long)B * 512, ns, L->bf(false), L->bf(true), L->h,
                                    L->Wh(false), L->bh(false), L->Wh(true), L->bh(true), A, L->d.dueling ? 1 : 0, L->Npad, act, rew, done, wgt, L->gamma_n, B,
                                    L->loss, L->q_o, L->q_t, L->draw, L->state, L->dh, stream));
    }
    const bool quantile = L->d.algo == A0_ALGO_IQN || L->d.algo == A0_ALGO_FQF;
    if (quantile) {
        const int N = L->qo.n_tau;
        // ---- dense backward over the B * n_tau rows of the differentiated pass (DeviceLearner._backward_dense, quantile branch)
        const int R = (int)L->qo.R;
        A0_CHECK(a0_dueling_bwd(L->qo.dq, L->draw, L->Npad, R, A, 1, L->d.dueling ? 1 : 0, stream));
        A0_CHECK(a0_dense_dgrad(L->draw, L->Wh(false), L->qo.h, L->dh, R, L->Npad, 512, stream));
        if (a0_dense_dgrad_hadamard_ok(R, 512, L->feat, N)) {      // round 6: dx = dh W never reaches HBM (DeviceLearner._backward_dense takes the same branch)
            A0_CHECK(a0_dense_dgrad_hadamard(L->dh, L->Wf(false), L->qo.emb, L->act3_o, L->qo.demb, L->d3, R, 512, L->feat, N, stream));
        } else {
            A0_CHECK(a0_dense_dgrad(L->dh, L->Wf(false), nullptr, L->qo.dx, R, 512, L->feat, stream));
            A0_CHECK(a0_hadamard_bwd(L->qo.dx, L->qo.emb, L->act3_o, L->qo.demb, L->d3, B, N, L->feat, stream));
        }
        {
            const float* dY[3] = {L->draw, L->dh, L->qo.demb};
            const float* X[3] = {L->qo.h, L->qo.x, L->qo.cosx};
            const int ldx[3] = {512, L->feat, 64}, Rr[3] = {R, R, R}, Nn[3] = {L->Npad, 512, L->feat}, Kk[3] = {512, L->feat, 64};
            float* G[3] = {L->grads + L->head.off, L->grads + L->fc1.off, L->grads + L->cos.off};
            A0_CHECK(a0_dense_wgrad_multi(3, dY, X, ldx, G, Rr, Nn, Kk, L->slabs, L->slab_off3, pp, stream));
        }
    }
    // ---- backward (agent.py:153-155): fc1's data gradient, the dense weight gradients with one slab reduction, the encoder
    if (!quantile) {
        const float* dY[2] = {L->draw, L->dh};
        const float* X[2] = {L->h, L->act3_o};
        const int ldx[2] = {512, L->feat}, R[2] = {B, B}, N[2] = {L->Npad, 512}, K[2] = {512, L->feat};
        float* G[2] = {L->grads + L->head.off, L->grads + L->fc1.off};
        if (head_wgrad_done) {
            if (a0_dense_dgrad_wgrad_ok(B, 512, L->feat)) {
                A0_CHECK(a0_dense_dgrad_wgrad(L->dh, L->Wf(false), L->act3_o, L->feat, L->d3, G[1], B, 512, L->feat, stream));
            } else {
                A0_CHECK(a0_dense_dgrad(L->dh, L->Wf(false), L->act3_o, L->d3, B, 512, L->feat, stream));
                A0_CHECK(a0_dense_wgrad_multi(1, dY + 1, X + 1, ldx + 1, G + 1, R + 1, N + 1, K + 1, L->slabs, L->slab_off + 1, pp, stream));
            }
        } else if (a0_dense_dgrad_wgrad2_ok(B, 512, L->feat, L->Npad, 512)) {
            // fc1's data gradient, fc1's weight gradient and the head's weight gradient — all three wait for the loss kernel only — as ONE launch (DeviceLearner._backward_dense)
            A0_CHECK(a0_dense_dgrad_wgrad2(L->dh, L->Wf(false), L->act3_o, L->feat, L->d3, G[1], B, 512, L->feat, L->draw, L->h, 512, G[0], L->Npad, 512, stream));
        } else if (a0_dense_dgrad_wgrad_ok(B, 512, L->feat)) {      // fc1's data gradient and weight gradient as one launch; the head's weight gradient alone
            A0_CHECK(a0_dense_dgrad_wgrad(L->dh, L->Wf(false), L->act3_o, L->feat, L->d3, G[1], B, 512, L->feat, stream));
            A0_CHECK(a0_dense_wgrad_multi(1, dY, X, ldx, G, R, N, K, L->slabs, L->slab_off, pp, stream));
        } else {
            A0_CHECK(a0_dense_dgrad(L->dh, L->Wf(false), L->act3_o, L->d3, B, 512, L->feat, stream));
            A0_CHECK(a0_dense_wgrad_multi(2, dY, X, ldx, G, R, N, K, L->slabs, L->slab_off, pp, stream));
        }
    }
    auto sigma_grads = [&]() -> int {      // d sigma = d eff * eps, from the reduced gradients in the mu blocks (model.py:78-87 differentiated)
        const float *gmu[3], *nin[3], *nw[3], *nb[3];
        float* gs[3];
        int N[3], K[3], r0[3], r1[3];
        for (int k = 0; k < L->n_mods; ++k) {
            const a0_noise_mod& m = L->mods[k];
            const Blk &bm = m.block ? L->head : L->fc1, &bs = m.block ? L->head_sigma : L->fc1_sigma;
            gmu[k] = L->grads + bm.off; gs[k] = L->grads + bs.off; N[k] = bm.N; K[k] = bm.K; r0[k] = m.r0; r1[k] = m.r1;
            nin[k] = L->noise + m.off_in; nw[k] = L->noise + m.off_w; nb[k] = L->noise + m.off_b;
        }
        return a0_noisy_multi(1, L->n_mods, gmu, nullptr, gs, N, K, r0, r1, nin, nw, nb, stream);
    };
    const long long conv_end = L->fc1.off;      // [0, conv_end): convolution blocks; [conv_end, n_pad]: dense blocks + the NaN flag (layout.py; the two buckets of dist.RcclGradAllReduce)
    if (dp) {
        // DeviceLearner.exchange_begin: every dense gradient (and the NaN flag riding behind them) is final here; their SUM over the ranks runs on the side stream
        // while the encoder backward computes the convolution gradients on this one
        if (L->d.noisy) A0_CHECK(sigma_grads());
        A0_CHECK(a0_nan_flag_export(L->state, L->grads + L->n_pad, stream));
        if (dp_inline) {
            A0_CHECK(a0_dp_allreduce(L->dp_comm, L->grads + conv_end, L->n_pad + 1 - conv_end, stream));
        } else {
            A0_HIP_THROW(hipEventRecord(L->dp_ev[0], (hipStream_t)stream));
            A0_HIP_THROW(hipStreamWaitEvent(L->dp_side, L->dp_ev[0], 0));
            A0_CHECK(a0_dp_allreduce(L->dp_comm, L->grads + conv_end, L->n_pad + 1 - conv_end, L->dp_side));
        }
    }
    A0_CHECK(a0_net_encoder_dgrad_fused(L->C, L->H, L->W, L->wt_on, L->d3, L->act1, L->act2, B, L->d2, L->d1, stream));
    A0_CHECK(a0_net_encoder_wgrad(L->net, &w_on, &f_obs, B, L->act1, L->act2, L->d3, L->d2, L->d1, L->grads + L->conv1.off, L->grads + L->conv2.off, L->grads + L->conv3.off,
                                  L->slabs + L->enc_slab_off, &pend, stream));
    if (dp) {
        // DeviceLearner.exchange_end: the convolution bucket behind the dense one on the same communicator and stream (one total order on every rank), then the join
        if (dp_inline) {
            A0_CHECK(a0_dp_allreduce(L->dp_comm, L->grads, conv_end, stream));
        } else {
            A0_HIP_THROW(hipEventRecord(L->dp_ev[1], (hipStream_t)stream));
            A0_HIP_THROW(hipStreamWaitEvent(L->dp_side, L->dp_ev[1], 0));
            A0_CHECK(a0_dp_allreduce(L->dp_comm, L->grads, conv_end, L->dp_side));
            A0_HIP_THROW(hipEventRecord(L->dp_ev[2], L->dp_side));
            A0_HIP_THROW(hipStreamWaitEvent((hipStream_t)stream, L->dp_ev[2], 0));
        }
    } else if (L->d.noisy) {
        A0_CHECK(sigma_grads());
    }
    if (loss_out) A0_HIP_THROW(hipMemcpyAsync(loss_out, L->loss, (size_t)B * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (L->d.algo == A0_ALGO_FQF)      // unconditional, like the reference's fqf_optimizer.step() in front of the NaN guard (agent.py:139-148); lr / 2e4, alpha 0.95, eps 1e-5
        A0_CHECK(a0_rmsprop_step(on + L->frac.off, L->grads + L->frac.off, L->rms_sq, L->frac.size(), L->d.lr / 2e4, 0.95, 1e-5, L->d.max_grad_norm > 0.0 ? L->d.max_grad_norm : -1.0, L->clip, stream));
    // ---- Adam (eps = 1e-2 / B unless given), NaN guard, update counter, target copy every target_update_freq updates, weight-copy refresh (agent.py:102-106,152-161)
    const double eps = L->d.adam_eps > 0.0 ? L->d.adam_eps : 1e-2 / (double)B;
    A0_CHECK(a0_adam_step_sync_wt(on, L->grads, L->m, L->v, L->n_adam, L->state, L->scalars, L->d.lr, 0.9, 0.999, eps, L->d.target_update_freq, tg, L->n_pad, dp ? L->grads + L->n_pad : nullptr, &w_on, L->C,
                                  L->wt_on, L->wt_tg, L->loss, B, L->loss_ring, L->loss_ring_cap, stream));
    return A0_OK;
    A0_CATCH
}
