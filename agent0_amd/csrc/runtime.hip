// The rest of the actor -> replay -> learner loop behind opaque handles with library-owned HBM, so that a host in ANY language can run
// BASELINE configs[1] with a handful of C calls per iteration (tests/c_host_demo.c does, in plain C):
//
//   a0_rbuf   ReplayDataset (reference agent0/deepq/replay.py:14-59 + the sampling of trainer.py:63-72,91-96): the HBM ring of st || st_next rows with its
//             metadata, the sampler state (uniform: the DataLoader's shuffled epochs as a Feistel permutation; prioritized: the sum-tree) and the schedules
//   a0_actor  Actor (agent.py:19-90) on the device-resident synthetic env: observations, n-step ring, episode statistics; one call = one sample_steps rollout
//             whose transitions land in the replay ring
//
// Host-side bookkeeping (cursors, epochs, Philox offsets, beta) is the part of agent0_amd/deepq/{replay,agent}.py and common/utils.py that these handles
// restate in C++; every device operation is one of the entry points declared above them in include/agent0_hip.h, in the order the Python classes issue
// them — the rings, parameters and statistics of a run driven through the handles are bit-identical to the Python Trainer's
// (tests/test_gpu_trainer.py::test_native_handles_run_the_loop_like_the_python_trainer).
#include "learner_state.h"

namespace {

typedef a0_host_rng Rng;      // learner_state.h
constexpr int STREAM_EGREEDY_U = 1, STREAM_EGREEDY_A = 2, STREAM_SUMTREE = 5, STREAM_PERM = 6;

struct Owned {
    std::vector<void*> ptrs;
    template <class T> T* alloc(long long n, bool zero = true) {
        void* p = nullptr;
        A0_HIP_THROW(hipMalloc(&p, (size_t)(n > 0 ? n : 1) * sizeof(T)));
        ptrs.push_back(p);
        if (zero) A0_HIP_THROW(hipMemset(p, 0, (size_t)(n > 0 ? n : 1) * sizeof(T)));
        return (T*)p;
    }
    ~Owned() { for (void* p : ptrs) (void)hipFree(p); }
};

// max_p^alpha as the new leaves' value (replay.py:51-52: `max_p ** alpha`, in float64 like torch's `.double() ** alpha`)
__global__ void a0_pow_scalar_kernel(const float* __restrict__ p, double alpha, float* __restrict__ out) {
    const double x = (double)p[0];
    out[0] = (float)(alpha == 0.5 ? sqrt(x) : pow(x, alpha));
}
__global__ void a0_fill_one_kernel(float* __restrict__ p, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

// ================================================================================================ replay
struct a0_rbuf {
    a0_rbuf_desc d;
    long long size = 0, top = 0, written = 0, cap2 = 1;
    int obs_bytes = 0, B = 0;
    long long row_bytes = 0;
    bool prio = false;             // prioritized replay: sum-tree (prioritize == 1) or the reference's flat priority vector (prioritize == 2, `flat`)
    bool flat = false;             // replay.py:45-59 / trainer.py:91-104 to the letter: uniform permutation batches, priority[-n:] = max_p^alpha on extend, whole-capacity sum
    float *prio_vec = nullptr, *psum = nullptr, *sum_scratch = nullptr;
    double beta_use = 0.0, sched_cur = 0.0, beta_inc = 0.0;      // importance exponent in use / LinearSchedule(beta0, 1, total_steps).current / its increment per transition
    struct { bool open = false; long long top = 0, nb = 0, pos = 0; unsigned seed = 0; } ep;
    Rng rng;
    Owned mem;
    uint8_t* frames = nullptr;
    int* act = nullptr; float *rew = nullptr, *done = nullptr;
    float *tree = nullptr, *pstate = nullptr, *val = nullptr, *ones = nullptr;
    bool top_stale = false;        // update_priority left tree[1 .. 2047] to the next sample's launch (a0_sumtree_set_from_loss(defer_top)); a0_rbuf_read brings them up to date
    long long* b_idx = nullptr; int *b_slot = nullptr, *b_act = nullptr; float *b_rew = nullptr, *b_done = nullptr, *b_prio = nullptr, *b_w = nullptr;
    // a0_rbuf_sample_block: up to 32 batches' buffers [32][B], allocated on first use
    long long* m_idx = nullptr; int *m_slot = nullptr, *m_act = nullptr; float *m_rew = nullptr, *m_done = nullptr, *m_prio = nullptr;
    long long head() const { return written > size ? written % size : 0; }
};

extern "C" int a0_rbuf_create(const a0_rbuf_desc* d, a0_rbuf** out) { return a0_rbuf_create_on(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out); }

extern "C" int a0_rbuf_create_on(const a0_rbuf_desc* d, uint8_t* frames, int* act, float* rew, float* done, float* tree, float* max_p, a0_rbuf** out) {
    A0_TRY
    if (!d || !out) return a0_fail(A0_EINVAL, "a0_rbuf_create: null argument");
    if (d->size < 2 || d->obs_bytes < 16 || (d->obs_bytes % 16) || d->B < 1 || d->B > d->size || d->prioritize < 0 || d->prioritize > 2 ||
        (d->prioritize && (!(d->alpha > 0.0) || d->total_steps < 1)) || (d->prioritize == 1 && d->B > 1024))
        return a0_fail(A0_EINVAL, "a0_rbuf_create: bad description (observation bytes a multiple of 16; prioritize 0 / 1 / 2; sum-tree batches of at most 1024)");
    a0_rbuf* R = new a0_rbuf();
    try {
        R->d = *d; R->size = d->size; R->obs_bytes = d->obs_bytes; R->row_bytes = 2LL * d->obs_bytes; R->B = d->B; R->prio = d->prioritize != 0; R->flat = d->prioritize == 2;
        R->rng.init(d->seed, 0);
        R->frames = frames ? frames : R->mem.alloc<uint8_t>(R->size * R->row_bytes, false);
        R->act = act ? act : R->mem.alloc<int>(R->size); R->rew = rew ? rew : R->mem.alloc<float>(R->size); R->done = done ? done : R->mem.alloc<float>(R->size);
        const int B = d->B;
        R->b_idx = R->mem.alloc<long long>(B); R->b_slot = R->mem.alloc<int>(B); R->b_act = R->mem.alloc<int>(B); R->b_rew = R->mem.alloc<float>(B);
        R->b_done = R->mem.alloc<float>(B); R->b_prio = R->mem.alloc<float>(B); R->b_w = R->mem.alloc<float>(B); R->ones = R->mem.alloc<float>(B);
        R->pstate = max_p ? max_p : R->mem.alloc<float>(1); R->val = R->mem.alloc<float>(4);
        if (!R->prio) {      // a0_rbuf_sample_block's buffers (uniform replay only): allocated here, so that no call after create allocates or synchronises
            R->m_idx = R->mem.alloc<long long>(32LL * B); R->m_slot = R->mem.alloc<int>(32LL * B); R->m_act = R->mem.alloc<int>(32LL * B); R->m_rew = R->mem.alloc<float>(32LL * B);
            R->m_done = R->mem.alloc<float>(32LL * B); R->m_prio = R->mem.alloc<float>(32LL * B);
        }
        hipLaunchKernelGGL(a0_fill_one_kernel, dim3((B + 255) / 256), dim3(256), 0, 0, R->ones, (long long)B, 1.0f);
        if (!max_p) hipLaunchKernelGGL(a0_fill_one_kernel, dim3(1), dim3(256), 0, 0, R->pstate, 1LL, 1.0f);           // max_p = 1 (replay.py:20)
        if (R->flat) {       // `tree` names the caller's priority vector [size] here (torch.ones(size), replay.py:19)
            R->prio_vec = tree ? tree : R->mem.alloc<float>(R->size, false);
            if (!tree) hipLaunchKernelGGL(a0_fill_one_kernel, dim3((unsigned)((R->size + 255) / 256)), dim3(256), 0, 0, R->prio_vec, R->size, 1.0f);
            R->psum = R->mem.alloc<float>(4); R->sum_scratch = R->mem.alloc<float>(256);
            R->beta_use = R->sched_cur = d->beta0;
            R->beta_inc = (1.0 - d->beta0) / (double)d->total_steps;
        } else if (R->prio) {
            while (R->cap2 < R->size) R->cap2 <<= 1;
            R->tree = tree ? tree : R->mem.alloc<float>(2 * R->cap2);
            R->beta_use = R->sched_cur = d->beta0;                                                         // LinearSchedule(beta0, 1, total_steps), utils.py:12-28
            R->beta_inc = (1.0 - d->beta0) / (double)d->total_steps;
        }
        A0_HIP_THROW(hipDeviceSynchronize());
    } catch (...) { delete R; throw; }
    *out = R;
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_rbuf_destroy(a0_rbuf* R) { delete R; return A0_OK; }
extern "C" long long a0_rbuf_len(const a0_rbuf* R) { return R ? R->top : 0; }
extern "C" long long a0_rbuf_write_cursor(const a0_rbuf* R) { return R ? R->written % R->size : 0; }
extern "C" int a0_rbuf_info(const a0_rbuf* R, long long* top, long long* written, double* beta) {
    if (!R) return a0_fail(A0_EINVAL, "a0_rbuf_info: null handle");
    if (top) *top = R->top;
    if (written) *written = R->written;
    if (beta) *beta = R->beta_use;
    return A0_OK;
}

extern "C" int a0_rbuf_buffers(a0_rbuf* R, uint8_t** frames, int** act, float** rew, float** done, float** tree, float** max_p) {
    if (!R) return a0_fail(A0_EINVAL, "a0_rbuf_buffers: null handle");
    if (frames) *frames = R->frames;
    if (act) *act = R->act;
    if (rew) *rew = R->rew;
    if (done) *done = R->done;
    if (tree) *tree = R->flat ? R->prio_vec : R->tree;       // flat mode: the priority vector [size]. (levels with < 2048 nodes may be stale after a0_rbuf_update_priority until the next a0_rbuf_sample / a0_rbuf_commit: use a0_rbuf_read for a consistent copy)
    if (max_p) *max_p = R->pstate;
    return A0_OK;
}

// copies of what the ring holds into caller buffers (any pointer may be NULL; device pointers): rows [0, rows) of frames / act / rew / done, the sum-tree, max_p
extern "C" int a0_rbuf_read(const a0_rbuf* R, long long rows, uint8_t* frames_out, int* act_out, float* rew_out, float* done_out, float* tree_out, float* max_p_out, void* stream) {
    A0_TRY
    if (!R || rows < 0 || rows > R->size) return a0_fail(A0_EINVAL, "a0_rbuf_read: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (frames_out) A0_HIP_THROW(hipMemcpyAsync(frames_out, R->frames, (size_t)(rows * R->row_bytes), hipMemcpyDeviceToDevice, st));
    if (act_out) A0_HIP_THROW(hipMemcpyAsync(act_out, R->act, (size_t)rows * 4, hipMemcpyDeviceToDevice, st));
    if (rew_out) A0_HIP_THROW(hipMemcpyAsync(rew_out, R->rew, (size_t)rows * 4, hipMemcpyDeviceToDevice, st));
    if (done_out) A0_HIP_THROW(hipMemcpyAsync(done_out, R->done, (size_t)rows * 4, hipMemcpyDeviceToDevice, st));
    if (tree_out && R->tree && R->top_stale) A0_CHECK(a0_sumtree_top_rebuild(R->tree, R->cap2, stream));      // (the flag stays: idempotent, and R is const here)
    if (tree_out && R->tree) A0_HIP_THROW(hipMemcpyAsync(tree_out, R->tree, (size_t)(2 * R->cap2) * 4, hipMemcpyDeviceToDevice, st));
    if (tree_out && R->flat) A0_HIP_THROW(hipMemcpyAsync(tree_out, R->prio_vec, (size_t)R->size * 4, hipMemcpyDeviceToDevice, st));      // flat mode: the priority vector [size]
    if (max_p_out) A0_HIP_THROW(hipMemcpyAsync(max_p_out, R->pstate, 4, hipMemcpyDeviceToDevice, st));
    return A0_OK;
    A0_CATCH
}

// ReplayDataset.extend (replay.py:45-53) for n transitions an actor has already written into the ring at the write cursor: counters, and for prioritized
// replay the new leaves at max_p^alpha (one ring range, one launch) and the beta schedule's step (the value BEFORE the increment is the one in use)
extern "C" int a0_rbuf_commit(a0_rbuf* R, long long n, void* stream) {
    A0_TRY
    if (!R || n < 1) return a0_fail(A0_EINVAL, "a0_rbuf_commit: bad argument");
    R->written += n;
    R->top = R->top + n < R->size ? R->top + n : R->size;
    if (R->flat) {
        // replay.py:51-52 as written: the roll's result is discarded and the n new priorities max_p^alpha land in the TAIL of the vector (quirk Q1)
        A0_CHECK(a0_priority_tail(R->prio_vec, R->size, n < R->size ? n : R->size, R->pstate, (float)R->d.alpha, stream));
        R->beta_use = R->sched_cur;
        const double nxt = R->sched_cur + R->beta_inc * (double)n;
        R->sched_cur = nxt < 1.0 ? nxt : 1.0;
    } else if (R->prio) {
        hipLaunchKernelGGL(a0_pow_scalar_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, R->pstate, R->d.alpha, R->val);
        const long long k = n < R->size ? n : R->size;
        A0_CHECK(a0_sumtree_set_range(R->tree, R->cap2, (R->written - k) % R->size, k, R->size, R->val, stream));
        R->top_stale = false;
        R->beta_use = R->sched_cur;                               // beta = beta_schedule(n): the schedule's value BEFORE it advances (utils.py:25-28)
        const double nxt = R->sched_cur + R->beta_inc * (double)n;
        R->sched_cur = nxt < 1.0 ? nxt : 1.0;
    }
    return A0_OK;
    A0_CATCH
}

// ReplayDataset.extend for a rollout that landed in ANOTHER ring (round 5; the launch schedule, launch.py:47-62: the learner books a finished rollout while the next
// one is in flight): rows [start_row, start_row + n) of `stage` — an a0_rbuf an asynchronous actor rolled out into — are copied to this ring's write cursor (at most two
// pieces: the ring may wrap), then committed like a0_rbuf_commit.  Stream-ordered on `stream`; the caller has made sure the rollout is complete.
extern "C" int a0_rbuf_extend_from(a0_rbuf* R, const a0_rbuf* S, long long start_row, long long n, void* stream) {
    A0_TRY
    if (!R || !S || n < 1 || start_row < 0 || start_row + n > S->size || n > R->size || S->row_bytes != R->row_bytes)
        return a0_fail(A0_EINVAL, "a0_rbuf_extend_from: rows [start_row, start_row + n) of a stage ring with the same row format, n <= ring size");
    hipStream_t st = (hipStream_t)stream;
    long long done = 0;
    while (done < n) {
        const long long c = (R->written + done) % R->size, k = std::min(n - done, R->size - c), a = start_row + done;
        A0_HIP_THROW(hipMemcpyAsync(R->frames + c * R->row_bytes, S->frames + a * S->row_bytes, (size_t)(k * R->row_bytes), hipMemcpyDeviceToDevice, st));
        A0_HIP_THROW(hipMemcpyAsync(R->act + c, S->act + a, (size_t)k * 4, hipMemcpyDeviceToDevice, st));
        A0_HIP_THROW(hipMemcpyAsync(R->rew + c, S->rew + a, (size_t)k * 4, hipMemcpyDeviceToDevice, st));
        A0_HIP_THROW(hipMemcpyAsync(R->done + c, S->done + a, (size_t)k * 4, hipMemcpyDeviceToDevice, st));
        done += k;
    }
    return a0_rbuf_commit(R, n, stream);
    A0_CATCH
}

// One batch (trainer.py:63-72 / 91-96): uniform — element pos * B + b of the epoch's permutation of range(top) (an epoch is opened when the previous one has
// no whole batch left BUT ONE: the reference's prefetcher never returns its last batch, utils.py:51-56) — or proportional from the sum-tree with importance
// weights.  The pointers in `out` are the handle's persistent batch buffers (valid until the next sample).
extern "C" int a0_rbuf_sample(a0_rbuf* R, a0_batch* out, void* stream) {
    A0_TRY
    a0_trace_scope range("sample");
    if (!R || !out) return a0_fail(A0_EINVAL, "a0_rbuf_sample: null argument");
    const int B = R->B;
    if (R->prio && !R->flat) {
        const unsigned long long off = R->rng.reserve(STREAM_SUMTREE, B);
        A0_CHECK(a0_sumtree_sample_batch(R->rng.seed, STREAM_SUMTREE, off, R->tree, R->cap2, B, R->top, R->size, (float)R->beta_use, R->act, R->rew, R->done, R->b_idx, R->b_slot,
                                         R->b_act, R->b_rew, R->b_done, R->b_prio, R->b_w, R->top_stale ? 1 : 0, stream));
        R->top_stale = false;
        *out = a0_batch{R->b_idx, R->b_slot, R->b_act, R->b_rew, R->b_done, R->b_prio, R->b_w};
        return A0_OK;
    }
    if (!R->ep.open || R->ep.pos + 1 >= R->ep.nb) {
        R->ep.top = R->top; R->ep.nb = (R->top + B - 1) / B; R->ep.pos = 0; R->ep.seed = R->rng.next_seed32(STREAM_PERM); R->ep.open = true;
        if (R->ep.nb < 2) { R->ep.open = false; return a0_fail(A0_EINVAL, "a0_rbuf_sample: the ring holds fewer than two batches (the reference's fetcher cannot return one either)"); }
    }
    const long long start = R->ep.pos * B;
    R->ep.pos += 1;
    A0_CHECK(a0_replay_sample_slots((unsigned long long)start, (unsigned long long)R->ep.top, R->ep.seed, R->top, R->head(), R->size, R->act, R->rew, R->done, R->flat ? R->prio_vec : nullptr, B,
                                    R->b_idx, R->b_slot, R->b_act, R->b_rew, R->b_done, R->b_prio, stream));
    if (R->flat) {       // trainer.py:91-94: probs = p / priority.sum() over the WHOLE capacity (quirk Q7), w = (top * probs)^-beta / max
        A0_CHECK(a0_sum_f32(R->prio_vec, R->size, R->sum_scratch, R->psum, stream));
        A0_CHECK(a0_is_weights(R->b_prio, B, R->psum, R->top, (float)R->beta_use, R->b_w, stream));
    }
    *out = a0_batch{R->b_idx, R->b_slot, R->b_act, R->b_rew, R->b_done, R->b_prio, R->flat ? R->b_w : R->ones};
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_rbuf_sample_block(a0_rbuf* R, int n, a0_batch* out, void* stream) {
    A0_TRY
    a0_trace_scope range("sample");
    if (!R || !out || n < 1 || n > 32) return a0_fail(A0_EINVAL, "a0_rbuf_sample_block: 1..32 batches");
    if (R->prio) return a0_fail(A0_EINVAL, "a0_rbuf_sample_block: prioritized batches depend on the update before them (a0_rbuf_sample)");
    const int B = R->B;
    unsigned long long start[32], n_perm[32];
    unsigned int seed[32];
    for (int g = 0; g < n; ++g) {        // the epoch bookkeeping of n consecutive a0_rbuf_sample calls (the ring does not change inside an update block)
        if (!R->ep.open || R->ep.pos + 1 >= R->ep.nb) {
            R->ep.top = R->top; R->ep.nb = (R->top + B - 1) / B; R->ep.pos = 0; R->ep.seed = R->rng.next_seed32(STREAM_PERM); R->ep.open = true;
            if (R->ep.nb < 2) { R->ep.open = false; return a0_fail(A0_EINVAL, "a0_rbuf_sample_block: the ring holds fewer than two batches"); }
        }
        start[g] = (unsigned long long)(R->ep.pos * B); n_perm[g] = (unsigned long long)R->ep.top; seed[g] = R->ep.seed;
        R->ep.pos += 1;
    }
    A0_CHECK(a0_replay_sample_slots_multi(n, start, n_perm, seed, R->top, R->head(), R->size, R->act, R->rew, R->done, B, R->m_idx, R->m_slot, R->m_act, R->m_rew, R->m_done,
                                          R->m_prio, stream));
    for (int g = 0; g < n; ++g) {
        const long long o = (long long)g * B;
        out[g] = a0_batch{R->m_idx + o, R->m_slot + o, R->m_act + o, R->m_rew + o, R->m_done + o, R->m_prio + o, R->ones};
    }
    return A0_OK;
    A0_CATCH
}

// ReplayDataset.update_priority (replay.py:55-59) with the last batch's indices: leaf = (loss + eps)^alpha, max_p = max(max_p, max loss); a no-op for uniform
// replay and when the learner skipped the update on a NaN (learner_state[3], agent.py:152-158 / trainer.py:103)
extern "C" int a0_rbuf_update_priority(a0_rbuf* R, const float* loss, const int* learner_state, void* stream) {
    A0_TRY
    if (!R || !loss) return a0_fail(A0_EINVAL, "a0_rbuf_update_priority: null argument");
    if (!R->prio) return A0_OK;
    if (R->flat) return a0_priority_update(R->prio_vec, R->b_idx, loss, R->B, (float)R->d.eps, (float)R->d.alpha, R->pstate, learner_state, stream);      // priority[ids] = (loss + eps)^alpha
    if (a0_sumtree_set_from_loss_ok(R->cap2)) {
        R->top_stale = true;
        return a0_sumtree_set_from_loss(R->tree, R->cap2, R->b_idx, loss, R->B, (float)R->d.eps, (float)R->d.alpha, R->pstate, learner_state, 1, stream);
    }
    A0_CHECK(a0_priority_from_loss(loss, R->B, (float)R->d.eps, (float)R->d.alpha, R->b_prio, R->pstate, learner_state, stream));
    return a0_sumtree_set(R->tree, R->cap2, R->b_idx, R->b_prio, R->B, learner_state, stream);
    A0_CATCH
}

// ================================================================================================ actor
struct a0_actor {
    a0_actor_desc d;
    int E = 0, T = 0, n = 1, K = 2, cur = 0, feat = 3136, obs_bytes = 4 * 84 * 84;
    unsigned g = 0;
    long long steps = 0;
    Rng rng;
    Owned mem;
    std::vector<uint8_t*> obs;
    float *ep_ret = nullptr, *qmax_all = nullptr, *stat_mask = nullptr, *stat_ret = nullptr, *qs = nullptr, *ring_rew = nullptr, *ring_done = nullptr, *act3 = nullptr, *scratch = nullptr;
    int *action = nullptr, *ring_act = nullptr;
    std::vector<float> h_mask, h_ret;
    float *p_mask = nullptr, *p_ret = nullptr, *p_qs = nullptr;       // page-locked: a0_actor_collect_begin / _end
    hipEvent_t stats_ev = nullptr;
    ~a0_actor() {
        if (p_mask) (void)hipHostFree(p_mask);
        if (p_ret) (void)hipHostFree(p_ret);
        if (p_qs) (void)hipHostFree(p_qs);
        if (stats_ev) (void)hipEventDestroy(stats_ev);
    }
    // distributional heads (c51): fc1 output, the head GEMM's split-K slabs, fc1's split-K scratch — sized at the first rollout from the learner's shapes
    float *h = nullptr, *head_slabs = nullptr, *fwd_scratch = nullptr;
    int dist_Npad = 0;
    // quantile heads (iqn): E * K rows — tau draws, cosine features, embedding x features, fc1 output
    float *q_taus = nullptr, *q_cosx = nullptr, *q_x = nullptr;
    float *f_logits = nullptr, *f_tau_all = nullptr;        // fqf: fraction logits [E][32], taus [E][F + 1] (q_taus holds the tau-hats)
    bool bound = false;
    // round 5, the launch schedule (launch.py:34-36,58-62): the actor's OWN copy of the network — packed parameters, the fused kernels' weight copies, and under NoisyNet
    // its own noise vectors and composed weights — refreshed by a0_actor_snapshot when a rollout is issued.  Without it (main schedule) the actor acts with the learner's
    // online network, NoisyNet buffers included: the reference's train actor SHARES the learner's module there (trainer.py:41-44).
    float *own_flat = nullptr, *own_wt = nullptr, *own_eff = nullptr, *own_noise = nullptr;
    unsigned int* w_planes = nullptr;     // quantile actors (round 6): fc1's weights as bf16 term planes (a0_split_planes), refreshed when a rollout starts and after every NoisyNet compose
};

// the network an actor acts with: the learner's online network, or the actor's own snapshot of it
struct a0_actor_net {
    const float* flat; const float* wt; float* eff; float* noise;
    const a0_learner* L;
    const float* Wf() const { return L->d.noisy ? eff + L->eff_fc1.w() : flat + L->fc1.w(); }
    const float* bf() const { return L->d.noisy ? eff + L->eff_fc1.b() : flat + L->fc1.b(); }
    const float* Wh() const { return L->d.noisy ? eff + L->eff_head.w() : flat + L->head.w(); }
    const float* bh() const { return L->d.noisy ? eff + L->eff_head.b() : flat + L->head.b(); }
};
static a0_actor_net a0_actor_view(const a0_actor* a, a0_learner* L) {
    if (a->own_flat) return a0_actor_net{a->own_flat, a->own_wt, a->own_eff, a->own_noise, L};
    return a0_actor_net{L->online, L->wt_on, L->eff_on, L->noise, L};
}

// Every workspace a rollout with `learner` needs — the distributional / quantile heads' buffers, sized by the learner's head — and, own_network != 0, the actor's own
// copy of the network (a0_actor_snapshot fills it).  Call it once at set-up: a0_actor_rollout then neither allocates nor synchronises (an unbound actor is bound at its
// first rollout, which does both).
extern "C" int a0_actor_bind(a0_actor* a, const a0_learner* L, int own_network) {
    A0_TRY
    if (!a || !L) return a0_fail(A0_EINVAL, "a0_actor_bind: null argument");
    if (L->d.A != a->d.A || (L->d.dueling != 0) != (a->d.dueling != 0)) return a0_fail(A0_EINVAL, "a0_actor_bind: actor and learner were created for different heads");
    const int E = a->E;
    const bool dist = L->d.algo == A0_ALGO_C51 || L->d.algo == A0_ALGO_QR, fqf = L->d.algo == A0_ALGO_FQF, quant = L->d.algo == A0_ALGO_IQN || fqf;
    if (a->bound && a->dist_Npad != L->Npad) return a0_fail(A0_EINVAL, "a0_actor_bind: this actor was sized for another head");
    if (!a->bound) {
        const int nt = fqf ? L->F : (quant ? L->d.iqn_K : 1);
        if (quant) {
            const long long R = (long long)E * nt;
            if (a0_dense_fwd_scratch((int)R, L->feat, 64) != 0) return a0_fail(A0_EINVAL, "a0_actor_bind: E * K rows too few for the embedding kernel this path takes");
            a->h = a->mem.alloc<float>(R * 512);
            a->head_slabs = a->mem.alloc<float>((long long)a0_dense_fwd_partial_slabs((int)R, L->Npad, 512) * R * L->Npad);
            long long sc = a0_dense_fwd_scratch((int)R, 512, L->feat);
            if (fqf && a0_dense_fwd_scratch(E, 32, L->feat) > sc) sc = a0_dense_fwd_scratch(E, 32, L->feat);
            a->fwd_scratch = a->mem.alloc<float>(sc > 4 ? sc : 4);
            a->q_taus = a->mem.alloc<float>(ceil_to(R, 4)); a->q_cosx = a->mem.alloc<float>(R * 64); a->q_x = a->mem.alloc<float>(R * L->feat);
            if (fqf) { a->f_logits = a->mem.alloc<float>((long long)E * 32); a->f_tau_all = a->mem.alloc<float>((long long)E * (nt + 1)); }
            static const bool no_planes = getenv("A0_NO_WPLANES") != nullptr;      // tuning aid (same bits)
            if (!no_planes && a0_dense_fwd_wplanes_ok((int)R, 512, L->feat)) a->w_planes = a->mem.alloc<unsigned int>(a0_weight_planes_words(512, L->feat), false);
        } else if (dist) {
            a->h = a->mem.alloc<float>((long long)E * 512);
            a->head_slabs = a->mem.alloc<float>((long long)a0_dense_fwd_partial_slabs(E, L->Npad, 512) * E * L->Npad);
            const long long sc = a0_dense_fwd_scratch(E, 512, L->feat);
            a->fwd_scratch = a->mem.alloc<float>(sc > 4 ? sc : 4);
        }
        a->dist_Npad = L->Npad;
        a->bound = true;
    }
    if (own_network && !a->own_flat) {
        a->own_flat = a->mem.alloc<float>(L->n_pad); a->own_wt = a->mem.alloc<float>(L->wt_floats);
        if (L->d.noisy) { a->own_eff = a->mem.alloc<float>(L->n_eff); a->own_noise = a->mem.alloc<float>(L->noise_len); }
    }
    A0_HIP_THROW(hipDeviceSynchronize());
    return A0_OK;
    A0_CATCH
}

// actor.futures.sample(eps, state_dict) (launch.py:34-36,58-62): the actor's own network := the learner's online network as it is at this point of `stream` — packed
// parameters, weight copies, NoisyNet noise and composed weights (DeviceNet.copy_from).  Needs a0_actor_bind(actor, learner, 1).
extern "C" int a0_actor_snapshot(a0_actor* a, const a0_learner* L, void* stream) {
    A0_TRY
    if (!a || !L || !a->own_flat) return a0_fail(A0_EINVAL, "a0_actor_snapshot: an actor bound with its own network (a0_actor_bind(actor, learner, 1))");
    hipStream_t st = (hipStream_t)stream;
    A0_HIP_THROW(hipMemcpyAsync(a->own_flat, L->online, (size_t)L->n_pad * 4, hipMemcpyDeviceToDevice, st));
    A0_HIP_THROW(hipMemcpyAsync(a->own_wt, L->wt_on, (size_t)L->wt_floats * 4, hipMemcpyDeviceToDevice, st));
    if (L->d.noisy) {
        A0_HIP_THROW(hipMemcpyAsync(a->own_eff, L->eff_on, (size_t)L->n_eff * 4, hipMemcpyDeviceToDevice, st));
        A0_HIP_THROW(hipMemcpyAsync(a->own_noise, L->noise, (size_t)L->noise_len * 4, hipMemcpyDeviceToDevice, st));
    }
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_actor_create(const a0_actor_desc* d, a0_actor** out) {
    A0_TRY
    if (!d || !out) return a0_fail(A0_EINVAL, "a0_actor_create: null argument");
    if (d->E < 1 || d->T < 1 || d->A < 1 || d->n_step < 1 || d->reset_noise_freq < 0 || !(d->discount > 0.0) || d->env_task < A0_ENV_TASK_STREAM || d->env_task > A0_ENV_TASK_CHASE || (d->env_task == A0_ENV_TASK_CHASE && d->A < 4))
        return a0_fail(A0_EINVAL, "a0_actor_create: bad description");
    a0_actor* a = new a0_actor();
    try {
        a->d = *d; a->E = d->E; a->T = d->T; a->n = d->n_step;
        a->rng.init(d->seed, d->rank);
        // observation buffers: two for 1-step transitions; for n-step the last n observations stay addressable (n + 1 buffers, or the next divisor of the rollout
        // length so that the buffer pattern of a rollout repeats: agent0_amd/deepq/agent.py, DeviceSynthVecEnv.set_history)
        a->K = 2;
        if (a->n > 1) {
            a->K = a->n + 1;
            for (int r = a->n + 1; r < 2 * a->n + 3; ++r) if (a->T % r == 0) { a->K = r; break; }
        }
        const long long E = a->E, T = a->T;
        for (int i = 0; i < a->K; ++i) a->obs.push_back(a->mem.alloc<uint8_t>(E * a->obs_bytes));
        a->ep_ret = a->mem.alloc<float>(E); a->qmax_all = a->mem.alloc<float>(T * E); a->stat_mask = a->mem.alloc<float>(T * E); a->stat_ret = a->mem.alloc<float>(T * E);
        a->qs = a->mem.alloc<float>(T); a->action = a->mem.alloc<int>(E);
        a->ring_act = a->mem.alloc<int>((long long)a->n * E); a->ring_rew = a->mem.alloc<float>((long long)a->n * E); a->ring_done = a->mem.alloc<float>((long long)a->n * E);
        a->act3 = a->mem.alloc<float>(E * a->feat); a->scratch = a->mem.alloc<float>(a0_actor_qhead_scratch(a->E, a->feat));
        a->h_mask.resize((size_t)(T * E)); a->h_ret.resize((size_t)(T * E));
        if (a0_env_synth_reset_task(d->seed, d->rank, a->E, a->obs[0], a->ep_ret, d->env_task, nullptr) != A0_OK) { delete a; return A0_EINVAL; }      // Actor.__init__: self.obs = envs.reset()
        A0_HIP_THROW(hipDeviceSynchronize());
    } catch (...) { delete a; throw; }
    *out = a;
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_actor_destroy(a0_actor* a) { delete a; return A0_OK; }

// Actor.sample (agent.py:44-90) with the learner's online network: T steps of [encoder, fc1 GEMM, tail + env step + n-step bookkeeping + replay row], the
// rows written straight into the ring at its write cursor (call a0_rbuf_commit(replay, T * E) afterwards: ReplayDataset.extend), then the per-step mean max-Q.
// Asynchronous like everything else; a0_actor_collect waits and returns the statistics.
// the online network's effective weights from its parameters and the noise vectors it currently holds (DeviceNet.compose_noise)
static int a0_actor_compose(const a0_actor_net& V, void* stream) {
    const a0_learner* L = V.L;
    const float *mu[3], *sg[3], *nin[3], *nw[3], *nb[3];
    float* eff[3];
    int N[3], K[3], r0[3], r1[3];
    for (int k = 0; k < L->n_mods; ++k) {
        const a0_noise_mod& m = L->mods[k];
        const Blk &bm = m.block ? L->head : L->fc1, &bs = m.block ? L->head_sigma : L->fc1_sigma, &be = m.block ? L->eff_head : L->eff_fc1;
        mu[k] = V.flat + bm.off; sg[k] = V.flat + bs.off; eff[k] = V.eff + be.off; N[k] = bm.N; K[k] = bm.K; r0[k] = m.r0; r1[k] = m.r1;
        nin[k] = V.noise + m.off_in; nw[k] = V.noise + m.off_w; nb[k] = V.noise + m.off_b;
    }
    return a0_noisy_multi(0, L->n_mods, mu, sg, eff, N, K, r0, r1, nin, nw, nb, stream);
}

extern "C" int a0_actor_rollout(a0_actor* a, a0_learner* L, a0_rbuf* R, float epsilon, void* stream) {
    A0_TRY
    a0_trace_scope range("rollout");
    if (!a || !L || !R) return a0_fail(A0_EINVAL, "a0_actor_rollout: null argument");
    if (L->d.A != a->d.A || (L->d.dueling != 0) != (a->d.dueling != 0) || R->obs_bytes != a->obs_bytes || R->size < a->E)
        return a0_fail(A0_EINVAL, "a0_actor_rollout: actor, learner and replay were created for different shapes");
    const int E = a->E, A = a->d.A;
    const bool dist = L->d.algo == A0_ALGO_C51 || L->d.algo == A0_ALGO_QR, fqf = L->d.algo == A0_ALGO_FQF, quant = L->d.algo == A0_ALGO_IQN || fqf;
    const int freq = a->d.reset_noise_freq > 0 ? a->d.reset_noise_freq : 4;
    const int nt = fqf ? L->F : (quant ? L->d.iqn_K : 1);  // fractions per env and step (agent.py:25-39 with IQNHead.qval / FQFHead.qval, model.py:253-257,280-284)
    if (!a->bound || a->dist_Npad != L->Npad) A0_CHECK(a0_actor_bind(a, L, 0));      // (a host that wants no allocation after set-up binds there)
    const a0_actor_net V = a0_actor_view(a, L);
    const long long start = R->written % R->size;
    a0_encoder_weights w = L->enc(V.flat);
    // NoisyLinear.forward composes mu + sigma * eps with the parameters as they are NOW (model.py:54-62): a rollout that does not start on a noise reset
    // recomposes the copies once (agent0_amd/deepq/agent.py Actor._rollout)
    if (L->d.noisy && a->steps % freq != 0) A0_CHECK(a0_actor_compose(V, stream));
    // quantile actors (round 6): fc1's weight operand as term planes for the rollout's 80 GEMMs of E * K rows — split once here (and after every noise reset below)
    // instead of in every tile of every GEMM; the same exact terms, the same bits
    const bool planes = quant && a->w_planes && a0_dense_fwd_wplanes_ok(E * nt, 512, L->feat);
    if (planes && !(L->d.noisy && a->steps % freq == 0)) A0_CHECK(a0_split_planes(V.Wf(), a->w_planes, 512, L->feat, stream));
    // scalar heads (Actor._rollout): the tail + env-step launch of step t also encodes the env's new observation (a0_actor_qhead_env_step_enc), so that step t + 1
    // starts with its features in place; the convolution weights do not change inside a rollout, the last step has no next one
    static const bool step_enc_on = getenv("A0_NO_X9") == nullptr && (getenv("A0_STEP_ENC") == nullptr || atoi(getenv("A0_STEP_ENC")) != 0);
    // (not for an actor with its own network — the launch schedule: its rollout runs beside the update block, the critical path there, and a workgroup that holds a CU's
    // LDS from the tail to the end of the encoder takes more from the block than the saved boundary gives: 9.43 -> 9.75 ms)
    const bool step_enc = step_enc_on && !a->own_flat;
    bool feat_ready = false;
    for (int t = 0; t < a->T; ++t) {
        if (L->d.noisy && a->steps % freq == 0) {      // agent.py:52-53: self.model.reset_noise() every reset_noise_freq steps, from the ACTOR's stream
            A0_CHECK(a0_rng_normal(a->rng.seed, 4 /* STREAM_NOISE */, a->rng.reserve(4, L->noise_len), 0.1f, V.noise, L->noise_len, stream));
            A0_CHECK(a0_actor_compose(V, stream));
            if (planes) A0_CHECK(a0_split_planes(V.Wf(), a->w_planes, 512, L->feat, stream));
        }
        const uint8_t* cur_obs = a->obs[a->cur];
        a0_frames_arg f{cur_obs, nullptr, (long long)a->obs_bytes, 0};
        if (!feat_ready) A0_CHECK(a0_net_encoder_fwd_fused(L->C, L->H, L->W, V.wt, &w, &f, E, nullptr, nullptr, a->act3, stream));
        feat_ready = false;
        const long long back = (a->steps + 1 < a->n ? a->steps + 1 : a->n) - 1;                 // first observation of the emitted n-step transition
        const uint8_t* obs0 = a->obs[((a->cur - back) % a->K + a->K) % a->K];
        if (quant) {
            // Actor._quant_tail_args + act_step_commit(kind = "quantile"): the step's K fractions per env from the actor's Philox stream 3, cosine features, the embedding
            // times the features in the embedding GEMM's epilogue, fc1, the head GEMM's slabs, then ONE launch for slab sum + bias, dueling, the mean over the fractions,
            // first-max argmax, epsilon-greedy, env step, n-step bookkeeping and the replay row
            const int rows = E * nt;
            const float* on = V.flat;
            if (fqf) {      // FQFHead.prop_taus (model.py:268-278): the fraction net on the step's features; no draws
                A0_CHECK(a0_dense_fwd(a->act3, L->feat, on + L->frac.w(), on + L->frac.b(), a->f_logits, E, 32, L->feat, 0, a->fwd_scratch, stream));
                A0_CHECK(a0_fqf_taus_cos(a->f_logits, 32, a->f_tau_all, a->q_taus, a->q_cosx, 64, E, nt, stream));       // (round 6) + the cosine features of the tau_hats
            } else {       // (round 6) the draw and its cosine features in one launch
                A0_CHECK(a0_tau_cos_features(a->rng.seed, 3 /* STREAM_TAUS */, a->rng.reserve(3, rows), nullptr, 0, a->q_taus, a->q_cosx, rows, 64, stream));
            }
            A0_CHECK(a0_dense_fwd_mul(a->q_cosx, 64, on + L->cos.w(), on + L->cos.b(), a->act3, nt, a->q_x, rows, L->feat, 64, 1, stream));
            if (planes) A0_CHECK(a0_dense_fwd_wplanes(a->q_x, L->feat, a->w_planes, V.bf(), a->h, rows, 512, L->feat, 1, stream));
            else A0_CHECK(a0_dense_fwd(a->q_x, L->feat, V.Wf(), V.bf(), a->h, rows, 512, L->feat, 1, a->fwd_scratch, stream));
            const int ns = a0_dense_fwd_partial_slabs(rows, L->Npad, 512);
            A0_CHECK(a0_dense_fwd_partial(a->h, 512, V.Wh(), rows, L->Npad, 512, a->head_slabs, stream));
            const unsigned long long oa = a->rng.reserve(STREAM_EGREEDY_A, E), ou = a->rng.reserve(STREAM_EGREEDY_U, E);
            const int nx = (a->cur + 1) % a->K;
            a->g += 1;
            if (step_enc && t + 1 < a->T) {      // (round 6) the tail's workgroups go on to encode their env's new observation: the next step starts with its features
                A0_CHECK(a0_actor_quantile_tail_env_step_enc(a->head_slabs, (long long)rows * L->Npad, ns, V.bh(), L->Npad, A, nt, a->d.dueling ? 1 : 0, fqf ? 3 : 1, fqf ? a->f_tau_all : nullptr, E, a->rng.seed,
                                                             STREAM_EGREEDY_A, STREAM_EGREEDY_U, oa, ou, epsilon, nullptr, nullptr, a->action, a->qmax_all + (long long)t * E, a->d.seed,
                                                             a->d.rank, a->g, cur_obs, a->obs[nx], a->ep_ret, a->stat_mask + (long long)t * E, a->stat_ret + (long long)t * E, a->n, a->steps,
                                                             a->d.discount, a->ring_act, a->ring_rew, a->ring_done, obs0, R->frames, R->size, (start + (long long)t * E) % R->size, R->act,
                                                             R->rew, R->done, a->d.env_task, V.wt, &w, a->act3, stream));
                feat_ready = true;
            } else
            A0_CHECK(a0_actor_quantile_tail_env_step(a->head_slabs, (long long)rows * L->Npad, ns, V.bh(), L->Npad, A, nt, a->d.dueling ? 1 : 0, fqf ? 3 : 1, fqf ? a->f_tau_all : nullptr, E, a->rng.seed,
                                                     STREAM_EGREEDY_A, STREAM_EGREEDY_U, oa, ou, epsilon, nullptr, nullptr, a->action, a->qmax_all + (long long)t * E, a->d.seed,
                                                     a->d.rank, a->g, cur_obs, a->obs[nx], a->ep_ret, a->stat_mask + (long long)t * E, a->stat_ret + (long long)t * E, a->n, a->steps,
                                                     a->d.discount, a->ring_act, a->ring_rew, a->ring_done, obs0, R->frames, R->size, (start + (long long)t * E) % R->size, R->act,
                                                     R->rew, R->done, a->d.env_task, stream));
            a->cur = nx;
            a->steps += 1;
            continue;
        }
        if (dist) {
            // fc1, the head GEMM's slabs, then ONE launch: slab sum + bias, dueling, expectation over the support, first-max argmax, epsilon-greedy, env step,
            // n-step bookkeeping and the replay row (Actor._dist_tail_args + act_step_commit(kind = "dist"))
            A0_CHECK(a0_dense_fwd(a->act3, L->feat, V.Wf(), V.bf(), a->h, E, 512, L->feat, 1, a->fwd_scratch, stream));
            const int ns = a0_dense_fwd_partial_slabs(E, L->Npad, 512);
            A0_CHECK(a0_dense_fwd_partial(a->h, 512, V.Wh(), E, L->Npad, 512, a->head_slabs, stream));
            const unsigned long long oa = a->rng.reserve(STREAM_EGREEDY_A, E), ou = a->rng.reserve(STREAM_EGREEDY_U, E);
            const int nx = (a->cur + 1) % a->K;
            a->g += 1;
            if (4LL * ((long long)A * L->T + L->T) * 4 > 160 * 1024) return a0_fail(A0_EINVAL, "a0_actor_rollout: head too wide for the distributional tail kernel");
            if (step_enc && t + 1 < a->T) {
                A0_CHECK(a0_actor_dist_tail_env_step_enc(a->head_slabs, (long long)E * L->Npad, ns, V.bh(), L->Npad, A, L->T, a->d.dueling ? 1 : 0, L->d.algo == A0_ALGO_C51 ? 2 : 1,
                                                         L->d.algo == A0_ALGO_C51 ? L->atoms : nullptr, E, a->rng.seed,
                                                         STREAM_EGREEDY_A, STREAM_EGREEDY_U, oa, ou, epsilon, nullptr, nullptr, a->action, a->qmax_all + (long long)t * E, a->d.seed, a->d.rank,
                                                         a->g, cur_obs, a->obs[nx], a->ep_ret, a->stat_mask + (long long)t * E, a->stat_ret + (long long)t * E, a->n, a->steps, a->d.discount,
                                                         a->ring_act, a->ring_rew, a->ring_done, obs0, R->frames, R->size, (start + (long long)t * E) % R->size, R->act, R->rew, R->done,
                                                         a->d.env_task, V.wt, &w, a->act3, stream));
                feat_ready = true;
            } else
            A0_CHECK(a0_actor_dist_tail_env_step(a->head_slabs, (long long)E * L->Npad, ns, V.bh(), L->Npad, A, L->T, a->d.dueling ? 1 : 0, L->d.algo == A0_ALGO_C51 ? 2 : 1,
                                                 L->d.algo == A0_ALGO_C51 ? L->atoms : nullptr, E, a->rng.seed,
                                                 STREAM_EGREEDY_A, STREAM_EGREEDY_U, oa, ou, epsilon, nullptr, nullptr, a->action, a->qmax_all + (long long)t * E, a->d.seed, a->d.rank,
                                                 a->g, cur_obs, a->obs[nx], a->ep_ret, a->stat_mask + (long long)t * E, a->stat_ret + (long long)t * E, a->n, a->steps, a->d.discount,
                                                 a->ring_act, a->ring_rew, a->ring_done, obs0, R->frames, R->size, (start + (long long)t * E) % R->size, R->act, R->rew, R->done,
                                                 a->d.env_task, stream));
            a->cur = nx;
            a->steps += 1;
            continue;
        }
        if (A + (a->d.dueling ? 1 : 0) > 24) return a0_fail(A0_EINVAL, "a0_actor_rollout: scalar heads with A + dueling <= 24 actions");
        const unsigned long long off_a = a->rng.reserve(STREAM_EGREEDY_A, E), off_u = a->rng.reserve(STREAM_EGREEDY_U, E);
        const int nxt = (a->cur + 1) % a->K;
        a->g += 1;
        if (step_enc && t + 1 < a->T) {
            A0_CHECK(a0_actor_qhead_env_step_enc(a->act3, E, a->feat, V.Wf(), V.bf(), V.Wh(), V.bh(), A, a->d.dueling ? 1 : 0,
                                                 a->scratch, a->rng.seed, STREAM_EGREEDY_A, STREAM_EGREEDY_U, off_a, off_u, epsilon, nullptr, nullptr, a->action, a->qmax_all + (long long)t * E,
                                                 a->d.seed, a->d.rank, a->g, cur_obs, a->obs[nxt], a->ep_ret, a->stat_mask + (long long)t * E, a->stat_ret + (long long)t * E, a->n, a->steps,
                                                 a->d.discount, a->ring_act, a->ring_rew, a->ring_done, obs0, R->frames, R->size, (start + (long long)t * E) % R->size, R->act, R->rew, R->done,
                                                 a->d.env_task, V.wt, &w, a->act3, stream));
            feat_ready = true;
        } else
        A0_CHECK(a0_actor_qhead_env_step(a->act3, E, a->feat, V.Wf(), V.bf(), V.Wh(), V.bh(), A, a->d.dueling ? 1 : 0,
                                         a->scratch, a->rng.seed, STREAM_EGREEDY_A, STREAM_EGREEDY_U, off_a, off_u, epsilon, nullptr, nullptr, a->action, a->qmax_all + (long long)t * E,
                                         a->d.seed, a->d.rank, a->g, cur_obs, a->obs[nxt], a->ep_ret, a->stat_mask + (long long)t * E, a->stat_ret + (long long)t * E, a->n, a->steps,
                                         a->d.discount, a->ring_act, a->ring_rew, a->ring_done, obs0, R->frames, R->size, (start + (long long)t * E) % R->size, R->act, R->rew, R->done,
                                         a->d.env_task, stream));
        a->cur = nxt;
        a->steps += 1;
    }
    return a0_mean_rows(a->qmax_all, a->T, E, a->qs, stream);
    A0_CATCH
}

// Waits for the stream, then hands back what Actor.sample returns besides the transitions (agent.py:85-90): the per-step mean max-Q (qs_host [T]) and the
// returns of the episodes that finished during the rollout, in the reference's order (step-major, env-major); *n_returns = how many there were (at most
// max_returns are stored).  The one device -> host read of a rollout.
extern "C" int a0_actor_collect(a0_actor* a, float* qs_host, float* returns_host, int max_returns, int* n_returns, void* stream) {
    A0_TRY
    if (!a) return a0_fail(A0_EINVAL, "a0_actor_collect: null handle");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)a->T * a->E;
    if (qs_host) A0_HIP_THROW(hipMemcpyAsync(qs_host, a->qs, (size_t)a->T * 4, hipMemcpyDeviceToHost, st));
    A0_HIP_THROW(hipMemcpyAsync(a->h_mask.data(), a->stat_mask, n * 4, hipMemcpyDeviceToHost, st));
    A0_HIP_THROW(hipMemcpyAsync(a->h_ret.data(), a->stat_ret, n * 4, hipMemcpyDeviceToHost, st));
    A0_HIP_THROW(hipStreamSynchronize(st));
    int k = 0;
    for (size_t i = 0; i < n; ++i)
        if (a->h_mask[i] != 0.f) { if (returns_host && k < max_returns) returns_host[k] = a->h_ret[i]; ++k; }
    if (n_returns) *n_returns = k;
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_actor_collect_begin(a0_actor* a, void* stream) {
    A0_TRY
    if (!a) return a0_fail(A0_EINVAL, "a0_actor_collect_begin: null handle");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)a->T * a->E;
    if (!a->p_mask) {
        A0_HIP_THROW(hipHostMalloc((void**)&a->p_mask, n * 4, hipHostMallocDefault));
        A0_HIP_THROW(hipHostMalloc((void**)&a->p_ret, n * 4, hipHostMallocDefault));
        A0_HIP_THROW(hipHostMalloc((void**)&a->p_qs, (size_t)a->T * 4, hipHostMallocDefault));
        A0_HIP_THROW(hipEventCreateWithFlags(&a->stats_ev, hipEventDisableTiming));
    }
    A0_HIP_THROW(hipMemcpyAsync(a->p_qs, a->qs, (size_t)a->T * 4, hipMemcpyDeviceToHost, st));
    A0_HIP_THROW(hipMemcpyAsync(a->p_mask, a->stat_mask, n * 4, hipMemcpyDeviceToHost, st));
    A0_HIP_THROW(hipMemcpyAsync(a->p_ret, a->stat_ret, n * 4, hipMemcpyDeviceToHost, st));
    A0_HIP_THROW(hipEventRecord(a->stats_ev, st));
    return A0_OK;
    A0_CATCH
}

extern "C" int a0_actor_collect_end(a0_actor* a, float* qs_host, float* returns_host, int max_returns, int* n_returns) {
    A0_TRY
    if (!a || !a->stats_ev) return a0_fail(A0_EINVAL, "a0_actor_collect_end: no a0_actor_collect_begin before it");
    A0_HIP_THROW(hipEventSynchronize(a->stats_ev));
    const size_t n = (size_t)a->T * a->E;
    if (qs_host) for (int t = 0; t < a->T; ++t) qs_host[t] = a->p_qs[t];
    int k = 0;
    for (size_t i = 0; i < n; ++i)
        if (a->p_mask[i] != 0.f) { if (returns_host && k < max_returns) returns_host[k] = a->p_ret[i]; ++k; }
    if (n_returns) *n_returns = k;
    return A0_OK;
    A0_CATCH
}
