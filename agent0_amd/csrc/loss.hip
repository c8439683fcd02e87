// Head epilogues and TD losses: dueling combine, greedy-action selection, DQN Huber, C51 projection +
// cross-entropy, quantile-Huber (QR / IQN / FQF) — forward value and gradient w.r.t. the head output in one pass.
// Restates reference agent0/deepq/agent.py:110-114, 173-190, 219-269, 273-293, 297-327 and the dueling / qval
// arithmetic of agent0/deepq/model.py:123-131, 163-177, 190-192, 219-233, 253-257, 280-284.
#include "a0_internal.h"
#include "actor_tail.h"      // a0_wave_sum / a0_wave_max, the actor tails' device functions

// ------------------------------------------------------------------------------------------------ dueling combine
// raw [R][ld]: columns [0, A*T) advantage / plain q (action-major), [A*T, A*T+T) value stream when dueling.
__global__ void a0_dueling_fwd_kernel(const float* __restrict__ raw, int ld, float* __restrict__ q, int R, int A, int T, int dueling) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)R * T) return;
    const int r = (int)(i / T), t = (int)(i % T);
    const float* x = raw + (long long)r * ld;
    float* o = q + (long long)r * A * T;
    if (!dueling) {
        for (int a = 0; a < A; ++a) o[a * T + t] = x[a * T + t];
        return;
    }
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += x[a * T + t];
    const float mean = s / (float)A;
    const float v = x[A * T + t];
    for (int a = 0; a < A; ++a) o[a * T + t] = v + (x[a * T + t] - mean);
}

__global__ void a0_dueling_bwd_kernel(const float* __restrict__ dq, float* __restrict__ draw, int ld, int R, int A, int T, int dueling) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)R * ld) return;
    const int r = (int)(i / ld), c = (int)(i % ld);
    const float* g = dq + (long long)r * A * T;
    float out = 0.f;
    if (c < A * T) {
        out = g[c];
        if (dueling) {
            const int t = c % T;
            float s = 0.f;
            for (int a = 0; a < A; ++a) s += g[a * T + t];
            out -= s / (float)A;
        }
    } else if (dueling && c < A * T + T) {
        const int t = c - A * T;
        float s = 0.f;
        for (int a = 0; a < A; ++a) s += g[a * T + t];
        out = s;
    }
    draw[i] = out;
}

extern "C" int a0_dueling_fwd(const float* raw, int ld, float* q, int R, int A, int T, int dueling, void* stream) {
    if (!raw || !q || R < 1 || A < 1 || T < 1 || ld < A * T + (dueling ? T : 0)) return a0_fail(A0_EINVAL, "a0_dueling_fwd: bad argument");
    long long n = (long long)R * T;
    hipLaunchKernelGGL(a0_dueling_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, ld, q, R, A, T, dueling);
    return a0_fail_hip((int)hipGetLastError(), "a0_dueling_fwd");
}

extern "C" int a0_dueling_bwd(const float* dq, float* draw, int ld, int R, int A, int T, int dueling, void* stream) {
    if (!dq || !draw || R < 1 || A < 1 || T < 1 || ld < A * T + (dueling ? T : 0)) return a0_fail(A0_EINVAL, "a0_dueling_bwd: bad argument");
    long long n = (long long)R * ld;
    hipLaunchKernelGGL(a0_dueling_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dq, draw, ld, R, A, T, dueling);
    return a0_fail_hip((int)hipGetLastError(), "a0_dueling_bwd");
}

// ------------------------------------------------------------------------------------------------ action values + argmax
// One wave per sample.  x(b,a,t) = x[b*sb + a*sa + t*st].  mode: 0 identity (T=1), 1 mean over t (QR atoms / IQN taus),
// 2 C51 expectation sum_t softmax_t * atoms[t], 3 FQF sum_t (tau[b][t+1]-tau[b][t]) * x.
__global__ __launch_bounds__(64) void a0_select_action_kernel(const float* __restrict__ x, long long sb, long long sa, long long st,
                                                               int B, int A, int T, int mode, const float* __restrict__ aux,
                                                               int* __restrict__ a_star, float* __restrict__ qsel, float* __restrict__ qmax) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    float best = 0.f;
    int besta = 0;
    for (int a = 0; a < A; ++a) {
        const float* p = x + (long long)b * sb + (long long)a * sa;
        float v;
        if (mode == 0) {
            v = p[0];
        } else if (mode == 1) {
            float s = 0.f;
            for (int t = lane; t < T; t += 64) s += p[(long long)t * st];
            v = a0_wave_sum(s) / (float)T;
        } else if (mode == 2) {
            float mx = -INFINITY;
            for (int t = lane; t < T; t += 64) mx = fmaxf(mx, p[(long long)t * st]);
            mx = a0_wave_max(mx);
            float se = 0.f, sz = 0.f;
            for (int t = lane; t < T; t += 64) {
                float e = expf(p[(long long)t * st] - mx);
                se += e;
                sz += e * aux[t];
            }
            se = a0_wave_sum(se);
            sz = a0_wave_sum(sz);
            v = sz / se;
        } else {
            const float* tau = aux + (long long)b * (T + 1);
            float s = 0.f;
            for (int t = lane; t < T; t += 64) s += (tau[t + 1] - tau[t]) * p[(long long)t * st];
            v = a0_wave_sum(s);
        }
        if (qsel && lane == 0) qsel[(long long)b * A + a] = v;
        if (a == 0 || v > best) { best = v; besta = a; }   // first maximum wins, like torch.argmax on CPU
    }
    if (lane == 0) {
        if (a_star) a_star[b] = besta;
        if (qmax) qmax[b] = best;
    }
}

extern "C" int a0_select_action(const float* x, long long sb, long long sa, long long st, int B, int A, int T, int mode,
                                const float* aux, int* a_star, float* qsel, float* qmax, void* stream) {
    if (!x || B < 1 || A < 1 || T < 1 || mode < 0 || mode > 3 || ((mode == 2 || mode == 3) && !aux)) return a0_fail(A0_EINVAL, "a0_select_action: bad argument");
    hipLaunchKernelGGL(a0_select_action_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, sb, sa, st, B, A, T, mode, aux, a_star, qsel, qmax);
    return a0_fail_hip((int)hipGetLastError(), "a0_select_action");
}

// ------------------------------------------------------------------------------------------------ actor tail for distributional heads
// Everything between the head GEMM and the chosen action of Actor.act (reference agent.py:25-39 with model.py:163-177 / 190-192
// behind it) for c51 and qr: the head GEMM leaves its split-K slabs (a0_dense_fwd_partial) and this kernel sums them in slab order and
// adds the bias (== a0_reduce_bias_act_kernel), applies the dueling combine per atom (== a0_dueling_fwd_kernel), takes the expectation
// (c51: softmax over atoms x support; qr: mean over quantiles) and the first maximum (== a0_select_action_kernel) and makes the
// epsilon-greedy draw from the actor's Philox streams (== a0_egreedy_rng_kernel).  Same arithmetic, statement for statement, as the four
// kernels it replaces; one wave per environment, the row's head outputs live in LDS.
#include "philox.h"
__global__ __launch_bounds__(256) void a0_actor_dist_tail_kernel(const float* __restrict__ slabs, long long slab_stride, int nslab, const float* __restrict__ bias,
                                                                 int ld, int A, int T, int dueling, int mode, const float* __restrict__ atoms, int E,
                                                                 unsigned long long seed, uint32_t stream_a, uint32_t stream_u, unsigned long long off_a,
                                                                 unsigned long long off_u, float eps, const long long* __restrict__ ctrl,
                                                                 const float* __restrict__ eps_ptr, int* __restrict__ action, float* __restrict__ qmax) {
    extern __shared__ float xs_all[];                    // [4 waves][A*T + T]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + wave;
    const int er = e < E ? e : E - 1;
    const int NC = A * T + (dueling ? T : 0);
    float* xs = xs_all + (size_t)wave * (A * T + T);
    // head output = slab sum in slab order + bias
    const float* sp = slabs + (long long)er * ld;
    for (int c0 = lane; c0 < NC; c0 += 256) {          // four columns x eight slabs requested before any is added; additions in slab order
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nslab; z += 8) {
            float t[8][4];
#pragma unroll
            for (int zz = 0; zz < 8; ++zz)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + 64 * j;
                    t[zz][j] = (z + zz < nslab && c < NC) ? sp[(long long)(z + zz) * slab_stride + c] : 0.f;
                }
#pragma unroll
            for (int zz = 0; zz < 8; ++zz)
                if (z + zz < nslab) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += t[zz][j];
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + 64 * j;
            if (c < NC) xs[c] = acc[j] + bias[c];
        }
    }
    __syncthreads();
    if (dueling) {
        for (int t = lane; t < T; t += 64) {
            float s = 0.f;
            for (int a = 0; a < A; ++a) s += xs[a * T + t];
            const float mean = s / (float)A;
            const float v = xs[A * T + t];
            for (int a = 0; a < A; ++a) xs[a * T + t] = v + (xs[a * T + t] - mean);
        }
    }
    __syncthreads();
    float best = 0.f;
    int besta = 0;
    for (int a = 0; a < A; ++a) {
        const float* p = xs + a * T;
        float v;
        if (mode == 1) {
            float s = 0.f;
            for (int t = lane; t < T; t += 64) s += p[t];
            v = a0_wave_sum(s) / (float)T;
        } else {
            float mx = -INFINITY;
            for (int t = lane; t < T; t += 64) mx = fmaxf(mx, p[t]);
            mx = a0_wave_max(mx);
            float se = 0.f, sz = 0.f;
            for (int t = lane; t < T; t += 64) {
                float ex = expf(p[t] - mx);
                se += ex;
                sz += ex * atoms[t];
            }
            se = a0_wave_sum(se);
            sz = a0_wave_sum(sz);
            v = sz / se;
        }
        if (a == 0 || v > best) { best = v; besta = a; }   // first maximum wins, like torch.argmax on CPU
    }
    if (lane != 0 || e >= E) return;
    if (ctrl) { off_a += (unsigned long long)ctrl[A0_CTRL_RNG_ACTION]; off_u += (unsigned long long)ctrl[A0_CTRL_RNG_UNIFORM]; }
    if (eps_ptr) eps = eps_ptr[0];
    const int ra = (int)(a0_philox_word(seed, stream_a, off_a + (unsigned long long)e) % (uint32_t)A);
    const float u = (float)(a0_philox_word(seed, stream_u, off_u + (unsigned long long)e) >> 8) * 0x1.0p-24f;
    action[e] = (u > eps) ? besta : ra;
    qmax[e] = best;
}

// ---- the same tail AND the synthetic env's step in one launch, a workgroup per env (the distributional counterpart of
// a0_actor_qhead_env_kernel, net.hip): all four waves sum the head's slabs into LDS (one column per thread: the same slab-order additions),
// then wave 0 alone applies the dueling combine, takes the expectation and the first maximum, draws the action and does the env's scalar work
// with it, while waves 1-3 already write the new frame, the shifted stack and the replay row.  Same bytes as a0_actor_dist_tail +
// a0_env_synth_step_commit.
// (a0_dtenv_args + a0_actor_dist_tail_env_body live in actor_tail.h: shared with the kernel that goes on to encode the next observation, encoder_fused.hip)
__global__ __launch_bounds__(512) void a0_actor_dist_tail_env_kernel(a0_dtenv_args P) {
    extern __shared__ float xs[];                        // [max(A*T + T, ld)] head outputs of this env
    __shared__ int s_chase_cell;
    a0_actor_dist_tail_env_body(P, xs, &s_chase_cell);
}

extern "C" int a0_actor_dist_tail_env_step(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                           const float* atoms, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                           unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                           unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                           float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                           const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task, void* stream) {
    if (!slabs || !bias || !action || !qmax || E < 1 || A < 1 || T < 1 || nslab < 1 || ld < A * T + (dueling ? T : 0) || slab_stride < (long long)E * ld ||
        (mode != 1 && mode != 2) || (mode == 2 && !atoms))
        return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step: bad argument");
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !ring_act || !ring_rew || !ring_done || !obs0 || !frames || !r_act ||
        !r_rew || !r_done || n < 1 || steps < 0 || cap < E || start_slot < 0 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE || (task == A0_ENV_TASK_CHASE && A < 4))
        return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step: bad env argument");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step: buffers must be 16-byte aligned");
    const bool vec4 = !(ld & 3) && !(slab_stride & 3) && !((((uintptr_t)slabs) | ((uintptr_t)bias)) & 15);
    const size_t lds = (size_t)((A * T + T) > ld || !vec4 ? (A * T + T) : ld) * sizeof(float);
    if (lds > 160 * 1024) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step: head too wide for LDS");
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)a0_actor_dist_tail_env_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step: LDS");
        configured = lds;
    }
    a0_dtenv_args P;
    P.slabs = slabs; P.slab_stride = slab_stride; P.nslab = nslab; P.bias = bias; P.ld = ld; P.A = A; P.T = T; P.dueling = dueling; P.mode = mode; P.atoms = atoms; P.E = E;
    P.rng_seed = seed; P.stream_a = stream_a; P.stream_u = stream_u; P.off_a = off_a; P.off_u = off_u; P.eps = eps; P.ctrl = ctrl; P.eps_ptr = eps_ptr;
    P.action = action; P.qmax = qmax;
    P.env_seed = env_seed; P.rank = rank; P.g = g; P.obs_in = obs_in; P.obs_out = obs_out; P.ep_ret = ep_ret; P.final_mask = final_mask; P.final_ret = final_ret;
    P.n = n; P.steps = steps; P.gamma = gamma; P.ring_act = ring_act; P.ring_rew = ring_rew; P.ring_done = ring_done; P.obs0 = obs0; P.frames = frames;
    P.cap = cap; P.start = start_slot % cap; P.r_act = r_act; P.r_rew = r_rew; P.r_done = r_done; P.task = task;
    P.kt = 0; P.taus = nullptr; P.vec4 = vec4 ? 1 : 0;
    hipLaunchKernelGGL(a0_actor_dist_tail_env_kernel, dim3(E), dim3(512), lds, (hipStream_t)stream, P);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_dist_tail_env_step");
}

// (round 5) the same step whose kernel goes on to encode the env's new observation into act3_next — the next actor step's features (a0_actor_dist_step_enc_kernel,
// encoder_fused.hip; the scalar heads' a0_actor_qhead_env_step_enc): a c51 / qr actor step is encoder-less between a rollout's first and last step
extern "C" int a0_actor_dist_tail_env_step_enc(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                           const float* atoms, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                           unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                           unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                           float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                           const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task,
                                               const float* wt, const a0_encoder_weights* w, float* act3_next, void* stream) {
    if (!slabs || !bias || !action || !qmax || E < 1 || A < 1 || T < 1 || nslab < 1 || ld < A * T + (dueling ? T : 0) || slab_stride < (long long)E * ld ||
        (mode != 1 && mode != 2) || (mode == 2 && !atoms))
        return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: bad argument");
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !ring_act || !ring_rew || !ring_done || !obs0 || !frames || !r_act ||
        !r_rew || !r_done || n < 1 || steps < 0 || cap < E || start_slot < 0 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE || (task == A0_ENV_TASK_CHASE && A < 4))
        return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: bad env argument");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: buffers must be 16-byte aligned");
    const bool vec4 = !(ld & 3) && !(slab_stride & 3) && !((((uintptr_t)slabs) | ((uintptr_t)bias)) & 15);
    const size_t lds = (size_t)((A * T + T) > ld || !vec4 ? (A * T + T) : ld) * sizeof(float);
    if (lds > 160 * 1024) return a0_fail(A0_EINVAL, "a0_actor_dist_tail_env_step_enc: head too wide for LDS");
    a0_dtenv_args P;
    P.slabs = slabs; P.slab_stride = slab_stride; P.nslab = nslab; P.bias = bias; P.ld = ld; P.A = A; P.T = T; P.dueling = dueling; P.mode = mode; P.atoms = atoms; P.E = E;
    P.rng_seed = seed; P.stream_a = stream_a; P.stream_u = stream_u; P.off_a = off_a; P.off_u = off_u; P.eps = eps; P.ctrl = ctrl; P.eps_ptr = eps_ptr;
    P.action = action; P.qmax = qmax;
    P.env_seed = env_seed; P.rank = rank; P.g = g; P.obs_in = obs_in; P.obs_out = obs_out; P.ep_ret = ep_ret; P.final_mask = final_mask; P.final_ret = final_ret;
    P.n = n; P.steps = steps; P.gamma = gamma; P.ring_act = ring_act; P.ring_rew = ring_rew; P.ring_done = ring_done; P.obs0 = obs0; P.frames = frames;
    P.cap = cap; P.start = start_slot % cap; P.r_act = r_act; P.r_rew = r_rew; P.r_done = r_done; P.task = task;
    P.kt = 0; P.taus = nullptr; P.vec4 = vec4 ? 1 : 0;
    return a0_actor_dist_step_enc_launch(P, lds, wt, w, act3_next, (hipStream_t)stream);
}

// The quantile networks' actor tail (iqn: mean over the K sampled quantiles, mode 1; fqf: sum over the F fractions weighted by their widths, mode 3) and
// the synthetic env's step in one launch: the head GEMM over E * T rows leaves its split-K slabs [nslab][E * T][ld] (a0_dense_fwd_partial); per env one
// workgroup sums them in slab order and adds the bias (== a0_reduce_bias_act), applies the dueling combine per quantile (== a0_dueling_fwd), takes the
// action values and the first maximum (== a0_select_action modes 1 / 3), draws epsilon-greedy (== a0_egreedy_rng) and steps the env (==
// a0_env_synth_step_commit): five launches of reference agent.py:25-39 / 44-90's per-step work become one.
extern "C" int a0_actor_quantile_tail_env_step(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                               const float* taus, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                               unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                               unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                               float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                               const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task, void* stream) {
    if (!slabs || !bias || !action || !qmax || E < 1 || A < 1 || T < 1 || nslab < 1 || ld < A + (dueling ? 1 : 0) || slab_stride < (long long)E * T * ld ||
        (mode != 1 && mode != 3) || (mode == 3 && !taus))
        return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step: bad argument");
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !ring_act || !ring_rew || !ring_done || !obs0 || !frames || !r_act ||
        !r_rew || !r_done || n < 1 || steps < 0 || cap < E || start_slot < 0 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE || (task == A0_ENV_TASK_CHASE && A < 4))
        return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step: bad env argument");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step: buffers must be 16-byte aligned");
    const size_t lds = (size_t)(A * T + T) * sizeof(float);
    if (lds > 64 * 1024) return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step: head too wide for LDS");
    a0_dtenv_args P;
    P.slabs = slabs; P.slab_stride = slab_stride; P.nslab = nslab; P.bias = bias; P.ld = ld; P.A = A; P.T = T; P.dueling = dueling; P.mode = mode; P.atoms = nullptr; P.E = E;
    P.rng_seed = seed; P.stream_a = stream_a; P.stream_u = stream_u; P.off_a = off_a; P.off_u = off_u; P.eps = eps; P.ctrl = ctrl; P.eps_ptr = eps_ptr;
    P.action = action; P.qmax = qmax;
    P.env_seed = env_seed; P.rank = rank; P.g = g; P.obs_in = obs_in; P.obs_out = obs_out; P.ep_ret = ep_ret; P.final_mask = final_mask; P.final_ret = final_ret;
    P.n = n; P.steps = steps; P.gamma = gamma; P.ring_act = ring_act; P.ring_rew = ring_rew; P.ring_done = ring_done; P.obs0 = obs0; P.frames = frames;
    P.cap = cap; P.start = start_slot % cap; P.r_act = r_act; P.r_rew = r_rew; P.r_done = r_done; P.task = task;
    P.kt = 1; P.taus = taus;
    P.vec4 = (!(ld & 3) && !(slab_stride & 3) && !((((uintptr_t)slabs) | ((uintptr_t)bias)) & 15)) ? 1 : 0;
    hipLaunchKernelGGL(a0_actor_dist_tail_env_kernel, dim3(E), dim3(512), lds, (hipStream_t)stream, P);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_quantile_tail_env_step");
}

// a0_actor_quantile_tail_env_step whose workgroups go on to encode their env's new observation into act3_next [E][3136] (round 6; a0_actor_dist_tail_env_step_enc's kernel
// with the quantile layout of the head's slabs): the NEXT step's features in the same launch
extern "C" int a0_actor_quantile_tail_env_step_enc(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                                   const float* taus, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                                   unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                                   unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                                   float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                                   const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task,
                                                   const float* wt, const a0_encoder_weights* w, float* act3_next, void* stream) {
    if (!slabs || !bias || !action || !qmax || E < 1 || A < 1 || T < 1 || nslab < 1 || ld < A + (dueling ? 1 : 0) || slab_stride < (long long)E * T * ld ||
        (mode != 1 && mode != 3) || (mode == 3 && !taus))
        return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step_enc: bad argument");
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !ring_act || !ring_rew || !ring_done || !obs0 || !frames || !r_act ||
        !r_rew || !r_done || n < 1 || steps < 0 || cap < E || start_slot < 0 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE || (task == A0_ENV_TASK_CHASE && A < 4))
        return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step_enc: bad env argument");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step_enc: buffers must be 16-byte aligned");
    const size_t lds = (size_t)(A * T + T) * sizeof(float);
    if (lds > 64 * 1024) return a0_fail(A0_EINVAL, "a0_actor_quantile_tail_env_step_enc: head too wide for LDS");
    a0_dtenv_args P;
    P.slabs = slabs; P.slab_stride = slab_stride; P.nslab = nslab; P.bias = bias; P.ld = ld; P.A = A; P.T = T; P.dueling = dueling; P.mode = mode; P.atoms = nullptr; P.E = E;
    P.rng_seed = seed; P.stream_a = stream_a; P.stream_u = stream_u; P.off_a = off_a; P.off_u = off_u; P.eps = eps; P.ctrl = ctrl; P.eps_ptr = eps_ptr;
    P.action = action; P.qmax = qmax;
    P.env_seed = env_seed; P.rank = rank; P.g = g; P.obs_in = obs_in; P.obs_out = obs_out; P.ep_ret = ep_ret; P.final_mask = final_mask; P.final_ret = final_ret;
    P.n = n; P.steps = steps; P.gamma = gamma; P.ring_act = ring_act; P.ring_rew = ring_rew; P.ring_done = ring_done; P.obs0 = obs0; P.frames = frames;
    P.cap = cap; P.start = start_slot % cap; P.r_act = r_act; P.r_rew = r_rew; P.r_done = r_done; P.task = task;
    P.kt = 1; P.taus = taus;
    P.vec4 = (!(ld & 3) && !(slab_stride & 3) && !((((uintptr_t)slabs) | ((uintptr_t)bias)) & 15)) ? 1 : 0;
    return a0_actor_dist_step_enc_launch(P, lds, wt, w, act3_next, (hipStream_t)stream);
}

// mode 1: mean over the T quantiles (qr); mode 2: C51 expectation with `atoms` [T]
extern "C" int a0_actor_dist_tail(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                  const float* atoms, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                  unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax, void* stream) {
    if (!slabs || !bias || !action || !qmax || E < 1 || A < 1 || T < 1 || nslab < 1 || ld < A * T + (dueling ? T : 0) || slab_stride < (long long)E * ld ||
        (mode != 1 && mode != 2) || (mode == 2 && !atoms))
        return a0_fail(A0_EINVAL, "a0_actor_dist_tail: bad argument");
    const size_t lds = (size_t)4 * (A * T + T) * sizeof(float);
    if (lds > 160 * 1024) return a0_fail(A0_EINVAL, "a0_actor_dist_tail: head too wide for LDS");
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)a0_actor_dist_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_actor_dist_tail: LDS");
        configured = lds;
    }
    hipLaunchKernelGGL(a0_actor_dist_tail_kernel, dim3((E + 3) / 4), dim3(256), lds, (hipStream_t)stream, slabs, slab_stride, nslab, bias, ld, A, T, dueling, mode, atoms, E,
                       seed, stream_a, stream_u, off_a, off_u, eps, ctrl, eps_ptr, action, qmax);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_dist_tail");
}

// ------------------------------------------------------------------------------------------------ DQN
// loss[b] = smooth_l1(q[b][a_b] - y_b), y_b = r + gamma_n*(1-d)*q_next[b][a*_b];  dq = w * clamp(q - y, -1, 1) at a_b.
__global__ void a0_dqn_loss_kernel(const float* __restrict__ q, const float* __restrict__ q_next, int A, const int* __restrict__ act,
                                   const int* __restrict__ a_star, const float* __restrict__ rew, const float* __restrict__ done,
                                   const float* __restrict__ wgt, float gamma_n, int B, float* __restrict__ loss, float* __restrict__ dq,
                                   int* __restrict__ nan_flag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float qn = q_next[(long long)b * A + a_star[b]];
    const float y = rew[b] + (gamma_n * (1.f - done[b])) * qn;
    const int a = act[b];
    const float d = q[(long long)b * A + a] - y;
    const float ad = fabsf(d);
    const float l = (ad < 1.f) ? 0.5f * d * d : ad - 0.5f;
    loss[b] = l;
    if (l != l) atomicOr(nan_flag, 1);
    const float g = wgt[b] * fminf(fmaxf(d, -1.f), 1.f);
    for (int k = 0; k < A; ++k) dq[(long long)b * A + k] = (k == a) ? g : 0.f;
}

extern "C" int a0_loss_dqn(const float* q, const float* q_next, int A, const int* act, const int* a_star, const float* rew,
                           const float* done, const float* wgt, float gamma_n, int B, float* loss, float* dq, int* nan_flag, void* stream) {
    if (!q || !q_next || !act || !a_star || !rew || !done || !wgt || !loss || !dq || !nan_flag || B < 1 || A < 1) return a0_fail(A0_EINVAL, "a0_loss_dqn: bad argument");
    hipLaunchKernelGGL(a0_dqn_loss_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, q, q_next, A, act, a_star, rew, done, wgt, gamma_n, B, loss, dq, nan_flag);
    return a0_fail_hip((int)hipGetLastError(), "a0_loss_dqn");
}

// ------------------------------------------------------------------------------------------------ DQN: heads + loss + head gradient
// Everything between the fc1 activations and the gradient w.r.t. the raw head outputs of DQNLearner.train_step (reference
// agent.py:173-190 with model.py:108-131 behind it) in one kernel, one wave per sample: q head of the online net on h(s), of the
// target net on h'(s') and — double-Q — of the online net on h(s'); dueling combine; first-max argmax; smooth-L1 against
// y = r + gamma^n (1-d) q'(s', a*); dq; dueling backward into draw [B][ld].  Replaces nine launches (2 x (GEMM, reduce, dueling),
// select, loss, dueling backward) that together move a few hundred kilobytes.
__global__ __launch_bounds__(256) void a0_dqn_head_loss_kernel(const float* __restrict__ h_on, const float* __restrict__ h_tg, const float* __restrict__ h_sel,
                                                               const float* __restrict__ W_on, const float* __restrict__ b_on, const float* __restrict__ W_tg,
                                                               const float* __restrict__ b_tg, int A, int dueling, int ld, const int* __restrict__ act,
                                                               const float* __restrict__ rew, const float* __restrict__ done, const float* __restrict__ wgt,
                                                               float gamma_n, int B, float* __restrict__ loss, float* __restrict__ q_on_out,
                                                               float* __restrict__ q_tg_out, float* __restrict__ draw, int* __restrict__ nan_flag) {
    extern __shared__ float wsm[];                 // [online rows | target rows], NQ x 512 each
    __shared__ float raw[4][3][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NQ = A + (dueling ? 1 : 0);
    for (int i = threadIdx.x; i < NQ * 128; i += 256) { ((a0_f4*)wsm)[i] = ((const a0_f4*)W_on)[i]; ((a0_f4*)wsm)[NQ * 128 + i] = ((const a0_f4*)W_tg)[i]; }
    const int b = blockIdx.x * 4 + wave;
    const int br = b < B ? b : B - 1;
    float ho[8], ht[8], hs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ho[i] = h_on[(long long)br * 512 + lane + 64 * i];
        ht[i] = h_tg[(long long)br * 512 + lane + 64 * i];
        hs[i] = h_sel ? h_sel[(long long)br * 512 + lane + 64 * i] : 0.f;
    }
    __syncthreads();
    if (b >= B) return;
    for (int a = 0; a < NQ; ++a) {
        const float* wo = wsm + a * 512;
        const float* wt = wsm + (NQ + a) * 512;
        float so = 0.f, st = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float w1 = wo[lane + 64 * i];
            so = fmaf(ho[i], w1, so);
            ss = fmaf(hs[i], w1, ss);
            st = fmaf(ht[i], wt[lane + 64 * i], st);
        }
        so = a0_wave_sum(so); st = a0_wave_sum(st); ss = a0_wave_sum(ss);
        if (lane == 0) { raw[wave][0][a] = so + b_on[a]; raw[wave][1][a] = st + b_tg[a]; raw[wave][2][a] = ss + b_on[a]; }
    }
    if (lane != 0) return;
    const int nsel = h_sel ? 2 : 1;
    float mean[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
    if (dueling)
        for (int s3 = 0; s3 < 3; ++s3) {
            float t = 0.f;
            for (int a = 0; a < A; ++a) t += raw[wave][s3][a];
            mean[s3] = t / (float)A;
            v[s3] = raw[wave][s3][A];
        }
    auto q = [&](int s3, int a) { return dueling ? v[s3] + (raw[wave][s3][a] - mean[s3]) : raw[wave][s3][a]; };
    float best = 0.f;
    int a_star = 0;
    for (int a = 0; a < A; ++a) {
        const float x = q(nsel, a);
        if (a == 0 || x > best) { best = x; a_star = a; }       // first maximum wins, like torch.argmax on CPU
        q_on_out[(long long)b * A + a] = q(0, a);
        if (q_tg_out) q_tg_out[(long long)b * A + a] = q(1, a);
    }
    const float qn = q(1, a_star);
    const float y = rew[b] + (gamma_n * (1.f - done[b])) * qn;
    const int ab = act[b];
    const float d = q(0, ab) - y;
    const float ad = fabsf(d);
    const float l = (ad < 1.f) ? 0.5f * d * d : ad - 0.5f;
    loss[b] = l;
    if (l != l) atomicOr(nan_flag, 1);
    const float g = wgt[b] * fminf(fmaxf(d, -1.f), 1.f);
    // dueling backward of dq = g * e_ab (a0_dueling_bwd_kernel): advantage column c gets dq[c] - sum(dq)/A, the value column sum(dq)
    float* o = draw + (long long)b * ld;
    for (int c = 0; c < ld; ++c) {
        float out = 0.f;
        if (c < A) {
            out = (c == ab) ? g : 0.f;
            if (dueling) {
                float s = 0.f;
                for (int a = 0; a < A; ++a) s += (a == ab) ? g : 0.f;
                out -= s / (float)A;
            }
        } else if (dueling && c == A) {
            float s = 0.f;
            for (int a = 0; a < A; ++a) s += (a == ab) ? g : 0.f;
            out = s;
        }
        o[c] = out;
    }
}

extern "C" int a0_dqn_head_loss(const float* h_on, const float* h_tg, const float* h_sel, const float* W_on, const float* b_on, const float* W_tg,
                                const float* b_tg, int A, int dueling, int ld, const int* act, const float* rew, const float* done, const float* wgt,
                                float gamma_n, int B, float* loss, float* q_on_out, float* q_tg_out, float* draw, int* nan_flag, void* stream) {
    const int NQ = A + (dueling ? 1 : 0);
    if (!h_on || !h_tg || !W_on || !b_on || !W_tg || !b_tg || !act || !rew || !done || !wgt || !loss || !q_on_out || !draw || !nan_flag || B < 1 || A < 1 || NQ > 24 ||
        ld < NQ)
        return a0_fail(A0_EINVAL, "a0_dqn_head_loss: bad argument (A + dueling <= 24)");
    const size_t lds = (size_t)2 * NQ * 512 * sizeof(float);
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)a0_dqn_head_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_dqn_head_loss: LDS");
        configured = lds;
    }
    hipLaunchKernelGGL(a0_dqn_head_loss_kernel, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, h_on, h_tg, h_sel, W_on, b_on, W_tg, b_tg, A, dueling, ld, act,
                       rew, done, wgt, gamma_n, B, loss, q_on_out, q_tg_out, draw, nan_flag);
    return a0_fail_hip((int)hipGetLastError(), "a0_dqn_head_loss");
}

// Variant that also finishes fc1: the three fc1 GEMMs (online on s, target on s', online on s' for double-Q) leave their split-K slabs
// behind (a0_dense_fwd_partial) and this kernel sums them in slab order, adds the bias and applies the ReLU exactly like
// a0_reduce_bias_act_kernel would (bit-identical), writing only the online activations h(s) that the backward pass needs.  Two or three
// reduction launches per update disappear.
// Round 5, MDQN = true: MDQNLearner.train_step (agent.py:193-215) through the same kernel — the third pass is the TARGET network on the current observation
// (its fc1 bias and head rows are the target's), and the loss is the Munchausen one (a0_mdqn_target below, shared with a0_mdqn_loss_kernel); q_sel_out receives it.
struct a0_mdqn_par { float tau, lo; float* q_sel_out; };
template <class QN, class QC>
A0_D float a0_mdqn_target(QN qn, QC qc, int A, int a, float rew, float done, float gamma_n, float tau, float lo);
template <bool MDQN>
__global__ __launch_bounds__(256) void a0_dqn_head_loss_slabs_kernel(const float* __restrict__ s_on, const float* __restrict__ s_tg, const float* __restrict__ s_sel,
                                                                     long long slab_stride, int nslab, const float* __restrict__ b1_on,
                                                                     const float* __restrict__ b1_tg, float* __restrict__ h_on_out,
                                                                     const float* __restrict__ W_on, const float* __restrict__ b_on, const float* __restrict__ W_tg,
                                                                     const float* __restrict__ b_tg, int A, int dueling, int ld, const int* __restrict__ act,
                                                                     const float* __restrict__ rew, const float* __restrict__ done, const float* __restrict__ wgt,
                                                                     float gamma_n, int B, float* __restrict__ loss, float* __restrict__ q_on_out,
                                                                     float* __restrict__ q_tg_out, float* __restrict__ draw, int* __restrict__ nan_flag,
                                                                     float* __restrict__ dh_out, a0_mdqn_par M) {
    extern __shared__ float wsm[];                 // [online rows | target rows], NQ x 512 each
    __shared__ float raw[4][3][32];
    __shared__ float dsh[4][32];                   // this row's head gradient (draw), for the fused head data gradient
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NQ = A + (dueling ? 1 : 0);
    for (int i = threadIdx.x; i < NQ * 128; i += 256) { ((a0_f4*)wsm)[i] = ((const a0_f4*)W_on)[i]; ((a0_f4*)wsm)[NQ * 128 + i] = ((const a0_f4*)W_tg)[i]; }
    const int b = blockIdx.x * 4 + wave;
    const int br = b < B ? b : B - 1;
    float ho[8], ht[8], hs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ho[i] = 0.f; ht[i] = 0.f; hs[i] = 0.f; }
    const long long ro = (long long)br * 512 + lane;
    // four slabs per trip: every column of the trip is requested before any is added, so 64 (96) loads overlap instead of queueing;
    // the additions stay in slab order
    auto sum4 = [&](const float* __restrict__ sp, float (&h)[8]) {
        int z = 0;
        for (; z + 4 <= nslab; z += 4) {
            float t[4][8];
#pragma unroll
            for (int zz = 0; zz < 4; ++zz)
#pragma unroll
                for (int i = 0; i < 8; ++i) t[zz][i] = sp[(long long)(z + zz) * slab_stride + ro + 64 * i];
#pragma unroll
            for (int zz = 0; zz < 4; ++zz)
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] += t[zz][i];
        }
        for (; z < nslab; ++z) {
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = sp[(long long)z * slab_stride + ro + 64 * i];
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] += t[i];
        }
    };
    sum4(s_on, ho);
    sum4(s_tg, ht);
    if (s_sel) sum4(s_sel, hs);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float bo = b1_on[lane + 64 * i], bt = b1_tg[lane + 64 * i];
        float v = ho[i] + bo; ho[i] = v < 0.f ? 0.f : v;
        v = ht[i] + bt; ht[i] = v < 0.f ? 0.f : v;
        v = hs[i] + (MDQN ? bt : bo); hs[i] = v < 0.f ? 0.f : v;
    }
    __syncthreads();
    if (b >= B) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) h_on_out[(long long)b * 512 + lane + 64 * i] = ho[i];
    for (int a = 0; a < NQ; ++a) {
        const float* wo = wsm + a * 512;
        const float* wt = wsm + (NQ + a) * 512;
        float so = 0.f, st = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float w1 = wo[lane + 64 * i], w2 = wt[lane + 64 * i];
            so = fmaf(ho[i], w1, so);
            ss = fmaf(hs[i], MDQN ? w2 : w1, ss);
            st = fmaf(ht[i], w2, st);
        }
        so = a0_wave_sum(so); st = a0_wave_sum(st); ss = a0_wave_sum(ss);
        if (lane == 0) { raw[wave][0][a] = so + b_on[a]; raw[wave][1][a] = st + b_tg[a]; raw[wave][2][a] = ss + (MDQN ? b_tg[a] : b_on[a]); }
    }
    if (lane == 0) {
    const int nsel = (s_sel && !MDQN) ? 2 : 1;
    float mean[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
    if (dueling)
        for (int s3 = 0; s3 < 3; ++s3) {
            float t = 0.f;
            for (int a = 0; a < A; ++a) t += raw[wave][s3][a];
            mean[s3] = t / (float)A;
            v[s3] = raw[wave][s3][A];
        }
    auto q = [&](int s3, int a) { return dueling ? v[s3] + (raw[wave][s3][a] - mean[s3]) : raw[wave][s3][a]; };
    float best = 0.f;
    int a_star = 0;
    for (int a = 0; a < A; ++a) {
        const float x = q(nsel, a);
        if (a == 0 || x > best) { best = x; a_star = a; }       // first maximum wins, like torch.argmax on CPU
        q_on_out[(long long)b * A + a] = q(0, a);
        if (q_tg_out) q_tg_out[(long long)b * A + a] = q(1, a);
        if (MDQN && M.q_sel_out) M.q_sel_out[(long long)b * A + a] = q(2, a);
    }
    const int ab = act[b];
    float y;
    if (MDQN) {
        y = a0_mdqn_target([&](int k) { return q(1, k); }, [&](int k) { return q(2, k); }, A, ab, rew[b], done[b], gamma_n, M.tau, M.lo);
    } else {
        const float qn = q(1, a_star);
        y = rew[b] + (gamma_n * (1.f - done[b])) * qn;
    }
    const float d = q(0, ab) - y;
    const float ad = fabsf(d);
    const float l = (ad < 1.f) ? 0.5f * d * d : ad - 0.5f;
    loss[b] = l;
    if (l != l) atomicOr(nan_flag, 1);
    const float g = wgt[b] * fminf(fmaxf(d, -1.f), 1.f);
    float* o = draw + (long long)b * ld;
    for (int c = 0; c < ld; ++c) {
        float out = 0.f;
        if (c < A) {
            out = (c == ab) ? g : 0.f;
            if (dueling) {
                float s = 0.f;
                for (int a = 0; a < A; ++a) s += (a == ab) ? g : 0.f;
                out -= s / (float)A;
            }
        } else if (dueling && c == A) {
            float s = 0.f;
            for (int a = 0; a < A; ++a) s += (a == ab) ? g : 0.f;
            out = s;
        }
        o[c] = out;
        if (c < 32) dsh[wave][c] = out;
    }
    }
    // head data gradient, fused: dh[k] = (h[k] > 0) * sum_c draw[c] * W_on[c][k] over the head's rows (what a0_dense_dgrad computes from
    // draw, W_on and the ReLU mask h) — the row's draw values come from lane 0 through LDS, the weights are already staged, h is in registers
    if (dh_out) {
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): lane 0's LDS stores of dsh have landed (one wave: lock step, LDS in order)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float acc = 0.f;
            for (int c = 0; c < NQ; ++c) acc = fmaf(dsh[wave][c], wsm[c * 512 + lane + 64 * i], acc);
            dh_out[(long long)b * 512 + lane + 64 * i] = ho[i] > 0.f ? acc : 0.f;
        }
    }
}

extern "C" int a0_dqn_head_loss_slabs(const float* slabs_on, const float* slabs_tg, const float* slabs_sel, long long slab_stride, int nslab, const float* b1_on,
                                      const float* b1_tg, float* h_on_out, const float* W_on, const float* b_on, const float* W_tg, const float* b_tg, int A,
                                      int dueling, int ld, const int* act, const float* rew, const float* done, const float* wgt, float gamma_n, int B, float* loss,
                                      float* q_on_out, float* q_tg_out, float* draw, int* nan_flag, float* dh_out, void* stream) {
    const int NQ = A + (dueling ? 1 : 0);
    if (!slabs_on || !slabs_tg || !b1_on || !b1_tg || !h_on_out || !W_on || !b_on || !W_tg || !b_tg || !act || !rew || !done || !wgt || !loss || !q_on_out || !draw ||
        !nan_flag || B < 1 || A < 1 || NQ > 24 || ld < NQ || nslab < 1 || slab_stride < (long long)B * 512)
        return a0_fail(A0_EINVAL, "a0_dqn_head_loss_slabs: bad argument (A + dueling <= 24)");
    const size_t lds = (size_t)2 * NQ * 512 * sizeof(float);
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)a0_dqn_head_loss_slabs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_dqn_head_loss_slabs: LDS");
        configured = lds;
    }
    hipLaunchKernelGGL(a0_dqn_head_loss_slabs_kernel<false>, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, slabs_on, slabs_tg, slabs_sel, slab_stride, nslab, b1_on, b1_tg,
                       h_on_out, W_on, b_on, W_tg, b_tg, A, dueling, ld, act, rew, done, wgt, gamma_n, B, loss, q_on_out, q_tg_out, draw, nan_flag, dh_out, a0_mdqn_par{0.f, 0.f, nullptr});
    return a0_fail_hip((int)hipGetLastError(), "a0_dqn_head_loss_slabs");
}

extern "C" int a0_mdqn_head_loss_slabs(const float* slabs_on, const float* slabs_tg, const float* slabs_cur, long long slab_stride, int nslab, const float* b1_on,
                                       const float* b1_tg, float* h_on_out, const float* W_on, const float* b_on, const float* W_tg, const float* b_tg, int A,
                                       int dueling, int ld, const int* act, const float* rew, const float* done, const float* wgt, float gamma_n, float tau, float lo,
                                       int B, float* loss, float* q_on_out, float* q_tg_out, float* q_cur_out, float* draw, int* nan_flag, float* dh_out, void* stream) {
    const int NQ = A + (dueling ? 1 : 0);
    if (!slabs_on || !slabs_tg || !slabs_cur || !b1_on || !b1_tg || !h_on_out || !W_on || !b_on || !W_tg || !b_tg || !act || !rew || !done || !wgt || !loss || !q_on_out ||
        !draw || !nan_flag || B < 1 || A < 1 || NQ > 24 || ld < NQ || nslab < 1 || slab_stride < (long long)B * 512 || !(tau > 0.f))
        return a0_fail(A0_EINVAL, "a0_mdqn_head_loss_slabs: bad argument (A + dueling <= 24, tau > 0)");
    const size_t lds = (size_t)2 * NQ * 512 * sizeof(float);
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)a0_dqn_head_loss_slabs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_mdqn_head_loss_slabs: LDS");
        configured = lds;
    }
    hipLaunchKernelGGL(a0_dqn_head_loss_slabs_kernel<true>, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, slabs_on, slabs_tg, slabs_cur, slab_stride, nslab, b1_on, b1_tg,
                       h_on_out, W_on, b_on, W_tg, b_tg, A, dueling, ld, act, rew, done, wgt, gamma_n, B, loss, q_on_out, q_tg_out, draw, nan_flag, dh_out, a0_mdqn_par{tau, lo, q_cur_out});
    return a0_fail_hip((int)hipGetLastError(), "a0_mdqn_head_loss_slabs");
}

// ------------------------------------------------------------------------------------------------ Munchausen DQN
// MDQNLearner.train_step (reference agent.py:194-215, log_softmax_stable 116-119), per sample:
//   lp(x)  = z - tau * logsumexp(z / tau),  z = x - max(x)
//   v_next = sum_a softmax(q')_a * (q'_a - lp(q')_a)            (softmax at temperature 1, as the reference has it)
//   y      = r + tau * clamp(lp(q_tgt(obs))[a], lo, 0) + gamma_n * (1 - d) * v_next        (alpha is unused in the reference, Q17)
template <class QN, class QC>
A0_D float a0_mdqn_target(QN qn, QC qc, int A, int a, float rew, float done, float gamma_n, float tau, float lo) {
    float mx = qn(0), mc = qc(0);
    for (int k = 1; k < A; ++k) { mx = fmaxf(mx, qn(k)); mc = fmaxf(mc, qc(k)); }
    float se_t = 0.f, se_1 = 0.f, sc_t = 0.f;
    for (int k = 0; k < A; ++k) { se_t += expf((qn(k) - mx) / tau); se_1 += expf(qn(k) - mx); sc_t += expf((qc(k) - mc) / tau); }
    const float lse_t = logf(se_t), lsc_t = logf(sc_t);
    float v_next = 0.f;
    for (int k = 0; k < A; ++k) {
        const float lp = (qn(k) - mx) - tau * lse_t;
        v_next += (expf(qn(k) - mx) / se_1) * (qn(k) - lp);
    }
    float add_on = (qc(a) - mc) - tau * lsc_t;
    add_on = fminf(fmaxf(add_on, lo), 0.f);
    return rew + tau * add_on + (gamma_n * (1.f - done)) * v_next;
}

__global__ void a0_mdqn_loss_kernel(const float* __restrict__ q, const float* __restrict__ q_next, const float* __restrict__ q_cur_tgt, int A,
                                    const int* __restrict__ act, const float* __restrict__ rew, const float* __restrict__ done, const float* __restrict__ wgt,
                                    float gamma_n, float tau, float lo, int B, float* __restrict__ loss, float* __restrict__ dq, int* __restrict__ nan_flag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* qn = q_next + (long long)b * A;
    const float* qc = q_cur_tgt + (long long)b * A;
    const int a = act[b];
    const float y = a0_mdqn_target([&](int k) { return qn[k]; }, [&](int k) { return qc[k]; }, A, a, rew[b], done[b], gamma_n, tau, lo);
    const float d = q[(long long)b * A + a] - y;
    const float ad = fabsf(d);
    const float l = (ad < 1.f) ? 0.5f * d * d : ad - 0.5f;
    loss[b] = l;
    if (l != l) atomicOr(nan_flag, 1);
    const float g = wgt[b] * fminf(fmaxf(d, -1.f), 1.f);
    for (int k = 0; k < A; ++k) dq[(long long)b * A + k] = (k == a) ? g : 0.f;
}

extern "C" int a0_loss_mdqn(const float* q, const float* q_next, const float* q_cur_tgt, int A, const int* act, const float* rew, const float* done,
                            const float* wgt, float gamma_n, float tau, float lo, int B, float* loss, float* dq, int* nan_flag, void* stream) {
    if (!q || !q_next || !q_cur_tgt || !act || !rew || !done || !wgt || !loss || !dq || !nan_flag || B < 1 || A < 1 || !(tau > 0.f)) return a0_fail(A0_EINVAL, "a0_loss_mdqn: bad argument");
    hipLaunchKernelGGL(a0_mdqn_loss_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, q, q_next, q_cur_tgt, A, act, rew, done, wgt, gamma_n, tau, lo, B, loss, dq, nan_flag);
    return a0_fail_hip((int)hipGetLastError(), "a0_loss_mdqn");
}

// ------------------------------------------------------------------------------------------------ C51
// One wave per sample, one lane per atom (T <= 64).  The projection is computed in gather form — bin j sums the
// contributions of the atoms whose lo (then up) index equals j, in ascending atom order — which is the order the
// reference's two index_add_ calls produce on the CPU, so the result is deterministic (quirk Q20 resolved).
__global__ __launch_bounds__(64) void a0_c51_loss_kernel(const float* __restrict__ logits, const float* __restrict__ tgt_logits,
                                                          int A, int T, const int* __restrict__ act, const int* __restrict__ a_star,
                                                          const float* __restrict__ rew, const float* __restrict__ done,
                                                          const float* __restrict__ wgt, const float* __restrict__ atoms,
                                                          float gamma_n, float vmin, float vmax, float delta, int B,
                                                          float* __restrict__ loss, float* __restrict__ dlogits, float* __restrict__ m_out,
                                                          int* __restrict__ nan_flag) {
    __shared__ int s_lo[64], s_up[64];
    __shared__ float s_wl[64], s_wu[64];
    const int b = blockIdx.x, t = threadIdx.x;
    const bool on = t < T;
    // softmax of the target net's logits at the greedy next action
    const float* tp = tgt_logits + ((long long)b * A + a_star[b]) * T;
    const float xt = on ? tp[t] : -INFINITY;
    const float mxt = a0_wave_max(xt);
    const float et = on ? expf(xt - mxt) : 0.f;
    const float p = et / a0_wave_sum(et);
    // Bellman-shifted, clamped support and its fractional bin position
    float tz = rew[b] + (gamma_n * (1.f - done[b])) * (on ? atoms[t] : 0.f);
    tz = fminf(fmaxf(tz, vmin), vmax);
    const float bp = (tz - vmin) / delta;
    int lo = (int)floorf(bp), up = (int)ceilf(bp);
    if (up > 0 && lo == up) lo -= 1;
    if (lo < T - 1 && lo == up) up += 1;
    s_lo[t] = on ? lo : -1;
    s_up[t] = on ? up : -1;
    s_wl[t] = p * ((float)up - bp);
    s_wu[t] = p * (bp - (float)lo);
    __syncthreads();
    float m = 0.f;
    if (on) {
        for (int k = 0; k < T; ++k) if (s_lo[k] == t) m += s_wl[k];
        for (int k = 0; k < T; ++k) if (s_up[k] == t) m += s_wu[k];
    }
    if (m_out && on) m_out[(long long)b * T + t] = m;
    // cross entropy against the online net's log-softmax at the taken action
    const int a = act[b];
    const float* op = logits + ((long long)b * A + a) * T;
    const float xo = on ? op[t] : -INFINITY;
    const float mxo = a0_wave_max(xo);
    const float eo = on ? expf(xo - mxo) : 0.f;
    const float so = a0_wave_sum(eo);
    const float logp = xo - mxo - logf(so);
    const float l = -a0_wave_sum(on ? m * logp : 0.f);
    const float msum = a0_wave_sum(m);
    if (t == 0) {
        loss[b] = l;
        if (l != l) atomicOr(nan_flag, 1);
    }
    if (on) {
        const float w = wgt[b];
        const float g = w * ((eo / so) * msum - m);
        for (int k = 0; k < A; ++k) dlogits[((long long)b * A + k) * T + t] = (k == a) ? g : 0.f;
    }
}

extern "C" int a0_loss_c51(const float* logits, const float* tgt_logits, int A, int T, const int* act, const int* a_star,
                           const float* rew, const float* done, const float* wgt, const float* atoms, float gamma_n,
                           float vmin, float vmax, int B, float* loss, float* dlogits, float* m_out, int* nan_flag, void* stream) {
    if (!logits || !tgt_logits || !act || !a_star || !rew || !done || !wgt || !atoms || !loss || !dlogits || !nan_flag || B < 1 || A < 1 || T < 2 || T > 64)
        return a0_fail(A0_EINVAL, "a0_loss_c51: bad argument (2 <= num_atoms <= 64)");
    const float delta = (float)(((double)vmax - (double)vmin) / (double)(T - 1));
    hipLaunchKernelGGL(a0_c51_loss_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, tgt_logits, A, T, act, a_star, rew, done, wgt,
                       atoms, gamma_n, vmin, vmax, delta, B, loss, dlogits, m_out, nan_flag);
    return a0_fail_hip((int)hipGetLastError(), "a0_loss_c51");
}

// ------------------------------------------------------------------------------------------------ C51: head tail + loss from the head GEMMs' slabs
// Round 4.  C51Learner.train_step (reference agent.py:219-269) from the head GEMMs on, in ONE launch: the online head GEMM over [s ; s'] rows (s' only
// under double-Q) and the target head GEMM over s' leave their split-K slabs (a0_dense_fwd_partial); per sample one wave
//   sums them in slab order and adds the bias                                    (== a0_reduce_bias_act_kernel, three times)
//   applies the dueling combine per atom                                         (== a0_dueling_fwd_kernel, three times; model.py:163-177)
//   takes the expectation under softmax and its first maximum                    (== a0_select_action_kernel mode 2; agent.py:225-231)
//   projects the target distribution at that action and takes the cross entropy  (== a0_c51_loss_kernel; agent.py:233-267)
//   forms d loss / d logits and carries it back through the dueling combine      (== a0_dueling_bwd_kernel)
// — nine launches of ~5 us of latency each — statement for statement the same arithmetic, so the update's numbers do not change.  Four samples per
// workgroup; the workgroup first stages the 3 x (A + 1) x T head outputs of its samples in LDS (every thread a few columns, all slabs of a column
// requested before any is added), then each wave works on its sample with one lane per atom.
struct a0_c51hl_args {
    const float* s_on; long long stride_on; int nslab_on;      // online head slabs [nslab_on][R_on][ld]: rows [0, B) = s, rows [sel_off, sel_off + B) = s'
    const float* s_tg; long long stride_tg; int nslab_tg;      // target head slabs [nslab_tg][B][ld] on s'
    int sel_off;                                               // < 0: no double-Q (the greedy next action comes from the target's own expectation)
    const float *bias_on, *bias_tg; int ld, A, T, dueling;
    const int* act; const float *rew, *done, *wgt, *atoms; float gamma_n, vmin, vmax, delta; int B;
    float *loss, *draw, *q_on_out, *q_tg_out, *m_out; int* a_star_out; int* nan_flag;
};
__global__ __launch_bounds__(256) void a0_c51_head_loss_slabs_kernel(a0_c51hl_args P) {
    extern __shared__ float xs[];                  // [4 samples][3 passes: online(s), target(s'), online(s')][ld]
    __shared__ __attribute__((aligned(16))) int s_lo[4][64], s_up[4][64];
    __shared__ __attribute__((aligned(16))) float s_wl[4][64], s_wu[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int A = P.A, T = P.T, NQ = A + (P.dueling ? 1 : 0), NC = NQ * T, ld = P.ld;
    const int b0 = blockIdx.x * 4;
    // this wave's sample: its scalars are requested before the staging loads, so that nothing in the arithmetic below waits for memory again
    const int bme = (b0 + wave) < P.B ? b0 + wave : P.B - 1;
    const int act_b = P.act[bme];
    const float rew_b = P.rew[bme], done_b = P.done[bme], wgt_b = P.wgt[bme];
    const float atom = lane < T ? P.atoms[lane] : 0.f;
    // staging: 16 bytes per lane and slab over the padded row (ld columns: the pad columns are summed too and never read), two row pieces x eight slabs
    // requested before anything is added — ~2 us of L2 / MALL latency per dependent batch is what this phase costs, so it is cut to two or three batches
    const int npass = P.sel_off < 0 ? 2 : 3;
    const int ld4 = ld >> 2, per = 4 * ld4, total = npass * per;
    for (int it0 = threadIdx.x; it0 < total; it0 += 512) {
        a0_f4 t[2][8];
        int ns[2], dst[2];
        const a0_f4* bp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = it0 + 256 * u;
            const bool valid = it < total;
            const int p = valid ? it / per : 0, r = valid ? it - p * per : 0;
            const int sidx = r / ld4, c4 = r - sidx * ld4;
            int b = b0 + sidx;
            b = b < P.B ? b : P.B - 1;
            const float* base = (p == 1) ? P.s_tg : P.s_on;
            const long long st4 = ((p == 1) ? P.stride_tg : P.stride_on) >> 2;
            ns[u] = valid ? ((p == 1) ? P.nslab_tg : P.nslab_on) : 0;
            const a0_f4* sp = (const a0_f4*)(base + (long long)(((p == 2) ? P.sel_off : 0) + b) * ld) + c4;
            bp[u] = (const a0_f4*)((p == 1) ? P.bias_tg : P.bias_on) + c4;
            dst[u] = (sidx * 3 + p) * ld + 4 * c4;
#pragma unroll
            for (int zz = 0; zz < 8; ++zz) t[u][zz] = (zz < ns[u]) ? sp[(long long)zz * st4] : a0_zero4();
            // more than eight slabs (never the case for a 512-deep head): the rest joins slab by slab below
            if (ns[u] > 8) {
                a0_f4 acc = a0_zero4();
#pragma unroll
                for (int zz = 0; zz < 8; ++zz) { acc.x += t[u][zz].x; acc.y += t[u][zz].y; acc.z += t[u][zz].z; acc.w += t[u][zz].w; }
                for (int z = 8; z < ns[u]; ++z) { const a0_f4 v = sp[(long long)z * st4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
                t[u][0] = acc;
                ns[u] = 1;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (it0 + 256 * u >= total) continue;
            a0_f4 acc = a0_zero4();
#pragma unroll
            for (int zz = 0; zz < 8; ++zz)
                if (zz < ns[u]) { acc.x += t[u][zz].x; acc.y += t[u][zz].y; acc.z += t[u][zz].z; acc.w += t[u][zz].w; }
            const a0_f4 bv = *bp[u];
            acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
            *(a0_f4*)(xs + dst[u]) = acc;
        }
    }
    __syncthreads();
    const int b = b0 + wave;
    if (b >= P.B) return;
    float* X = xs + (long long)wave * 3 * ld;      // pass p of this wave's sample at X + p * ld
    const int t = lane;
    const bool on = t < T;
    if (P.dueling && on) {
        for (int p = 0; p < npass; ++p) {
            float* x = X + p * ld;
            float s = 0.f;
#pragma unroll 4
            for (int a = 0; a < A; ++a) s += x[a * T + t];
            const float mean = s / (float)A;
            const float v = x[A * T + t];
#pragma unroll 4
            for (int a = 0; a < A; ++a) x[a * T + t] = v + (x[a * T + t] - mean);
        }
    }
    // each lane reads back only what it wrote itself (column t of every action), so no fence is needed up to here
    if (on) {
        if (P.q_on_out) for (int a = 0; a < A; ++a) P.q_on_out[((long long)b * A + a) * T + t] = X[a * T + t];
        if (P.q_tg_out) for (int a = 0; a < A; ++a) P.q_tg_out[((long long)b * A + a) * T + t] = X[ld + a * T + t];
    }
    // greedy next action: expectation of the selecting network's distribution, first maximum (a0_select_action_kernel, mode 2)
    const float* xsel = X + (P.sel_off < 0 ? 1 : 2) * ld;
    float best = 0.f;
    int a_star = 0;
    for (int a = 0; a < A; ++a) {
        const float xv = on ? xsel[a * T + t] : -INFINITY;
        const float mx = a0_wave_max(xv);
        float se = 0.f, sz = 0.f;
        if (on) {
            const float ex = expf(xv - mx);
            se += ex;
            sz += ex * atom;
        }
        se = a0_wave_sum(se);
        sz = a0_wave_sum(sz);
        const float v = sz / se;
        if (a == 0 || v > best) { best = v; a_star = a; }
    }
    if (P.a_star_out && lane == 0) P.a_star_out[b] = a_star;
    // projection of the target distribution at a* (a0_c51_loss_kernel)
    const float xt = on ? X[ld + a_star * T + t] : -INFINITY;
    const float mxt = a0_wave_max(xt);
    const float et = on ? expf(xt - mxt) : 0.f;
    const float pr = et / a0_wave_sum(et);
    float tz = rew_b + (P.gamma_n * (1.f - done_b)) * atom;
    tz = fminf(fmaxf(tz, P.vmin), P.vmax);
    const float bp = (tz - P.vmin) / P.delta;
    int lo = (int)floorf(bp), up = (int)ceilf(bp);
    if (up > 0 && lo == up) lo -= 1;
    if (lo < T - 1 && lo == up) up += 1;
    s_lo[wave][t] = on ? lo : -1;
    s_up[wave][t] = on ? up : -1;
    s_wl[wave][t] = pr * ((float)up - bp);
    s_wu[wave][t] = pr * (bp - (float)lo);
    // one wave: its LDS writes above are ordered before its LDS reads below (in-order LDS queue); the fences keep the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float m = 0.f;
    if (on) {
        // ascending atom order, as in a0_c51_loss_kernel; four table entries per LDS read (entries k >= T hold -1 and never match)
        const int4* lo4 = (const int4*)s_lo[wave];
        const int4* up4 = (const int4*)s_up[wave];
        const a0_f4* wl4 = (const a0_f4*)s_wl[wave];
        const a0_f4* wu4 = (const a0_f4*)s_wu[wave];
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4)
            if (4 * k4 < T) {
                const int4 L = lo4[k4]; const a0_f4 Wv = wl4[k4];
                if (L.x == t) m += Wv.x;
                if (L.y == t) m += Wv.y;
                if (L.z == t) m += Wv.z;
                if (L.w == t) m += Wv.w;
            }
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4)
            if (4 * k4 < T) {
                const int4 U = up4[k4]; const a0_f4 Wv = wu4[k4];
                if (U.x == t) m += Wv.x;
                if (U.y == t) m += Wv.y;
                if (U.z == t) m += Wv.z;
                if (U.w == t) m += Wv.w;
            }
    }
    if (P.m_out && on) P.m_out[(long long)b * T + t] = m;
    const int a = act_b;
    const float xo = on ? X[a * T + t] : -INFINITY;
    const float mxo = a0_wave_max(xo);
    const float eo = on ? expf(xo - mxo) : 0.f;
    const float so = a0_wave_sum(eo);
    const float logp = xo - mxo - logf(so);
    const float l = -a0_wave_sum(on ? m * logp : 0.f);
    const float msum = a0_wave_sum(m);
    if (lane == 0) {
        P.loss[b] = l;
        if (l != l) atomicOr(P.nan_flag, 1);
    }
    // d loss / d logits at the taken action, back through the dueling combine (a0_dueling_bwd_kernel), straight into the head GEMM's output gradient
    float* o = P.draw + (long long)b * ld;
    if (on) {
        const float g = wgt_b * ((eo / so) * msum - m);
        float sdl = 0.f;
        for (int k = 0; k < A; ++k) sdl += (k == a) ? g : 0.f;
        for (int k = 0; k < A; ++k) {
            float out = (k == a) ? g : 0.f;
            if (P.dueling) out -= sdl / (float)A;
            o[k * T + t] = out;
        }
        if (P.dueling) o[A * T + t] = sdl;
    }
    for (int c = NC + lane; c < ld; c += 64) o[c] = 0.f;
}

extern "C" int a0_c51_head_loss_slabs(const float* slabs_on, long long stride_on, int nslab_on, int rows_on, const float* slabs_tg, long long stride_tg, int nslab_tg,
                                      int sel_off, const float* bias_on, const float* bias_tg, int ld, int A, int T, int dueling, const int* act, const float* rew,
                                      const float* done, const float* wgt, const float* atoms, float gamma_n, float vmin, float vmax, int B, float* loss, float* draw,
                                      float* q_on_out, float* q_tg_out, float* m_out, int* a_star_out, int* nan_flag, void* stream) {
    const int NQ = A + (dueling ? 1 : 0);
    if (!slabs_on || !slabs_tg || !bias_on || !bias_tg || !act || !rew || !done || !wgt || !atoms || !loss || !draw || !nan_flag || B < 1 || A < 1 || T < 2 || T > 64 ||
        ld < NQ * T || nslab_on < 1 || nslab_tg < 1 || rows_on < B || stride_on < (long long)rows_on * ld || stride_tg < (long long)B * ld ||
        (sel_off >= 0 && (sel_off < B || sel_off + B > rows_on)))
        return a0_fail(A0_EINVAL, "a0_c51_head_loss_slabs: bad argument (2 <= num_atoms <= 64; the s' rows of the online slabs must lie behind the s rows)");
    const size_t lds = (size_t)4 * 3 * ld * sizeof(float);
    if (lds > 150 * 1024) return a0_fail(A0_EINVAL, "a0_c51_head_loss_slabs: head too wide for LDS");
    if ((ld & 3) || (stride_on & 3) || (stride_tg & 3) || ((((uintptr_t)slabs_on) | ((uintptr_t)slabs_tg) | ((uintptr_t)bias_on) | ((uintptr_t)bias_tg)) & 15))
        return a0_fail(A0_EINVAL, "a0_c51_head_loss_slabs: slabs and biases must be 16-byte aligned, ld and the slab strides multiples of 4 floats");
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)a0_c51_head_loss_slabs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_c51_head_loss_slabs: LDS");
        configured = lds;
    }
    a0_c51hl_args P;
    P.s_on = slabs_on; P.stride_on = stride_on; P.nslab_on = nslab_on; P.s_tg = slabs_tg; P.stride_tg = stride_tg; P.nslab_tg = nslab_tg; P.sel_off = sel_off;
    P.bias_on = bias_on; P.bias_tg = bias_tg; P.ld = ld; P.A = A; P.T = T; P.dueling = dueling; P.act = act; P.rew = rew; P.done = done; P.wgt = wgt; P.atoms = atoms;
    P.gamma_n = gamma_n; P.vmin = vmin; P.vmax = vmax; P.delta = (float)(((double)vmax - (double)vmin) / (double)(T - 1)); P.B = B;
    P.loss = loss; P.draw = draw; P.q_on_out = q_on_out; P.q_tg_out = q_tg_out; P.m_out = m_out; P.a_star_out = a_star_out; P.nan_flag = nan_flag;
    hipLaunchKernelGGL(a0_c51_head_loss_slabs_kernel, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, P);
    return a0_fail_hip((int)hipGetLastError(), "a0_c51_head_loss_slabs");
}

// ------------------------------------------------------------------------------------------------ quantile Huber
// y[b][j] = r + gamma_n*(1-d) * qn(b, j, a*_b),  qn(b,j,a) = q_next[b*sb + j*sj + a*sa]
__global__ void a0_quantile_target_kernel(const float* __restrict__ q_next, long long sb, long long sj, long long sa,
                                          const int* __restrict__ a_star, const float* __restrict__ rew, const float* __restrict__ done,
                                          float gamma_n, int B, int Nd, float* __restrict__ y) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * Nd) return;
    const int b = (int)(i / Nd), j = (int)(i % Nd);
    y[i] = rew[b] + (gamma_n * (1.f - done[b])) * q_next[(long long)b * sb + (long long)j * sj + (long long)a_star[b] * sa];
}

// One online quantile q_i against the N' targets held in LDS (16-byte aligned), in target order: al = sum_j huber(q_i - T_j) |tau_i - 1{T_j < q_i}| and
// ag = sum_j clamp(q_i - T_j, -1, 1) |tau_i - 1{T_j < q_i}| (reference agent.py:110-114 and its derivative w.r.t. q_i).  Shared by the stand-alone loss kernel and
// by the kernel that runs QRLearner.train_step from the head GEMMs' slabs, so the two cannot differ.  Four targets per LDS read (every lane reads the same address: a
// broadcast).  Eight vector instructions per pair (round 4: eleven), none of them changing a rounding:
//   * the two values |tau - 1| and |tau - 0| the weight can take are formed once per quantile (kept out of the loop by an empty asm: the compiler would otherwise
//     sink the abs behind the select and pay it per pair);
//   * with c = min(|d|, 1) the Huber value (|d| < 1 ? 0.5 d d : |d| - 0.5) is c * (|d| - 0.5 c): for |d| < 1 the bracket is |d| - 0.5 |d| = 0.5 |d| exactly (an
//     exponent step) and the product |d| * (0.5 |d|) is the same correctly rounded product as (0.5 d) * d; for |d| >= 1 it is 1 * (|d| - 0.5);
//   * both sums advance in one packed fma.
A0_D void a0_qh_sweep(float qi, float tau, const float* __restrict__ s_t, int Nd, float& al, float& ag) {
    float w_lt = fabsf(tau - 1.f), w_ge = fabsf(tau - 0.f);
    asm volatile("" : "+v"(w_lt), "+v"(w_ge));
    float l = 0.f, g = 0.f;
    auto pair = [&](float tj) {
        const float d = qi - tj;
        const float ad = fabsf(d);
        const float c = fminf(ad, 1.f);
        const float h = c * __builtin_fmaf(-0.5f, c, ad);
        const float wq = (tj < qi) ? w_lt : w_ge;
        l += h * wq;
        g += fminf(fmaxf(d, -1.f), 1.f) * wq;
    };
    int j = 0;
    for (; j + 4 <= Nd; j += 4) {
        const a0_f4 t = *(const a0_f4*)(s_t + j);
        pair(t.x); pair(t.y); pair(t.z); pair(t.w);
    }
    for (; j < Nd; ++j) pair(s_t[j]);
    al = l; ag = g;
}

// One workgroup per sample; thread i owns online quantile i and sweeps the N' targets held in LDS, so the
// B x N' x N pairwise tensor of the reference (agent.py:110-114) is never materialised.
// q(b,i,a) = q[b*sb + i*si + a*sa]; taus[b*tb + i] (tb = 0: shared fixed midpoints).
// dq gets w_b/N' * sum_j clamp(q_i - T_j, -1, 1) * |tau_i - 1{T_j < q_i}| at the taken action (dq pre-zeroed by caller).
__global__ __launch_bounds__(256) void a0_quantile_huber_kernel(const float* __restrict__ q, long long sb, long long si, long long sa,
                                                                 const float* __restrict__ y, const float* __restrict__ taus, long long tb,
                                                                 const int* __restrict__ act, const float* __restrict__ wgt,
                                                                 int B, int N, int Nd, float* __restrict__ loss, float* __restrict__ dq,
                                                                 int* __restrict__ nan_flag) {
    extern __shared__ __attribute__((aligned(16))) float s_t[];       // Nd targets
    __shared__ float s_part[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < Nd; j += blockDim.x) s_t[j] = y[(long long)b * Nd + j];
    __syncthreads();
    const int a = act[b];
    float total = 0.f;
    for (int i = tid; i < N; i += blockDim.x) {
        const long long qi_off = (long long)b * sb + (long long)i * si + (long long)a * sa;
        float al, ag;
        a0_qh_sweep(q[qi_off], taus[(long long)b * tb + i], s_t, Nd, al, ag);
        total += al;
        dq[qi_off] = wgt[b] * ag / (float)Nd;
    }
    total = a0_wave_sum(total);
    if ((tid & 63) == 0) s_part[tid >> 6] = total;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += s_part[w];
        const float l = s / (float)Nd;
        loss[b] = l;
        if (l != l) atomicOr(nan_flag, 1);
    }
}

// ------------------------------------------------------------------------------------------------ QR: head tail + loss from the head GEMMs' slabs
// Round 5.  QRLearner.train_step (reference agent.py:272-293) from the head GEMMs on, in ONE launch — the quantile-regression counterpart of
// a0_c51_head_loss_slabs_kernel, same slab layout: the online head GEMM over [s ; s'] rows (s' only under double-Q) and the target head GEMM over s' leave their
// split-K slabs; per sample one workgroup of four waves
//   sums them in slab order and adds the bias                                    (== a0_reduce_bias_act_kernel, three times)
//   applies the dueling combine per quantile                                     (== a0_dueling_fwd_kernel, three times; model.py:163-177 through QRHead 180-192)
//   takes the mean over the quantiles and its first maximum                      (== a0_select_action_kernel mode 1; agent.py:277-280)
//   forms the target quantiles r + gamma^n (1 - d) q'(a*)                        (== a0_quantile_target_kernel; agent.py:281-286)
//   sweeps the N x N pairs of the quantile Huber loss                            (== a0_quantile_huber_kernel; agent.py:110-114,288-292)
//   and carries d loss / d q back through the dueling combine                    (== the dq memset + a0_dueling_bwd_kernel)
// — eleven launches of ~5 us of latency each around a 10 us loss kernel — statement for statement the same arithmetic, so the update's numbers do not change.
// The staged head outputs (3 passes x ld floats) and the N targets live in LDS; thread i owns online quantile i (strided when N > 256).
struct a0_qrhl_args {
    const float* s_on; long long stride_on; int nslab_on;      // online head slabs [nslab_on][R_on][ld]: rows [0, B) = s, rows [sel_off, sel_off + B) = s'
    const float* s_tg; long long stride_tg; int nslab_tg;      // target head slabs [nslab_tg][B][ld] on s'
    int sel_off;                                               // < 0: no double-Q (the greedy next action comes from the target's own quantile mean)
    const float *bias_on, *bias_tg; int ld, A, T, dueling;
    const int* act; const float *rew, *done, *wgt, *taus; float gamma_n; int B;
    float *loss, *draw, *q_on_out, *q_tg_out; int* a_star_out; int* nan_flag;
};
__global__ __launch_bounds__(256) void a0_qr_head_loss_slabs_kernel(a0_qrhl_args P) {
    extern __shared__ __attribute__((aligned(16))) float xq[];     // [3 passes: online(s), target(s'), online(s')][ld], then the T targets (padded to 4)
    __shared__ float s_val[32];
    __shared__ float s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = P.A, T = P.T, NQ = A + (P.dueling ? 1 : 0), NC = NQ * T, ld = P.ld;
    const int b = blockIdx.x;
    // the sample's scalars are requested before the staging loads
    const int act_b = P.act[b];
    const float rew_b = P.rew[b], done_b = P.done[b], wgt_b = P.wgt[b];
    float* s_y = xq + 3 * ld;
    // staging: 16 bytes per lane and slab over the padded row (the pad columns are summed too and never read); a pass at a time, so that the slab addresses are
    // wave-uniform offsets from one base; all slabs of a piece (at most eight) requested before any is added, additions in slab order
    const int npass = P.sel_off < 0 ? 2 : 3;
    const int ld4 = ld >> 2;
    for (int p = 0; p < npass; ++p) {
        const float* base = (p == 1) ? P.s_tg : P.s_on;
        const long long st4 = ((p == 1) ? P.stride_tg : P.stride_on) >> 2;
        const int ns = (p == 1) ? P.nslab_tg : P.nslab_on;
        const a0_f4* row = (const a0_f4*)(base + (long long)(((p == 2) ? P.sel_off : 0) + b) * ld);
        const a0_f4* bias4 = (const a0_f4*)((p == 1) ? P.bias_tg : P.bias_on);
        for (int c4 = tid; c4 < ld4; c4 += 256) {
            a0_f4 t[8];
#pragma unroll
            for (int zz = 0; zz < 8; ++zz) t[zz] = (zz < ns) ? row[(long long)zz * st4 + c4] : a0_zero4();
            a0_f4 acc = a0_zero4();
#pragma unroll
            for (int zz = 0; zz < 8; ++zz)
                if (zz < ns) { acc.x += t[zz].x; acc.y += t[zz].y; acc.z += t[zz].z; acc.w += t[zz].w; }
            const a0_f4 bv = bias4[c4];
            acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
            *(a0_f4*)(xq + p * ld + 4 * c4) = acc;
        }
    }
    __syncthreads();
    // dueling combine: thread t owns column t of every action in every pass (it reads back only what it wrote itself)
    for (int t = tid; t < T; t += 256) {
        if (P.dueling)
            for (int p = 0; p < npass; ++p) {
                float* x = xq + p * ld;
                float s = 0.f;
#pragma unroll 4
                for (int a = 0; a < A; ++a) s += x[a * T + t];
                const float mean = s / (float)A;
                const float v = x[A * T + t];
#pragma unroll 4
                for (int a = 0; a < A; ++a) x[a * T + t] = v + (x[a * T + t] - mean);
            }
        if (P.q_on_out) for (int a = 0; a < A; ++a) P.q_on_out[((long long)b * A + a) * T + t] = xq[a * T + t];
        if (P.q_tg_out) for (int a = 0; a < A; ++a) P.q_tg_out[((long long)b * A + a) * T + t] = xq[ld + a * T + t];
    }
    __syncthreads();
    // greedy next action: mean over the quantiles of the selecting network's values, first maximum (a0_select_action_kernel, mode 1: lane-strided partial sums,
    // then the wave's butterfly) — wave w takes actions w, w + 4, ...
    const float* xsel = xq + (P.sel_off < 0 ? 1 : 2) * ld;
    for (int a = wave; a < A; a += 4) {
        const float* p = xsel + a * T;
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += p[t];
        const float v = a0_wave_sum(s) / (float)T;
        if (lane == 0) s_val[a] = v;
    }
    __syncthreads();
    float best = 0.f;
    int a_star = 0;
    for (int a = 0; a < A; ++a) {
        const float v = s_val[a];
        if (a == 0 || v > best) { best = v; a_star = a; }
    }
    if (P.a_star_out && tid == 0) P.a_star_out[b] = a_star;
    // target quantiles (a0_quantile_target_kernel)
    for (int j = tid; j < T; j += 256) s_y[j] = rew_b + (P.gamma_n * (1.f - done_b)) * xq[ld + a_star * T + j];
    __syncthreads();
    // the pairwise sweep (a0_quantile_huber_kernel) and, with each quantile's gradient at hand, its way back through the dueling combine (a0_dueling_bwd_kernel on a dq
    // that is zero outside the taken action) straight into the head GEMM's output gradient
    float* o = P.draw + (long long)b * ld;
    float tot = 0.f;
    for (int i = tid; i < T; i += 256) {
        float al, ag;
        a0_qh_sweep(xq[act_b * T + i], P.taus[i], s_y, T, al, ag);
        tot += al;
        const float g = wgt_b * ag / (float)T;
        float sdl = 0.f;
        for (int k = 0; k < A; ++k) sdl += (k == act_b) ? g : 0.f;
        for (int k = 0; k < A; ++k) {
            float out = (k == act_b) ? g : 0.f;
            if (P.dueling) out -= sdl / (float)A;
            o[k * T + i] = out;
        }
        if (P.dueling) o[A * T + i] = sdl;
    }
    for (int c = NC + tid; c < ld; c += 256) o[c] = 0.f;
    tot = a0_wave_sum(tot);
    if (lane == 0) s_part[wave] = tot;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += s_part[w];
        const float l = s / (float)T;
        P.loss[b] = l;
        if (l != l) atomicOr(P.nan_flag, 1);
    }
}

extern "C" int a0_qr_head_loss_slabs(const float* slabs_on, long long stride_on, int nslab_on, int rows_on, const float* slabs_tg, long long stride_tg, int nslab_tg,
                                     int sel_off, const float* bias_on, const float* bias_tg, int ld, int A, int T, int dueling, const int* act, const float* rew,
                                     const float* done, const float* wgt, const float* taus, float gamma_n, int B, float* loss, float* draw, float* q_on_out,
                                     float* q_tg_out, int* a_star_out, int* nan_flag, void* stream) {
    const int NQ = A + (dueling ? 1 : 0);
    if (!slabs_on || !slabs_tg || !bias_on || !bias_tg || !act || !rew || !done || !wgt || !taus || !loss || !draw || !nan_flag || B < 1 || A < 1 || A > 32 || T < 1 ||
        ld < NQ * T || nslab_on < 1 || nslab_tg < 1 || nslab_on > 8 || nslab_tg > 8 || rows_on < B || stride_on < (long long)rows_on * ld || stride_tg < (long long)B * ld ||
        (sel_off >= 0 && (sel_off < B || sel_off + B > rows_on)))
        return a0_fail(A0_EINVAL, "a0_qr_head_loss_slabs: bad argument (A <= 32; at most 8 slabs per head; the s' rows of the online slabs must lie behind the s rows)");
    const size_t lds = ((size_t)3 * ld + (size_t)((T + 3) / 4 * 4)) * sizeof(float);
    if (lds > 150 * 1024) return a0_fail(A0_EINVAL, "a0_qr_head_loss_slabs: head too wide for LDS");
    if ((ld & 3) || (stride_on & 3) || (stride_tg & 3) || ((((uintptr_t)slabs_on) | ((uintptr_t)slabs_tg) | ((uintptr_t)bias_on) | ((uintptr_t)bias_tg)) & 15))
        return a0_fail(A0_EINVAL, "a0_qr_head_loss_slabs: slabs and biases must be 16-byte aligned, ld and the slab strides multiples of 4 floats");
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)a0_qr_head_loss_slabs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return a0_fail(A0_EINVAL, "a0_qr_head_loss_slabs: LDS");
        configured = lds;
    }
    a0_qrhl_args P;
    P.s_on = slabs_on; P.stride_on = stride_on; P.nslab_on = nslab_on; P.s_tg = slabs_tg; P.stride_tg = stride_tg; P.nslab_tg = nslab_tg; P.sel_off = sel_off;
    P.bias_on = bias_on; P.bias_tg = bias_tg; P.ld = ld; P.A = A; P.T = T; P.dueling = dueling; P.act = act; P.rew = rew; P.done = done; P.wgt = wgt; P.taus = taus;
    P.gamma_n = gamma_n; P.B = B; P.loss = loss; P.draw = draw; P.q_on_out = q_on_out; P.q_tg_out = q_tg_out; P.a_star_out = a_star_out; P.nan_flag = nan_flag;
    hipLaunchKernelGGL(a0_qr_head_loss_slabs_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, P);
    return a0_fail_hip((int)hipGetLastError(), "a0_qr_head_loss_slabs");
}

extern "C" int a0_quantile_target(const float* q_next, long long sb, long long sj, long long sa, const int* a_star, const float* rew,
                                  const float* done, float gamma_n, int B, int Nd, float* y, void* stream) {
    if (!q_next || !a_star || !rew || !done || !y || B < 1 || Nd < 1) return a0_fail(A0_EINVAL, "a0_quantile_target: bad argument");
    long long n = (long long)B * Nd;
    hipLaunchKernelGGL(a0_quantile_target_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q_next, sb, sj, sa, a_star, rew, done, gamma_n, B, Nd, y);
    return a0_fail_hip((int)hipGetLastError(), "a0_quantile_target");
}

extern "C" int a0_loss_quantile_huber(const float* q, long long sb, long long si, long long sa, const float* y, const float* taus,
                                      long long tb, const int* act, const float* wgt, int B, int N, int Nd, float* loss, float* dq,
                                      int* nan_flag, void* stream) {
    if (!q || !y || !taus || !act || !wgt || !loss || !dq || !nan_flag || B < 1 || N < 1 || Nd < 1 || Nd > 8192) return a0_fail(A0_EINVAL, "a0_loss_quantile_huber: bad argument");
    int threads = ((N + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    hipLaunchKernelGGL(a0_quantile_huber_kernel, dim3(B), dim3(threads), (size_t)Nd * sizeof(float), (hipStream_t)stream, q, sb, si, sa, y, taus, tb, act, wgt, B, N, Nd, loss, dq, nan_flag);
    return a0_fail_hip((int)hipGetLastError(), "a0_loss_quantile_huber");
}
