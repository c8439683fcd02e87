// The scalar-head actor tail + the synthetic env's step as DEVICE functions, shared by the kernel that runs them alone (a0_actor_qhead_env_kernel, net.hip) and the
// kernel that goes on to encode the NEXT observation in the same launch (a0_actor_step_enc_kernel, encoder_fused.hip: round 5).
#pragma once
#include "synth_env.h"
// One wave finishes fc1 for env `er` from the GEMM's slabs (slab sum + bias + ReLU, in slab order: bit-identical to a0_reduce_bias_act_kernel),
// evaluates the q head rows staged in `w2s`, the dueling combine, the first maximum and the epsilon-greedy draw.  `raw`: 64 floats of LDS
// owned by this wave.  Lane 0 returns the action and max_a q; call with all 64 lanes.
// `issued`: called once, behind the issue of the first slab loads and before anything waits for them (a0_actor_step_enc2_kernel: the wave joins a workgroup barrier
// there with its loads in flight).
template <class Issued>
A0_D void a0_qhead_wave(const float* __restrict__ slabs, long long slab_stride, int nslab, const float* __restrict__ b1, const float* __restrict__ w2s,
                        const float* __restrict__ b2, int A, int dueling, int er, int lane, float* __restrict__ raw, unsigned long long seed, uint32_t stream_a,
                        uint32_t stream_u, unsigned long long off_a, unsigned long long off_u, float eps, int& out_action, float& out_best, Issued&& issued) {
    const int NQ = A + (dueling ? 1 : 0);
    bool hooked = false;
    // lane holds k = lane + 64 i.  All eight columns of a slab are requested before any is added: the loads overlap instead of queueing.
    float h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = 0.f;
    const float* sp = slabs + (long long)er * 512 + lane;
    int z = 0;
    for (; z + 8 <= nslab; z += 8) {        // the actor's usual eight slabs: ONE trip, all 64 loads of a lane in flight together (same order of additions)
        float t[8][8];
#pragma unroll
        for (int zz = 0; zz < 8; ++zz)
#pragma unroll
            for (int i = 0; i < 8; ++i) t[zz][i] = sp[(long long)(z + zz) * slab_stride + 64 * i];
        if (!hooked) { issued(); hooked = true; }
#pragma unroll
        for (int zz = 0; zz < 8; ++zz)
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] += t[zz][i];
    }
    if (!hooked) issued();
    for (; z + 4 <= nslab; z += 4) {
        float t[4][8];
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
            for (int i = 0; i < 8; ++i) t[zz][i] = sp[(long long)(z + zz) * slab_stride + 64 * i];
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] += t[zz][i];
    }
    for (; z < nslab; ++z) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = sp[(long long)z * slab_stride + 64 * i];
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] += t[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float sv = h[i] + b1[lane + 64 * i];
        h[i] = sv < 0.f ? 0.f : sv;
    }
    for (int a = 0; a < NQ; ++a) {
        const float* w = w2s + a * 512;
        float sa = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sa = fmaf(h[i], w[lane + 64 * i], sa);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sa += __shfl_xor(sa, o, 64);
        if (lane == 0) raw[a] = sa + b2[a];
    }
    if (lane != 0) return;
    float mean = 0.f, v = 0.f;
    if (dueling) {
        float t = 0.f;
        for (int a = 0; a < A; ++a) t += raw[a];
        mean = t / (float)A;
        v = raw[A];
    }
    float best = 0.f;
    int besta = 0;
    for (int a = 0; a < A; ++a) {
        const float q = dueling ? v + (raw[a] - mean) : raw[a];
        if (a == 0 || q > best) { best = q; besta = a; }
    }
    const int ra = (int)(a0_philox_word(seed, stream_a, off_a + (unsigned long long)er) % (uint32_t)A);
    const float u = (float)(a0_philox_word(seed, stream_u, off_u + (unsigned long long)er) >> 8) * 0x1.0p-24f;
    out_action = (u > eps) ? besta : ra;
    out_best = best;
}

A0_D void a0_qhead_wave(const float* __restrict__ slabs, long long slab_stride, int nslab, const float* __restrict__ b1, const float* __restrict__ w2s,
                        const float* __restrict__ b2, int A, int dueling, int er, int lane, float* __restrict__ raw, unsigned long long seed, uint32_t stream_a,
                        uint32_t stream_u, unsigned long long off_a, unsigned long long off_u, float eps, int& out_action, float& out_best) {
    a0_qhead_wave(slabs, slab_stride, nslab, b1, w2s, b2, A, dueling, er, lane, raw, seed, stream_a, stream_u, off_a, off_u, eps, out_action, out_best, [] {});
}

// The actor tail AND the synthetic env's step in one launch, one workgroup per env (reference agent.py:25-39 followed by agent.py:52-81 for
// that env): wave 0 finishes fc1, evaluates the head and draws the action exactly as a0_actor_qhead_kernel does, then its lane 0 does the
// env's scalar work with that action (episode statistics, n-step bookkeeping, the replay row's a / R / D); meanwhile the other three waves
// are already producing the new frame, the shifted stack and the replay row's frames — none of which needs the action — and wave 0 joins
// them when it is done.  Same bytes as a0_actor_qhead + a0_env_synth_step_commit; one kernel boundary less on the actor's critical path,
// and the env's 29 MB of HBM traffic overlaps the latency-bound head.
struct a0_qenv_args {
    const float* slabs; long long slab_stride; int nslab; const float *b1, *W2, *b2; int A, dueling, E;
    unsigned long long rng_seed; uint32_t stream_a, stream_u; unsigned long long off_a, off_u; float eps; const long long* ctrl; const float* eps_ptr;
    int* action; float* qmax;
    unsigned long long env_seed; uint32_t rank, g; const uint8_t* obs_in; uint8_t* obs_out; float *ep_ret, *final_mask, *final_ret;
    int n; long long steps; double gamma; int* ring_act; float *ring_rew, *ring_done; const uint8_t* obs0; uint8_t* frames; long long cap, start;
    int* r_act; float *r_rew, *r_done;
    int task;
};
// Round 4: EIGHT waves.  Wave 0 is the tail alone (it also stages the head's rows for itself: no workgroup barrier anywhere in the kernel), waves 1-7 are 448
// lanes for the 441 sixteen-byte groups of a frame: every lane issues its four loads and twelve stores ONCE and at once, instead of two or three trips of
// 256 lanes that wave 0 joined only after the tail (11.1 -> see profiles/r04_experiments.md).
// The body of a0_actor_qhead_env_kernel for env blockIdx.x (512 threads).  raw: 64 floats of LDS, w2s: (A + dueling) x 512 floats of LDS for the head's rows,
// s_chase_cell: one int of LDS.  Every wave returns (no barrier at the end): a caller that goes on to read what the workgroup wrote synchronises first.
A0_D void a0_actor_qhead_env_body(const a0_qenv_args& P, float* __restrict__ raw, float* __restrict__ w2s, int* __restrict__ s_chase_cell) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t e = blockIdx.x;
    const int NQ = P.A + (P.dueling ? 1 : 0);
    uint32_t g = P.g; long long steps = P.steps, start = P.start; unsigned long long off_a = P.off_a, off_u = P.off_u; float eps = P.eps;
    if (P.ctrl) {
        g += (uint32_t)P.ctrl[A0_CTRL_ENV_STEP]; steps += P.ctrl[A0_CTRL_ACTOR_STEPS]; start += P.ctrl[A0_CTRL_REPLAY_SLOT];
        off_a += (unsigned long long)P.ctrl[A0_CTRL_RNG_ACTION]; off_u += (unsigned long long)P.ctrl[A0_CTRL_RNG_UNIFORM];
    }
    if (P.eps_ptr) eps = P.eps_ptr[0];
    const long long slot = (start + e) % P.cap;
    // the env's Philox draws on the VECTOR unit (every lane the same): on uniform inputs the compiler runs the ten rounds on the scalar unit and keeps their
    // partial products in scalar registers for the rest of the kernel (the source of its scalar-register spills)
    uint32_t e_v = e;
    asm volatile("" : "+v"(e_v));
    const a0_u4 x = a0_philox4x32_10(e_v, g, 0u, 0x454E56u, (uint32_t)P.env_seed, (uint32_t)(P.env_seed >> 32) ^ P.rank);
    const bool term = (x.y % 500u) == 0u;
    // A0_ENV_TASK_CHASE: the action moves the block, so the new frame needs it — the frame waves wait at a workgroup barrier for wave 0's tail and read the block's new
    // cell from LDS (the other tasks keep the barrier-free overlap: their frames do not depend on the action)
    const bool chase = P.task == A0_ENV_TASK_CHASE;
    if (wave != 0) {
        int cell = -1;
        if (chase) { __syncthreads(); cell = *s_chase_cell; }
        a0_env_commit_frames(P.env_seed, e, g, term, P.obs_in, P.obs_out, P.obs0, P.frames + slot * (8LL * A0_ENV_PIX), (int)threadIdx.x - 64, 448, cell);
    } else {
        // the env's scalar work needs the running return and the n-step ring's previous entries: requested now, ahead of the head's loads
        a0_env_pre Z;
        a0_env_commit_prefetch(Z, e, P.E, P.n, steps, P.ep_ret, P.ring_act, P.ring_rew, P.ring_done);
        a0_env_pre_to_vgpr(Z);
        const a0_env_out O = a0_env_out_vgpr(P.ep_ret, P.final_mask, P.final_ret, P.ring_act, P.ring_rew, P.ring_done, P.r_act, P.r_rew, P.r_done);
        A0_TO_VGPR(steps); A0_TO_VGPR(off_a); A0_TO_VGPR(off_u); A0_TO_VGPR(eps);
        int n_v = P.n, E_v = P.E, task_v = P.task; double gamma_v = P.gamma; int* action_v = P.action; float* qmax_v = P.qmax;
        A0_TO_VGPR(n_v); A0_TO_VGPR(E_v); A0_TO_VGPR(task_v); A0_TO_VGPR(gamma_v); A0_TO_VGPR(action_v); A0_TO_VGPR(qmax_v);
#pragma unroll 4
        for (int i = lane; i < NQ * 128; i += 64) ((a0_f4*)w2s)[i] = ((const a0_f4*)P.W2)[i];
        // one wave: its LDS writes above are ordered before its LDS reads below (in-order LDS queue); the fences keep the compiler from moving them
        // (staging these rows BEHIND the issue of the slab loads — one memory round trip instead of two — measured slower: 34.9 -> 36.4 us in the merged kernel)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int act = 0; float best = 0.f;
        a0_qhead_wave(P.slabs, P.slab_stride, P.nslab, P.b1, w2s, P.b2, P.A, P.dueling, (int)e, lane, raw, P.rng_seed, P.stream_a, P.stream_u, off_a, off_u, eps, act, best);
        if (lane == 0) {
            action_v[e] = act; qmax_v[e] = best;
            float r_chase = 0.f;
            if (chase) *s_chase_cell = a0_chase_step(a0_chase_cell(P.obs_in + ((size_t)e * 4 + 3) * A0_ENV_PIX, e), act, x.w, r_chase);
            a0_env_commit_finish(Z, x, e, g, task_v, P.A, E_v, n_v, steps, gamma_v, act, O.ep_ret, O.final_mask, O.final_ret, O.ring_act, O.ring_rew, O.ring_done, O.r_act, O.r_rew,
                                 O.r_done, slot, r_chase);
        }
        if (chase) __syncthreads();
    }
}

// ---- the distributional / quantile counterpart (c51, qr; iqn, fqf): a0_actor_dist_tail_env_kernel's body (loss.hip)
A0_D float a0_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
A0_D float a0_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

struct a0_dtenv_args {
    const float* slabs; long long slab_stride; int nslab; const float* bias; int ld, A, T, dueling, mode; const float* atoms; int E;
    unsigned long long rng_seed; uint32_t stream_a, stream_u; unsigned long long off_a, off_u; float eps; const long long* ctrl; const float* eps_ptr;
    int* action; float* qmax;
    unsigned long long env_seed; uint32_t rank, g; const uint8_t* obs_in; uint8_t* obs_out; float *ep_ret, *final_mask, *final_ret;
    int n; long long steps; double gamma; int* ring_act; float *ring_rew, *ring_done; const uint8_t* obs0; uint8_t* frames; long long cap, start;
    int* r_act; float *r_rew, *r_done;
    int kt;                  // 1: quantile networks (iqn / fqf) — the head's rows are (env, quantile), its columns the actions (+ value): element (a, t) of env e at
                             //    slabs[(e * T + t) * ld + a], bias per column; 0: distributional heads (c51 / qr) — one row per env, columns (a, t)
    int task;                // synthetic env's reward task (A0_ENV_TASK_*)
    int vec4;                // kt == 0 and rows / slabs / bias 16-byte aligned: the slab sum runs 16 bytes wide over the padded row
    const float* taus;       // mode 3 (fqf): [E][T + 1] fraction boundaries, value(a) = sum_t (tau[t + 1] - tau[t]) q(t, a)  (== a0_select_action_kernel mode 3)
};
// Wave 0's part of the body below: head slab sum, dueling, expectation / quantile mean, first maximum, epsilon-greedy draw, the env's scalar work with the action
// (and, chase task, the block's new cell into *s_chase_cell_p).  No workgroup barrier inside.  g / slot / x: the step's counter, the replay slot and the env's Philox
// draw as the body derives them.
A0_D void a0_actor_dist_tail_wave0(const a0_dtenv_args& P, float* __restrict__ xs, int* __restrict__ s_chase_cell_p, uint32_t g, long long slot, const a0_u4& x) {
    const int lane = threadIdx.x & 63;
    const uint32_t e = blockIdx.x;
    const int A = P.A, T = P.T, NC = A * T + (P.dueling ? T : 0);
    const bool chase = P.task == A0_ENV_TASK_CHASE;
    // everything the env's scalar work will need from memory is requested now, ahead of the slab loads: the control words, epsilon, the env's running
    // return and the n-step ring's previous entries (a0_env_commit_prefetch) — the arithmetic behind the action then never waits for memory again
    long long steps = P.steps; unsigned long long off_a = P.off_a, off_u = P.off_u; float eps = P.eps;
    if (P.ctrl) { steps += P.ctrl[A0_CTRL_ACTOR_STEPS]; off_a += (unsigned long long)P.ctrl[A0_CTRL_RNG_ACTION]; off_u += (unsigned long long)P.ctrl[A0_CTRL_RNG_UNIFORM]; }
    if (P.eps_ptr) eps = P.eps_ptr[0];
    a0_env_pre Z;
    a0_env_commit_prefetch(Z, e, P.E, P.n, steps, P.ep_ret, P.ring_act, P.ring_rew, P.ring_done);
    a0_env_pre_to_vgpr(Z);
    const a0_env_out O = a0_env_out_vgpr(P.ep_ret, P.final_mask, P.final_ret, P.ring_act, P.ring_rew, P.ring_done, P.r_act, P.r_rew, P.r_done);
    A0_TO_VGPR(steps); A0_TO_VGPR(off_a); A0_TO_VGPR(off_u); A0_TO_VGPR(eps);
    int n_v = P.n, E_v = P.E, task_v = P.task; double gamma_v = P.gamma; unsigned long long seed_v = P.rng_seed; uint32_t sa_v = P.stream_a, su_v = P.stream_u;
    int* action_v = P.action; float* qmax_v = P.qmax;
    A0_TO_VGPR(n_v); A0_TO_VGPR(E_v); A0_TO_VGPR(task_v); A0_TO_VGPR(gamma_v); A0_TO_VGPR(seed_v); A0_TO_VGPR(sa_v); A0_TO_VGPR(su_v); A0_TO_VGPR(action_v); A0_TO_VGPR(qmax_v);
    // head output = slab sum in slab order + bias (all slabs of a column requested before any is added)
    const float* sp = P.slabs + (long long)e * (P.kt ? (long long)T * P.ld : (long long)P.ld);
    if (P.vec4) {
        // 16 bytes per lane and slab: one padded row per env (kt = 0), or the env's T rows of ld columns, one contiguous block (kt = 1: stored transposed
        // into the (a, t) order the tail reads); the pad columns are summed too and never read
        const int ld4 = P.ld >> 2, n4 = P.kt ? T * ld4 : ld4;
        const long long st4 = P.slab_stride >> 2;
        for (int c4 = lane; c4 < n4; c4 += 64) {
            const a0_f4* p4 = (const a0_f4*)sp + c4;
            a0_f4 acc = a0_zero4();
            for (int z = 0; z < P.nslab; z += 8) {
                a0_f4 t[8];
#pragma unroll
                for (int zz = 0; zz < 8; ++zz) t[zz] = (z + zz < P.nslab) ? p4[(long long)(z + zz) * st4] : a0_zero4();
#pragma unroll
                for (int zz = 0; zz < 8; ++zz)
                    if (z + zz < P.nslab) { acc.x += t[zz].x; acc.y += t[zz].y; acc.z += t[zz].z; acc.w += t[zz].w; }
            }
            if (!P.kt) {
                const a0_f4 bv = ((const a0_f4*)P.bias)[c4];            // the bias block is ld long for row-per-env heads
                acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
                ((a0_f4*)xs)[c4] = acc;
                continue;
            }
            const int q = c4 / ld4, a0 = 4 * (c4 - q * ld4);
            const int NQ = A + (P.dueling ? 1 : 0);
            if (a0 < NQ) xs[a0 * T + q] = acc.x + P.bias[a0];
            if (a0 + 1 < NQ) xs[(a0 + 1) * T + q] = acc.y + P.bias[a0 + 1];
            if (a0 + 2 < NQ) xs[(a0 + 2) * T + q] = acc.z + P.bias[a0 + 2];
            if (a0 + 3 < NQ) xs[(a0 + 3) * T + q] = acc.w + P.bias[a0 + 3];
        }
    } else
    for (int c = lane; c < NC; c += 64) {
        int src = c, col = c;
        if (P.kt) { const int a = c / T, q = c - a * T; src = q * P.ld + a; col = a; }      // LDS keeps the (a, t) order either way
        float acc = 0.f;
        for (int z = 0; z < P.nslab; z += 8) {
            float t[8];
#pragma unroll
            for (int zz = 0; zz < 8; ++zz) t[zz] = (z + zz < P.nslab) ? sp[(long long)(z + zz) * P.slab_stride + src] : 0.f;
#pragma unroll
            for (int zz = 0; zz < 8; ++zz)
                if (z + zz < P.nslab) acc += t[zz];
        }
        xs[c] = acc + P.bias[col];
    }
    // one wave: its LDS writes above are ordered before its LDS reads below (in-order LDS queue); the fences keep the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        if (P.dueling) {
            for (int t = lane; t < T; t += 64) {
                float s = 0.f;
                for (int a = 0; a < A; ++a) s += xs[a * T + t];
                const float mean = s / (float)A;
                const float v = xs[A * T + t];
                for (int a = 0; a < A; ++a) xs[a * T + t] = v + (xs[a * T + t] - mean);
            }
            // one wave: its LDS writes above are ordered before its LDS reads below (in-order LDS queue); the fence keeps the compiler from moving them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        float best = 0.f;
        int besta = 0;
        for (int a = 0; a < A; ++a) {
            const float* p = xs + a * T;
            float v;
            if (P.mode == 1) {
                float s = 0.f;
                for (int t = lane; t < T; t += 64) s += p[t];
                v = a0_wave_sum(s) / (float)T;
            } else if (P.mode == 3) {
                const float* tau = P.taus + (long long)e * (T + 1);
                float s = 0.f;
                for (int t = lane; t < T; t += 64) s += (tau[t + 1] - tau[t]) * p[t];
                v = a0_wave_sum(s);
            } else {
                float mx = -INFINITY;
                for (int t = lane; t < T; t += 64) mx = fmaxf(mx, p[t]);
                mx = a0_wave_max(mx);
                float se = 0.f, sz = 0.f;
                for (int t = lane; t < T; t += 64) {
                    float ex = expf(p[t] - mx);
                    se += ex;
                    sz += ex * P.atoms[t];
                }
                se = a0_wave_sum(se);
                sz = a0_wave_sum(sz);
                v = sz / se;
            }
            if (a == 0 || v > best) { best = v; besta = a; }   // first maximum wins, like torch.argmax on CPU
        }
        if (lane == 0) {
            const int ra = (int)(a0_philox_word(seed_v, sa_v, off_a + (unsigned long long)e) % (uint32_t)A);
            const float u = (float)(a0_philox_word(seed_v, su_v, off_u + (unsigned long long)e) >> 8) * 0x1.0p-24f;
            const int act = (u > eps) ? besta : ra;
            action_v[e] = act; qmax_v[e] = best;
            float r_chase = 0.f;
            if (chase) *s_chase_cell_p = a0_chase_step(a0_chase_cell(P.obs_in + ((size_t)e * 4 + 3) * A0_ENV_PIX, e), act, x.w, r_chase);
            a0_env_commit_finish(Z, x, e, g, task_v, A, E_v, n_v, steps, gamma_v, act, O.ep_ret, O.final_mask, O.final_ret, O.ring_act, O.ring_rew, O.ring_done, O.r_act, O.r_rew,
                                 O.r_done, slot, r_chase);
        }
    }
}

// Round 4: EIGHT waves, no workgroup barrier.  Wave 0 is the tail alone (it sums the head's slabs for itself: for row-per-env heads 16 bytes per lane and slab
// over the padded row, all slabs requested before any is added), waves 1-7 are 448 lanes for the 441 sixteen-byte groups of a frame, each issuing its four
// loads and twelve stores once and at once (was: 256 lanes, two or three trips, wave 0 joining after the tail; 16.4 us -> profiles/r04_experiments.md).
// The body of a0_actor_dist_tail_env_kernel for env blockIdx.x (512 threads).  xs: max(A*T + T, ld) floats of LDS, s_chase_cell: one int of LDS.  Every wave returns (no
// barrier at the end): a caller that goes on to read what the workgroup wrote synchronises first.
A0_D void a0_actor_dist_tail_env_body(const a0_dtenv_args& P, float* __restrict__ xs, int* __restrict__ s_chase_cell_p) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t e = blockIdx.x;
    uint32_t g = P.g; long long start = P.start;
    if (P.ctrl) { g += (uint32_t)P.ctrl[A0_CTRL_ENV_STEP]; start += P.ctrl[A0_CTRL_REPLAY_SLOT]; }
    const long long slot = (start + e) % P.cap;
    // the env's Philox draws on the VECTOR unit (every lane the same): on uniform inputs the compiler runs the ten rounds on the scalar unit and keeps their
    // partial products in scalar registers for the rest of the kernel (the source of its scalar-register spills)
    uint32_t e_v = e;
    asm volatile("" : "+v"(e_v));
    const a0_u4 x = a0_philox4x32_10(e_v, g, 0u, 0x454E56u, (uint32_t)P.env_seed, (uint32_t)(P.env_seed >> 32) ^ P.rank);
    const bool term = (x.y % 500u) == 0u;
    // A0_ENV_TASK_CHASE: the frame waves wait at a workgroup barrier for wave 0's action and read the block's new cell from LDS (a0_actor_qhead_env_kernel, net.hip)
    const bool chase = P.task == A0_ENV_TASK_CHASE;
    if (wave != 0) {
        int cell = -1;
        if (chase) { __syncthreads(); cell = *s_chase_cell_p; }
        a0_env_commit_frames(P.env_seed, e, g, term, P.obs_in, P.obs_out, P.obs0, P.frames + slot * (8LL * A0_ENV_PIX), (int)threadIdx.x - 64, 448, cell);
        return;
    }
    a0_actor_dist_tail_wave0(P, xs, s_chase_cell_p, g, slot, x);
    if (chase) __syncthreads();
}


// encoder_fused.hip: the body above, then the fused encoder over the observation the workgroup has just produced (obs_out of env blockIdx.x) into act3 — the NEXT actor
// step's features in the same launch.  Returns 0, or an A0_E* code after a0_fail.
struct a0_encoder_weights;
int a0_actor_step_enc_launch(const a0_qenv_args& Q, const float* wt, const a0_encoder_weights* w, float* act3, hipStream_t st);
int a0_actor_dist_step_enc_launch(const a0_dtenv_args& Q, size_t tail_lds, const float* wt, const a0_encoder_weights* w, float* act3, hipStream_t st);
