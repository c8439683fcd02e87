// Device functions of the synthetic vector env that more than one translation unit needs (synth_env.hip: the env's own kernels; net.hip: the
// actor tail that also performs the env step).  Byte-exact twin of oracle/synth_env.c.
#pragma once
#include "a0_defs.h"
#include "philox.h"

#define A0_ENV_H 84
#define A0_ENV_W 84
#define A0_ENV_PIX (A0_ENV_H * A0_ENV_W)

A0_D uint32_t a0_env_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// Reward of env e for the action `a` taken at step g (the step that produces frame(e, g)); `x` = the step's Philox draws.
//   task 0 (A0_ENV_TASK_STREAM): x0 % 1000 < 50 -> -1, < 100 -> +1, else 0 — independent of the action (the bench workload)
//   task 1 (A0_ENV_TASK_BLOCK):  the NEWEST frame of the observation the action was chosen on is frame(e, g - 1), whose bright 8x8 block sits at
//        (by, bx) = ((3 (g-1) + 11 e) % 77, (5 (g-1) + 7 e) % 77); target = (2 [by >= 39] + [bx >= 39]) % A (the block's quadrant);
//        +1 for the target action, -1 for (target + 1) % A, 0 otherwise — chance level 0, optimum +1 per step, learnable from the pixels
A0_D float a0_env_reward(const a0_u4& x, int task, int A, uint32_t e, uint32_t g, int a) {
    if (task == A0_ENV_TASK_BLOCK) {
        const uint32_t gp = g - 1u;
        const uint32_t by = (3u * gp + 11u * e) % 77u, bx = (5u * gp + 7u * e) % 77u;
        const int target = (int)((2u * (by >= 39u ? 1u : 0u) + (bx >= 39u ? 1u : 0u)) % (uint32_t)A);
        const int wrong = (target + 1) % A;
        return a == target ? 1.0f : (a == wrong ? -1.0f : 0.0f);
    }
    const uint32_t rw = x.x % 1000u;
    return rw < 50u ? -1.0f : (rw < 100u ? 1.0f : 0.0f);
}

// `chase`: the background never reaches 255 under A0_ENV_TASK_CHASE (a lit pixel of value 255 becomes 254), so that 255 means "block" and the env can read its own
// state — the block's cell — back from the newest frame of the observation
A0_D uint8_t a0_env_pixel(uint32_t base, uint32_t by, uint32_t bx, uint32_t pix, bool chase = false) {
    const uint32_t y = pix / A0_ENV_W, x = pix - y * A0_ENV_W;
    const uint32_t h = a0_env_mix32(base ^ (pix * 0x85EBCA77u));
    uint8_t v = (((h >> 8) & 3u) == 0u) ? (uint8_t)(h & 255u) : (uint8_t)0;
    if (chase && v == 255) v = 254;
    if (y >= by && y < by + 8 && x >= bx && x < bx + 8) v = 255;
    return v;
}

// ---- task 2 (A0_ENV_TASK_CHASE): a task with TEMPORAL credit.  The bright 8x8 block lives on a 4 x 4 lattice — cell c = 4 cy + cx, top-left pixel (4 + 22 cy, 4 + 22 cx) —
// and the ACTION MOVES IT: a % 4 = 0 up, 1 down, 2 left, 3 right (clamped at the walls).  The env carries that state in its own output: the current cell is the one
// whose probe pixel (7 + 22 cy, 7 + 22 cx) of the observation's newest frame is 255.  Reward +1 only on ARRIVAL at the target cell 15 (bottom right), after which the
// block respawns at one of the ten cells at Manhattan distance >= 3 (the step's fourth Philox word picks it): a reward needs three to six correct moves, none of
// which pays by itself — a learner without a bootstrap term, with a mis-indexed n-step window or without discounting cannot find the policy (tests/test_gpu_learning.py).
// Optimum: one reward per 4.0 steps on average (the mean spawn distance).
#define A0_CHASE_TARGET 15
A0_D uint32_t a0_chase_start_cell(uint32_t e) { return (7u * e + 3u) % 15u; }
A0_D void a0_chase_pos(int cell, uint32_t& by, uint32_t& bx) { by = 4u + 22u * (uint32_t)(cell >> 2); bx = 4u + 22u * (uint32_t)(cell & 3); }
A0_D int a0_chase_cell(const uint8_t* __restrict__ newest /* one 84 x 84 frame */, uint32_t e) {
    int found = -1;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (newest[(7 + 22 * (c >> 2)) * A0_ENV_W + 7 + 22 * (c & 3)] == 255) found = found < 0 ? c : found;
    return found < 0 ? (int)a0_chase_start_cell(e) : found;
}
A0_D int a0_chase_step(int cell, int a, uint32_t x3, float& reward) {
    int cy = cell >> 2, cx = cell & 3;
    const int m = a & 3;
    if (m == 0) cy = cy > 0 ? cy - 1 : 0;
    else if (m == 1) cy = cy < 3 ? cy + 1 : 3;
    else if (m == 2) cx = cx > 0 ? cx - 1 : 0;
    else cx = cx < 3 ? cx + 1 : 3;
    int nc = 4 * cy + cx;
    reward = 0.f;
    if (nc == A0_CHASE_TARGET) {
        reward = 1.f;
        // the cells with cy + cx <= 3, in row-major order: 0 1 2 3 | 4 5 6 | 8 9 | 12
        const uint32_t k = x3 % 10u;
        nc = (int)(k < 4u ? k : (k < 7u ? k : (k < 9u ? k + 1u : 12u)));
    }
    return nc;
}


// n-step bookkeeping of one env and step (a0_nstep_kernel; truncated is always 0 for this env) and the emitted transition's (a, R, D).
// Split in two so that a fused kernel can request everything this needs from memory EARLY (a0_env_commit_prefetch: the env's running return, the ring
// entries of the previous n - 1 steps, the oldest action) and do the arithmetic once the action is known (a0_env_commit_finish) without waiting for
// memory again; a0_env_commit_scalars is the two back to back.  Up to eight ring entries are prefetched, deeper n-step windows load the rest in the loop.
struct a0_env_pre { float ep_prev; float dk[8], rk[8]; int act_old, count; };

A0_D void a0_env_commit_prefetch(a0_env_pre& Z, uint32_t e, int E, int n, long long steps, const float* __restrict__ ep_ret, const int* __restrict__ ring_act,
                                 const float* __restrict__ ring_rew, const float* __restrict__ ring_done) {
    const long long have = steps + 1;
    Z.count = have < n ? (int)have : n;
    Z.ep_prev = ep_ret[e];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        Z.dk[k] = 0.f; Z.rk[k] = 0.f;
        if (k < Z.count) {
            const int idx = (int)(((steps - k) % n + n) % n);
            Z.dk[k] = ring_done[(long long)idx * E + e];
            Z.rk[k] = ring_rew[(long long)idx * E + e];
        }
    }
    Z.dk[0] = 0.f; Z.rk[0] = 0.f;
    const int oldest = (int)(((steps - (Z.count - 1)) % n + n) % n);
    Z.act_old = (Z.count > 1) ? ring_act[(long long)oldest * E + e] : 0;      // a slot other than the one this step writes (count <= n)
}

// The prefetched values and the pointers of the scalar work are wave-uniform, so the compiler keeps them in SCALAR registers from the prefetch to the end of
// the kernel — more than there are; moved to vector registers (of which these kernels use a third) they cost nothing.
#define A0_TO_VGPR(x) asm volatile("" : "+v"(x))
A0_D void a0_env_pre_to_vgpr(a0_env_pre& Z) {
    A0_TO_VGPR(Z.ep_prev); A0_TO_VGPR(Z.act_old); A0_TO_VGPR(Z.count);
#pragma unroll
    for (int k = 1; k < 8; ++k) { A0_TO_VGPR(Z.dk[k]); A0_TO_VGPR(Z.rk[k]); }
}
// the output pointers of a0_env_commit_finish, held in vector registers
struct a0_env_out { float *ep_ret, *final_mask, *final_ret; int* ring_act; float *ring_rew, *ring_done; int* r_act; float *r_rew, *r_done; };
A0_D a0_env_out a0_env_out_vgpr(float* ep_ret, float* final_mask, float* final_ret, int* ring_act, float* ring_rew, float* ring_done, int* r_act, float* r_rew, float* r_done) {
    a0_env_out O{ep_ret, final_mask, final_ret, ring_act, ring_rew, ring_done, r_act, r_rew, r_done};
    A0_TO_VGPR(O.ep_ret); A0_TO_VGPR(O.final_mask); A0_TO_VGPR(O.final_ret); A0_TO_VGPR(O.ring_act); A0_TO_VGPR(O.ring_rew); A0_TO_VGPR(O.ring_done);
    A0_TO_VGPR(O.r_act); A0_TO_VGPR(O.r_rew); A0_TO_VGPR(O.r_done);
    return O;
}

A0_D void a0_env_nstep_row_pre(const a0_env_pre& Z, uint32_t e, int E, int n, long long steps, double gamma, float r, bool done, int a_now, int* __restrict__ ring_act,
                               float* __restrict__ ring_rew, float* __restrict__ ring_done, int* __restrict__ r_act, float* __restrict__ r_rew,
                               float* __restrict__ r_done, long long slot) {
#pragma clang fp contract(off)      // R = R * gamma * (1 - d) + r in separately rounded steps, like numpy (agent.py:64-69) and oracle/core.py
    const int cur = (int)(steps % n);
    ring_act[(long long)cur * E + e] = a_now;
    ring_rew[(long long)cur * E + e] = r;
    ring_done[(long long)cur * E + e] = done ? 1.f : 0.f;
    const int count = Z.count;
    double R = 0.0;
    bool D = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < count) {
            const float dk = (k == 0) ? (done ? 1.f : 0.f) : Z.dk[k];
            const float rk = (k == 0) ? r : Z.rk[k];
            D = D || (dk != 0.f);
            R = R * gamma * (double)(1 - (dk != 0.f ? 1 : 0)) + (double)rk;
        }
    }
    for (int k = 8; k < count; ++k) {
        const int idx = (int)(((steps - k) % n + n) % n);
        const float dk = ring_done[(long long)idx * E + e];
        const float rk = ring_rew[(long long)idx * E + e];
        D = D || (dk != 0.f);
        R = R * gamma * (double)(1 - (dk != 0.f ? 1 : 0)) + (double)rk;
    }
    r_act[slot] = (count == 1) ? a_now : Z.act_old;
    r_rew[slot] = (float)R;
    r_done[slot] = D ? 1.f : 0.f;
}

// episode statistics + n-step row of env e at step g: what ONE thread per env does once the action is known
A0_D void a0_env_commit_finish(const a0_env_pre& Z, const a0_u4& x, uint32_t e, uint32_t g, int task, int A, int E, int n, long long steps, double gamma, int a_now,
                               float* __restrict__ ep_ret, float* __restrict__ final_mask, float* __restrict__ final_ret, int* __restrict__ ring_act,
                               float* __restrict__ ring_rew, float* __restrict__ ring_done, int* __restrict__ r_act, float* __restrict__ r_rew,
                               float* __restrict__ r_done, long long slot, float r_chase = 0.f) {
#pragma clang fp contract(off)
    const bool term = (x.y % 500u) == 0u;
    const float r = task == A0_ENV_TASK_CHASE ? r_chase : a0_env_reward(x, task, A, e, g, a_now);
    const bool life = (!term) && ((x.z % 200u) == 0u);
    const float ret = Z.ep_prev + r;
    final_mask[e] = term ? 1.f : 0.f;
    final_ret[e] = term ? ret : 0.f;
    ep_ret[e] = term ? 0.f : ret;
    a0_env_nstep_row_pre(Z, e, E, n, steps, gamma, r, term || life, a_now, ring_act, ring_rew, ring_done, r_act, r_rew, r_done, slot);
}

A0_D void a0_env_commit_scalars(const a0_u4& x, uint32_t e, uint32_t g, int task, int A, int E, int n, long long steps, double gamma, int a_now, float* __restrict__ ep_ret,
                                float* __restrict__ final_mask, float* __restrict__ final_ret, int* __restrict__ ring_act, float* __restrict__ ring_rew,
                                float* __restrict__ ring_done, int* __restrict__ r_act, float* __restrict__ r_rew, float* __restrict__ r_done, long long slot,
                                float r_chase = 0.f) {
    a0_env_pre Z;
    a0_env_commit_prefetch(Z, e, E, n, steps, ep_ret, ring_act, ring_rew, ring_done);
    a0_env_commit_finish(Z, x, e, g, task, A, E, n, steps, gamma, a_now, ep_ret, final_mask, final_ret, ring_act, ring_rew, ring_done, r_act, r_rew, r_done, slot, r_chase);
}

// the frame work of env e at step g for the 16-byte groups j = j0, j0 + jstride, ...: new frame, stack shift into obs_out, and the replay row
// [st (obs0) | st_next] at `row`
// (chase_cell >= 0: A0_ENV_TASK_CHASE — the block is drawn at that lattice cell, the background clamped below 255)
A0_D void a0_env_commit_frames(unsigned long long seed, uint32_t e, uint32_t g, bool term, const uint8_t* __restrict__ obs_in, uint8_t* __restrict__ obs_out,
                               const uint8_t* __restrict__ obs0, uint8_t* __restrict__ row, int j0, int jstride, int chase_cell = -1) {
    const uint32_t base = (uint32_t)seed ^ a0_env_mix32(e * 0x9E3779B1u + g);
    uint32_t by = (3u * g + 11u * e) % 77u, bx = (5u * g + 7u * e) % 77u;
    const bool chase = chase_cell >= 0;
    if (chase) a0_chase_pos(chase_cell, by, bx);
    const int q = A0_ENV_PIX / 16;
    const uint4* in16 = (const uint4*)(obs_in + (size_t)e * 4 * A0_ENV_PIX);
    const uint4* o016 = (const uint4*)(obs0 + (size_t)e * 4 * A0_ENV_PIX);
    uint4* out16 = (uint4*)(obs_out + (size_t)e * 4 * A0_ENV_PIX);
    uint4* row16 = (uint4*)row;
    for (int j = j0; j < q; j += jstride) {
        const uint4 i0 = in16[j], i1 = in16[q + j], i2 = in16[2 * q + j], i3 = in16[3 * q + j];
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = 16u * (uint32_t)j + 4u * (uint32_t)k;
            w[k] = (uint32_t)a0_env_pixel(base, by, bx, p, chase) | ((uint32_t)a0_env_pixel(base, by, bx, p + 1, chase) << 8) |
                   ((uint32_t)a0_env_pixel(base, by, bx, p + 2, chase) << 16) | ((uint32_t)a0_env_pixel(base, by, bx, p + 3, chase) << 24);
        }
        const uint4 nw = uint4{w[0], w[1], w[2], w[3]};
        uint4 n0, n1, n2, n3;
        if (term) { n0 = nw; n1 = nw; n2 = nw; n3 = nw; } else { n0 = i1; n1 = i2; n2 = i3; n3 = nw; }
        out16[j] = n0; out16[q + j] = n1; out16[2 * q + j] = n2; out16[3 * q + j] = n3;
        if (obs0 == obs_in) { row16[j] = i0; row16[q + j] = i1; row16[2 * q + j] = i2; row16[3 * q + j] = i3; }
        else { row16[j] = o016[j]; row16[q + j] = o016[q + j]; row16[2 * q + j] = o016[2 * q + j]; row16[3 * q + j] = o016[3 * q + j]; }
        row16[4 * q + j] = n0; row16[5 * q + j] = n1; row16[6 * q + j] = n2; row16[7 * q + j] = n3;
    }
}
