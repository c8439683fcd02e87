// Device-resident synthetic Atari-shaped vector environment (gfx950).  Byte-exact twin of oracle/synth_env.c; see that
// file for the definition.  Stands in for reference agent0/common/atari_wrappers.py:59-69 (gymnasium + ale-py are not
// available on the GPU box); honours the obs/reward/terminal/truncated/life_loss/episode-return contract that
// agent0/deepq/agent.py:55-62,85-88 consumes.  It is NOT Atari.
#include "a0_internal.h"
#include "philox.h"

#pragma clang fp contract(off)

#define A0_ENV_H 84
#define A0_ENV_W 84
#define A0_ENV_PIX (A0_ENV_H * A0_ENV_W)

A0_D uint32_t a0_env_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

A0_D uint8_t a0_env_pixel(uint32_t base, uint32_t by, uint32_t bx, uint32_t pix) {
    const uint32_t y = pix / A0_ENV_W, x = pix - y * A0_ENV_W;
    const uint32_t h = a0_env_mix32(base ^ (pix * 0x85EBCA77u));
    uint8_t v = (((h >> 8) & 3u) == 0u) ? (uint8_t)(h & 255u) : (uint8_t)0;
    if (y >= by && y < by + 8 && x >= bx && x < bx + 8) v = 255;
    return v;
}

// grid (chunks, E); each thread produces 4 consecutive pixels of the new frame and moves the matching 4-byte groups
// of the three older frames.  g = step index since reset (identical for every env: they step in lockstep).
__global__ __launch_bounds__(256) void a0_env_step_kernel(unsigned long long seed, uint32_t rank, int E, uint32_t g, const uint8_t* __restrict__ obs_in,
                                                           uint8_t* __restrict__ obs_out, float* __restrict__ ep_ret, float* __restrict__ reward,
                                                           float* __restrict__ terminal, float* __restrict__ truncated, float* __restrict__ life_loss,
                                                           float* __restrict__ final_mask, float* __restrict__ final_ret, int reset,
                                                           const long long* __restrict__ ctrl) {
    if (ctrl) g += (uint32_t)ctrl[A0_CTRL_ENV_STEP];
    const uint32_t e = blockIdx.y;
    bool term = reset != 0;
    if (!reset) {
        const a0_u4 x = a0_philox4x32_10(e, g, 0u, 0x454E56u, (uint32_t)seed, (uint32_t)(seed >> 32) ^ rank);
        term = (x.y % 500u) == 0u;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const uint32_t rw = x.x % 1000u;
            const float r = rw < 50u ? -1.0f : (rw < 100u ? 1.0f : 0.0f);
            const bool life = (!term) && ((x.z % 200u) == 0u);
            reward[e] = r; terminal[e] = term ? 1.f : 0.f; truncated[e] = 0.f; life_loss[e] = life ? 1.f : 0.f;
            const float ret = ep_ret[e] + r;
            final_mask[e] = term ? 1.f : 0.f;
            final_ret[e] = term ? ret : 0.f;
            ep_ret[e] = term ? 0.f : ret;
        }
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        ep_ret[e] = 0.f;
    }
    const uint32_t base = (uint32_t)seed ^ a0_env_mix32(e * 0x9E3779B1u + g);
    const uint32_t by = (3u * g + 11u * e) % 77u, bx = (5u * g + 7u * e) % 77u;
    const uint32_t* in4 = (const uint32_t*)(obs_in + (size_t)e * 4 * A0_ENV_PIX);
    uint32_t* out4 = (uint32_t*)(obs_out + (size_t)e * 4 * A0_ENV_PIX);
    const int q = A0_ENV_PIX / 4;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < q; j += gridDim.x * blockDim.x) {
        const uint32_t p = 4u * (uint32_t)j;
        const uint32_t nw = (uint32_t)a0_env_pixel(base, by, bx, p) | ((uint32_t)a0_env_pixel(base, by, bx, p + 1) << 8) |
                            ((uint32_t)a0_env_pixel(base, by, bx, p + 2) << 16) | ((uint32_t)a0_env_pixel(base, by, bx, p + 3) << 24);
        if (term) {
            out4[j] = nw; out4[q + j] = nw; out4[2 * q + j] = nw; out4[3 * q + j] = nw;
        } else {
            out4[j] = in4[q + j]; out4[q + j] = in4[2 * q + j]; out4[2 * q + j] = in4[3 * q + j]; out4[3 * q + j] = nw;
        }
    }
}

extern "C" int a0_env_synth_reset(unsigned long long seed, unsigned int rank, int E, uint8_t* obs, float* ep_ret, void* stream) {
    if (!obs || !ep_ret || E < 1) return a0_fail(A0_EINVAL, "a0_env_synth_reset: bad argument");
    hipLaunchKernelGGL(a0_env_step_kernel, dim3(7, E), dim3(256), 0, (hipStream_t)stream, seed, rank, E, 0u, obs, obs, ep_ret,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 1, (const long long*)nullptr);
    return a0_fail_hip((int)hipGetLastError(), "a0_env_synth_reset");
}

extern "C" int a0_env_synth_step(unsigned long long seed, unsigned int rank, int E, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out,
                                 float* ep_ret, float* reward, float* terminal, float* truncated, float* life_loss, float* final_mask,
                                 float* final_ret, const long long* ctrl, void* stream) {
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !reward || !terminal || !truncated || !life_loss || !final_mask || !final_ret || E < 1)
        return a0_fail(A0_EINVAL, "a0_env_synth_step: bad argument (obs_in and obs_out must differ)");
    hipLaunchKernelGGL(a0_env_step_kernel, dim3(7, E), dim3(256), 0, (hipStream_t)stream, seed, rank, E, g, obs_in, obs_out, ep_ret, reward, terminal,
                       truncated, life_loss, final_mask, final_ret, 0, ctrl);
    return a0_fail_hip((int)hipGetLastError(), "a0_env_synth_step");
}


// ------------------------------------------------------------------------------------------------ env step + n-step + replay commit
// One launch for the three actor-side stages that follow action selection when the synthetic env drives the rollout:
// a0_env_synth_step (new frame, reward, terminal / life-loss flags, episode statistics), a0_actor_nstep (done flag, n-step return and
// emitted action; reference agent.py:57-73) and a0_replay_insert (st || st_next row + metadata into the ring; agent.py:78-81,
// replay.py:45-53).  Same arithmetic, same draws, same bytes as the three separate kernels — checked byte for byte against the
// oracle actor in tests/test_gpu_trainer.py — but the 28 KB observation is read once instead of twice and two launches disappear
// from every actor step.  obs0 = the observation the emitted transition starts from (obs_in itself for n = 1, the ring entry of
// n - 1 steps ago otherwise).  Grid (2, E): 16 bytes per lane.
__global__ __launch_bounds__(256) void a0_env_step_commit_kernel(unsigned long long seed, uint32_t rank, int E, uint32_t g, const uint8_t* __restrict__ obs_in,
                                                                  uint8_t* __restrict__ obs_out, float* __restrict__ ep_ret, float* __restrict__ final_mask,
                                                                  float* __restrict__ final_ret, int n, long long steps, double gamma,
                                                                  const int* __restrict__ action, int* __restrict__ ring_act, float* __restrict__ ring_rew,
                                                                  float* __restrict__ ring_done, const uint8_t* __restrict__ obs0, uint8_t* __restrict__ frames,
                                                                  long long cap, long long start, int* __restrict__ r_act, float* __restrict__ r_rew,
                                                                  float* __restrict__ r_done, const long long* __restrict__ ctrl) {
    if (ctrl) { g += (uint32_t)ctrl[A0_CTRL_ENV_STEP]; steps += ctrl[A0_CTRL_ACTOR_STEPS]; start += ctrl[A0_CTRL_REPLAY_SLOT]; }
    const uint32_t e = blockIdx.y;
    const long long slot = (start + e) % cap;
    const a0_u4 x = a0_philox4x32_10(e, g, 0u, 0x454E56u, (uint32_t)seed, (uint32_t)(seed >> 32) ^ rank);
    const bool term = (x.y % 500u) == 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t rw = x.x % 1000u;
        const float r = rw < 50u ? -1.0f : (rw < 100u ? 1.0f : 0.0f);
        const bool life = (!term) && ((x.z % 200u) == 0u);
        const float ret = ep_ret[e] + r;
        final_mask[e] = term ? 1.f : 0.f;
        final_ret[e] = term ? ret : 0.f;
        ep_ret[e] = term ? 0.f : ret;
        // n-step bookkeeping (a0_nstep_kernel; truncated is always 0 for this env)
        const bool done = term || life;
        const int cur = (int)(steps % n);
        const int a_now = action[e];
        ring_act[(long long)cur * E + e] = a_now;
        ring_rew[(long long)cur * E + e] = r;
        ring_done[(long long)cur * E + e] = done ? 1.f : 0.f;
        const long long have = steps + 1;
        const int count = have < n ? (int)have : n;
        double R = 0.0;
        bool D = false;
        for (int k = 0; k < count; ++k) {
            const int idx = (int)(((steps - k) % n + n) % n);
            const float dk = (k == 0) ? (done ? 1.f : 0.f) : ring_done[(long long)idx * E + e];
            const float rk = (k == 0) ? r : ring_rew[(long long)idx * E + e];
            D = D || (dk != 0.f);
            R = R * gamma * (double)(1 - (dk != 0.f ? 1 : 0)) + (double)rk;
        }
        const int oldest = (int)(((steps - (count - 1)) % n + n) % n);
        r_act[slot] = (count == 1) ? a_now : ring_act[(long long)oldest * E + e];
        r_rew[slot] = (float)R;
        r_done[slot] = D ? 1.f : 0.f;
    }
    const uint32_t base = (uint32_t)seed ^ a0_env_mix32(e * 0x9E3779B1u + g);
    const uint32_t by = (3u * g + 11u * e) % 77u, bx = (5u * g + 7u * e) % 77u;
    // 16 bytes per lane: 441 lanes cover a frame (two workgroups per env); every load and store is a full-width vector access
    const int q = A0_ENV_PIX / 16;
    const uint4* in16 = (const uint4*)(obs_in + (size_t)e * 4 * A0_ENV_PIX);
    const uint4* o016 = (const uint4*)(obs0 + (size_t)e * 4 * A0_ENV_PIX);
    uint4* out16 = (uint4*)(obs_out + (size_t)e * 4 * A0_ENV_PIX);
    uint4* row16 = (uint4*)(frames + slot * (8LL * A0_ENV_PIX));          // [st (4 frames) | st_next (4 frames)]
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < q; j += gridDim.x * blockDim.x) {
        const uint4 i0 = in16[j], i1 = in16[q + j], i2 = in16[2 * q + j], i3 = in16[3 * q + j];
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = 16u * (uint32_t)j + 4u * (uint32_t)k;
            w[k] = (uint32_t)a0_env_pixel(base, by, bx, p) | ((uint32_t)a0_env_pixel(base, by, bx, p + 1) << 8) |
                   ((uint32_t)a0_env_pixel(base, by, bx, p + 2) << 16) | ((uint32_t)a0_env_pixel(base, by, bx, p + 3) << 24);
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

extern "C" int a0_env_synth_step_commit(unsigned long long seed, unsigned int rank, int E, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                        float* final_mask, float* final_ret, int n, long long steps, double gamma, const int* action, int* ring_act,
                                        float* ring_rew, float* ring_done, const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act,
                                        float* r_rew, float* r_done, const long long* ctrl, void* stream) {
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !action || !ring_act || !ring_rew || !ring_done || !obs0 || !frames ||
        !r_act || !r_rew || !r_done || E < 1 || n < 1 || steps < 0 || cap < E || start_slot < 0)
        return a0_fail(A0_EINVAL, "a0_env_synth_step_commit: bad argument");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_env_synth_step_commit: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(a0_env_step_commit_kernel, dim3(2, E), dim3(256), 0, (hipStream_t)stream, seed, rank, E, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n,
                       steps, gamma, action, ring_act, ring_rew, ring_done, obs0, frames, cap, start_slot % cap, r_act, r_rew, r_done, ctrl);
    return a0_fail_hip((int)hipGetLastError(), "a0_env_synth_step_commit");
}
