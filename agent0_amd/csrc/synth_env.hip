// Device-resident synthetic Atari-shaped vector environment (gfx950).  Byte-exact twin of oracle/synth_env.c; see that
// file for the definition.  Stands in for reference agent0/common/atari_wrappers.py:59-69 (gymnasium + ale-py are not
// available on the GPU box); honours the obs/reward/terminal/truncated/life_loss/episode-return contract that
// agent0/deepq/agent.py:55-62,85-88 consumes.  It is NOT Atari.
#include "a0_internal.h"

#pragma clang fp contract(off)
#include "synth_env.h"

// grid (chunks, E); each thread produces 4 consecutive pixels of the new frame and moves the matching 4-byte groups
// of the three older frames.  g = step index since reset (identical for every env: they step in lockstep).
__global__ __launch_bounds__(256) void a0_env_step_kernel(unsigned long long seed, uint32_t rank, int E, uint32_t g, const uint8_t* __restrict__ obs_in,
                                                           uint8_t* __restrict__ obs_out, float* __restrict__ ep_ret, float* __restrict__ reward,
                                                           float* __restrict__ terminal, float* __restrict__ truncated, float* __restrict__ life_loss,
                                                           float* __restrict__ final_mask, float* __restrict__ final_ret, int reset,
                                                           const long long* __restrict__ ctrl, const int* __restrict__ action, int A, int task) {
    if (ctrl) g += (uint32_t)ctrl[A0_CTRL_ENV_STEP];
    const uint32_t e = blockIdx.y;
    bool term = reset != 0;
    int chase_cell = (task == A0_ENV_TASK_CHASE) ? (int)a0_chase_start_cell(e) : -1;      // (reset: the start cell)
    float r_chase = 0.f;
    if (!reset) {
        const a0_u4 x = a0_philox4x32_10(e, g, 0u, 0x454E56u, (uint32_t)seed, (uint32_t)(seed >> 32) ^ rank);
        term = (x.y % 500u) == 0u;
        if (task == A0_ENV_TASK_CHASE) chase_cell = a0_chase_step(a0_chase_cell(obs_in + ((size_t)e * 4 + 3) * A0_ENV_PIX, e), action[e], x.w, r_chase);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const float r = task == A0_ENV_TASK_CHASE ? r_chase : a0_env_reward(x, task, A, e, g, task == A0_ENV_TASK_BLOCK ? action[e] : 0);
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
    uint32_t by = (3u * g + 11u * e) % 77u, bx = (5u * g + 7u * e) % 77u;
    const bool chase = chase_cell >= 0;
    if (chase) a0_chase_pos(chase_cell, by, bx);
    const uint32_t* in4 = (const uint32_t*)(obs_in + (size_t)e * 4 * A0_ENV_PIX);
    uint32_t* out4 = (uint32_t*)(obs_out + (size_t)e * 4 * A0_ENV_PIX);
    const int q = A0_ENV_PIX / 4;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < q; j += gridDim.x * blockDim.x) {
        const uint32_t p = 4u * (uint32_t)j;
        const uint32_t nw = (uint32_t)a0_env_pixel(base, by, bx, p, chase) | ((uint32_t)a0_env_pixel(base, by, bx, p + 1, chase) << 8) |
                            ((uint32_t)a0_env_pixel(base, by, bx, p + 2, chase) << 16) | ((uint32_t)a0_env_pixel(base, by, bx, p + 3, chase) << 24);
        if (term) {
            out4[j] = nw; out4[q + j] = nw; out4[2 * q + j] = nw; out4[3 * q + j] = nw;
        } else {
            out4[j] = in4[q + j]; out4[q + j] = in4[2 * q + j]; out4[2 * q + j] = in4[3 * q + j]; out4[3 * q + j] = nw;
        }
    }
}

extern "C" int a0_env_synth_reset(unsigned long long seed, unsigned int rank, int E, uint8_t* obs, float* ep_ret, void* stream) {
    return a0_env_synth_reset_task(seed, rank, E, obs, ep_ret, A0_ENV_TASK_STREAM, stream);
}

// (round 5) the reset of a given task: under A0_ENV_TASK_CHASE the first frame shows the block at the env's start cell over a background clamped below 255
extern "C" int a0_env_synth_reset_task(unsigned long long seed, unsigned int rank, int E, uint8_t* obs, float* ep_ret, int task, void* stream) {
    if (!obs || !ep_ret || E < 1 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE) return a0_fail(A0_EINVAL, "a0_env_synth_reset: bad argument");
    hipLaunchKernelGGL(a0_env_step_kernel, dim3(7, E), dim3(256), 0, (hipStream_t)stream, seed, rank, E, 0u, obs, obs, ep_ret,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 1, (const long long*)nullptr, (const int*)nullptr, 1,
                       task == A0_ENV_TASK_CHASE ? A0_ENV_TASK_CHASE : A0_ENV_TASK_STREAM);
    return a0_fail_hip((int)hipGetLastError(), "a0_env_synth_reset");
}

extern "C" int a0_env_synth_step(unsigned long long seed, unsigned int rank, int E, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out,
                                 float* ep_ret, float* reward, float* terminal, float* truncated, float* life_loss, float* final_mask,
                                 float* final_ret, const int* action, int A, int task, const long long* ctrl, void* stream) {
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !reward || !terminal || !truncated || !life_loss || !final_mask || !final_ret || E < 1 ||
        task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE || (task == A0_ENV_TASK_BLOCK && (!action || A < 1)) || (task == A0_ENV_TASK_CHASE && (!action || A < 4)))
        return a0_fail(A0_EINVAL, "a0_env_synth_step: bad argument (obs_in and obs_out must differ; the chase task needs at least four actions)");
    hipLaunchKernelGGL(a0_env_step_kernel, dim3(7, E), dim3(256), 0, (hipStream_t)stream, seed, rank, E, g, obs_in, obs_out, ep_ret, reward, terminal,
                       truncated, life_loss, final_mask, final_ret, 0, ctrl, action, A, task);
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
                                                                  float* __restrict__ r_done, const long long* __restrict__ ctrl, int A, int task) {
    if (ctrl) { g += (uint32_t)ctrl[A0_CTRL_ENV_STEP]; steps += ctrl[A0_CTRL_ACTOR_STEPS]; start += ctrl[A0_CTRL_REPLAY_SLOT]; }
    const uint32_t e = blockIdx.y;
    const long long slot = (start + e) % cap;
    const a0_u4 x = a0_philox4x32_10(e, g, 0u, 0x454E56u, (uint32_t)seed, (uint32_t)(seed >> 32) ^ rank);
    const bool term = (x.y % 500u) == 0u;
    int chase_cell = -1;
    float r_chase = 0.f;
    if (task == A0_ENV_TASK_CHASE) chase_cell = a0_chase_step(a0_chase_cell(obs_in + ((size_t)e * 4 + 3) * A0_ENV_PIX, e), action[e], x.w, r_chase);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        a0_env_commit_scalars(x, e, g, task, A, E, n, steps, gamma, action[e], ep_ret, final_mask, final_ret, ring_act, ring_rew, ring_done, r_act, r_rew, r_done, slot, r_chase);
    // 16 bytes per lane: 441 lanes cover a frame (two workgroups per env); every load and store is a full-width vector access
    a0_env_commit_frames(seed, e, g, term, obs_in, obs_out, obs0, frames + slot * (8LL * A0_ENV_PIX), blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, chase_cell);
}

extern "C" int a0_env_synth_step_commit(unsigned long long seed, unsigned int rank, int E, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                        float* final_mask, float* final_ret, int n, long long steps, double gamma, const int* action, int* ring_act,
                                        float* ring_rew, float* ring_done, const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act,
                                        float* r_rew, float* r_done, int A, int task, const long long* ctrl, void* stream) {
    if (!obs_in || !obs_out || obs_in == obs_out || !ep_ret || !final_mask || !final_ret || !action || !ring_act || !ring_rew || !ring_done || !obs0 || !frames ||
        !r_act || !r_rew || !r_done || E < 1 || n < 1 || steps < 0 || cap < E || start_slot < 0 || task < A0_ENV_TASK_STREAM || task > A0_ENV_TASK_CHASE ||
        (task == A0_ENV_TASK_BLOCK && A < 1) || (task == A0_ENV_TASK_CHASE && A < 4))
        return a0_fail(A0_EINVAL, "a0_env_synth_step_commit: bad argument");
    if ((((uintptr_t)obs_in) | ((uintptr_t)obs_out) | ((uintptr_t)obs0) | ((uintptr_t)frames)) & 15) return a0_fail(A0_EINVAL, "a0_env_synth_step_commit: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(a0_env_step_commit_kernel, dim3(2, E), dim3(256), 0, (hipStream_t)stream, seed, rank, E, g, obs_in, obs_out, ep_ret, final_mask, final_ret, n,
                       steps, gamma, action, ring_act, ring_rew, ring_done, obs0, frames, cap, start_slot % cap, r_act, r_rew, r_done, ctrl, A, task);
    return a0_fail_hip((int)hipGetLastError(), "a0_env_synth_step_commit");
}
