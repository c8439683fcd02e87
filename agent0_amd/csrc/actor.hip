// Actor-side fused elementwise kernels: epsilon-greedy selection and the n-step return scan (gfx950).
// Restates reference agent0/deepq/agent.py:25-39 (act) and agent.py:57-73 (done logic + n-step accumulation over a
// deque(maxlen=n) that is never cleared, quirk Q9).  n-step sums are carried in fp64 like the reference's numpy
// arrays and rounded to fp32 once, where the reference's Trainer casts them (trainer.py:88-90).
#include "a0_internal.h"
#include "philox.h"

#pragma clang fp contract(off)

// action = u > eps ? greedy : random;  qs_out[0] = mean_e qmax[e]   (single workgroup)
__global__ __launch_bounds__(256) void a0_egreedy_kernel(const int* __restrict__ greedy, const int* __restrict__ rand_action, const float* __restrict__ u,
                                                          float eps, int E, int* __restrict__ action, const float* __restrict__ qmax, float* __restrict__ qs_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) {
        action[e] = (u[e] > eps) ? greedy[e] : rand_action[e];
        if (qmax) s += qmax[e];
    }
    if (!qs_out) return;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) qs_out[0] = red[0] / (float)E;
}

extern "C" int a0_actor_egreedy(const int* greedy, const int* rand_action, const float* u, float eps, int E, int* action, const float* qmax,
                                float* qs_out, void* stream) {
    if (!greedy || !rand_action || !u || !action || E < 1) return a0_fail(A0_EINVAL, "a0_actor_egreedy: bad argument");
    hipLaunchKernelGGL(a0_egreedy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, greedy, rand_action, u, eps, E, action, qmax, qs_out);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_egreedy");
}

// Same selection with the two draws generated in-kernel from the Philox streams (element e of stream_a / stream_u at the
// given offsets — identical values to a0_rng_randint + a0_rng_uniform followed by a0_actor_egreedy, in one launch).
__global__ __launch_bounds__(256) void a0_egreedy_rng_kernel(const int* __restrict__ greedy, unsigned long long seed, uint32_t stream_a, uint32_t stream_u,
                                                              unsigned long long off_a, unsigned long long off_u, int A, float eps, int E,
                                                              int* __restrict__ action, const float* __restrict__ qmax, float* __restrict__ qs_out,
                                                              const long long* __restrict__ ctrl, const float* __restrict__ eps_ptr) {
    __shared__ float red[256];
    if (ctrl) { off_a += (unsigned long long)ctrl[A0_CTRL_RNG_ACTION]; off_u += (unsigned long long)ctrl[A0_CTRL_RNG_UNIFORM]; }
    if (eps_ptr) eps = eps_ptr[0];
    float s = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) {
        const int ra = (int)(a0_philox_word(seed, stream_a, off_a + (unsigned long long)e) % (uint32_t)A);
        const float u = (float)(a0_philox_word(seed, stream_u, off_u + (unsigned long long)e) >> 8) * 0x1.0p-24f;
        action[e] = (u > eps) ? greedy[e] : ra;
        if (qmax) s += qmax[e];
    }
    if (!qs_out) return;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) qs_out[0] = red[0] / (float)E;
}

extern "C" int a0_actor_egreedy_rng(const int* greedy, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                    unsigned long long off_u, int A, float eps, int E, int* action, const float* qmax, float* qs_out,
                                    const long long* ctrl, const float* eps_ptr, void* stream) {
    if (!greedy || !action || E < 1 || A < 1) return a0_fail(A0_EINVAL, "a0_actor_egreedy_rng: bad argument");
    hipLaunchKernelGGL(a0_egreedy_rng_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, greedy, seed, stream_a, stream_u, off_a, off_u, A, eps, E, action, qmax, qs_out, ctrl, eps_ptr);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_egreedy_rng");
}

// Ring of the last n (action, reward, done) per env; entry for step t lives at t % n.
//   done_t = (terminal | life_loss) & ~truncated                      agent.py:57-62
//   R = 0; D = 0; for k = newest .. oldest: D |= d_k; R = R*gamma*(1-d_k) + r_k      agent.py:64-69
//   emitted action = action of the oldest entry                       agent.py:70-71
// steps = number of env steps taken BEFORE this one (so this step is written at steps % n).
__global__ void a0_nstep_kernel(int E, int n, long long steps, double gamma, const int* __restrict__ action, const float* __restrict__ reward,
                                const float* __restrict__ terminal, const float* __restrict__ truncated, const float* __restrict__ life_loss,
                                int* __restrict__ ring_act, float* __restrict__ ring_rew, float* __restrict__ ring_done,
                                int* __restrict__ out_act, float* __restrict__ out_rew, float* __restrict__ out_done, const long long* __restrict__ ctrl) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    if (ctrl) steps += ctrl[A0_CTRL_ACTOR_STEPS];
    const bool done = ((terminal[e] != 0.f) || (life_loss && life_loss[e] != 0.f)) && !(truncated[e] != 0.f);
    const int cur = (int)(steps % n);
    ring_act[(long long)cur * E + e] = action[e];
    ring_rew[(long long)cur * E + e] = reward[e];
    ring_done[(long long)cur * E + e] = done ? 1.f : 0.f;
    const long long have = steps + 1;
    const int count = have < n ? (int)have : n;
    double R = 0.0;
    bool D = false;
    for (int k = 0; k < count; ++k) {
        const int idx = (int)(((steps - k) % n + n) % n);
        const float dk = (k == 0) ? (done ? 1.f : 0.f) : ring_done[(long long)idx * E + e];
        const float rk = (k == 0) ? reward[e] : ring_rew[(long long)idx * E + e];
        D = D || (dk != 0.f);
        R = R * gamma * (double)(1 - (dk != 0.f ? 1 : 0)) + (double)rk;
    }
    const int oldest = (int)(((steps - (count - 1)) % n + n) % n);
    out_act[e] = (count == 1) ? action[e] : ring_act[(long long)oldest * E + e];
    out_rew[e] = (float)R;
    out_done[e] = D ? 1.f : 0.f;
}

extern "C" int a0_actor_nstep(int E, int n, long long steps, double gamma, const int* action, const float* reward, const float* terminal,
                              const float* truncated, const float* life_loss, int* ring_act, float* ring_rew, float* ring_done, int* out_act,
                              float* out_rew, float* out_done, const long long* ctrl, void* stream) {
    if (E < 1 || n < 1 || steps < 0 || !action || !reward || !terminal || !truncated || !ring_act || !ring_rew || !ring_done || !out_act || !out_rew || !out_done)
        return a0_fail(A0_EINVAL, "a0_actor_nstep: bad argument");
    hipLaunchKernelGGL(a0_nstep_kernel, dim3((E + 127) / 128), dim3(128), 0, (hipStream_t)stream, E, n, steps, gamma, action, reward, terminal, truncated,
                       life_loss, ring_act, ring_rew, ring_done, out_act, out_rew, out_done, ctrl);
    return a0_fail_hip((int)hipGetLastError(), "a0_actor_nstep");
}

// Device frame stack of the host-environment front-end.  The reference stacks frames on the host (gymnasium
// FrameStack, atari_wrappers.py:63) and copies the whole (E, 4, 84, 84) batch to the device every step (agent.py:27); here only the
// newest frame of an env crosses PCIe when its stack merely advanced, and the stack is rebuilt from the previous observation the
// device already holds:  out[e] = prev[e][1:] ‖ newest[e]  where advance[e] != 0.  Rows with advance[e] == 0 were uploaded whole.
__global__ __launch_bounds__(256) void a0_frame_stack_kernel(const uint4* __restrict__ prev, const uint4* __restrict__ newest, const float* __restrict__ advance,
                                                              uint4* __restrict__ out, int nstack, int fv) {
    const int e = blockIdx.y;
    if (advance[e] == 0.f) return;
    const int keep = (nstack - 1) * fv, total = nstack * fv;
    const uint4* p = prev + (long long)e * total + fv;
    const uint4* n = newest + (long long)e * fv;
    uint4* o = out + (long long)e * total;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) o[i] = i < keep ? p[i] : n[i - keep];
}

extern "C" int a0_env_frame_stack(const uint8_t* prev, const uint8_t* newest, const float* advance, uint8_t* out, int E, int nstack, long long frame_bytes,
                                  void* stream) {
    if (!prev || !newest || !advance || !out || prev == out || E < 1 || nstack < 2 || frame_bytes < 16 || (frame_bytes % 16) ||
        ((((uintptr_t)prev) | ((uintptr_t)newest) | ((uintptr_t)out)) % 16))
        return a0_fail(A0_EINVAL, "a0_env_frame_stack: bad argument (frames of a multiple of 16 bytes, 16-byte aligned, out != prev)");
    const int fv = (int)(frame_bytes / 16);
    int gx = (nstack * fv + 255) / 256; if (gx > 4) gx = 4;
    hipLaunchKernelGGL(a0_frame_stack_kernel, dim3(gx, E), dim3(256), 0, (hipStream_t)stream, (const uint4*)prev, (const uint4*)newest, advance, (uint4*)out, nstack, fv);
    return a0_fail_hip((int)hipGetLastError(), "a0_env_frame_stack");
}

// The host-environment front-end's two PCIe legs as ONE library call each (round 4): the Python thread's eager copies and launches were on the
// critical path between "workers finished" and "workers see the next actions" (~130 us of a ~310 us step, profiles/r04_experiments.md).
//   upload: newest frames + scalars (+ the few whole stacks) host -> device, then the device frame stack — agent.py:27's torch.from_numpy(obs).to(device);
//   send:   actions and the step word device -> host, the direction gymnasium's AsyncVectorEnv.step_async pickles through pipes (atari_wrappers.py:59-69).
// The send stores straight into the page-locked shared block from two tiny kernels (system-scope relaxed stores, no fence inside a kernel: the kernel
// boundary orders the actions before the word, and a worker that sees the new word therefore sees its actions) instead of two DMA copies.
__global__ void a0_pool_send_actions_kernel(const int* __restrict__ action, int* act_host, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) __hip_atomic_store(act_host + i, action[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void a0_pool_send_word_kernel(long long* ctl_host, long long word) {
    __hip_atomic_store(ctl_host, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int a0_host_device_pointer(void* host, void** dev) {
    if (!host || !dev) return a0_fail(A0_EINVAL, "a0_host_device_pointer: null argument");
    return a0_fail_hip((int)hipHostGetDevicePointer(dev, host, 0), "a0_host_device_pointer (is the block page-locked with hipHostRegister?)");
}

extern "C" int a0_env_pool_send(const int* action, int* act_host_dev, int E, long long* ctl_host_dev, long long word, void* stream) {
    if (!action || !act_host_dev || !ctl_host_dev || E < 1 || (((uintptr_t)ctl_host_dev) % 8))
        return a0_fail(A0_EINVAL, "a0_env_pool_send: bad argument");
    hipLaunchKernelGGL(a0_pool_send_actions_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, action, act_host_dev, E);
    hipLaunchKernelGGL(a0_pool_send_word_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctl_host_dev, word);
    return a0_fail_hip((int)hipGetLastError(), "a0_env_pool_send");
}

extern "C" int a0_env_pool_upload(const uint8_t* new_host, uint8_t* new_dev, const float* scal_host, float* scal_dev, int n_scal, int advance_row,
                                  const uint8_t* obs_host, const uint8_t* prev, uint8_t* out, int E, int nstack, long long frame_bytes, int* n_whole,
                                  void* stream) {
    if (!new_host || !new_dev || !scal_host || !scal_dev || !obs_host || !prev || !out || prev == out || E < 1 || nstack < 2 || n_scal < 1 || advance_row < 0 ||
        advance_row >= n_scal || frame_bytes < 16 || (frame_bytes % 16) || ((((uintptr_t)prev) | ((uintptr_t)new_dev) | ((uintptr_t)out)) % 16))
        return a0_fail(A0_EINVAL, "a0_env_pool_upload: bad argument (frames of a multiple of 16 bytes, 16-byte aligned, out != prev)");
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = hipMemcpyAsync(new_dev, new_host, (size_t)E * frame_bytes, hipMemcpyHostToDevice, s);
    if (err == hipSuccess) err = hipMemcpyAsync(scal_dev, scal_host, (size_t)n_scal * E * sizeof(float), hipMemcpyHostToDevice, s);
    // whole stacks for the envs that did not merely advance (the workers have finished: the host rows are final); adjacent envs in one copy
    const float* adv = scal_host + (size_t)advance_row * E;
    const long long ob = (long long)nstack * frame_bytes;
    int whole = 0;
    for (int e = 0; e < E && err == hipSuccess;) {
        if (adv[e] != 0.f) { ++e; continue; }
        int j = e;
        while (j + 1 < E && adv[j + 1] == 0.f) ++j;
        err = hipMemcpyAsync(out + e * ob, obs_host + e * ob, (size_t)(j + 1 - e) * ob, hipMemcpyHostToDevice, s);
        whole += j + 1 - e;
        e = j + 1;
    }
    if (n_whole) *n_whole = whole;
    if (err != hipSuccess) return a0_fail_hip((int)err, "a0_env_pool_upload: hipMemcpyAsync");
    return a0_env_frame_stack(prev, new_dev, scal_dev + (size_t)advance_row * E, out, E, nstack, frame_bytes, stream);
}
