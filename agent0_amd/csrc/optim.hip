// Optimizer, NaN guard and target-network sync over the flat parameter buffer (gfx950).
// Restates reference agent0/deepq/agent.py:102-106 (Adam lr 5e-4, eps 1e-2/B), :333-338 (RMSprop for the FQF
// fraction net), :152-158 (skip the step when any per-sample loss is NaN) and :160-161 (target <- online every
// target_update_freq successful updates) without the two host syncs per update the reference pays (quirk Q14):
// the NaN flag, the step counter and the "sync now" decision all live in a small device-side state block.
#include "a0_internal.h"

// state block (ints): see a0_learner_state in include/agent0_hip.h
//   [0] nan_flag      set by the loss kernels (atomicOr) when a per-sample loss is NaN
//   [1] update_steps  successful optimizer steps so far (reference BaseLearner.update_steps)
//   [2] skipped       number of updates skipped because of NaN
//   [3] skip_now      decision for the update in flight (1 = NaN seen, leave the parameters alone)
//   [4] sync_now      1 if update_steps % target_update_freq == 0 after this update
// scalars (floats): [0] step_size = lr / (1 - b1^t), [1] bc2_sqrt = sqrt(1 - b2^t)
// extra_flag (optional): a float that is nonzero when ANY data-parallel rank saw a NaN — the sum over ranks of a0_nan_flag_export's
// output, which travels at the tail of the dense gradient bucket instead of in an all-reduce of its own.
__global__ void a0_adam_prep_kernel(int* __restrict__ state, float* __restrict__ scal, double lr, double b1, double b2, int target_freq,
                                    const float* __restrict__ extra_flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int skip = (state[0] != 0) || (extra_flag && extra_flag[0] != 0.f);
    int steps = state[1];
    if (!skip) steps += 1; else state[2] += 1;
    const int t = steps > 0 ? steps : 1;
    scal[0] = (float)(lr / (1.0 - pow(b1, (double)t)));
    scal[1] = (float)sqrt(1.0 - pow(b2, (double)t));
    state[1] = steps;
    state[3] = skip;
    state[4] = (target_freq > 0 && (steps % target_freq) == 0) ? 1 : 0;   // evaluated even after a skipped step, like the reference
    state[0] = 0;
}

__global__ void a0_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               long long n, const int* __restrict__ state, const float* __restrict__ scal,
                               float w1, float b2, float w2, float eps) {
    if (state[3]) return;
    const float step_size = scal[0], bc2_sqrt = scal[1];
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * w1;
        vi = vi * b2 + (w2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int a0_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int* state, float* scalars,
                            double lr, double beta1, double beta2, double eps, int target_update_freq, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !state || !scalars || n < 1) return a0_fail(A0_EINVAL, "a0_adam_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(a0_adam_prep_kernel, dim3(1), dim3(1), 0, st, state, scalars, lr, beta1, beta2, target_update_freq, (const float*)nullptr);
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(a0_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, state, scalars,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps);
    return a0_fail_hip((int)hipGetLastError(), "a0_adam_step");
}

// Adam with the target copy folded in: when the prep kernel decided "sync now" (update_steps % target_update_freq == 0, agent.py:160-161)
// every element's NEW value is also written to the target buffer, over [0, n_total) — n_total > n covers blocks Adam does not own (FQF's
// fraction net).  A skipped (NaN) step leaves the parameters alone but still syncs, like the reference.  == a0_adam_step + a0_target_sync.
// FOLD: no a0_adam_prep_kernel in front.  Every workgroup derives the step's decisions and scalars itself from state[0] (NaN flag), state[1]
// (update_steps BEFORE this step) and the data-parallel flag — words nobody writes during this kernel — and workgroup 0 publishes them:
// state[2..4] as the prep kernel does, the scalars, and the NEW step count in state[5].  state[1] <- state[5] and state[0] <- 0 are
// committed by the next kernel on the stream (a0_conv_wt_kernel with `commit`), because other workgroups of this one may still have to read them.
struct a0_adam_fold { double lr, b1, b2; int target_freq; const float* extra_flag; int* state_w; float* scal_w; const float* loss; int loss_n; float* loss_ring; int ring_cap; };
template <bool FOLD>
__global__ void a0_adam_sync_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                    long long n, const int* __restrict__ state, const float* __restrict__ scal,
                                    float w1, float b2, float w2, float eps, float* __restrict__ target, long long n_total, int vec4, a0_adam_fold F) {
    bool skip, sync;
    float step_size, bc2_sqrt;
    if constexpr (FOLD) {
        __shared__ float sh_f[2];
        __shared__ int sh_i[2];
        // round 4: workgroup 0 also takes the batch mean of this update's per-sample losses (the Trainer's `loss` statistic, trainer.py:99,111-113) into a ring
        // slot indexed by the free-running call counter state[6] — the same reduction, statement for statement, as a0_mean_rows_kernel, whose launch per update
        // (4.8 us of pure latency) it replaces; skipped steps are recorded too (their mean is whatever the loss kernel wrote, NaN included)
        if (blockIdx.x == 0 && F.loss) {
            __shared__ float red[256];
            float s = 0.f;
            for (int e = threadIdx.x; e < F.loss_n; e += 256) s += F.loss[e];
            red[threadIdx.x] = s;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
            if (threadIdx.x == 0) { const int c = F.state_w[6]; F.loss_ring[c % F.ring_cap] = red[0] / (float)F.loss_n; F.state_w[6] = c + 1; }
        }
        if (threadIdx.x == 0) {
            const int sk = (state[0] != 0) || (F.extra_flag && F.extra_flag[0] != 0.f);
            const int steps = state[1] + (sk ? 0 : 1);
            const int t = steps > 0 ? steps : 1;
            const float ss = (float)(F.lr / (1.0 - pow(F.b1, (double)t))), bc = (float)sqrt(1.0 - pow(F.b2, (double)t));
            const int sy = (F.target_freq > 0 && (steps % F.target_freq) == 0) ? 1 : 0;
            sh_f[0] = ss; sh_f[1] = bc; sh_i[0] = sk; sh_i[1] = sy;
            if (blockIdx.x == 0) {
                if (sk) F.state_w[2] += 1;
                F.state_w[3] = sk; F.state_w[4] = sy; F.state_w[5] = steps;
                F.scal_w[0] = ss; F.scal_w[1] = bc;
            }
        }
        __syncthreads();
        skip = sh_i[0] != 0; sync = sh_i[1] != 0; step_size = sh_f[0]; bc2_sqrt = sh_f[1];
    } else {
        skip = state[3] != 0; sync = state[4] != 0; step_size = scal[0]; bc2_sqrt = scal[1];
    }
    if (skip && !sync) return;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long end = sync ? n_total : n;
    if (vec4) {                     // n, n_total multiples of four and every buffer 16-byte aligned: 16 bytes per lane, same arithmetic per element
        const long long n4 = n >> 2, end4 = end >> 2;
        for (; i < end4; i += stride) {
            a0_f4 pv = ((a0_f4*)p)[i];
            if (i < n4 && !skip) {
                const a0_f4 gv = ((const a0_f4*)g)[i];
                a0_f4 mv = ((a0_f4*)m)[i], vv = ((a0_f4*)v)[i];
                float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gi = gp[e];
                    float mi = mp[e], vi = vp[e];
                    mi = mi + (gi - mi) * w1;
                    vi = vi * b2 + (w2 * gi) * gi;
                    const float denom = sqrtf(vi) / bc2_sqrt + eps;
                    pp[e] = pp[e] - step_size * (mi / denom);
                    mp[e] = mi; vp[e] = vi;
                }
                ((a0_f4*)p)[i] = pv; ((a0_f4*)m)[i] = mv; ((a0_f4*)v)[i] = vv;
            }
            if (sync) ((a0_f4*)target)[i] = pv;
        }
        return;
    }
    for (; i < end; i += stride) {
        float pi = p[i];
        if (i < n && !skip) {
            const float gi = g[i];
            float mi = m[i], vi = v[i];
            mi = mi + (gi - mi) * w1;
            vi = vi * b2 + (w2 * gi) * gi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pi = pi - step_size * (mi / denom);
            p[i] = pi;
            m[i] = mi;
            v[i] = vi;
        }
        if (sync) target[i] = pi;
    }
}

extern "C" int a0_adam_step_sync(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int* state, float* scalars,
                                 double lr, double beta1, double beta2, double eps, int target_update_freq, float* target, long long n_total,
                                 const float* extra_nan_flag, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !state || !scalars || !target || n < 1 || n_total < n)
        return a0_fail(A0_EINVAL, "a0_adam_step_sync: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(a0_adam_prep_kernel, dim3(1), dim3(1), 0, st, state, scalars, lr, beta1, beta2, target_update_freq, extra_nan_flag);
    const int vec4 = ((n | n_total) % 4 == 0) &&
                     ((((uintptr_t)params) | ((uintptr_t)grads) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq) | ((uintptr_t)target)) % 16 == 0);
    long long blocks = ((vec4 ? n_total / 4 : n_total) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(a0_adam_sync_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, state, scalars,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, target, n_total, vec4, a0_adam_fold{0.0, 0.0, 0.0, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 1});
    return a0_fail_hip((int)hipGetLastError(), "a0_adam_step_sync");
}

// The optimizer tail of a network with fused-kernel weight copies, in TWO launches: Adam with the step's bookkeeping folded in (no one-thread
// prep kernel in front), then the refresh of the online net's weight copies (mirrored into the target's on a sync step), which also commits
// the step counter.  Same results as a0_adam_step_sync + a0_net_conv_wt_refresh_sync.
int a0_conv_wt_refresh_commit(const a0_encoder_weights* w, int C, float* wt, float* wt_target, int* state, hipStream_t st);      // encoder_fused.hip
extern "C" int a0_adam_step_sync_wt(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int* state, float* scalars,
                                    double lr, double beta1, double beta2, double eps, int target_update_freq, float* target, long long n_total,
                                    const float* extra_nan_flag, const a0_encoder_weights* w, int C, float* wt, float* wt_target, const float* loss, int loss_n,
                                    float* loss_ring, int ring_cap, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !state || !scalars || !target || n < 1 || n_total < n || !w || !wt || !wt_target || C < 1)
        return a0_fail(A0_EINVAL, "a0_adam_step_sync_wt: bad argument");
    if (loss && (!loss_ring || loss_n < 1 || ring_cap < 1)) return a0_fail(A0_EINVAL, "a0_adam_step_sync_wt: loss statistics need a ring of at least one slot");
    hipStream_t st = (hipStream_t)stream;
    const int vec4 = ((n | n_total) % 4 == 0) &&
                     ((((uintptr_t)params) | ((uintptr_t)grads) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq) | ((uintptr_t)target)) % 16 == 0);
    long long blocks = ((vec4 ? n_total / 4 : n_total) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(a0_adam_sync_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, (const int*)state, (const float*)scalars,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, target, n_total, vec4,
                       a0_adam_fold{lr, beta1, beta2, target_update_freq, extra_nan_flag, state, scalars, loss, loss_n, loss_ring, ring_cap > 0 ? ring_cap : 1});
    int e = a0_fail_hip((int)hipGetLastError(), "a0_adam_step_sync_wt");
    if (e != A0_OK) return e;
    return a0_conv_wt_refresh_commit(w, C, wt, wt_target, state, st);
}

// out[0] = 1.0f if this rank's NaN flag (state[0], set by the loss kernels) is up, else 0.0f — a float so that it can ride along in
// the SUM all-reduce of a gradient bucket (dist.GradAllReduce)
__global__ void a0_nan_flag_export_kernel(const int* __restrict__ state, float* __restrict__ out) { out[0] = state[0] != 0 ? 1.f : 0.f; }

extern "C" int a0_nan_flag_export(const int* state, float* out, void* stream) {
    if (!state || !out) return a0_fail(A0_EINVAL, "a0_nan_flag_export: bad argument");
    hipLaunchKernelGGL(a0_nan_flag_export_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, out);
    return a0_fail_hip((int)hipGetLastError(), "a0_nan_flag_export");
}

// RMSprop(lr, alpha, eps), no momentum, not centered (torch.optim.RMSprop defaults otherwise) — runs unconditionally,
// like the reference's fqf_optimizer.step() which sits before the NaN guard (agent.py:139-148).
__global__ void a0_rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, long long n,
                                  float lr, float alpha, float w, float eps, const float* __restrict__ clip_coef) {
    const float c = clip_coef ? clip_coef[0] : 1.f;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i] * c;
        float s = sq[i] * alpha + (w * gi) * gi;
        sq[i] = s;
        p[i] = p[i] - lr * (gi / (sqrtf(s) + eps));
    }
}

// clip_coef[0] = min(1, max_norm / (||g|| + 1e-6))   (torch.nn.utils.clip_grad_norm_), single workgroup
__global__ __launch_bounds__(256) void a0_clip_coef_kernel(const float* __restrict__ g, long long n, float max_norm, float* __restrict__ coef) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float c = max_norm / (sqrtf(red[0]) + 1e-6f);
        coef[0] = c < 1.f ? c : 1.f;
    }
}

extern "C" int a0_rmsprop_step(float* params, const float* grads, float* square_avg, long long n, double lr, double alpha, double eps,
                               double max_grad_norm, float* clip_scratch, void* stream) {
    if (!params || !grads || !square_avg || n < 1) return a0_fail(A0_EINVAL, "a0_rmsprop_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const float* coef = nullptr;
    if (max_grad_norm > 0) {
        if (!clip_scratch) return a0_fail(A0_EINVAL, "a0_rmsprop_step: clipping needs a 1-float scratch");
        hipLaunchKernelGGL(a0_clip_coef_kernel, dim3(1), dim3(256), 0, st, grads, n, (float)max_grad_norm, clip_scratch);
        coef = clip_scratch;
    }
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(a0_rmsprop_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, square_avg, n, (float)lr, (float)alpha,
                       (float)(1.0 - alpha), (float)eps, coef);
    return a0_fail_hip((int)hipGetLastError(), "a0_rmsprop_step");
}

// target <- online when state[4] says so (agent.py:160-161: deepcopy of the whole module, buffers included)
__global__ void a0_target_sync_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n, const int* __restrict__ state, int force) {
    if (!force && !state[4]) return;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;
    const a0_f4* s4 = (const a0_f4*)src;
    a0_f4* d4 = (a0_f4*)dst;
    for (long long j = i; j < n4; j += stride) d4[j] = s4[j];
    for (long long j = (n4 << 2) + i; j < n; j += stride) dst[j] = src[j];
}

extern "C" int a0_target_sync(float* target, const float* online, long long n, const int* state, int force, void* stream) {
    if (!target || !online || n < 1 || (!force && !state)) return a0_fail(A0_EINVAL, "a0_target_sync: bad argument");
    if ((((uintptr_t)target) | ((uintptr_t)online)) & 15) return a0_fail(A0_EINVAL, "a0_target_sync: buffers must be 16-byte aligned");
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(a0_target_sync_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, target, online, n, state, force);
    return a0_fail_hip((int)hipGetLastError(), "a0_target_sync");
}

// ------------------------------------------------------------------------------------------------ NoisyNet
// W = mu + sigma * (f(eps_out) x f(eps_in)),  b = mu_b + sigma_b * f(eps_b),  f(x) = sign(x) sqrt|x|
// (reference agent0/deepq/model.py:54-62,73-87).  Blocks are [W (N*K) | b (N)]; one call handles the rows [r0, r1) that
// belong to one NoisyLinear module (q_head and value_head share a packed block but have their own noise vectors).
A0_D float a0_noise_f(float x) { return (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)) * sqrtf(fabsf(x)); }

__global__ void a0_noisy_compose_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, float* __restrict__ eff, int N, int K,
                                        int r0, int r1, const float* __restrict__ noise_in, const float* __restrict__ noise_out_w,
                                        const float* __restrict__ noise_out_b) {
    const long long rows = r1 - r0;
    const long long total = rows * K + rows;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        long long off; float e;
        if (i < rows * K) {
            const int n = (int)(i / K), k = (int)(i % K);
            off = (long long)(r0 + n) * K + k;
            e = a0_noise_f(noise_out_w[n]) * a0_noise_f(noise_in[k]);
        } else {
            const int n = (int)(i - rows * K);
            off = (long long)N * K + r0 + n;
            e = a0_noise_f(noise_out_b[n]);
        }
        eff[off] = mu[off] + sigma[off] * e;
    }
}

// gradient fan-out: the weight-gradient kernels write d(eff) into the mu block (d mu = d eff); d sigma = d eff * eps
__global__ void a0_noisy_grad_sigma_kernel(const float* __restrict__ gmu, float* __restrict__ gsigma, int N, int K, int r0, int r1,
                                           const float* __restrict__ noise_in, const float* __restrict__ noise_out_w,
                                           const float* __restrict__ noise_out_b) {
    const long long rows = r1 - r0;
    const long long total = rows * K + rows;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        long long off; float e;
        if (i < rows * K) {
            const int n = (int)(i / K), k = (int)(i % K);
            off = (long long)(r0 + n) * K + k;
            e = a0_noise_f(noise_out_w[n]) * a0_noise_f(noise_in[k]);
        } else {
            const int n = (int)(i - rows * K);
            off = (long long)N * K + r0 + n;
            e = a0_noise_f(noise_out_b[n]);
        }
        gsigma[off] = gmu[off] * e;
    }
}

// Up to six NoisyLinear modules (first_dense, q_head, value_head of the online AND the target network: round 4) in one launch: workgroups [first_block[m], first_block[m+1]) serve module m.
struct a0_noisy_mod { const float* mu; const float* sigma; float* out; int N, K, r0, r1; const float* noise_in; const float* noise_out_w; const float* noise_out_b; };
struct a0_noisy_multi_args { a0_noisy_mod mod[6]; int first_block[7]; };

template <bool GRAD>
__global__ void a0_noisy_multi_kernel(a0_noisy_multi_args A) {
    int mi = 0;
#pragma unroll
    for (int k = 1; k < 6; ++k) mi += ((int)blockIdx.x >= A.first_block[k]) ? 1 : 0;
    const a0_noisy_mod M = A.mod[mi];
    const int nblk = A.first_block[mi + 1] - A.first_block[mi];
    const long long rows = M.r1 - M.r0;
    const long long total = rows * M.K + rows;
    long long i = (long long)((int)blockIdx.x - A.first_block[mi]) * blockDim.x + threadIdx.x;
    const long long stride = (long long)nblk * blockDim.x;
    for (; i < total; i += stride) {
        long long off; float e;
        if (i < rows * M.K) {
            const int n = (int)(i / M.K), k = (int)(i % M.K);
            off = (long long)(M.r0 + n) * M.K + k;
            e = a0_noise_f(M.noise_out_w[n]) * a0_noise_f(M.noise_in[k]);
        } else {
            const int n = (int)(i - rows * M.K);
            off = (long long)M.N * M.K + M.r0 + n;
            e = a0_noise_f(M.noise_out_b[n]);
        }
        if (GRAD) M.out[off] = M.mu[off] * e;                 // d sigma = d eff * eps (mu = the gradient written into the mu block)
        else M.out[off] = M.mu[off] + M.sigma[off] * e;       // eff = mu + sigma * eps
    }
}

// The same, four k per lane: 16-byte loads and stores, 32-bit index arithmetic (the element-wise kernel spends its time in a 64-bit division per element; this one
// runs at the HBM rate).  Every element is formed by the same expression as above; needs K % 4 == 0 and 16-byte aligned blocks / noise vectors (the packed layout's).
template <bool GRAD>
__global__ __launch_bounds__(256) void a0_noisy_multi_v4_kernel(a0_noisy_multi_args A) {
    int mi = 0;
#pragma unroll
    for (int k = 1; k < 6; ++k) mi += ((int)blockIdx.x >= A.first_block[k]) ? 1 : 0;
    const a0_noisy_mod M = A.mod[mi];
    const unsigned nblk = (unsigned)(A.first_block[mi + 1] - A.first_block[mi]);
    const unsigned rows = (unsigned)(M.r1 - M.r0), K4 = (unsigned)M.K >> 2, nw = rows * K4, total = nw + rows;
    const unsigned stride = nblk * 256u;
    for (unsigned i = ((unsigned)blockIdx.x - (unsigned)A.first_block[mi]) * 256u + threadIdx.x; i < total; i += stride) {
        if (i < nw) {
            const unsigned n = i / K4, k4 = i - n * K4;
            const long long off = (long long)(M.r0 + (int)n) * M.K + 4 * k4;
            const float eo = a0_noise_f(M.noise_out_w[n]);
            const a0_f4 ni = *(const a0_f4*)(M.noise_in + 4 * k4);
            const float e0 = eo * a0_noise_f(ni.x), e1 = eo * a0_noise_f(ni.y), e2 = eo * a0_noise_f(ni.z), e3 = eo * a0_noise_f(ni.w);
            const a0_f4 mu = *(const a0_f4*)(M.mu + off);
            a0_f4 o;
            if (GRAD) { o.x = mu.x * e0; o.y = mu.y * e1; o.z = mu.z * e2; o.w = mu.w * e3; }
            else {
                const a0_f4 sg = *(const a0_f4*)(M.sigma + off);
                o.x = mu.x + sg.x * e0; o.y = mu.y + sg.y * e1; o.z = mu.z + sg.z * e2; o.w = mu.w + sg.w * e3;
            }
            *(a0_f4*)(M.out + off) = o;
        } else {
            const unsigned n = i - nw;
            const long long off = (long long)M.N * M.K + M.r0 + (int)n;
            const float e = a0_noise_f(M.noise_out_b[n]);
            if (GRAD) M.out[off] = M.mu[off] * e;
            else M.out[off] = M.mu[off] + M.sigma[off] * e;
        }
    }
}

// nmod <= 6 modules; arrays are host arrays.  grad = 0: eff[m] = mu[m] + sigma[m] * eps (a0_noisy_compose per module);
// grad = 1: gsigma[m] (passed as eff) = gmu[m] (passed as mu) * eps (a0_noisy_grad_sigma per module; sigma unused).
extern "C" int a0_noisy_multi(int grad, int nmod, const float* const* mu, const float* const* sigma, float* const* eff, const int* N, const int* K, const int* r0,
                              const int* r1, const float* const* noise_in, const float* const* noise_out_w, const float* const* noise_out_b, void* stream) {
    if (nmod < 1 || nmod > 6 || !mu || !eff || !N || !K || !r0 || !r1 || !noise_in || !noise_out_w || !noise_out_b || (!grad && !sigma))
        return a0_fail(A0_EINVAL, "a0_noisy_multi: bad argument");
    a0_noisy_multi_args A;
    int blocks = 0;
    static const bool scalar_only = getenv("A0_NOISY_SCALAR") != nullptr;       // tuning aid: the element-wise kernel
    bool v4 = !scalar_only;
    for (int m = 0; m < nmod && v4; ++m)
        v4 = mu[m] && eff[m] && noise_in[m] && K[m] % 4 == 0 && (long long)N[m] * K[m] < (1LL << 31) &&
             (((uintptr_t)mu[m] | (uintptr_t)eff[m] | (uintptr_t)noise_in[m] | (grad ? 0 : (uintptr_t)sigma[m])) & 15) == 0;
    for (int m = 0; m < 6; ++m) {
        A.first_block[m] = blocks;
        if (m < nmod) {
            if (!mu[m] || !eff[m] || (!grad && !sigma[m]) || !noise_in[m] || !noise_out_w[m] || !noise_out_b[m] || N[m] < 1 || K[m] < 1 || r0[m] < 0 || r1[m] <= r0[m] || r1[m] > N[m])
                return a0_fail(A0_EINVAL, "a0_noisy_multi: bad module");
            A.mod[m] = a0_noisy_mod{mu[m], grad ? nullptr : sigma[m], eff[m], N[m], K[m], r0[m], r1[m], noise_in[m], noise_out_w[m], noise_out_b[m]};
            long long b = v4 ? ((long long)(r1[m] - r0[m]) * (K[m] / 4 + 1) + 255) / 256 : ((long long)(r1[m] - r0[m]) * (K[m] + 1) + 255) / 256;
            if (b > 2048) b = 2048;
            blocks += (int)b;
        } else {
            A.mod[m] = a0_noisy_mod{nullptr, nullptr, nullptr, 1, 1, 0, 0, nullptr, nullptr, nullptr};
        }
    }
    for (int m = nmod + 1; m <= 6; ++m) A.first_block[m] = 0x7fffffff;       // unused modules are never selected
    A.first_block[nmod] = blocks;                                              // end of the last real module
    if (v4 && grad) hipLaunchKernelGGL(a0_noisy_multi_v4_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A);
    else if (v4) hipLaunchKernelGGL(a0_noisy_multi_v4_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A);
    else if (grad) hipLaunchKernelGGL(a0_noisy_multi_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(a0_noisy_multi_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A);
    return a0_fail_hip((int)hipGetLastError(), "a0_noisy_multi");
}

extern "C" int a0_noisy_compose(const float* mu, const float* sigma, float* eff, int N, int K, int r0, int r1, const float* noise_in,
                                const float* noise_out_w, const float* noise_out_b, void* stream) {
    if (!mu || !sigma || !eff || !noise_in || !noise_out_w || !noise_out_b || N < 1 || K < 1 || r0 < 0 || r1 <= r0 || r1 > N) return a0_fail(A0_EINVAL, "a0_noisy_compose: bad argument");
    long long total = (long long)(r1 - r0) * (K + 1), blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(a0_noisy_compose_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mu, sigma, eff, N, K, r0, r1, noise_in, noise_out_w, noise_out_b);
    return a0_fail_hip((int)hipGetLastError(), "a0_noisy_compose");
}

extern "C" int a0_noisy_grad_sigma(const float* gmu, float* gsigma, int N, int K, int r0, int r1, const float* noise_in,
                                   const float* noise_out_w, const float* noise_out_b, void* stream) {
    if (!gmu || !gsigma || !noise_in || !noise_out_w || !noise_out_b || N < 1 || K < 1 || r0 < 0 || r1 <= r0 || r1 > N) return a0_fail(A0_EINVAL, "a0_noisy_grad_sigma: bad argument");
    long long total = (long long)(r1 - r0) * (K + 1), blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(a0_noisy_grad_sigma_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gmu, gsigma, N, K, r0, r1, noise_in, noise_out_w, noise_out_b);
    return a0_fail_hip((int)hipGetLastError(), "a0_noisy_grad_sigma");
}
