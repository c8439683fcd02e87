// IQN / FQF head pieces around the dense GEMMs (gfx950): cosine features, the tau-embedding Hadamard product and its
// backward, FQF's fraction proposal (softmax -> cumsum) and the fraction-loss surrogate gradient.
// Restates reference agent0/deepq/model.py:235-251 (IQNHead.feature_emb), model.py:268-278 (FQFHead.prop_taus) and
// agent0/deepq/agent.py:371-387 (fraction loss).
#include "a0_internal.h"
#include "philox.h"

// cosx[r][i] = cos((pi * (i+1)) * tau_r), i < D   — pi*(i+1) rounded to fp32 first, as `np.pi * torch.arange(1, D+1)` is
__global__ void a0_cos_features_kernel(const float* __restrict__ taus, float* __restrict__ out, long long R, int D) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * D) return;
    const long long r = i / D;
    const int d = (int)(i % D);
    const float ipi = 3.14159274101257324f * (float)(d + 1);
    out[i] = cosf(ipi * taus[r]);
}

extern "C" int a0_cos_features(const float* taus, float* out, long long R, int D, void* stream) {
    if (!taus || !out || R < 1 || D < 1) return a0_fail(A0_EINVAL, "a0_cos_features: bad argument");
    long long n = R * D;
    hipLaunchKernelGGL(a0_cos_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, taus, out, R, D);
    return a0_fail_hip((int)hipGetLastError(), "a0_cos_features");
}

// The actor's IQN step (round 6): the step's R = E * K fractions drawn from the Philox stream (element r of a0_rng_uniform: word (r & 3) of block (offset + r) >> 2) AND
// their cosine features in one launch — the same two formulas as a0_rng_uniform_kernel and a0_cos_features_kernel, hence the same bits.
__global__ void a0_tau_cos_features_kernel(unsigned long long seed, uint32_t stream, unsigned long long offset, const long long* __restrict__ ctrl, int ctrl_idx,
                                           float* __restrict__ taus, float* __restrict__ out, long long R, int D) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * D) return;
    if (ctrl) offset += (unsigned long long)ctrl[ctrl_idx];
    const long long r = i / D;
    const int d = (int)(i % D);
    const float tau = (float)(a0_philox_word(seed, stream, offset + (unsigned long long)r) >> 8) * 0x1.0p-24f;
    if (d == 0) taus[r] = tau;
    const float ipi = 3.14159274101257324f * (float)(d + 1);
    out[i] = cosf(ipi * tau);
}

extern "C" int a0_tau_cos_features(unsigned long long seed, unsigned int stream, unsigned long long offset, const long long* ctrl, int ctrl_idx, float* taus, float* out,
                                   long long R, int D, void* stream_h) {
    if (!taus || !out || R < 1 || D < 1 || (ctrl && (ctrl_idx < 0 || ctrl_idx >= A0_CTRL_WORDS))) return a0_fail(A0_EINVAL, "a0_tau_cos_features: bad argument");
    const long long n = R * D;
    hipLaunchKernelGGL(a0_tau_cos_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_h, seed, stream, offset, ctrl, ctrl_idx, taus, out, R, D);
    return a0_fail_hip((int)hipGetLastError(), "a0_tau_cos_features");
}

// x[(b,n)][d] = emb[(b,n)][d] * feat[b][d]       (16 B per lane; D % 4 == 0)
__global__ void a0_hadamard_fwd_kernel(const a0_f4* __restrict__ emb, const a0_f4* __restrict__ feat, a0_f4* __restrict__ x, int B, int n, int D4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * n * D4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long long row = i / D4;
        const int d = (int)(i % D4);
        const a0_f4 e = emb[i], f = feat[(row / n) * D4 + d];
        a0_f4 o; o.x = e.x * f.x; o.y = e.y * f.y; o.z = e.z * f.z; o.w = e.w * f.w;
        x[i] = o;
    }
}

extern "C" int a0_hadamard_fwd(const float* emb, const float* feat, float* x, int B, int n, int D, void* stream) {
    if (!emb || !feat || !x || B < 1 || n < 1 || D < 4 || (D & 3)) return a0_fail(A0_EINVAL, "a0_hadamard_fwd: bad argument");
    long long total = (long long)B * n * (D / 4), blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(a0_hadamard_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const a0_f4*)emb, (const a0_f4*)feat, (a0_f4*)x, B, n, D / 4);
    return a0_fail_hip((int)hipGetLastError(), "a0_hadamard_fwd");
}

// demb[(b,n)][d] = emb > 0 ? dx * feat[b][d] : 0       (gradient w.r.t. the cosine layer's pre-activation)
// d3[b][d]       = feat[b][d] > 0 ? sum_n dx[(b,n)][d] * emb[(b,n)][d] : 0     (w.r.t. conv3's pre-activation)
__global__ void a0_hadamard_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ emb, const float* __restrict__ feat,
                                       float* __restrict__ demb, float* __restrict__ d3, int B, int n, int D) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * D) return;
    const int b = (int)(i / D), d = (int)(i % D);
    const float f = feat[i];
    float s = 0.f;
    for (int k = 0; k < n; ++k) {
        const long long j = ((long long)b * n + k) * D + d;
        const float g = dx[j], e = emb[j];
        demb[j] = (e > 0.f) ? g * f : 0.f;
        s += g * e;
    }
    d3[i] = (f > 0.f) ? s : 0.f;
}

extern "C" int a0_hadamard_bwd(const float* dx, const float* emb, const float* feat, float* demb, float* d3, int B, int n, int D, void* stream) {
    if (!dx || !emb || !feat || !demb || !d3 || B < 1 || n < 1 || D < 1) return a0_fail(A0_EINVAL, "a0_hadamard_bwd: bad argument");
    long long total = (long long)B * D;
    hipLaunchKernelGGL(a0_hadamard_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, emb, feat, demb, d3, B, n, D);
    return a0_fail_hip((int)hipGetLastError(), "a0_hadamard_bwd");
}

// taus[b][0] = 0, taus[b][i+1] = cumsum_i softmax(logits[b]);  tau_hat[b][i] = (taus[i] + taus[i+1]) / 2     (F <= 64)
// cos_out (optional, round 6: the actor's step): the cosine features [B * F][D] of the tau_hats in the same launch (== a0_cos_features on tau_hat)
__global__ __launch_bounds__(64) void a0_fqf_taus_kernel(const float* __restrict__ logits, int ld, float* __restrict__ taus, float* __restrict__ tau_hat, int B, int F,
                                                          float* __restrict__ cos_out, int D) {
    __shared__ float p[64];
    __shared__ float th[64];
    const int b = blockIdx.x, t = threadIdx.x;
    const float x = (t < F) ? logits[(long long)b * ld + t] : -INFINITY;
    float mx = x;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = (t < F) ? expf(x - mx) : 0.f;
    float s = e;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float logp = x - mx - logf(s);
    p[t] = (t < F) ? expf(logp) : 0.f;
    __syncthreads();
    if (t == 0) {
        float c = 0.f;
        taus[(long long)b * (F + 1)] = 0.f;
        for (int i = 0; i < F; ++i) {
            const float prev = c;
            c += p[i];
            taus[(long long)b * (F + 1) + i + 1] = c;
            const float h = (prev + c) / 2.0f;
            tau_hat[(long long)b * F + i] = h;
            th[i] = h;
        }
    }
    if (cos_out) {
        __syncthreads();
        for (int k = t; k < F * D; k += 64) {
            const int i = k / D, d = k - i * D;
            const float ipi = 3.14159274101257324f * (float)(d + 1);
            cos_out[(long long)b * F * D + k] = cosf(ipi * th[i]);
        }
    }
}

extern "C" int a0_fqf_taus(const float* logits, int ld, float* taus, float* tau_hat, int B, int F, void* stream) {
    if (!logits || !taus || !tau_hat || B < 1 || F < 2 || F > 64 || ld < F) return a0_fail(A0_EINVAL, "a0_fqf_taus: bad argument");
    hipLaunchKernelGGL(a0_fqf_taus_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ld, taus, tau_hat, B, F, (float*)nullptr, 0);
    return a0_fail_hip((int)hipGetLastError(), "a0_fqf_taus");
}

extern "C" int a0_fqf_taus_cos(const float* logits, int ld, float* taus, float* tau_hat, float* cos_out, int D, int B, int F, void* stream) {
    if (!logits || !taus || !tau_hat || !cos_out || D < 1 || B < 1 || F < 2 || F > 64 || ld < F) return a0_fail(A0_EINVAL, "a0_fqf_taus_cos: bad argument");
    hipLaunchKernelGGL(a0_fqf_taus_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ld, taus, tau_hat, B, F, cos_out, D);
    return a0_fail_hip((int)hipGetLastError(), "a0_fqf_taus_cos");
}

// out[b][i] = taus[b][i+1], i < F-1     (taus[:, 1:-1])
__global__ void a0_fqf_inner_taus_kernel(const float* __restrict__ taus, float* __restrict__ out, int B, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (F - 1)) return;
    const int b = i / (F - 1), k = i % (F - 1);
    out[i] = taus[(long long)b * (F + 1) + k + 1];
}

extern "C" int a0_fqf_inner_taus(const float* taus, float* out, int B, int F, void* stream) {
    if (!taus || !out || B < 1 || F < 2) return a0_fail(A0_EINVAL, "a0_fqf_inner_taus: bad argument");
    const int n = B * (F - 1);
    hipLaunchKernelGGL(a0_fqf_inner_taus_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, taus, out, B, F);
    return a0_fail_hip((int)hipGetLastError(), "a0_fqf_inner_taus");
}

// Fraction loss (agent.py:371-387) and its gradient w.r.t. the fraction-net logits.  One wave per sample.
//   q   [B][F-1][A]  quantile values at the interior taus (no grad),  qh [B][F][A] values at the tau-hats
//   g_i = (q_i > prev_i ? v1 : -v1) + (q_i < next_i ? v2 : -v2),  v1 = q_i - qh_i,  v2 = q_i - qh_{i+1}
//   loss_b = sum_i g_i * tau_{i+1};   d(w_b loss_b)/dp_k = w_b sum_{i>=k} g_i;   dlogit_k = p_k (dp_k - sum_j p_j dp_j)
__global__ __launch_bounds__(64) void a0_fqf_fraction_loss_kernel(const float* __restrict__ q, const float* __restrict__ qh, const float* __restrict__ taus,
                                                                   const int* __restrict__ act, const float* __restrict__ wgt, int B, int F, int A, int ldl,
                                                                   float* __restrict__ loss, float* __restrict__ dlogits, const float* __restrict__ logits) {
    __shared__ float sq[64], sh[65], sg[64], sp[64], sdp[64];
    const int b = blockIdx.x, t = threadIdx.x, a = act[b];
    sq[t] = (t < F - 1) ? q[((long long)b * (F - 1) + t) * A + a] : 0.f;
    sh[t] = (t < F) ? qh[((long long)b * F + t) * A + a] : 0.f;
    // softmax of the fraction logits
    const float x = (t < F) ? logits[(long long)b * ldl + t] : -INFINITY;
    float mx = x;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = (t < F) ? expf(x - mx) : 0.f;
    float s = e;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    sp[t] = e / s;
    __syncthreads();
    float g = 0.f;
    if (t < F - 1) {
        const float qi = sq[t];
        const float prev = (t == 0) ? sh[0] : sq[t - 1];
        const float next = (t == F - 2) ? sh[F - 1] : sq[t + 1];
        const float v1 = qi - sh[t], v2 = qi - sh[t + 1];
        g = ((qi > prev) ? v1 : -v1) + ((qi < next) ? v2 : -v2);
    }
    sg[t] = g;
    __syncthreads();
    float l = (t < F - 1) ? g * taus[(long long)b * (F + 1) + t + 1] : 0.f;
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if (t == 0) loss[b] = l;
    // dp_k = w * sum_{i=k}^{F-2} g_i
    float dp = 0.f;
    if (t < F) for (int i = t; i < F - 1; ++i) dp += sg[i];
    dp *= wgt[b];
    sdp[t] = dp;
    float dot = (t < F) ? sp[t] * dp : 0.f;
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (t < ldl) dlogits[(long long)b * ldl + t] = (t < F) ? sp[t] * (dp - dot) : 0.f;
}

extern "C" int a0_fqf_fraction_loss(const float* q, const float* qh, const float* taus, const int* act, const float* wgt, int B, int F, int A,
                                    int ldl, float* loss, float* dlogits, const float* logits, void* stream) {
    if (!q || !qh || !taus || !act || !wgt || !loss || !dlogits || !logits || B < 1 || F < 2 || F > 64 || ldl < F || ldl > 64 || A < 1)
        return a0_fail(A0_EINVAL, "a0_fqf_fraction_loss: bad argument (2 <= F <= ldl <= 64)");
    hipLaunchKernelGGL(a0_fqf_fraction_loss_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, q, qh, taus, act, wgt, B, F, A, ldl, loss, dlogits, logits);
    return a0_fail_hip((int)hipGetLastError(), "a0_fqf_fraction_loss");
}
