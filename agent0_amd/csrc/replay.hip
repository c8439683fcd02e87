// HBM-resident replay: ring-buffer insert, indexed gather, flat-priority bookkeeping, importance weights,
// uniform-permutation index generator, fp32 sum-tree (gfx950).
//
// Replaces reference agent0/deepq/replay.py:14-59 (deque of lz4 blobs + torch priority vector on the host),
// the DataLoader/DataPrefetcher path of agent0/deepq/trainer.py:63-72 + agent0/common/utils.py:31-61 and the
// importance-weight block trainer.py:91-96.  Frames are stored uncompressed exactly as the actor packs them
// (agent.py:78-81): slot -> [8][H][W] u8 = st || st_next; 56 448 B per transition at 84x84, so 1 M transitions
// = 56.4 GB of the 288 GB HBM.  Integer / index results are bit-exact against oracle/sumtree.c.
#include "a0_internal.h"
#include "philox.h"

#include <cstdlib>

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------ insert / gather
// One workgroup column per transition, 16 B per lane.  obs / obs_next: [n][obs_bytes] u8; ring slot (start+i) % cap.
__global__ __launch_bounds__(256) void a0_replay_insert_kernel(uint8_t* __restrict__ frames, long long cap, int obs_bytes, long long start, int n,
                                                                const uint8_t* __restrict__ obs, const uint8_t* __restrict__ obs_next,
                                                                const int* __restrict__ act, const float* __restrict__ rew, const float* __restrict__ done,
                                                                int* __restrict__ r_act, float* __restrict__ r_rew, float* __restrict__ r_done,
                                                                const long long* __restrict__ ctrl) {
    const int i = blockIdx.y;
    if (ctrl) start += ctrl[A0_CTRL_REPLAY_SLOT];
    const long long slot = (start + i) % cap;
    const int q = obs_bytes >> 4;   // 16-byte groups per observation
    const uint4* s0 = (const uint4*)(obs + (long long)i * obs_bytes);
    const uint4* s1 = (const uint4*)(obs_next + (long long)i * obs_bytes);
    uint4* d = (uint4*)(frames + slot * (2LL * obs_bytes));
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < 2 * q; j += gridDim.x * blockDim.x) d[j] = (j < q) ? s0[j] : s1[j - q];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        r_act[slot] = act[i];
        r_rew[slot] = rew[i];
        r_done[slot] = done[i];
    }
}

extern "C" int a0_replay_insert(uint8_t* frames, long long cap, int obs_bytes, long long start_slot, int n, const uint8_t* obs, const uint8_t* obs_next,
                                const int* act, const float* rew, const float* done, int* r_act, float* r_rew, float* r_done, const long long* ctrl,
                                void* stream) {
    if (!frames || !obs || !obs_next || !act || !rew || !done || !r_act || !r_rew || !r_done || cap < 1 || n < 1 || n > cap || (obs_bytes & 15) || start_slot < 0)
        return a0_fail(A0_EINVAL, "a0_replay_insert: bad argument (obs_bytes must be a multiple of 16)");
    if ((((uintptr_t)frames) | ((uintptr_t)obs) | ((uintptr_t)obs_next)) & 15) return a0_fail(A0_EINVAL, "a0_replay_insert: buffers must be 16-byte aligned");
    const int q2 = 2 * (obs_bytes >> 4);
    int gx = (q2 + 255) / 256; if (gx > 8) gx = 8;
    hipLaunchKernelGGL(a0_replay_insert_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, frames, cap, obs_bytes, start_slot % cap, n, obs, obs_next,
                       act, rew, done, r_act, r_rew, r_done, ctrl);
    return a0_fail_hip((int)hipGetLastError(), "a0_replay_insert");
}

// logical deque index -> ring slot (reference deque(maxlen) semantics): slot = (head + idx % top) % cap
__global__ void a0_replay_slots_kernel(const long long* __restrict__ idx, int B, long long top, long long head, long long cap, int* __restrict__ slot,
                                       const int* __restrict__ r_act, const float* __restrict__ r_rew, const float* __restrict__ r_done,
                                       const float* __restrict__ priority, int* __restrict__ act, float* __restrict__ rew, float* __restrict__ done,
                                       float* __restrict__ prio, long long* __restrict__ idx_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long long li = idx[b] % top;
    const long long s = (head + li) % cap;
    slot[b] = (int)s;
    act[b] = r_act[s];
    rew[b] = r_rew[s];
    done[b] = r_done[s];
    if (prio) prio[b] = priority ? priority[li] : 1.f;
    if (idx_out) idx_out[b] = li;
}

extern "C" int a0_replay_lookup(const long long* idx, int B, long long top, long long head, long long cap, int* slot, const int* r_act,
                                const float* r_rew, const float* r_done, const float* priority, int* act, float* rew, float* done, float* prio,
                                long long* idx_out, void* stream) {
    if (!idx || !slot || !r_act || !r_rew || !r_done || !act || !rew || !done || B < 1 || top < 1 || cap < top || cap > 2147483647LL) return a0_fail(A0_EINVAL, "a0_replay_lookup: bad argument");
    hipLaunchKernelGGL(a0_replay_slots_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, idx, B, top, head, cap, slot, r_act, r_rew, r_done,
                       priority, act, rew, done, prio, idx_out);
    return a0_fail_hip((int)hipGetLastError(), "a0_replay_lookup");
}

// out[b] = frames[slot[b]]  (row_bytes each, 16 B per lane, rows are contiguous so every load is a full line)
__global__ __launch_bounds__(256) void a0_replay_gather_kernel(const uint8_t* __restrict__ frames, int row_bytes, const int* __restrict__ slot,
                                                                uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int q = row_bytes >> 4;
    const uint4* s = (const uint4*)(frames + (long long)slot[b] * row_bytes);
    uint4* d = (uint4*)(out + (long long)b * row_bytes);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < q; j += gridDim.x * blockDim.x) d[j] = s[j];
}

extern "C" int a0_replay_gather(const uint8_t* frames, int row_bytes, const int* slot, int B, uint8_t* out, void* stream) {
    if (!frames || !slot || !out || B < 1 || (row_bytes & 15)) return a0_fail(A0_EINVAL, "a0_replay_gather: bad argument");
    int gx = ((row_bytes >> 4) + 255) / 256; if (gx > 4) gx = 4;
    hipLaunchKernelGGL(a0_replay_gather_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, frames, row_bytes, slot, out);
    return a0_fail_hip((int)hipGetLastError(), "a0_replay_gather");
}

// ------------------------------------------------------------------------------------------------ flat priority vector (reference semantics)
__global__ void a0_fill_kernel(float* __restrict__ p, long long n, float v) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

extern "C" int a0_fill_f32(float* p, long long n, float v, void* stream) {
    if (!p || n < 1) return a0_fail(A0_EINVAL, "a0_fill_f32: bad argument");
    long long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(a0_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, n, v);
    return a0_fail_hip((int)hipGetLastError(), "a0_fill_f32");
}

// torch lowers pow(x, 0.5) to a correctly rounded sqrt; fp64 sqrt rounded to fp32 is correctly rounded too (53 >= 2*24 + 2),
// whatever the device's v_sqrt_f32 does
A0_D float a0_prio_pow(float x, float alpha) { return (alpha == 0.5f) ? (float)sqrt((double)x) : powf(x, alpha); }

// priority[ids] = (loss + eps)^alpha in batch order (a later duplicate wins); max_p[0] = max(max_p, max loss)
// pstate: [0] max_p (float).  Single workgroup so that the duplicate rule is deterministic.
__global__ __launch_bounds__(1024) void a0_priority_update_kernel(float* __restrict__ priority, const long long* __restrict__ ids,
                                                                   const float* __restrict__ loss, int B, float eps, float alpha,
                                                                   float* __restrict__ pstate, const int* __restrict__ state) {
    if (state && state[3]) return;   // the update was skipped (NaN): the reference would have raised here; we leave priorities alone
    __shared__ float red[1024];
    float mx = -INFINITY;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const long long id = ids[b];
        bool last = true;
        for (int c = b + 1; c < B; ++c) if (ids[c] == id) { last = false; break; }
        const float l = loss[b];
        if (last) priority[id] = a0_prio_pow(l + eps, alpha);
        mx = fmaxf(mx, l);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) pstate[0] = fmaxf(pstate[0], red[0]);
}

extern "C" int a0_priority_update(float* priority, const long long* ids, const float* loss, int B, float eps, float alpha, float* pstate,
                                  const int* state, void* stream) {
    if (!priority || !ids || !loss || !pstate || B < 1) return a0_fail(A0_EINVAL, "a0_priority_update: bad argument");
    hipLaunchKernelGGL(a0_priority_update_kernel, dim3(1), dim3(B >= 1024 ? 1024 : ((B + 63) / 64) * 64), 0, (hipStream_t)stream, priority, ids, loss, B, eps, alpha, pstate, state);
    return a0_fail_hip((int)hipGetLastError(), "a0_priority_update");
}

// priority[-n:] = max_p ^ alpha  (the reference's tail write, replay.py:51-52, quirk Q1)
__global__ void a0_priority_tail_kernel(float* __restrict__ priority, long long size, long long n, const float* __restrict__ pstate, float alpha) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    priority[size - n + i] = (float)pow((double)pstate[0], (double)alpha);   // python float ** float, then cast to fp32
}

extern "C" int a0_priority_tail(float* priority, long long size, long long n, const float* pstate, float alpha, void* stream) {
    if (!priority || !pstate || n < 1 || n > size) return a0_fail(A0_EINVAL, "a0_priority_tail: bad argument");
    hipLaunchKernelGGL(a0_priority_tail_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, priority, size, n, pstate, alpha);
    return a0_fail_hip((int)hipGetLastError(), "a0_priority_tail");
}

// sum of a float vector, deterministic two-level tree (fixed grid), result in out[0]
__global__ __launch_bounds__(256) void a0_sum_stage1(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void a0_sum_stage2(const float* __restrict__ part, int np, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

extern "C" int a0_sum_f32(const float* x, long long n, float* scratch256, float* out, void* stream) {
    if (!x || !scratch256 || !out || n < 1) return a0_fail(A0_EINVAL, "a0_sum_f32: bad argument");
    int blocks = (int)((n + 255) / 256); if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(a0_sum_stage1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scratch256);
    hipLaunchKernelGGL(a0_sum_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch256, blocks, out);
    return a0_fail_hip((int)hipGetLastError(), "a0_sum_f32");
}

// w = (top * p / sum)^(-beta);  w /= (max w + 1e-8)     (trainer.py:91-94).  Single workgroup.
__global__ __launch_bounds__(1024) void a0_is_weights_kernel(const float* __restrict__ prio, int B, const float* __restrict__ psum, float top, float beta,
                                                              float* __restrict__ w) {
    __shared__ float red[1024];
    const float total = psum[0];
    float mx = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float probs = prio[b] / total;
        const float v = powf(top * probs, -beta);
        w[b] = v;
        mx = fmaxf(mx, v);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    const float denom = red[0] + 1e-8f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) w[b] = w[b] / denom;
}

extern "C" int a0_is_weights(const float* prio, int B, const float* psum, long long top, float beta, float* w, void* stream) {
    if (!prio || !psum || !w || B < 1 || top < 1) return a0_fail(A0_EINVAL, "a0_is_weights: bad argument");
    int threads = 64; while (threads < B && threads < 1024) threads <<= 1;
    hipLaunchKernelGGL(a0_is_weights_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, prio, B, psum, (float)top, beta, w);
    return a0_fail_hip((int)hipGetLastError(), "a0_is_weights");
}

// ------------------------------------------------------------------------------------------------ uniform permutation
// 4-round Feistel bijection on [0, 2^(2h)) with cycle walking — must equal oracle/sumtree.c a0o_perm_index.
A0_D uint32_t a0_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

__global__ void a0_perm_kernel(unsigned long long start, int count, unsigned long long n, uint32_t seed, long long* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint32_t h = 1;
    while ((1ull << (2 * h)) < n) ++h;
    const uint32_t mask = (uint32_t)((1ull << h) - 1);
    unsigned long long x = start + (unsigned long long)k;
    do {
        uint32_t l = (uint32_t)(x >> h) & mask, r = (uint32_t)x & mask;
        for (uint32_t round = 0; round < 4; ++round) {
            const uint32_t f = a0_mix32(r ^ (seed + 0x9E3779B9u * (round + 1))) & mask;
            const uint32_t nl = r, nr = l ^ f;
            l = nl; r = nr;
        }
        x = ((unsigned long long)l << h) | r;
    } while (x >= n);
    out[k] = (long long)x;
}

extern "C" int a0_perm_batch(unsigned long long start, int count, unsigned long long n, unsigned int seed, long long* out, void* stream) {
    if (!out || count < 1 || n < 1 || start + (unsigned long long)count > n) return a0_fail(A0_EINVAL, "a0_perm_batch: bad argument");
    hipLaunchKernelGGL(a0_perm_kernel, dim3((count + 127) / 128), dim3(128), 0, (hipStream_t)stream, start, count, n, seed, out);
    return a0_fail_hip((int)hipGetLastError(), "a0_perm_batch");
}

// ------------------------------------------------------------------------------------------------ sum-tree
// Contract: oracle/sumtree.c.  tree[1] root, leaf i at tree[cap2 + i]; ancestors recomputed as left + right.
// Top of the tree in LDS.  Every level costs one global round trip (store, fence, barrier, load) when it is recomputed in place; the
// levels with at most A0_ST_TOP nodes are instead recomputed WHOLE from a copy of level A0_ST_TOP staged in LDS (all of its parents, not
// only the touched ones — by the tree's invariant the untouched ones come out unchanged, bit for bit) and written back once: eleven of a
// 1 M-leaf tree's twenty levels become LDS work.  Call with level `s0` = min(cap2, A0_ST_TOP) up to date in global memory and a barrier
// behind its last writes.
constexpr int A0_ST_TOP = 2048;
A0_D void a0_sumtree_top(float* __restrict__ tree, long long s0, float* __restrict__ lds) {
    for (long long p = threadIdx.x; p < s0; p += blockDim.x) lds[s0 + p] = tree[s0 + p];
    __syncthreads();
    for (long long span = s0 >> 1; span >= 1; span >>= 1) {
        for (long long p = span + threadIdx.x; p < 2 * span; p += blockDim.x) lds[p] = lds[2 * p] + lds[2 * p + 1];
        __syncthreads();
    }
    for (long long p = 1 + threadIdx.x; p < s0; p += blockDim.x) tree[p] = lds[p];
}

__global__ __launch_bounds__(1024) void a0_sumtree_set_kernel(float* __restrict__ tree, long long cap2, const long long* __restrict__ idx,
                                                               const float* __restrict__ val, int n, const int* __restrict__ state) {
    if (state && state[3]) return;      // the update was skipped on a NaN loss (agent.py:152-158): nothing to write, like a0_priority_update
    // single workgroup: leaves first (a later duplicate wins), then one level per barrier, bottom-up.  The indices are staged in LDS
    // once: the duplicate scan and the twenty levels re-read them from there instead of from global memory.
    __shared__ long long sidx[1024];
    for (int i = threadIdx.x; i < n; i += blockDim.x) sidx[i] = idx[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long long id = sidx[i];
        bool last = true;
        for (int c = i + 1; c < n; ++c) if (sidx[c] == id) { last = false; break; }
        if (last) tree[cap2 + id] = val[i];
    }
    __threadfence_block();
    __syncthreads();
    __shared__ float top[2 * A0_ST_TOP];
    const long long s0 = cap2 < A0_ST_TOP ? cap2 : A0_ST_TOP;
    for (long long span = cap2 >> 1; span >= s0; span >>= 1) {  // span = number of nodes on the level being recomputed
        const int shift = __builtin_ctzll(cap2 / span);          // leaf -> ancestor on this level
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const long long p = (cap2 + sidx[i]) >> shift;
            tree[p] = tree[2 * p] + tree[2 * p + 1];             // identical value from every writer of p
        }
        __threadfence_block();
        __syncthreads();
    }
    a0_sumtree_top(tree, s0, top);
}

// Leaves (start + i) % size, i < n, all set to val[0] — what ReplayDataset.extend does for a rollout's new transitions (replay.py:45-53:
// new entries get max_p^alpha) — then every affected ancestor recomputed level by level.  The leaves are one ring range (two node ranges
// per level when it wraps), so a level is a contiguous loop instead of n scattered updates: one launch for a 20 480-transition rollout
// where the batch kernel needed twenty.  Same tree as a0_sumtree_set with those (idx, val) pairs (oracle/sumtree.c).
__global__ __launch_bounds__(1024) void a0_sumtree_set_range_kernel(float* __restrict__ tree, long long cap2, long long start, long long n, long long size,
                                                                     const float* __restrict__ val) {
    const float v = val[0];
    const long long n1 = (start + n <= size) ? n : size - start;       // [start, start + n1) and, when the ring wraps, [0, n - n1)
    for (long long i = threadIdx.x; i < n; i += blockDim.x) tree[cap2 + (i < n1 ? start + i : i - n1)] = v;
    __threadfence_block();
    __syncthreads();
    __shared__ float top[2 * A0_ST_TOP];
    const long long s0 = cap2 < A0_ST_TOP ? cap2 : A0_ST_TOP;
    for (long long span = cap2 >> 1; span >= s0; span >>= 1) {
        const int shift = __builtin_ctzll(cap2 / span);
        const long long a0 = (cap2 + start) >> shift, a1 = (cap2 + start + n1 - 1) >> shift;
        const long long ca = a1 - a0 + 1;
        long long cb = 0, b0 = 0;
        if (n1 < n) { b0 = cap2 >> shift; cb = ((cap2 + (n - n1) - 1) >> shift) - b0 + 1; }
        for (long long i = threadIdx.x; i < ca + cb; i += blockDim.x) {
            const long long p = i < ca ? a0 + i : b0 + (i - ca);
            tree[p] = tree[2 * p] + tree[2 * p + 1];             // a node covered by both ranges gets the same value twice
        }
        __threadfence_block();
        __syncthreads();
    }
    a0_sumtree_top(tree, s0, top);
}

extern "C" int a0_sumtree_set_range(float* tree, long long cap2, long long start, long long n, long long size, const float* val, void* stream) {
    if (!tree || !val || n < 1 || size < 1 || n > size || start < 0 || start >= size || cap2 < size || (cap2 & (cap2 - 1)))
        return a0_fail(A0_EINVAL, "a0_sumtree_set_range: bad argument");
    hipLaunchKernelGGL(a0_sumtree_set_range_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tree, cap2, start, n, size, val);
    return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_set_range");
}

// ---- the same update for a large tree in two launches and many workgroups.  Below the level of A0_ST_TOP nodes the tree is A0_ST_TOP
// independent subtrees of S = cap2 / A0_ST_TOP leaves.  One WAVE per updated leaf rebuilds the whole subtree that leaf belongs to: it reads
// the S leaves, applies EVERY update that falls into the subtree in batch order (a later duplicate wins, as in the single-workgroup
// kernel) — so it does not depend on what other workgroups have written — and recomputes all S - 1 internal nodes as left + right in
// registers / across lanes (untouched nodes come out unchanged, bit for bit).  Waves whose subtree an earlier entry of the batch already
// names exit at once.  No level needs a global round trip; the top A0_ST_TOP levels follow in a second launch (a0_sumtree_top).  20 levels
// x (store, fence, barrier, load) on one workgroup took 47-56 us for 512 leaves of a 1 M-leaf tree.
template <int M>       // leaves per lane: S = 64 * M
__global__ __launch_bounds__(64) void a0_sumtree_set_sub_kernel(float* __restrict__ tree, long long cap2, const long long* __restrict__ idx,
                                                                 const float* __restrict__ val, int n, const int* __restrict__ state,
                                                                 const float* __restrict__ loss, float eps, float alpha, float* __restrict__ pstate) {
    // loss != NULL (round 4, a0_sumtree_set_from_loss): the values are (loss + eps)^alpha, formed here while the batch is staged (== a0_sumtree_prio_kernel:
    // one launch less per prioritized update), and workgroup 0 keeps max_p; val is unused then
    if (state && state[3]) return;
    constexpr int S = 64 * M, LS = __builtin_ctz(S);
    __shared__ int sidx[1024];             // leaf indices fit 31 bits for every tree this path takes (S <= 1024: cap2 <= 2 M)
    __shared__ float sval[1024];
    __shared__ float leaf[S];
    __shared__ int win[S];
    const int lane = threadIdx.x, i = blockIdx.x;
    if (loss) {
        float mx = -INFINITY;
        for (int j = lane; j < n; j += 64) { const float l = loss[j]; sidx[j] = (int)idx[j]; sval[j] = a0_prio_pow(l + eps, alpha); mx = fmaxf(mx, l); }
        if (i == 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (lane == 0) pstate[0] = fmaxf(pstate[0], mx);
        }
    } else {
        for (int j = lane; j < n; j += 64) { sidx[j] = (int)idx[j]; sval[j] = val[j]; }
    }
    __syncthreads();                        // (one wave per workgroup: the barriers here only order its LDS accesses)
    const int t = sidx[i] >> LS;                                        // this wave's subtree
    bool dup = false;
    for (int j = lane; j < i; j += 64) dup |= (sidx[j] >> LS) == t;
    if (__any(dup)) return;                                             // an earlier entry names it: that wave does the work (wave-uniform)
    const long long leaf0 = cap2 + (long long)t * S;
    for (int k = lane; k < S; k += 64) { leaf[k] = tree[leaf0 + k]; win[k] = -1; }
    __syncthreads();
    // every update that falls into this subtree, in batch order: the LAST entry naming a leaf wins
    for (int j = lane; j < n; j += 64) if ((sidx[j] >> LS) == t) atomicMax(&win[sidx[j] & (S - 1)], j);
    __syncthreads();
    for (int j = lane; j < n; j += 64) if ((sidx[j] >> LS) == t && win[sidx[j] & (S - 1)] == j) leaf[sidx[j] & (S - 1)] = sval[j];
    __syncthreads();
    float v[M];
#pragma unroll
    for (int k = 0; k < M; ++k) { v[k] = leaf[lane * M + k]; tree[leaf0 + lane * M + k] = v[k]; }
    // levels inside a lane
    int level = 0;
#pragma unroll
    for (int w = M / 2; w >= 1; w /= 2) {
        ++level;
        const long long base = (cap2 >> level) + t * (S >> level) + (long long)lane * w;
#pragma unroll
        for (int k = 0; k < w; ++k) { v[k] = v[2 * k] + v[2 * k + 1]; tree[base + k] = v[k]; }
    }
    // levels across lanes: after step d the lanes with (lane & (2d - 1)) == 0 hold a node
    float x = v[0];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        ++level;
        const float o = __shfl_xor(x, d, 64);
        x = x + o;                                                     // fp32 addition commutes: left + right either way
        if ((lane & (2 * d - 1)) == 0) tree[(cap2 >> level) + t * (S >> level) + (lane / (2 * d))] = x;
    }
}

__global__ __launch_bounds__(1024) void a0_sumtree_top_kernel(float* __restrict__ tree, const int* __restrict__ state) {
    if (state && state[3]) return;
    __shared__ float top[2 * A0_ST_TOP];
    a0_sumtree_top(tree, A0_ST_TOP, top);
}

static bool a0_sumtree_sub_path(long long cap2) {
    const long long S = cap2 / A0_ST_TOP;        // leaves per subtree below the LDS-resident top
    static const bool one_wg = getenv("A0_SUMTREE_ONE_WG") != nullptr;      // tuning aid: the single-workgroup kernel for every size
    return !one_wg && (S == 64 || S == 128 || S == 256 || S == 512 || S == 1024);
}
static void a0_sumtree_sub_launch(float* tree, long long cap2, const long long* idx, const float* val, int n, const int* state, const float* loss, float eps, float alpha,
                                  float* pstate, hipStream_t st, bool defer_top = false) {
    switch ((int)(cap2 / A0_ST_TOP)) {
        case 64: hipLaunchKernelGGL(a0_sumtree_set_sub_kernel<1>, dim3(n), dim3(64), 0, st, tree, cap2, idx, val, n, state, loss, eps, alpha, pstate); break;
        case 128: hipLaunchKernelGGL(a0_sumtree_set_sub_kernel<2>, dim3(n), dim3(64), 0, st, tree, cap2, idx, val, n, state, loss, eps, alpha, pstate); break;
        case 256: hipLaunchKernelGGL(a0_sumtree_set_sub_kernel<4>, dim3(n), dim3(64), 0, st, tree, cap2, idx, val, n, state, loss, eps, alpha, pstate); break;
        case 512: hipLaunchKernelGGL(a0_sumtree_set_sub_kernel<8>, dim3(n), dim3(64), 0, st, tree, cap2, idx, val, n, state, loss, eps, alpha, pstate); break;
        default: hipLaunchKernelGGL(a0_sumtree_set_sub_kernel<16>, dim3(n), dim3(64), 0, st, tree, cap2, idx, val, n, state, loss, eps, alpha, pstate); break;
    }
    if (!defer_top) hipLaunchKernelGGL(a0_sumtree_top_kernel, dim3(1), dim3(1024), 0, st, tree, state);
}

// replay.py:55-59 on the sum-tree in TWO launches (was three): leaf[idx] = (loss + eps)^alpha in batch order, max_p = max(max_p, max loss), subtrees, then the top.
// Trees whose subtrees the per-wave kernel does not cover (a0_sumtree_set_from_loss_ok == 0) take a0_priority_from_loss + a0_sumtree_set.
extern "C" int a0_sumtree_set_from_loss_ok(long long cap2) { return (cap2 >= 1 && !(cap2 & (cap2 - 1)) && a0_sumtree_sub_path(cap2)) ? 1 : 0; }

extern "C" int a0_sumtree_set_from_loss(float* tree, long long cap2, const long long* idx, const float* loss, int n, float eps, float alpha, float* pstate, const int* state,
                                        int defer_top, void* stream) {
    if (!tree || !idx || !loss || !pstate || n < 1 || n > 1024 || cap2 < 1 || (cap2 & (cap2 - 1)) || !a0_sumtree_sub_path(cap2))
        return a0_fail(A0_EINVAL, "a0_sumtree_set_from_loss: bad argument (at most 1024 leaves per call, a tree a0_sumtree_set_from_loss_ok accepts)");
    a0_sumtree_sub_launch(tree, cap2, idx, nullptr, n, state, loss, eps, alpha, pstate, (hipStream_t)stream, defer_top != 0);
    return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_set_from_loss");
}

// The launch a0_sumtree_set_from_loss(defer_top = 1) leaves out, for a reader of the top levels other than a0_sumtree_sample_batch(rebuild_top = 1) and
// a0_sumtree_set_range (both recompute the top from level A0_ST_TOP themselves).  Idempotent.
extern "C" int a0_sumtree_top_rebuild(float* tree, long long cap2, void* stream) {
    if (!tree || cap2 < 1 || (cap2 & (cap2 - 1))) return a0_fail(A0_EINVAL, "a0_sumtree_top_rebuild: bad argument");
    if (cap2 < 2 * A0_ST_TOP) return a0_sumtree_rebuild(tree, cap2, stream);
    hipLaunchKernelGGL(a0_sumtree_top_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tree, (const int*)nullptr);
    return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_top_rebuild");
}

extern "C" int a0_sumtree_set(float* tree, long long cap2, const long long* idx, const float* val, int n, const int* state, void* stream) {
    if (!tree || !idx || !val || n < 1 || cap2 < 1 || (cap2 & (cap2 - 1))) return a0_fail(A0_EINVAL, "a0_sumtree_set: cap2 must be a power of two");
    if (n > 1024) return a0_fail(A0_EINVAL, "a0_sumtree_set: at most 1024 leaves per call (the indices are staged in one workgroup's LDS); split the batch in order");
    if (a0_sumtree_sub_path(cap2)) {
        a0_sumtree_sub_launch(tree, cap2, idx, val, n, state, nullptr, 0.f, 0.f, nullptr, (hipStream_t)stream);
        return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_set");
    }
    int threads = 64; while (threads < n && threads < 1024) threads <<= 1;
    hipLaunchKernelGGL(a0_sumtree_set_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, tree, cap2, idx, val, n, state);
    return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_set");
}

// transform-and-set used by the learner: val = (loss + eps)^alpha; also tracks max_p like a0_priority_update
__global__ __launch_bounds__(1024) void a0_sumtree_prio_kernel(const float* __restrict__ loss, int n, float eps, float alpha, float* __restrict__ val,
                                                                float* __restrict__ pstate, const int* __restrict__ state) {
    if (state && state[3]) return;      // NaN-skipped update: max_p and the staged values stay as they were
    __shared__ float red[1024];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const float l = loss[i]; val[i] = a0_prio_pow(l + eps, alpha); mx = fmaxf(mx, l); }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) pstate[0] = fmaxf(pstate[0], red[0]);
}

extern "C" int a0_priority_from_loss(const float* loss, int n, float eps, float alpha, float* val, float* pstate, const int* state, void* stream) {
    if (!loss || !val || !pstate || n < 1) return a0_fail(A0_EINVAL, "a0_priority_from_loss: bad argument");
    int threads = 64; while (threads < n && threads < 1024) threads <<= 1;
    hipLaunchKernelGGL(a0_sumtree_prio_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, loss, n, eps, alpha, val, pstate, state);
    return a0_fail_hip((int)hipGetLastError(), "a0_priority_from_loss");
}

// full rebuild, one launch per level (used after bulk fills)
__global__ void a0_sumtree_level_kernel(float* __restrict__ tree, long long first, long long count) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const long long p = first + i;
    tree[p] = tree[2 * p] + tree[2 * p + 1];
}

extern "C" int a0_sumtree_rebuild(float* tree, long long cap2, void* stream) {
    if (!tree || cap2 < 1 || (cap2 & (cap2 - 1))) return a0_fail(A0_EINVAL, "a0_sumtree_rebuild: bad argument");
    for (long long first = cap2 >> 1; first >= 1; first >>= 1)
        hipLaunchKernelGGL(a0_sumtree_level_kernel, dim3((unsigned)((first + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tree, first, first);
    return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_rebuild");
}

// stratified proportional sampling: u_k = (k + xi_k) * (total / B)
__global__ void a0_sumtree_sample_kernel(const float* __restrict__ tree, long long cap2, const float* __restrict__ xi, int B,
                                         long long* __restrict__ out_idx, float* __restrict__ out_p) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const float total = tree[1];
    const float seg = total / (float)B;
    float u = ((float)k + xi[k]) * seg;
    long long n = 1;
    while (n < cap2) {
        const float left = tree[2 * n];
        const float right = tree[2 * n + 1];
        if (u < left || !(right > 0.0f)) {
            n = 2 * n;
        } else {
            u -= left;
            n = 2 * n + 1;
        }
    }
    out_idx[k] = n - cap2;
    out_p[k] = tree[n];
}

extern "C" int a0_sumtree_sample(const float* tree, long long cap2, const float* xi, int B, long long* out_idx, float* out_p, void* stream) {
    if (!tree || !xi || !out_idx || !out_p || B < 1 || cap2 < 1 || (cap2 & (cap2 - 1))) return a0_fail(A0_EINVAL, "a0_sumtree_sample: bad argument");
    hipLaunchKernelGGL(a0_sumtree_sample_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, tree, cap2, xi, B, out_idx, out_p);
    return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_sample");
}


// The descent of a0_sumtree_sample_kernel, two levels per memory round trip: the four grandchildren of node n are the 16 aligned bytes tree[4n .. 4n + 3],
// and because every internal node IS left + right of its children (the tree is only ever recomputed from children, never updated by delta), the children's
// values are gc0 + gc1 and gc2 + gc3 bit for bit — so one 16-byte load replaces two dependent 8-byte ones and the comparisons see the same numbers.
// A 1 M-leaf tree is 20 dependent L2 round trips deep (~0.8 us each for a lone workgroup); this makes it 10.
A0_D long long a0_sumtree_descend(const float* __restrict__ tree, long long cap2, float& u, long long n = 1) {
    while (4 * n <= cap2) {              // at least two levels below n
        const a0_f4 g = *(const a0_f4*)(tree + 4 * n);
        const float left = g.x + g.y, right = g.z + g.w;
        if (u < left || !(right > 0.0f)) {
            n = 2 * n;
            if (u < g.x || !(g.y > 0.0f)) { n = 2 * n; } else { u -= g.x; n = 2 * n + 1; }
        } else {
            u -= left;
            n = 2 * n + 1;
            if (u < g.z || !(g.w > 0.0f)) { n = 2 * n; } else { u -= g.z; n = 2 * n + 1; }
        }
    }
    while (n < cap2) {
        const float left = tree[2 * n];
        const float right = tree[2 * n + 1];
        if (u < left || !(right > 0.0f)) { n = 2 * n; } else { u -= left; n = 2 * n + 1; }
    }
    return n;
}

// ------------------------------------------------------------------------------------------------ prioritized batch in one launch
// The stratified uniforms (element b of the sampler's Philox stream: the value a0_rng_uniform would write), the sum-tree descent, the
// slot / metadata lookup and the importance weights w = (top * p / total)^-beta / (max w + 1e-8) (trainer.py:91-94) for one batch:
// == a0_rng_uniform + a0_sumtree_sample + a0_replay_lookup + a0_is_weights, statement for statement.  Single workgroup, B <= 1024.
__global__ __launch_bounds__(1024) void a0_sumtree_batch_kernel(unsigned long long seed, uint32_t stream, unsigned long long offset, float* __restrict__ tree,
                                                                 long long cap2, int B, long long top, long long cap, float beta, const int* __restrict__ r_act,
                                                                 const float* __restrict__ r_rew, const float* __restrict__ r_done, long long* __restrict__ idx_out,
                                                                 int* __restrict__ slot_out, int* __restrict__ act, float* __restrict__ rew, float* __restrict__ done,
                                                                 float* __restrict__ prio, float* __restrict__ w, int rebuild_top) {
    __shared__ float red[1024];
    // Round 4: the levels with at most A0_ST_TOP nodes are staged in LDS first — recomputed from level A0_ST_TOP exactly as a0_sumtree_top does (and written back when
    // `rebuild_top`: the launch a0_sumtree_set_from_loss(defer_top = 1) left out; recomputing a top that is up to date changes no bit) — and the descent walks them
    // there: eleven of a 1 M-leaf tree's twenty levels cost LDS reads instead of five dependent L2 round trips, and the prioritized update is two launches
    // (subtrees; top + next batch) instead of three.  Same comparisons on the same node values: every node IS left + right of its children.
    __shared__ float top_l[2 * A0_ST_TOP];
    const long long s0 = cap2 < A0_ST_TOP ? cap2 : A0_ST_TOP;
    for (long long p = threadIdx.x; p < s0; p += blockDim.x) top_l[s0 + p] = tree[s0 + p];
    __syncthreads();
    for (long long span = s0 >> 1; span >= 1; span >>= 1) {
        for (long long p = span + threadIdx.x; p < 2 * span; p += blockDim.x) top_l[p] = top_l[2 * p] + top_l[2 * p + 1];
        __syncthreads();
    }
    if (rebuild_top) for (long long p = 1 + threadIdx.x; p < s0; p += blockDim.x) tree[p] = top_l[p];
    const float total = top_l[1];
    const float seg = total / (float)B;
    float mx = 0.f;
    for (int k = threadIdx.x; k < B; k += blockDim.x) {
        const float xi = (float)(a0_philox_word(seed, stream, offset + (unsigned long long)k) >> 8) * 0x1.0p-24f;
        float u = ((float)k + xi) * seg;
        long long n = 1;
        while (n < s0) {
            const float left = top_l[2 * n], right = top_l[2 * n + 1];
            if (u < left || !(right > 0.0f)) { n = 2 * n; } else { u -= left; n = 2 * n + 1; }
        }
        n = a0_sumtree_descend(tree, cap2, u, n);
        const long long li = (n - cap2) % cap;          // sum-tree leaves are addressed by ring slot (head = 0): logical index == slot
        const long long sl = li;
        const float p = tree[n];
        idx_out[k] = li; slot_out[k] = (int)sl; act[k] = r_act[sl]; rew[k] = r_rew[sl]; done[k] = r_done[sl];
        prio[k] = p;
        const float probs = p / total;
        const float v = powf((float)top * probs, -beta);
        w[k] = v;
        mx = fmaxf(mx, v);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    const float denom = red[0] + 1e-8f;
    for (int k = threadIdx.x; k < B; k += blockDim.x) w[k] = w[k] / denom;
}

extern "C" int a0_sumtree_sample_batch(unsigned long long seed, unsigned int stream, unsigned long long offset, float* tree, long long cap2, int B, long long top,
                                       long long cap, float beta, const int* r_act, const float* r_rew, const float* r_done, long long* idx_out, int* slot_out,
                                       int* act, float* rew, float* done, float* prio, float* w, int rebuild_top, void* stream_h) {
    if (!tree || !r_act || !r_rew || !r_done || !idx_out || !slot_out || !act || !rew || !done || !prio || !w || B < 1 || B > 1024 || top < 1 || cap < top ||
        cap2 < cap || (cap2 & (cap2 - 1)) || cap > 2147483647LL)
        return a0_fail(A0_EINVAL, "a0_sumtree_sample_batch: bad argument");
    int threads = 64; while (threads < B && threads < 1024) threads <<= 1;
    hipLaunchKernelGGL(a0_sumtree_batch_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream_h, seed, stream, offset, tree, cap2, B, top, cap, beta, r_act, r_rew, r_done,
                       idx_out, slot_out, act, rew, done, prio, w, rebuild_top);
    return a0_fail_hip((int)hipGetLastError(), "a0_sumtree_sample_batch");
}

// ------------------------------------------------------------------------------------------------ fused sample + gather
// One launch draws the batch indices AND copies the rows: workgroup column b computes its own index (mode 0: Feistel permutation
// element start + b of the current epoch, reference DataLoader semantics; mode 1: stratified sum-tree descent with xi[b]), maps it to
// a ring slot, writes the per-sample metadata and streams the 2*obs_bytes row with 16 B per lane.  Same results as
// a0_perm_batch / a0_sumtree_sample + a0_replay_lookup + a0_replay_gather (checked in tests/test_gpu_kernels.py).
__global__ __launch_bounds__(256) void a0_sample_gather_kernel(int mode, unsigned long long start, unsigned long long n_perm, uint32_t seed,
                                                                const float* __restrict__ tree, long long cap2, const float* __restrict__ xi,
                                                                long long top, long long head, long long cap, const uint8_t* __restrict__ frames, int row_bytes,
                                                                const int* __restrict__ r_act, const float* __restrict__ r_rew, const float* __restrict__ r_done,
                                                                const float* __restrict__ priority, int B, uint8_t* __restrict__ out, long long* __restrict__ idx_out,
                                                                int* __restrict__ slot_out, int* __restrict__ act, float* __restrict__ rew, float* __restrict__ done,
                                                                float* __restrict__ prio) {
    __shared__ long long s_slot;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        long long li, slot;
        float p = 1.f;
        if (mode == 0) {
            uint32_t h = 1;
            while ((1ull << (2 * h)) < n_perm) ++h;
            const uint32_t mask = (uint32_t)((1ull << h) - 1);
            unsigned long long x = start + (unsigned long long)b;
            do {
                uint32_t l = (uint32_t)(x >> h) & mask, r = (uint32_t)x & mask;
                for (uint32_t round = 0; round < 4; ++round) {
                    const uint32_t f = a0_mix32(r ^ (seed + 0x9E3779B9u * (round + 1))) & mask;
                    const uint32_t nl = r, nr = l ^ f;
                    l = nl; r = nr;
                }
                x = ((unsigned long long)l << h) | r;
            } while (x >= n_perm);
            li = (long long)(x % (unsigned long long)top);
            slot = (head + li) % cap;
            if (priority) p = priority[li];
        } else {
            const float total = tree[1];
            const float seg = total / (float)B;
            float u = ((float)b + xi[b]) * seg;
            long long n = 1;
            while (n < cap2) {
                const float left = tree[2 * n];
                const float right = tree[2 * n + 1];
                if (u < left || !(right > 0.0f)) { n = 2 * n; } else { u -= left; n = 2 * n + 1; }
            }
            li = n - cap2;
            slot = li;
            p = tree[n];
        }
        s_slot = slot;
        if (blockIdx.x == 0) {
            idx_out[b] = li; slot_out[b] = (int)slot; act[b] = r_act[slot]; rew[b] = r_rew[slot]; done[b] = r_done[slot];
            if (prio) prio[b] = p;
        }
    }
    __syncthreads();
    const int q = row_bytes >> 4;
    const uint4* s = (const uint4*)(frames + s_slot * row_bytes);
    uint4* d = (uint4*)(out + (long long)b * row_bytes);
    const int step = gridDim.x * blockDim.x;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    for (; j + 3 * step < q; j += 4 * step) {          // four 16-byte loads in flight per lane before the first store
        const uint4 v0 = s[j], v1 = s[j + step], v2 = s[j + 2 * step], v3 = s[j + 3 * step];
        d[j] = v0; d[j + step] = v1; d[j + 2 * step] = v2; d[j + 3 * step] = v3;
    }
    for (; j < q; j += step) d[j] = s[j];
}

// Uniform sampling without the row copy (the learner reads ring rows through the slot index): permutation element start + b of the
// epoch -> logical index -> ring slot + metadata, one thread per sample.  == a0_perm_batch + a0_replay_lookup in one launch.
__global__ void a0_sample_slots_kernel(unsigned long long start, unsigned long long n_perm, uint32_t seed, long long top, long long head, long long cap,
                                       const int* __restrict__ r_act, const float* __restrict__ r_rew, const float* __restrict__ r_done,
                                       const float* __restrict__ priority, int B, long long* __restrict__ idx_out, int* __restrict__ slot_out,
                                       int* __restrict__ act, float* __restrict__ rew, float* __restrict__ done, float* __restrict__ prio) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t h = 1;
    while ((1ull << (2 * h)) < n_perm) ++h;
    const uint32_t mask = (uint32_t)((1ull << h) - 1);
    unsigned long long x = start + (unsigned long long)b;
    do {
        uint32_t l = (uint32_t)(x >> h) & mask, r = (uint32_t)x & mask;
        for (uint32_t round = 0; round < 4; ++round) {
            const uint32_t f = a0_mix32(r ^ (seed + 0x9E3779B9u * (round + 1))) & mask;
            const uint32_t nl = r, nr = l ^ f;
            l = nl; r = nr;
        }
        x = ((unsigned long long)l << h) | r;
    } while (x >= n_perm);
    const long long li = (long long)(x % (unsigned long long)top);
    const long long s = (head + li) % cap;
    idx_out[b] = li; slot_out[b] = (int)s; act[b] = r_act[s]; rew[b] = r_rew[s]; done[b] = r_done[s];
    if (prio) prio[b] = priority ? priority[li] : 1.f;
}

// n batches of the same epoch structure in ONE launch: batch g = blockIdx.y takes permutation elements start[g] .. start[g] + B - 1 of the epoch (n_perm[g], seed[g]) and
// writes row g of the [n][B] outputs.  Uniform replay's batches do not depend on the updates between them (the DataLoader's shuffled epochs, trainer.py:63-72), so a
// whole update block's sampling is one 5 us launch instead of twenty.  Per batch == a0_sample_slots_kernel.
struct a0_slots_multi_args { unsigned long long start[32], n_perm[32]; uint32_t seed[32]; };
__global__ void a0_sample_slots_multi_kernel(a0_slots_multi_args S, long long top, long long head, long long cap, const int* __restrict__ r_act, const float* __restrict__ r_rew,
                                             const float* __restrict__ r_done, int B, long long* __restrict__ idx_out, int* __restrict__ slot_out, int* __restrict__ act,
                                             float* __restrict__ rew, float* __restrict__ done, float* __restrict__ prio) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (b >= B) return;
    const unsigned long long n_perm = S.n_perm[g];
    const uint32_t seed = S.seed[g];
    uint32_t h = 1;
    while ((1ull << (2 * h)) < n_perm) ++h;
    const uint32_t mask = (uint32_t)((1ull << h) - 1);
    unsigned long long x = S.start[g] + (unsigned long long)b;
    do {
        uint32_t l = (uint32_t)(x >> h) & mask, r = (uint32_t)x & mask;
        for (uint32_t round = 0; round < 4; ++round) {
            const uint32_t f = a0_mix32(r ^ (seed + 0x9E3779B9u * (round + 1))) & mask;
            const uint32_t nl = r, nr = l ^ f;
            l = nl; r = nr;
        }
        x = ((unsigned long long)l << h) | r;
    } while (x >= n_perm);
    const long long li = (long long)(x % (unsigned long long)top);
    const long long s = (head + li) % cap;
    const long long o = (long long)g * B + b;
    idx_out[o] = li; slot_out[o] = (int)s; act[o] = r_act[s]; rew[o] = r_rew[s]; done[o] = r_done[s];
    if (prio) prio[o] = 1.f;
}

extern "C" int a0_replay_sample_slots_multi(int n, const unsigned long long* start, const unsigned long long* n_perm, const unsigned int* seed, long long top, long long head,
                                            long long cap, const int* r_act, const float* r_rew, const float* r_done, int B, long long* idx_out, int* slot_out, int* act, float* rew,
                                            float* done, float* prio, void* stream) {
    if (n < 1 || n > 32 || !start || !n_perm || !seed || !r_act || !r_rew || !r_done || !idx_out || !slot_out || !act || !rew || !done || B < 1 || top < 1 || cap < top ||
        cap > 2147483647LL)
        return a0_fail(A0_EINVAL, "a0_replay_sample_slots_multi: bad argument (1..32 batches)");
    a0_slots_multi_args S;
    for (int g = 0; g < 32; ++g) {
        const int j = g < n ? g : 0;
        if (n_perm[j] < 1 || start[j] + (unsigned long long)B > n_perm[j]) return a0_fail(A0_EINVAL, "a0_replay_sample_slots_multi: batch outside its epoch");
        S.start[g] = start[j]; S.n_perm[g] = n_perm[j]; S.seed[g] = seed[j];
    }
    hipLaunchKernelGGL(a0_sample_slots_multi_kernel, dim3((B + 127) / 128, n), dim3(128), 0, (hipStream_t)stream, S, top, head, cap, r_act, r_rew, r_done, B, idx_out, slot_out, act,
                       rew, done, prio);
    return a0_fail_hip((int)hipGetLastError(), "a0_replay_sample_slots_multi");
}

extern "C" int a0_replay_sample_slots(unsigned long long start, unsigned long long n_perm, unsigned int seed, long long top, long long head, long long cap,
                                      const int* r_act, const float* r_rew, const float* r_done, const float* priority, int B, long long* idx_out, int* slot_out,
                                      int* act, float* rew, float* done, float* prio, void* stream) {
    if (!r_act || !r_rew || !r_done || !idx_out || !slot_out || !act || !rew || !done || B < 1 || top < 1 || cap < top || cap > 2147483647LL || n_perm < 1 ||
        start + (unsigned long long)B > n_perm)
        return a0_fail(A0_EINVAL, "a0_replay_sample_slots: bad argument");
    hipLaunchKernelGGL(a0_sample_slots_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, start, n_perm, seed, top, head, cap, r_act, r_rew, r_done, priority,
                       B, idx_out, slot_out, act, rew, done, prio);
    return a0_fail_hip((int)hipGetLastError(), "a0_replay_sample_slots");
}

extern "C" int a0_replay_sample_gather(int mode, unsigned long long start, unsigned long long n_perm, unsigned int seed, const float* tree, long long cap2,
                                       const float* xi, long long top, long long head, long long cap, const uint8_t* frames, int row_bytes, const int* r_act,
                                       const float* r_rew, const float* r_done, const float* priority, int B, uint8_t* out, long long* idx_out, int* slot_out,
                                       int* act, float* rew, float* done, float* prio, void* stream) {
    if (!frames || !r_act || !r_rew || !r_done || !out || !idx_out || !slot_out || !act || !rew || !done || B < 1 || (row_bytes & 15) || top < 1 || cap < top)
        return a0_fail(A0_EINVAL, "a0_replay_sample_gather: bad argument");
    if (mode == 0 && (n_perm < 1 || start + (unsigned long long)B > n_perm)) return a0_fail(A0_EINVAL, "a0_replay_sample_gather: permutation window out of range");
    if (mode == 1 && (!tree || !xi || cap2 < 1 || (cap2 & (cap2 - 1)))) return a0_fail(A0_EINVAL, "a0_replay_sample_gather: sum-tree arguments");
    if (mode != 0 && mode != 1) return a0_fail(A0_EINVAL, "a0_replay_sample_gather: mode");
    // three workgroups per row: 768 lanes x four 16-byte loads cover 3072 of a row's 3528 vectors in one unrolled trip (12.8 us per
    // 512-row batch = 2.25 TB/s sampled, 4.5 TB/s of HBM traffic; 14.4 us with four workgroups and one load in flight; non-temporal
    // loads / stores measured slower: tools/ubench_gather.py).  A0_GATHER_GX: tuning aid.
    static const int gx_cap = getenv("A0_GATHER_GX") ? atoi(getenv("A0_GATHER_GX")) : 3;
    int gx = ((row_bytes >> 4) + 255) / 256; if (gx > gx_cap) gx = gx_cap;
    hipLaunchKernelGGL(a0_sample_gather_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, mode, start, n_perm, seed, tree, cap2, xi, top, head, cap, frames,
                       row_bytes, r_act, r_rew, r_done, priority, B, out, idx_out, slot_out, act, rew, done, prio);
    return a0_fail_hip((int)hipGetLastError(), "a0_replay_sample_gather");
}
