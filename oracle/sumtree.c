/* CPU oracle: fp32 sum-tree for proportional replay sampling — TEST INFRASTRUCTURE.
 *
 * The reference (zhoubin-me/agent0) has NO sum-tree: agent0/deepq/replay.py:18-19 keeps a flat
 * torch.ones(size) priority vector and agent0/deepq/trainer.py:63-72 samples uniformly; the only
 * proportional sampler, replay.py:39-43 (torch.multinomial), is dead code.  This file therefore
 * DEFINES the contract the HIP kernels (agent0_amd/csrc/sumtree.hip) must match bit-for-bit
 * ("parity unpinned" against the reference):
 *
 *   layout   tree[1] root, children of n are 2n / 2n+1, leaf i at tree[cap2 + i], cap2 = 2^k >= size,
 *            tree[0] unused.  All fp32.
 *   set      write leaves in batch order (a later duplicate wins), then recompute every ancestor
 *            bottom-up as  tree[p] = tree[2p] + tree[2p+1]   (left + right, one fp32 add) — never by delta,
 *            so the result depends only on the leaf values.
 *   sample   descend from the root with u in [0,total):  go left if (u < left || right <= 0)
 *            else { u -= left; go right }.  A zero-sum subtree is never entered.
 *   strata   u_k = ((float)k + xi_k) * (total / (float)B),  xi_k in [0,1) supplied by the caller.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <stdint.h>
#include <stddef.h>

int64_t a0o_sumtree_cap2(int64_t size) {
    int64_t c = 1;
    while (c < size) c <<= 1;
    return c;
}

void a0o_sumtree_rebuild(float* tree, int64_t cap2) {
    for (int64_t p = cap2 - 1; p >= 1; --p) tree[p] = tree[2 * p] + tree[2 * p + 1];
}

void a0o_sumtree_set(float* tree, int64_t cap2, const int64_t* idx, const float* val, int64_t n) {
    for (int64_t i = 0; i < n; ++i) tree[cap2 + idx[i]] = val[i];
    for (int64_t i = 0; i < n; ++i) {
        int64_t p = (cap2 + idx[i]) >> 1;
        while (p >= 1) {
            tree[p] = tree[2 * p] + tree[2 * p + 1];
            p >>= 1;
        }
    }
}

float a0o_sumtree_total(const float* tree) { return tree[1]; }

int64_t a0o_sumtree_find(const float* tree, int64_t cap2, float u) {
    int64_t n = 1;
    while (n < cap2) {
        float left = tree[2 * n];
        float right = tree[2 * n + 1];
        if (u < left || !(right > 0.0f)) {
            n = 2 * n;
        } else {
            u -= left;
            n = 2 * n + 1;
        }
    }
    return n - cap2;
}

void a0o_sumtree_sample(const float* tree, int64_t cap2, const float* xi, int64_t B, int64_t* out_idx, float* out_p) {
    float total = tree[1];
    float seg = total / (float)B;
    for (int64_t k = 0; k < B; ++k) {
        float u = ((float)k + xi[k]) * seg;
        int64_t i = a0o_sumtree_find(tree, cap2, u);
        out_idx[k] = i;
        out_p[k] = tree[cap2 + i];
    }
}

/* Bijective pseudo-random permutation of [0, n) used for the uniform-permutation sampler
 * (restates the *semantics* of DataLoader(shuffle=True) -> RandomSampler, trainer.py:63-72: every index of
 * range(top) exactly once per epoch; the concrete permutation is ours because torch's CPU generator stream
 * cannot be reproduced on the GPU).  4-round Feistel network on 2*h bits (2^(2h) >= n) with cycle-walking. */
static uint32_t a0o_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

uint64_t a0o_perm_index(uint64_t i, uint64_t n, uint32_t seed) {
    uint32_t h = 1;
    while (((uint64_t)1 << (2 * h)) < n) ++h;
    uint32_t mask = (uint32_t)(((uint64_t)1 << h) - 1);
    uint64_t x = i;
    do {
        uint32_t l = (uint32_t)(x >> h) & mask, r = (uint32_t)x & mask;
        for (uint32_t round = 0; round < 4; ++round) {
            uint32_t f = a0o_mix32(r ^ (seed + 0x9E3779B9u * (round + 1))) & mask;
            uint32_t nl = r, nr = l ^ f;
            l = nl; r = nr;
        }
        x = ((uint64_t)l << h) | r;
    } while (x >= n);
    return x;
}

void a0o_perm_batch(uint64_t start, uint64_t count, uint64_t n, uint32_t seed, int64_t* out) {
    for (uint64_t k = 0; k < count; ++k) out[k] = (int64_t)a0o_perm_index(start + k, n, seed);
}
