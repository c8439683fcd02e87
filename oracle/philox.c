/* CPU oracle: Philox4x32-10 counter RNG (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
 * — TEST INFRASTRUCTURE.  No reference counterpart: the reference draws from numpy's global MT19937
 * (agent0/deepq/agent.py:29-36) and torch's CPU generator (agent0/deepq/model.py:74-76,238), streams a GPU
 * cannot reproduce; parity tests therefore inject the draws, and this file pins the device generator
 * (agent0_amd/csrc/rng.hip) integer-exactly for the uniform stream (known-answer vectors from the
 * Random123 distribution are checked in tests/test_oracle_core.py).
 */
#include <stdint.h>
#include <math.h>

void a0o_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* element i of stream (seed, stream): block = (offset+i)/4, word = (offset+i)%4, ctr = (block_lo, block_hi, stream, 0) */
static uint32_t a0o_word(uint64_t seed, uint32_t stream, uint64_t pos) {
    uint64_t blk = pos >> 2;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), stream, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    a0o_philox4x32_10(ctr, key, out);
    return out[pos & 3];
}

void a0o_rng_u32(uint64_t seed, uint32_t stream, uint64_t offset, uint32_t* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = a0o_word(seed, stream, offset + i);
}

/* [0,1): top 24 bits * 2^-24 (exact in fp32) */
void a0o_rng_uniform(uint64_t seed, uint32_t stream, uint64_t offset, float* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = (float)(a0o_word(seed, stream, offset + i) >> 8) * 0x1.0p-24f;
}

/* N(0, std^2) by Box-Muller on word pairs (2j, 2j+1) of the stream: element 2j -> cos branch, 2j+1 -> sin branch.
 * u1 in (0,1], u2 in [0,1).  (libm vs device transcendental rounding: compared with a tolerance, not bit-exact.) */
void a0o_rng_normal(uint64_t seed, uint32_t stream, uint64_t offset, float std, float* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t pos = offset + i;
        uint64_t pair = pos & ~(uint64_t)1;
        float u1 = (float)((a0o_word(seed, stream, pair) >> 8) + 1u) * 0x1.0p-24f;
        float u2 = (float)(a0o_word(seed, stream, pair + 1) >> 8) * 0x1.0p-24f;
        float rad = sqrtf(-2.0f * logf(u1));
        float ang = 6.283185307179586f * u2;
        out[i] = std * rad * ((pos & 1) ? sinf(ang) : cosf(ang));
    }
}
