// Philox4x32-10 (device).  Same arithmetic as oracle/philox.c.
#pragma once
#include "a0_defs.h"

struct a0_u4 { uint32_t x, y, z, w; };

A0_HD a0_u4 a0_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    a0_u4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

A0_HD uint32_t a0_philox_word(unsigned long long seed, uint32_t stream, unsigned long long pos) {
    const unsigned long long blk = pos >> 2;
    const a0_u4 o = a0_philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w = (uint32_t)(pos & 3);
    return w == 0 ? o.x : (w == 1 ? o.y : (w == 2 ? o.z : o.w));
}
