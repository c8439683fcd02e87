// Common definitions for the agent0_amd HIP library (gfx950 only).
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define A0_HD __host__ __device__ __forceinline__
#define A0_D __device__ __forceinline__
#else
#define A0_HD inline
#define A0_D inline
#endif

// 16-byte vector types usable from both hipcc and plain g++ (host emulation of the index math)
struct __attribute__((aligned(16))) a0_f4 { float x, y, z, w; };
struct __attribute__((aligned(16))) a0_i4 { int x, y, z, w; };

A0_HD a0_f4 a0_zero4() { a0_f4 v; v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; return v; }

// status codes returned by every exported function (include/agent0_hip.h)
enum {
    A0_OK = 0,
    A0_EINVAL = -1,   // bad argument (shape/alignment/null)
    A0_EHIP = -2,     // HIP runtime error, see a0_last_error()
    A0_ENOMEM = -3,
    A0_ESTATE = -4,   // call not valid in the current state
};

// Geometry of a "virtual convolution": row m = (b, oh, ow) reads
//   in[b][oh*stride - pad + kh][ow*stride - pad + kw][c]
// Used for forward im2col (pad = 0) and for gather-form data gradients (stride = 1, pad > 0).
struct a0_geom {
    int Hin, Win, C;          // input spatial size and channels
    int Hout, Wout;           // rows per sample = Hout*Wout
    int stride, pad;
    int HWout;
    long long sample_stride;  // elements (bytes for u8) between consecutive samples of the input
    unsigned hw_magic, w_magic;   // ceil(2^32 / HWout), ceil(2^32 / Wout): row index -> (sample, oh, ow) without integer division (a0_udiv)
};

// floor(m / d) for any 32-bit m, given magic = ceil(2^32 / d), d >= 2: the multiply-high overestimates by at most one.
A0_HD unsigned a0_udiv_magic(unsigned d) { return d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u; }
A0_HD int a0_udiv(int m, int d, unsigned magic) {
    if (d <= 1) return m;
    unsigned q = (unsigned)(((unsigned long long)(unsigned)m * magic) >> 32);
    if (q * (unsigned)d > (unsigned)m) --q;
    return (int)q;
}
