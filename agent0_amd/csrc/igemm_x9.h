// The implicit GEMM of igemm.h on the bf16 matrix pipe with BOTH fp32 operands split exactly into three bf16 terms.
//
// Same contract as a0_igemm_kernel (operand policies of operands.h, epilogues, split-K slabs, the row-sum by-product of the
// weight gradients) and the same fp32 result up to the association order of the additions: any fp32 value is EXACTLY
// hi + mid + lo with hi = trunc16(x), mid = trunc16(x - hi), lo = x - hi - mid (8 + 8 + 8 significand bits, each term exact in
// bf16), a product of two bf16 values is exact in fp32, and all nine cross products are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Per 32 k and 32x32 block that is 18 MFMAs of 32 matrix-pipe cycles instead of 16 fp32 MFMAs of 64.
//
// The split happens ONCE per element, when a tile moves from registers to LDS; LDS holds three bf16 term planes per operand:
//   A0_KC operands (k contiguous in memory):  plane[x][32 k], rows of 64 B at a pitch of 80 B; a fragment (8 consecutive k of one
//          row) is one aligned ds_read_b128, conflict-free (5 is odd, so any 16 rows of a b128 lane group hit 16 different 4-bank units).
//   A0_XC operands (x contiguous in memory):  plane[32 k][x] exactly as fetched, pitch = 16 or 48 dwords mod 64; the k-major
//          fragment is produced by the LDS itself: two ds_read_b64_tr_b16 (4 k x 16 x blocks, transposed per 16-lane group).
// Either way one float4 of fetched data becomes three 8-byte LDS writes.
//
// Tile shapes: WM x WN waves of MT x NT blocks of 32 x 32, over k tiles of 16 KS.  KS = 2 (32 k, two k-steps per barrier) is the default; KS = 1 halves the
// LDS per stage and makes room for the 256 x 128 tile (eight waves of 64 x 64: <4, 2, 2, 2, 1>, 110 KB double-buffered) that the backend picks for the
// quantile networks' 16 384 / 32 768-row layers: per MFMA a quarter less staging (fetch, split, LDS store) and a third fewer fragment reads than
// 128 x 128 x 32, the same 36 MFMAs per wave between barriers.
#pragma once
#include "igemm.h"
#if defined(__HIPCC__)
#include <hip/hip_ext.h>
#endif

#if defined(__HIPCC__)

typedef __bf16 a0_bf16x8g __attribute__((ext_vector_type(8)));
typedef short a0_s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t a0_u32x2g __attribute__((ext_vector_type(2)));
typedef uint32_t a0_u32x4g __attribute__((ext_vector_type(4)));

// One piece = one float4 of an operand tile per thread.  Its way from registers to the three term planes is cut into six
// micro-steps (zero-fill / finish, four element splits, pack + store) so that the kernel can place each of them behind an MFMA.
struct a0_x9_piece {
    a0_f4 v;
    uint32_t h[4], m[4], l[4];
    A0_D void split(int e) {
        const float x = e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
        h[e] = __float_as_uint(x);
        const float r1 = x - __uint_as_float(h[e] & 0xffff0000u);            // exact: at most 16 significant bits left
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & 0xffff0000u);           // exact: at most 8 significant bits left
        l[e] = __float_as_uint(r2);
    }
    // v_perm_b32: the upper halves of two dwords side by side (the truncation itself); element e of the float4 in bf16 slot e
    A0_D void pack(a0_u32x2g& hi, a0_u32x2g& mid, a0_u32x2g& lo) const {
        hi.x = __builtin_amdgcn_perm(h[1], h[0], 0x07060302u);  hi.y = __builtin_amdgcn_perm(h[3], h[2], 0x07060302u);
        mid.x = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u); mid.y = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
        lo.x = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);  lo.y = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
    }
};

// KS = k-steps of 16 per tile: 2 (tiles of 32 k, the default) or 1 (tiles of 16 k: half the LDS per stage, for the 256 x 128 workgroup tile)
template <int MODE, int BX, int KS = 2> struct a0_x9_image;
template <int BX, int KS> struct a0_x9_image<A0_KC, BX, KS> {
    static constexpr int PITCH = 32 * KS + 16;             // bytes per x row (16 KS bf16 + 16 B pad): 80 / 48, an odd multiple of 16 -> conflict-free b128 reads
    static constexpr int PLANE = BX * PITCH;
};
template <int BX, int KS> struct a0_x9_image<A0_XC, BX, KS> {
    static constexpr int PDW = ((BX / 2) % 32 == 16) ? (BX / 2) : (BX / 2 + 16);   // dwords per k row: 16 mod 32, so 4 rows x 16 dwords tile the 64 banks
    static constexpr int PITCH = 4 * PDW;
    static constexpr int PLANE = 16 * KS * PITCH;
};

// global -> registers (the policies' branch-free loads, as in igemm.h) and registers -> term planes, both in PIECES of one float4 per
// thread: the kernel threads a tile's pieces between its MFMAs, so the ~22 VALU instructions of a split run in the shadow of the
// matrix pipe instead of in front of it.
template <class OP, int BX, int NTH, int KS = 2, int MODE = OP::MODE> struct a0_x9_stager;

template <class OP, int BX, int NTH, int KS> struct a0_x9_stager<OP, BX, NTH, KS, A0_KC> {
    static constexpr bool PRESPLIT = false;
    static constexpr int TPR = 4 * KS;       // threads per row: eight (four) threads cover the 32 (16) k of one row
    static constexpr int RPP = NTH / TPR;    // rows per pass
    static constexpr int R = BX / RPP;
    static_assert(BX % RPP == 0, "tile rows per pass");
    typedef a0_x9_image<A0_KC, BX, KS> IM;
    typename OP::Row rows[R];
    typename OP::KInfo ki;        // gather-table entry of the tile being fetched (looked up when the previous tile's last piece was fetched)
    struct Slot { typename OP::Raw raw[R]; unsigned okmask; };
    A0_D void init(const typename OP::Params& P, int x0, int X, int kb, int ke, int tid) {
#pragma unroll
        for (int j = 0; j < R; ++j) rows[j] = OP::row(P, x0 + tid / TPR + RPP * j, X);
        ki = OP::kinfo(P, kb + 4 * (tid % TPR), ke);
    }
    // pieces of one tile are fetched in the order j = 0 .. R-1; the last one moves the k state on to the next tile
    // FULL (the body's launch-time choice: no partial k tile anywhere in the launch): nothing is zero-filled — rows past the edge read valid memory and feed only outputs
    // that are never stored, tiles past the k range are staged and never multiplied — so the ok bits are neither kept nor applied
    template <bool FULL = false> A0_D void fetch_piece(const typename OP::Params& P, Slot& s, int j, int k0, int ke, int tid) {
        bool ok;
        s.raw[j] = OP::load(P, rows[j], ki, ok);
        if constexpr (!FULL) s.okmask = (s.okmask & ~(1u << j)) | (ok ? (1u << j) : 0u);
        if (j == R - 1) ki = OP::kinfo(P, k0 + 16 * KS + 4 * (tid % TPR), ke);
    }
    template <bool RS, bool FULL = false> A0_D a0_f4 value(const Slot& s, int j, a0_f4&) const {
        static_assert(!RS, "row sums are taken from x-contiguous A operands");
        if constexpr (FULL) return OP::finish(s.raw[j], true);
        else return OP::finish(s.raw[j], (s.okmask >> j) & 1u);
    }
    A0_D void store(const a0_x9_piece& pc, int j, char* lds, int tid) const {
        const int r = tid / TPR + RPP * j;
        a0_u32x2g hi, mid, lo;
        pc.pack(hi, mid, lo);
        char* p = lds + r * IM::PITCH + 8 * (tid % TPR);
        *(a0_u32x2g*)(p) = hi;
        *(a0_u32x2g*)(p + IM::PLANE) = mid;
        *(a0_u32x2g*)(p + 2 * IM::PLANE) = lo;
    }
    // byte offset of this lane's fragment of block 0, k-step 0 (wave rows start at x = woff)
    A0_D static int frag_base(int lane, int woff) { return (woff + (lane & 31)) * IM::PITCH + (lane >> 5) * 16; }
    A0_D static a0_u32x4g frag(const char* plane, int base, int blk, int s) {
        return *(const a0_u32x4g*)(plane + base + blk * 32 * IM::PITCH + s * 32);
    }
};

template <class OP, int BX, int NTH, int KS> struct a0_x9_stager<OP, BX, NTH, KS, A0_XC> {
    static constexpr bool PRESPLIT = false;
    static constexpr int R = 4 * KS * BX / NTH;
    static constexpr int Q = BX / 4;   // 16-byte groups per k row
    static_assert((4 * KS * BX) % NTH == 0 && NTH % Q == 0, "tile columns per pass");
    typedef a0_x9_image<A0_XC, BX, KS> IM;
    typename OP::XInfo xi[R];
    struct Slot { typename OP::Raw raw[R]; unsigned okmask; };
    A0_D void init(const typename OP::Params& P, int x0, int X, int, int, int tid) {
#pragma unroll
        for (int j = 0; j < R; ++j) xi[j] = OP::xinfo(P, x0 + 4 * ((tid + NTH * j) % Q), X);
    }
    template <bool FULL = false> A0_D void fetch_piece(const typename OP::Params& P, Slot& s, int j, int k0, int ke, int tid) {
        bool ok;
        s.raw[j] = OP::load(P, k0 + (tid + NTH * j) / Q, ke, xi[j], ok);
        if constexpr (!FULL) s.okmask = (s.okmask & ~(1u << j)) | (ok ? (1u << j) : 0u);
    }
    template <bool RS, bool FULL = false> A0_D a0_f4 value(const Slot& s, int j, a0_f4& rowsum) const {
        const a0_f4 v = FULL ? OP::finish(s.raw[j], true) : OP::finish(s.raw[j], (s.okmask >> j) & 1u);
        if constexpr (RS) { rowsum.x += v.x; rowsum.y += v.y; rowsum.z += v.z; rowsum.w += v.w; }
        return v;
    }
    A0_D void store(const a0_x9_piece& pc, int j, char* lds, int tid) const {
        const int f = tid + NTH * j;
        a0_u32x2g hi, mid, lo;
        pc.pack(hi, mid, lo);
        char* p = lds + (f / Q) * IM::PITCH + 8 * (f % Q);
        *(a0_u32x2g*)(p) = hi;
        *(a0_u32x2g*)(p + IM::PLANE) = mid;
        *(a0_u32x2g*)(p + 2 * IM::PLANE) = lo;
    }
    // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row k0+q, columns 4p..4p+3 of the group's 4 x 16 block and
    // receives column (lane & 15), rows k0..k0+3.  Groups 0/1 take x 0-15 / 16-31 at k 0-7, groups 2/3 the same at k 8-15.
    A0_D static int frag_base(int lane, int woff) {
        return ((lane >> 5) * 8 + ((lane & 15) >> 2)) * IM::PITCH + 2 * (woff + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
    }
    A0_D static a0_u32x4g frag(const char* plane, int base, int blk, int s) {
        typedef __attribute__((address_space(3))) a0_s16x4 lds_s16x4;
        const char* p = plane + base + blk * 64 + s * 16 * IM::PITCH;
        const a0_s16x4 k03 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
        const a0_s16x4 k47 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * IM::PITCH));
        const a0_u32x2g a = __builtin_bit_cast(a0_u32x2g, k03), b = __builtin_bit_cast(a0_u32x2g, k47);
        a0_u32x4g r; r.x = a.x; r.y = a.y; r.z = b.x; r.w = b.y;
        return r;
    }
};

// ---- an operand that is ALREADY split (round 6): the three bf16 term planes of a weight matrix W [N][K], written once per optimizer step by a0_split_planes_kernel
// (net.hip) and reused by every GEMM until the weights change — the actor's fc1 runs 80 steps on one W, a learner pass has rows in the tens of thousands — so that
// this operand's tiles go from global memory to the LDS term planes without a vector instruction.  Layout: per row n and group g of four consecutive k six dwords
// { hi(k0,k1) hi(k2,k3) | mid .. | lo .. } = the three 8-byte LDS stores of a piece, 24 bytes apart.  K % 4 == 0; rows beyond Y and the k range past the split's end
// are NOT zero-filled (callers guarantee whole k tiles; rows past the edge only feed outputs that are never stored).
struct a0_planes_src { const uint32_t* p; int kgroups; };      // kgroups = K / 4
struct OpPlanesKC {
    static constexpr int MODE = A0_KC;
    typedef a0_planes_src Params;
    struct Row { const uint32_t* p; };
    struct KInfo { int g; };
    struct Raw { a0_u32x4g a; a0_u32x2g b; };
    A0_HD static Row row(const Params& P, int r, int R) { Row o; o.p = P.p + (long long)(r < R ? r : R - 1) * P.kgroups * 6; return o; }
    A0_HD static KInfo kinfo(const Params& P, int k, int Kend) { KInfo ki; ki.g = (k < Kend ? k : 0) >> 2; return ki; }
    A0_HD static Raw load(const Params&, const Row& r, const KInfo& ki) {
        Raw v;
        const uint32_t* q = r.p + (long long)ki.g * 6;
        v.a = *(const a0_u32x4g*)q;
        v.b = *(const a0_u32x2g*)(q + 4);
        return v;
    }
};
template <int BX, int NTH, int KS> struct a0_x9_stager<OpPlanesKC, BX, NTH, KS, A0_KC> {
    static constexpr bool PRESPLIT = true;
    static constexpr int TPR = 4 * KS;
    static constexpr int RPP = NTH / TPR;
    static constexpr int R = BX / RPP;
    static_assert(BX % RPP == 0, "tile rows per pass");
    typedef a0_x9_image<A0_KC, BX, KS> IM;
    typedef OpPlanesKC OP;
    OP::Row rows[R];
    OP::KInfo ki;
    struct Slot { OP::Raw raw[R]; unsigned okmask; };
    A0_D void init(const OP::Params& P, int x0, int X, int kb, int ke, int tid) {
#pragma unroll
        for (int j = 0; j < R; ++j) rows[j] = OP::row(P, x0 + tid / TPR + RPP * j, X);
        ki = OP::kinfo(P, kb + 4 * (tid % TPR), ke);
    }
    template <bool FULL = false> A0_D void fetch_piece(const OP::Params& P, Slot& s, int j, int k0, int ke, int tid) {
        s.raw[j] = OP::load(P, rows[j], ki);
        if (j == R - 1) ki = OP::kinfo(P, k0 + 16 * KS + 4 * (tid % TPR), ke);
    }
    template <bool RS, bool FULL = false> A0_D a0_f4 value(const Slot&, int, a0_f4&) const { return a0_zero4(); }      // (never called: PRESPLIT pieces skip the split micro-steps)
    A0_D void store(const a0_x9_piece&, int, char*, int) const {}
    A0_D void store_raw(const Slot& s, int j, char* lds, int tid) const {
        const int r = tid / TPR + RPP * j;
        char* p = lds + r * IM::PITCH + 8 * (tid % TPR);
        a0_u32x2g hi, mid; hi.x = s.raw[j].a.x; hi.y = s.raw[j].a.y; mid.x = s.raw[j].a.z; mid.y = s.raw[j].a.w;
        *(a0_u32x2g*)(p) = hi;
        *(a0_u32x2g*)(p + IM::PLANE) = mid;
        *(a0_u32x2g*)(p + 2 * IM::PLANE) = s.raw[j].b;
    }
    A0_D static int frag_base(int lane, int woff) { return (woff + (lane & 31)) * IM::PITCH + (lane >> 5) * 16; }
    A0_D static a0_u32x4g frag(const char* plane, int base, int blk, int s) {
        return *(const a0_u32x4g*)(plane + base + blk * 32 * IM::PITCH + s * 32);
    }
};

template <class OA, class OB, int WM, int WN, int MT, int NT, int KS = 2> struct a0_x9_geom {
    static constexpr int BX = WM * MT * 32, BY = WN * NT * 32;
    static constexpr int ABYTES = 3 * a0_x9_image<OA::MODE, BX, KS>::PLANE;
    static constexpr int BBYTES = 3 * a0_x9_image<OB::MODE, BY, KS>::PLANE;
    static constexpr int LDS_BYTES = 2 * (ABYTES + BBYTES);                 // double-buffered
};

// plain matrix operands (no gather tables, no padding semantics): eligible for the unmasked staging of FULL launches
template <class OP> struct a0_x9_plain { static constexpr bool value = false; };
template <> struct a0_x9_plain<OpMatKC> { static constexpr bool value = true; };
template <> struct a0_x9_plain<OpMatXC> { static constexpr bool value = true; };
struct OpPlanesKC;
template <> struct a0_x9_plain<OpPlanesKC> { static constexpr bool value = true; };
template <class EP> struct a0_is_hadamard { static constexpr bool value = false; };
template <> struct a0_is_hadamard<EpiHadamard> { static constexpr bool value = true; };

// Cross products of the three-term splits that are formed (NPR): 9 = all of them (every partial product of the fp32 fmaf chain, exactly: the strict mode), 6 = those
// with term orders i + j <= 2 — a1*b2, a2*b1 and a2*b2 are left out, each below 2^-24 of a*b, i.e. below the rounding the fp32 chain itself applies to every partial SUM
// (tools/check_bf16x9.hip, profiles/r06_x6_accuracy.txt: against fp64 the six-product sum is at least as close as the fmaf chain at every K of the path).  Chosen at run
// time (a0_x9_products(), A0_X9_PRODUCTS=9|6); both forms of every instantiation are in the library.
int a0_x9_products_now();
template <class OA, class OB, class EP, int WM, int WN, int MT, int NT, int KS = 2, int NPR = 9, bool FULL = false>
__device__ __forceinline__ void a0_igemm_x9_body(const typename OA::Params& pa, const typename OB::Params& pb,
                                                 const typename EP::Params& pe, int X, int Y, int K, int kchunk, int gx, int gy) {
    static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves per workgroup");
    static_assert(KS == 1 || KS == 2, "tiles of 16 or 32 k");
    typedef a0_x9_geom<OA, OB, WM, WN, MT, NT, KS> G;
    constexpr int BK = 16 * KS, BX = G::BX, BY = G::BY, NTH = WM * WN * 64;
    typedef a0_x9_stager<OA, BX, NTH, KS> SA;
    typedef a0_x9_stager<OB, BY, NTH, KS> SB;
    constexpr int APL = a0_x9_image<OA::MODE, BX, KS>::PLANE, BPL = a0_x9_image<OB::MODE, BY, KS>::PLANE;
    constexpr int RA = SA::R, RB = SB::R, NP = RA + RB;      // commit / fetch pieces per tile
    static_assert(NPR == 9 || NPR == 6, "nine or six cross products");
    constexpr int NG = NPR * KS;                             // product groups per tile: KS k-steps x NPR term pairs, MT*NT MFMAs each
    static_assert(!EP::ROWSUM_A || OA::MODE == A0_XC, "row sums need an x-contiguous A operand");
    extern __shared__ __attribute__((aligned(16))) char a0_x9_lds[];
    char* const As = a0_x9_lds;                       // [2 buffers][3 planes]
    char* const Bs = a0_x9_lds + 2 * G::ABYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order.  The grid is one-dimensional; hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each
    // with its own L2.  Ids are regrouped so that every XCD walks a CONTIGUOUS range of logical tiles, and within the range the
    // smaller grid dimension runs fastest: the workgroups an XCD runs side by side then share one tile of the operand that would
    // otherwise be re-read from HBM once per tile of the other dimension (fc1 over 32 768 quantile rows: 411 MB of activations x 8).
    int bx, by, bz;
    {
        const int n = gridDim.x, id = blockIdx.x, per = n >> 3;
        const int L = (id < (per << 3)) ? (id & 7) * per + (id >> 3) : id;
        if (gy <= gx) { by = L % gy; const int t = L / gy; bx = t % gx; bz = t / gx; }
        else          { bx = L % gx; const int t = L / gx; by = t % gy; bz = t / gy; }
    }
    const int x0 = bx * BX, y0 = by * BY;
    const int kb = bz * kchunk;
    const int ke = (K < kb + kchunk) ? K : (kb + kchunk);

    SA sa; SB sb;
    sa.init(pa, x0, X, kb, ke, tid);
    sb.init(pb, y0, Y, kb, ke, tid);

    a0_acc16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // EP::ROWSUM_A (weight gradients): the sums of A's rows over this split's k range are the bias gradient; they are taken from
    // the fetched registers on their way to LDS by the blockIdx.y == 0 column of workgroups (an extra add per element elsewhere
    // would be wasted, so the flag picks the instantiation).
    a0_f4 rowsum = a0_zero4();

    // piece p of a tile: p < RA -> A slot row-group p, else B slot row-group p - RA
    auto fetch_piece = [&](typename SA::Slot& s_a, typename SB::Slot& s_b, int p, int k0) {
        if (p < RA) sa.template fetch_piece<FULL>(pa, s_a, p, k0, ke, tid);
        else sb.template fetch_piece<FULL>(pb, s_b, p - RA, k0, ke, tid);
    };
    // micro-step u of a tile's staging work: piece u / 6 (A row-groups first, then B), step u % 6 = finish | split e0..e3 | pack + store + refetch
    a0_x9_piece pc;
    auto micro = [&](int u, const typename SA::Slot& s_a_c, const typename SB::Slot& s_b_c, typename SA::Slot& s_a, typename SB::Slot& s_b,
                     char* a_dst, char* b_dst, int k_fetch) {
        const int p = u / 6, st = u % 6;
        if constexpr (SB::PRESPLIT) {
            if (p >= RA) {      // an operand that arrives as term planes: no split, one micro-step (planes to LDS, refetch)
                if (st == 5) { sb.store_raw(s_b_c, p - RA, b_dst, tid); fetch_piece(s_a, s_b, p, k_fetch); }
                return;
            }
        }
        if (st == 0) {
            if (p < RA) pc.v = sa.template value<EP::ROWSUM_A, FULL>(s_a_c, p, rowsum);
            else pc.v = sb.template value<false, FULL>(s_b_c, p - RA, rowsum);
        } else if (st <= 4) {
            pc.split(st - 1);
        } else {
            if (p < RA) sa.store(pc, p, a_dst, tid);
            else sb.store(pc, p - RA, b_dst, tid);
            fetch_piece(s_a, s_b, p, k_fetch);
        }
    };

    const int abase = SA::frag_base(lane, wm * (MT * 32));
    const int bbase = SB::frag_base(lane, wn * (NT * 32));

    // One tile: multiply LDS buffer `buf` (tile t) while tile t+1 (slots s_a / s_b) is split and written into the other buffer and
    // tile t+3 is requested into the registers each piece frees.  The 6*NP micro-steps are dealt out evenly behind the tile's
    // 18*MT*NT MFMAs and pinned there with sched_barriers: an MFMA holds the vector issue port for 8 of its 32 cycles, so the five
    // or six VALU instructions of a micro-step issue in its shadow (left to itself the scheduler clusters the MFMAs and the splits
    // and the two run one after the other).
    auto mma_tile = [&](int buf, typename SA::Slot& s_a, typename SB::Slot& s_b, int k_fetch) {
        const char* ap = As + buf * G::ABYTES;
        const char* bp = Bs + buf * G::BBYTES;
        char* a_dst = As + (buf ^ 1) * G::ABYTES;
        char* b_dst = Bs + (buf ^ 1) * G::BBYTES;
        a0_u32x4g a[KS][MT][3], b[KS][NT][3];
        // LDS returns reads in issue order and the product groups start with the smallest terms (lo x lo): the lo planes are requested first,
        // so the first MFMA waits for MT + NT fragments instead of for a whole k-step's
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 2; t >= 0; --t) {
#pragma unroll
                for (int i = 0; i < MT; ++i) a[s][i][t] = SA::frag(ap + t * APL, abase, i, s);
                const int tq = NPR == 6 ? 2 - t : t;         // six products start with lo x hi: B's planes arrive in the opposite order
#pragma unroll
                for (int j = 0; j < NT; ++j) b[s][j][tq] = SB::frag(bp + tq * BPL, bbase, j, s);
            }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NM = NG * MT * NT, NU = 6 * NP;
#pragma unroll
        for (int n = 0; n < NM; ++n) {
            // MFMA n: product group g = n / (MT*NT) -> k-step g / 9 and term pair g % 9 in the order of increasing magnitude (lo*lo first, hi*hi last)
            const int g = n / (MT * NT), i = (n / NT) % MT, j = n % NT;
            const int s = g / NPR, q = g % NPR + (9 - NPR);            // six products: the three smallest (lo*lo, lo*mid, mid*lo) are not formed
            constexpr int TA[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
            constexpr int TB[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(a0_bf16x8g, a[s][i][TA[q]]),
                                                                 __builtin_bit_cast(a0_bf16x8g, b[s][j][TB[q]]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int u = n * NU / NM; u < (n + 1) * NU / NM; ++u) micro(u, s_a, s_b, s_a, s_b, a_dst, b_dst, k_fetch);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Prologue: tiles 0 and 1 into the two register slots, tile 0 into LDS buffer 0, tile 2 into the freed slot.  Fetches past the
    // end of the k range are harmless by construction (safe address, zero fill), so the steady state needs no conditionals.
    typename SA::Slot sa0, sa1;
    typename SB::Slot sb0, sb1;
    sa0.okmask = sa1.okmask = 0u; sb0.okmask = sb1.okmask = 0u;
#pragma unroll
    for (int p = 0; p < NP; ++p) fetch_piece(sa0, sb0, p, kb);
#pragma unroll
    for (int p = 0; p < NP; ++p) fetch_piece(sa1, sb1, p, kb + BK);
#pragma unroll
    for (int u = 0; u < 6 * NP; ++u) micro(u, sa0, sb0, sa0, sb0, As, Bs, kb + 2 * BK);
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += 2 * BK) {
        mma_tile(0, sa1, sb1, k0 + 3 * BK);
        __syncthreads();
        if (k0 + BK >= ke) break;
        mma_tile(1, sa0, sb0, k0 + 4 * BK);
        __syncthreads();
    }
    if constexpr (EP::ROWSUM_A) {
        if (by == 0) {                       // every thread's columns are x0 + 4*(tid % Q) .. +3 in all of its slots; the LDS is free (the loop ends with a barrier)
            constexpr int Q = BX / 4;
            a0_f4* red = (a0_f4*)a0_x9_lds;
            red[tid] = rowsum;
            __syncthreads();
            if (tid < BX && x0 + tid < X) {
                const float* rf = (const float*)a0_x9_lds;
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < NTH / Q; ++g) t += rf[4 * (g * Q + (tid >> 2)) + (tid & 3)];
                EP::store_rowsum(pe, x0 + tid, t, bz);
            }
        }
    }

    // C/D layout of v_mfma_f32_32x32x16_bf16 == that of 32x32x2_f32: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if constexpr (a0_is_hadamard<EP>::value) {
        // EpiHadamard: a wave holds MT * 32 consecutive rows = whole samples (n = 32: one per 32-row block; n = 64 with MT = 2: one per wave; the launcher checks), so the
        // sum over a sample's rows is 16 values per lane and block plus the partner lane that holds the other 16 rows of the block — no atomics, a fixed order
        static_assert(MT == 2, "EpiHadamard: waves of 64 rows");
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int y = y0 + wn * (NT * 32) + j * 32 + (lane & 31);
            const bool yok = y < Y;
            const int yy = yok ? y : 0;
            float s64 = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int xb = x0 + wm * (MT * 32) + i * 32;                  // first row of the block (a multiple of 32)
                const int bsmp = xb / pe.n;                                   // the sample these rows belong to
                const float f = pe.feat[(long long)bsmp * pe.ld + yy];
                float sacc = 0.f;
                float ev[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ev[r] = pe.emb[(long long)(xb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * pe.ld + yy];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long idx = (long long)(xb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * pe.ld + yy;
                    const float g = acc[i][j][r];
                    if (yok) pe.demb[idx] = (ev[r] > 0.f) ? g * f : 0.f;
                    sacc += g * ev[r];
                }
                sacc += __shfl_xor(sacc, 32, 64);
                if (pe.n == 32) { if (yok && lane < 32) pe.d3[(long long)bsmp * pe.ld + y] = (f > 0.f) ? sacc : 0.f; }
                else {
                    s64 += sacc;
                    if (i == MT - 1 && yok && lane < 32) pe.d3[(long long)bsmp * pe.ld + y] = (f > 0.f) ? s64 : 0.f;
                }
            }
        }
        return;
    }
    const int z = bz;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int y = y0 + wn * (NT * 32) + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (x < X && y < Y) EP::store(pe, x, y, acc[i][j][r], z);
            }
        }
}

template <class OA, class OB, class EP, int WM, int WN, int MT, int NT, int KS = 2, int NPR = 9, bool FULL = false>
__global__ __launch_bounds__(WM * WN * 64) void a0_igemm_x9_kernel(typename OA::Params pa, typename OB::Params pb,
                                                           typename EP::Params pe, int X, int Y, int K, int kchunk, int gx, int gy) {
    a0_igemm_x9_body<OA, OB, EP, WM, WN, MT, NT, KS, NPR, FULL>(pa, pb, pe, X, Y, K, kchunk, gx, gy);
}

// Up to three GEMMs of ONE shape (own operands and outputs each) in one launch: blockIdx.y names the problem.  The point is not the launch it saves but the split: a
// 512-row fc1 pass alone needs K cut eight ways to fill 256 CUs, and the ~10 us of ramp, prologue and slab epilogue then weigh as much as its k loop; two or three
// passes side by side fill the chip with half or a third of the splits — each workgroup's k loop is two or three times as long, the fixed part is paid once
// (the learner's target / online fc1 passes of one update: 2 x 17.1 us -> one launch; and the consumer sums half as many slabs).
template <class OA, class OB, class EP> struct a0_x9_group { typename OA::Params pa[3]; typename OB::Params pb[3]; typename EP::Params pe[3]; };
template <class OA, class OB, class EP, int WM, int WN, int MT, int NT, int KS = 2, int NPR = 9>
__global__ __launch_bounds__(WM * WN * 64) void a0_igemm_x9_group_kernel(a0_x9_group<OA, OB, EP> G, int X, int Y, int K, int kchunk, int gx, int gy) {
    const int g = blockIdx.y;
    a0_igemm_x9_body<OA, OB, EP, WM, WN, MT, NT, KS, NPR>(G.pa[g], G.pb[g], G.pe[g], X, Y, K, kchunk, gx, gy);
}

template <class OA, class OB, class EP, int WM, int WN, int MT, int NT, int KS = 2>
static inline hipError_t a0_igemm_x9_group_launch(hipStream_t st, int n, const a0_x9_group<OA, OB, EP>& grp, int X, int Y, int K, int splits, hipEvent_t ev0 = nullptr,
                                                  hipEvent_t ev1 = nullptr) {
    typedef a0_x9_geom<OA, OB, WM, WN, MT, NT, KS> G;
    const int six = a0_x9_products_now() == 6;
    auto kern = six ? a0_igemm_x9_group_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 6> : a0_igemm_x9_group_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 9>;
    static bool configured[2] = {false, false};
    if (!configured[six]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured[six] = true;
    }
    if (splits < 1) splits = 1;
    constexpr int BK = 16 * KS;
    const int ktiles = (K + BK - 1) / BK;
    const int kchunk = ((ktiles + splits - 1) / splits) * BK;
    const int gx = (X + G::BX - 1) / G::BX, gy = (Y + G::BY - 1) / G::BY;
    if (ev0) hipExtLaunchKernelGGL(kern, dim3((unsigned)(gx * gy * splits), (unsigned)n), dim3(WM * WN * 64), (uint32_t)G::LDS_BYTES, st, ev0, ev1, 0, grp, X, Y, K, kchunk, gx, gy);
    else hipLaunchKernelGGL(kern, dim3((unsigned)(gx * gy * splits), (unsigned)n), dim3(WM * WN * 64), G::LDS_BYTES, st, grp, X, Y, K, kchunk, gx, gy);
    return hipGetLastError();
}

// Two DIFFERENT GEMMs with the same tile shape and the same number of workgroups in one launch (blockIdx.y picks the body): fc1's data gradient and fc1's weight gradient of
// a 512-row batch are 392 tiles each on a chip with 512 workgroup slots — alone each leaves a quarter of the slots empty and ends on the CUs that hold two workgroups; together
// their 784 workgroups keep the slots filled (1.5 rounds instead of 2 x 1).  Same bodies, same tiles: bit-identical results.
template <class OA1, class OB1, class EP1, class OA2, class OB2, class EP2, int WM, int WN, int MT, int NT, int KS = 2, int NPR = 9>
__global__ __launch_bounds__(WM * WN * 64) void a0_igemm_x9_pair_kernel(typename OA1::Params pa1, typename OB1::Params pb1, typename EP1::Params pe1, int X1, int Y1, int K1, int kc1,
                                                                int gx1, int gy1, typename OA2::Params pa2, typename OB2::Params pb2, typename EP2::Params pe2, int X2, int Y2,
                                                                int K2, int kc2, int gx2, int gy2) {
    // (round 5: the two problems may have different numbers of tiles — the grid is as long as the longer one, and a workgroup whose logical tile, by the body's own
    // id -> tile map, does not exist in its problem leaves at once)
    const int n = gridDim.x, id = blockIdx.x, per = n >> 3;
    const int L = (id < (per << 3)) ? (id & 7) * per + (id >> 3) : id;
    if (blockIdx.y == 0) { if (L < gx1 * gy1) a0_igemm_x9_body<OA1, OB1, EP1, WM, WN, MT, NT, KS, NPR>(pa1, pb1, pe1, X1, Y1, K1, kc1, gx1, gy1); }
    else if (L < gx2 * gy2) a0_igemm_x9_body<OA2, OB2, EP2, WM, WN, MT, NT, KS, NPR>(pa2, pb2, pe2, X2, Y2, K2, kc2, gx2, gy2);
}

template <class OA1, class OB1, class EP1, class OA2, class OB2, class EP2, int WM, int WN, int MT, int NT, int KS = 2>
static inline hipError_t a0_igemm_x9_pair_launch(hipStream_t st, const typename OA1::Params& pa1, const typename OB1::Params& pb1, const typename EP1::Params& pe1, int X1, int Y1, int K1,
                                                 const typename OA2::Params& pa2, const typename OB2::Params& pb2, const typename EP2::Params& pe2, int X2, int Y2, int K2) {
    typedef a0_x9_geom<OA1, OB1, WM, WN, MT, NT, KS> G1;
    typedef a0_x9_geom<OA2, OB2, WM, WN, MT, NT, KS> G2;
    constexpr int LDS = G1::LDS_BYTES > G2::LDS_BYTES ? G1::LDS_BYTES : G2::LDS_BYTES;
    const int six = a0_x9_products_now() == 6;
    auto kern = six ? a0_igemm_x9_pair_kernel<OA1, OB1, EP1, OA2, OB2, EP2, WM, WN, MT, NT, KS, 6> : a0_igemm_x9_pair_kernel<OA1, OB1, EP1, OA2, OB2, EP2, WM, WN, MT, NT, KS, 9>;
    static bool configured[2] = {false, false};
    if (!configured[six]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured[six] = true;
    }
    constexpr int BK = 16 * KS;
    const int kc1 = ((K1 + BK - 1) / BK) * BK, kc2 = ((K2 + BK - 1) / BK) * BK;          // no reduction split
    const int gx1 = (X1 + G1::BX - 1) / G1::BX, gy1 = (Y1 + G1::BY - 1) / G1::BY, gx2 = (X2 + G2::BX - 1) / G2::BX, gy2 = (Y2 + G2::BY - 1) / G2::BY;
    const int gmax = gx1 * gy1 > gx2 * gy2 ? gx1 * gy1 : gx2 * gy2;
    hipLaunchKernelGGL(kern, dim3((unsigned)gmax, 2), dim3(WM * WN * 64), LDS, st, pa1, pb1, pe1, X1, Y1, K1, kc1, gx1, gy1, pa2, pb2, pe2, X2, Y2, K2, kc2, gx2, gy2);
    return hipGetLastError();
}

// The pair above plus a SECOND instance of its second body on a smaller problem (round 5): fc1's data gradient, fc1's weight gradient and the HEAD's weight gradient all
// depend on the loss kernel's outputs only, and the head's few tiles (8 for a scalar head, 56 for qr's 800 columns) fit into the slots the pair leaves empty — its launch
// (6.5 - 10.9 us alone, 20 times per block) disappears.  blockIdx.y names the problem; the third problem's grid column is as long as the pair's, workgroups without a tile
// of it leave at once.
template <class OA1, class OB1, class EP1, class OA2, class OB2, class EP2, int WM, int WN, int MT, int NT, int KS = 2, int NPR = 9>
__global__ __launch_bounds__(WM * WN * 64) void a0_igemm_x9_trio_kernel(typename OA1::Params pa1, typename OB1::Params pb1, typename EP1::Params pe1, int X1, int Y1, int K1, int kc1,
                                                                int gx1, int gy1, typename OA2::Params pa2, typename OB2::Params pb2, typename EP2::Params pe2, int X2, int Y2,
                                                                int K2, int kc2, int gx2, int gy2, typename OA2::Params pa3, typename OB2::Params pb3, typename EP2::Params pe3,
                                                                int X3, int Y3, int K3, int kc3, int gx3, int gy3) {
    // the small problem takes grid row 0: rows are dispatched in order, so its workgroups start with the launch instead of behind the pair's 784 (measured: in the last
    // row they began when the pair's slots drained and added their whole duration to the launch's tail)
    if (blockIdx.y == 1) a0_igemm_x9_body<OA1, OB1, EP1, WM, WN, MT, NT, KS, NPR>(pa1, pb1, pe1, X1, Y1, K1, kc1, gx1, gy1);
    else if (blockIdx.y == 2) a0_igemm_x9_body<OA2, OB2, EP2, WM, WN, MT, NT, KS, NPR>(pa2, pb2, pe2, X2, Y2, K2, kc2, gx2, gy2);
    else {
        const int n = gridDim.x, id = blockIdx.x, per = n >> 3;          // the body's own id -> logical tile map: only tiles [0, gx3 * gy3) exist
        const int L = (id < (per << 3)) ? (id & 7) * per + (id >> 3) : id;
        if (L >= gx3 * gy3) return;
        a0_igemm_x9_body<OA2, OB2, EP2, WM, WN, MT, NT, KS, NPR>(pa3, pb3, pe3, X3, Y3, K3, kc3, gx3, gy3);
    }
}

template <class OA1, class OB1, class EP1, class OA2, class OB2, class EP2, int WM, int WN, int MT, int NT, int KS = 2>
static inline hipError_t a0_igemm_x9_trio_launch(hipStream_t st, const typename OA1::Params& pa1, const typename OB1::Params& pb1, const typename EP1::Params& pe1, int X1, int Y1, int K1,
                                                 const typename OA2::Params& pa2, const typename OB2::Params& pb2, const typename EP2::Params& pe2, int X2, int Y2, int K2,
                                                 const typename OA2::Params& pa3, const typename OB2::Params& pb3, const typename EP2::Params& pe3, int X3, int Y3, int K3) {
    typedef a0_x9_geom<OA1, OB1, WM, WN, MT, NT, KS> G1;
    typedef a0_x9_geom<OA2, OB2, WM, WN, MT, NT, KS> G2;
    constexpr int LDS = G1::LDS_BYTES > G2::LDS_BYTES ? G1::LDS_BYTES : G2::LDS_BYTES;
    const int six = a0_x9_products_now() == 6;
    auto kern = six ? a0_igemm_x9_trio_kernel<OA1, OB1, EP1, OA2, OB2, EP2, WM, WN, MT, NT, KS, 6> : a0_igemm_x9_trio_kernel<OA1, OB1, EP1, OA2, OB2, EP2, WM, WN, MT, NT, KS, 9>;
    static bool configured[2] = {false, false};
    if (!configured[six]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured[six] = true;
    }
    constexpr int BK = 16 * KS;
    const int kc1 = ((K1 + BK - 1) / BK) * BK, kc2 = ((K2 + BK - 1) / BK) * BK, kc3 = ((K3 + BK - 1) / BK) * BK;          // no reduction split
    const int gx1 = (X1 + G1::BX - 1) / G1::BX, gy1 = (Y1 + G1::BY - 1) / G1::BY, gx2 = (X2 + G2::BX - 1) / G2::BX, gy2 = (Y2 + G2::BY - 1) / G2::BY;
    const int gx3 = (X3 + G2::BX - 1) / G2::BX, gy3 = (Y3 + G2::BY - 1) / G2::BY;
    if (gx1 * gy1 != gx2 * gy2 || gx3 * gy3 > gx1 * gy1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)(gx1 * gy1), 3), dim3(WM * WN * 64), LDS, st, pa1, pb1, pe1, X1, Y1, K1, kc1, gx1, gy1, pa2, pb2, pe2, X2, Y2, K2, kc2, gx2, gy2,
                       pa3, pb3, pe3, X3, Y3, K3, kc3, gx3, gy3);
    return hipGetLastError();
}

template <class OA, class OB, class EP, int WM, int WN, int MT, int NT, int KS = 2>
static inline hipError_t a0_igemm_x9_launch(hipStream_t st, const typename OA::Params& pa, const typename OB::Params& pb,
                                            const typename EP::Params& pe, int X, int Y, int K, int splits, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
    typedef a0_x9_geom<OA, OB, WM, WN, MT, NT, KS> G;
    if (splits < 1) splits = 1;
    constexpr int BK = 16 * KS;
    // FULL (round 6): no partial k tile anywhere in the launch (K a multiple of the tile depth) and plain matrix operands without the row-sum by-product: the staging
    // keeps and applies no validity masks (a fifth of its vector instructions; the vector issue port, not the matrix pipe, paces this kernel: profiles/r06_pmc_gemm.md)
    constexpr bool can_full = a0_x9_plain<OA>::value && a0_x9_plain<OB>::value && !EP::ROWSUM_A;
    static const bool full_off = getenv("A0_X9_NO_FULL") != nullptr;      // tuning aid (same bits)
    const int six = a0_x9_products_now() == 6;
    const int full = (can_full && !full_off && (K % BK) == 0) ? 1 : 0;
    typedef void (*a0_kern_t)(typename OA::Params, typename OB::Params, typename EP::Params, int, int, int, int, int, int);
    a0_kern_t kern;
    if constexpr (can_full) kern = full ? (six ? (a0_kern_t)a0_igemm_x9_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 6, true> : (a0_kern_t)a0_igemm_x9_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 9, true>)
                                        : (six ? (a0_kern_t)a0_igemm_x9_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 6> : (a0_kern_t)a0_igemm_x9_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 9>);
    else kern = six ? (a0_kern_t)a0_igemm_x9_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 6> : (a0_kern_t)a0_igemm_x9_kernel<OA, OB, EP, WM, WN, MT, NT, KS, 9>;
    static bool configured[4] = {false, false, false, false};           // per instantiation: more than 64 KB of dynamic LDS needs the attribute
    if (!configured[2 * full + six]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured[2 * full + six] = true;
    }
    const int ktiles = (K + BK - 1) / BK;
    const int kchunk = ((ktiles + splits - 1) / splits) * BK;
    const int gx = (X + G::BX - 1) / G::BX, gy = (Y + G::BY - 1) / G::BY;
    // ev0 / ev1 (the profiler probe, net.hip): the launch carries the event pair — the dispatch's own begin / end timestamps
    if (ev0) hipExtLaunchKernelGGL(kern, dim3((unsigned)(gx * gy * splits)), dim3(WM * WN * 64), (uint32_t)G::LDS_BYTES, st, ev0, ev1, 0, pa, pb, pe, X, Y, K, kchunk, gx, gy);
    else hipLaunchKernelGGL(kern, dim3((unsigned)(gx * gy * splits)), dim3(WM * WN * 64), G::LDS_BYTES, st, pa, pb, pe, X, Y, K, kchunk, gx, gy);
    return hipGetLastError();
}

// operands whose source is fp32 (everything but the u8 frames, which the fused conv1 kernels cover)
template <class OP> struct a0_x9_ok { static constexpr bool value = sizeof(typename OP::Raw) == sizeof(a0_f4); };

#endif  // __HIPCC__
