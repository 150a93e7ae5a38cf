// devutil.hpp -- device-side building blocks shared by the kernels (gfx950, wave64).
//
// Arithmetic contract (DESIGN.md "Numerics"): everything the forest and the
// approximate search compute is defined by
//   * wave_dot(): the 64-lane canonical dot product -- lane l owns elements
//     256*k + 4*l + c, one fmaf chain per c over k, folded (a0+a1)+(a2+a3),
//     lanes combined by an xor butterfly 32,16,8,4,2,1;
//   * ang_dist(): annoy's Angular::distance with its double literals;
//   * Kiss32 streams seeded per (tree, level, segment start, attempt).
// This file is compiled with -ffp-contract=off: a*b+c stays two roundings unless
// written as fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace morna {

// ---------------------------------------------------------------- hashing / RNG

__host__ __device__ inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// MurmurHash3_x86_32 (what mmh3.hash computes, morna.py:369), unsigned result.
__host__ __device__ inline uint32_t murmur3_32(const uint8_t *key, int64_t len, uint32_t seed)
{
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h1 = seed;
    const int64_t nblocks = len >> 2;
    for (int64_t i = 0; i < nblocks; i++) {
        const uint8_t *b = key + 4 * i;
        uint32_t k1 = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        k1 *= c1;
        k1 = rotl32(k1, 15);
        k1 *= c2;
        h1 ^= k1;
        h1 = rotl32(h1, 13);
        h1 = h1 * 5u + 0xe6546b64u;
    }
    const uint8_t *tail = key + nblocks * 4;
    uint32_t k1 = 0;
    const int rem = (int)(len & 3);
    if (rem == 3) k1 ^= (uint32_t)tail[2] << 16;
    if (rem >= 2) k1 ^= (uint32_t)tail[1] << 8;
    if (rem >= 1) {
        k1 ^= tail[0];
        k1 *= c1;
        k1 = rotl32(k1, 15);
        k1 *= c2;
        h1 ^= k1;
    }
    h1 ^= (uint32_t)len;
    return fmix32(h1);
}

// Python's floored `h % dim` for dim > 0 (morna.py:371).
__host__ __device__ inline int32_t floored_mod(int32_t h, int32_t dim)
{
    int32_t r = h % dim;
    return r < 0 ? r + dim : r;
}

// annoy's Kiss32Random
struct Kiss32 {
    uint32_t x, y, z, c;
    __host__ __device__ explicit Kiss32(uint32_t seed) : x(seed), y(362436000u), z(521288629u), c(7654321u) {}
    __host__ __device__ inline uint32_t next()
    {
        x = 69069u * x + 12345u;
        y ^= y << 13;
        y ^= y >> 17;
        y ^= y << 5;
        uint64_t t = 698769069ULL * z + c;
        c = (uint32_t)(t >> 32);
        z = (uint32_t)t;
        return x + y + z;
    }
    __host__ __device__ inline uint32_t index(uint32_t n) { return next() % n; }
};

// seed of the stream owned by one split attempt of one node
__host__ __device__ inline uint32_t node_seed(uint32_t seed, uint32_t tree, uint32_t level, uint32_t start,
                                              uint32_t attempt)
{
    uint32_t h = fmix32(seed + 0x9E3779B9u * (tree + 1u));
    h = fmix32(h ^ (level * 0x85ebca6bu + 0x27d4eb2fu));
    h = fmix32(h ^ start);
    h = fmix32(h + attempt * 0xc2b2ae35u + 0x165667b1u);
    return h ? h : 1u;
}

// coin flip for the item at position i of the node's segment (margin == 0, fallback)
__host__ __device__ inline int pos_flip(uint32_t nseed, uint32_t i)
{
    return (int)(fmix32(nseed ^ fmix32(i + 0x632BE5ABu)) & 1u);
}

// annoy's _split_imbalance
__host__ __device__ inline double split_imbalance(int64_t left, int64_t right)
{
    double ls = (float)left, rs = (float)right;
    float f = (float)(ls / (ls + rs + 1e-9));
    return f > 1 - f ? f : 1 - f;
}

// ------------------------------------------------------------------- arithmetic

// Angular::distance, T = float fields with double literals.
__device__ inline float ang_dist(float pp, float qq, float pq)
{
    float ppqq = pp * qq;
    if (ppqq > 0) return (float)(2.0 - 2.0 * (double)pq / (double)sqrtf(ppqq));
    return 2.0f;
}

// A wave-uniform look-up issued through the VECTOR memory path.  Scalar loads return out of order, so the first
// s_waitcnt lgkmcnt(0) after one -- every LDS read, every constant look-up -- waits for it in full; vector loads
// are counted in order and can stay in flight for a whole two_means step (its prefetches are a step or more ahead).
template <typename T>
__device__ inline T vgather(const T *base, uint32_t i)
{
    asm volatile("" : "+v"(i));   // the compiler no longer knows that all lanes hold the same index
    return base[i];
}

// One step of annoy's two_means centroid update for four elements, (c * n + x / |x|) / (n + 1), every
// operation rounded to fp32 as the scalar code rounds it, without the two IEEE division sequences per element.
//
// x / |x| goes through the fp64 reciprocal r1 = RN64(1 / |x|) (RowInfo): a quotient a / b of two floats is never
// exactly on, nor within 2^-49 (relative) of, a rounding boundary of float (a - b * mid is a non-zero multiple
// of ulp(b) * ulp(mid) for every midpoint mid), and RN64(a * r1) is within 2^-52 of a / b, so its nearest float
// IS RN32(a / b).  Below the normal range the spacing of floats is fixed and a quotient CAN be an exact tie;
// whether x_i / |x| can land there at all is a property of the row, found once when its norm is taken: such
// rows take the real divisions for the whole step (`force`).
//
// t / (n + 1), a division by a small integer m <= 202, stays in fp32: with y = RN32(1 / m), q = RN(t * y) is
// within ~2 ulp of t / m; e = q * m - t is then a multiple of ulp(q) no larger than ~400 ulp(q): the FMA
// delivers it exactly; and q - e * y differs from t / m by at most |e| / m * 2^-24 <= 2^-23 ulp(q), while t / m
// keeps at least ulp(q) / (2 m) away from every rounding boundary (t - m * mid is a non-zero multiple of half
// an ulp): o = RN(q - e * y) IS RN32(t / m).  Written as fma(q, m, -t) and fma(-e, y, q) the zero results keep
// the sign of t.  The argument needs finite operands and a normal result.  At an exact tie below FLT_MIN the
// route returns one of the two neighbours, and when it is the wrong (odd) one that is a non-zero subnormal; a
// non-finite t gives NaN.  So a result that is neither subnormal nor non-finite is right, and the rare others
// are recomputed with real divisions (wave-uniform branch, laid out off the hot path; kernels run with fp32
// denormals on).  All 64 lanes must call.  Three packed operations per two elements instead of six.

// the step as written: two real divisions per element
__device__ inline float4 centroid_div4(const float4 c, const float4 x, float f0, float f1, float norm)
{
    float4 o;
    o.x = (c.x * f0 + x.x / norm) / f1;
    o.y = (c.y * f0 + x.y / norm) / f1;
    o.z = (c.z * f0 + x.z / norm) / f1;
    o.w = (c.w * f0 + x.w / norm) / f1;
    return o;
}
// lanes whose v is a non-zero subnormal, an infinity or a NaN: one v_cmp_class_f32 straight into a lane mask
// (the bool route through __builtin_amdgcn_classf + ballot costs two more VALU operations per value)
__device__ inline unsigned long long suspect_lanes(float v)
{
    unsigned long long m;
    asm("v_cmp_class_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v), "s"(0x297));   // NaNs | infinities | denormals
    return m;
}
// r1 = RN64(1 / norm), y = RN32(1 / f1); `force`: all ones for a row whose x / |x| may have a subnormal element
__device__ inline float4 centroid_step4(const float4 c, const float4 x, float f0, float f1, float norm, double r1,
                                        float y, unsigned long long force)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float nx = (float)((double)x.x * r1), ny = (float)((double)x.y * r1);
    const float nz = (float)((double)x.z * r1), nw = (float)((double)x.w * r1);
    const f32x2 f02 = {f0, f0}, m2 = {f1, f1}, y2 = {y, y};
    const f32x2 t01 = (f32x2){c.x, c.y} * f02 + (f32x2){nx, ny};   // two roundings each (-ffp-contract=off)
    const f32x2 t23 = (f32x2){c.z, c.w} * f02 + (f32x2){nz, nw};
    const f32x2 q01 = t01 * y2, q23 = t23 * y2;
    const f32x2 e01 = __builtin_elementwise_fma(q01, m2, -t01), e23 = __builtin_elementwise_fma(q23, m2, -t23);
    const f32x2 o01 = __builtin_elementwise_fma(-e01, y2, q01), o23 = __builtin_elementwise_fma(-e23, y2, q23);
    float4 o = make_float4(o01.x, o01.y, o23.x, o23.y);
    const unsigned long long sus = ((suspect_lanes(o.x) | suspect_lanes(o.y)) | (suspect_lanes(o.z) | suspect_lanes(o.w))) | force;
    if (__builtin_expect(sus != 0, 0)) o = centroid_div4(c, x, f0, f1, norm);   // wave-uniform
    return o;
}

// order-preserving map float -> uint32 (total order of non-NaN values)
__device__ inline uint32_t f32_orderable(float f)
{
    if (f == 0.f) f = 0.f;   // -0.0 and +0.0 compare equal on the CPU: one key for both
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float f32_from_orderable(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ inline float wave_sum_xor(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, WAVE);
    return v;
}

// The value lane 0 holds after the xor butterfly 32,16,8,4,2,1 -- same binary tree, same
// roundings -- computed with VALU cross-lane moves only (v_permlane32_swap, v_permlane16_swap,
// DPP row_shl) instead of six dependent ds_bpermute round trips, then broadcast to the wave.
// At stage s only lanes < s matter, and lane l < s needs lane l + s: a half swap, a row swap and
// four in-row shifts deliver exactly those partners.
__device__ inline float wave_sum_lane0(float v)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    unsigned u = __float_as_uint(v);
    u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[1], lanes 0-31: lanes 32-63 of u
    v = v + __uint_as_float(r[1]);
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane16_swap(u, u, false, false);         // r[1], rows 0/2: rows 1/3 of u
    v = v + __uint_as_float(r[1]);
    // row_shl:n -- lane i reads lane i + n of its 16-lane row (0 when out of the row)
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x108, 0xf, 0xf, true));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0xf, true));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x102, 0xf, 0xf, true));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// GT wave sums at once, each with exactly the additions of wave_sum_lane0 (same pairs at every
// stage; an addition may have its operands swapped, which is exact).  At stage 32 two values share
// one v_permlane32_swap (the lower half of the wave goes on with one value, the upper half with the
// other), at stage 16 two registers share one v_permlane16_swap, at stage 8 two registers share a
// pair of DPP shifts: 8 values cost 18 VALU operations instead of 8 x 12, and no chain waits for
// another.  Value j ends in lane j * (64 / GT); the other lanes hold partial sums.
template <int CTRL>
__device__ inline float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ inline float swap32_add(float a, float b)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // r[0] = {a.lo, b.lo}, r[1] = {a.hi, b.hi}: lanes 0-31 add a's halves, lanes 32-63 b's
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float swap16_add(float a, float b)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // r[0] = rows {a0, b0, a2, b2}, r[1] = rows {a1, b1, a3, b3}
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int GT>
__device__ inline float wave_sum_multi(const float (&v)[GT], int lane)
{
    static_assert(GT == 2 || GT == 3 || GT == 4 || GT == 8, "wave_sum_multi: 2, 3, 4 or 8 values");
    float u;
    if constexpr (GT == 3) {   // as four values with a zero in the fourth place: value j ends in lane 16 j
        const float v4[4] = {v[0], v[1], v[2], 0.f};
        return wave_sum_multi<4>(v4, lane);
    }
    if constexpr (GT == 8) {
        const float s0 = swap32_add(v[0], v[4]), s1 = swap32_add(v[1], v[5]);
        const float s2 = swap32_add(v[2], v[6]), s3 = swap32_add(v[3], v[7]);
        const float t0 = swap16_add(s0, s2);   // rows: v0, v2, v4, v6
        const float t1 = swap16_add(s1, s3);   // rows: v1, v3, v5, v7
        const float x = t0 + dpp_move<0x108>(t0);   // row_shl:8 -- lanes 0-7 of a row: t0[l] + t0[l + 8]
        const float y = dpp_move<0x118>(t1) + t1;   // row_shr:8 -- lanes 8-15 of a row: t1[l - 8] + t1[l]
        u = (lane & 8) ? y : x;                     // 8-lane groups: v0, v1, v2, ... v7
    } else if constexpr (GT == 4) {
        const float s0 = swap32_add(v[0], v[2]), s1 = swap32_add(v[1], v[3]);
        u = swap16_add(s0, s1);                     // rows: v0, v1, v2, v3
        u = u + dpp_move<0x108>(u);
    } else {
        u = swap32_add(v[0], v[1]);                 // halves: v0, v1
        u = swap16_add(u, u);                       // rows 0 / 2: u[l] + u[l + 16]
        u = u + dpp_move<0x108>(u);
    }
    u = u + dpp_move<0x104>(u);
    u = u + dpp_move<0x102>(u);
    u = u + dpp_move<0x101>(u);
    return u;
}

// fold of the four per-lane chains, then the butterfly (lane 0's value, broadcast)
__device__ inline float wave_dot_finish(float a0, float a1, float a2, float a3)
{
    return wave_sum_lane0((a0 + a1) + (a2 + a3));
}

// The lane's four fmaf chains, held as two register PAIRS (chains 0,1 and 2,3) so that each
// k-step is two v_pk_fma_f32 on the (x,y) and (z,w) halves of the float4 operands as they
// sit in registers -- no operand shuffling.  Each component is an IEEE fmaf, exactly the
// scalar chains of the contract.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Acc4 {
    f32x2 lo, hi;
};
__device__ inline Acc4 acc4_zero()
{
    Acc4 a;
    a.lo = (f32x2){0.f, 0.f};
    a.hi = (f32x2){0.f, 0.f};
    return a;
}
__device__ inline void fma4(Acc4 &a, const float4 &x, const float4 &y)
{
    a.lo = __builtin_elementwise_fma((f32x2){x.x, x.y}, (f32x2){y.x, y.y}, a.lo);
    a.hi = __builtin_elementwise_fma((f32x2){x.z, x.w}, (f32x2){y.z, y.w}, a.hi);
}
__device__ inline float acc4_finish(const Acc4 &a) { return wave_dot_finish(a.lo.x, a.lo.y, a.hi.x, a.hi.y); }

// Canonical dot of two vectors of nvec float4 (both 16-byte aligned, zero padded
// to a multiple of 4 floats).  All 64 lanes of the wave must call it; every lane
// receives the result.  a and b may be in any address space the compiler can
// resolve (global rows, LDS images).
template <typename PA, typename PB>
__device__ inline float wave_dot(PA a, PB b, int nvec, int lane)
{
    Acc4 s = acc4_zero();
    int i = lane;
    // 4 independent 1-KiB loads in flight per operand before the first use (8 measured no faster)
    for (; i + 3 * WAVE < nvec; i += 4 * WAVE) {
        float4 x0 = a[i], x1 = a[i + WAVE], x2 = a[i + 2 * WAVE], x3 = a[i + 3 * WAVE];
        float4 y0 = b[i], y1 = b[i + WAVE], y2 = b[i + 2 * WAVE], y3 = b[i + 3 * WAVE];
        fma4(s, x0, y0);
        fma4(s, x1, y1);
        fma4(s, x2, y2);
        fma4(s, x3, y3);
    }
    for (; i < nvec; i += WAVE) {
        float4 x = a[i], y = b[i];
        fma4(s, x, y);
    }
    return acc4_finish(s);
}

// 64-bit butterfly helpers for (key) reductions
__device__ inline uint64_t shfl_xor_u64(uint64_t v, int off)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, off, WAVE);
    hi = __shfl_xor(hi, off, WAVE);
    return ((uint64_t)hi << 32) | lo;
}
__device__ inline uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint64_t o = shfl_xor_u64(v, off);
        v = o < v ? o : v;
    }
    return v;
}
// The wave's maximum without LDS round trips (v_permlane32_swap, v_permlane16_swap, DPP row shifts instead of the twelve
// ds_bpermute of the butterfly below); 0 is the identity (lanes shifted in from outside a row read 0).  Every lane gets it.
__device__ inline uint64_t wave_max_u64_fast(uint64_t v)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    auto take = [&](uint32_t olo, uint32_t ohi) {
        const uint64_t o = ((uint64_t)ohi << 32) | olo, m = ((uint64_t)hi << 32) | lo;
        if (o > m) { lo = olo; hi = ohi; }
    };
    {
        const u32x2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        take(a[1], b[1]);   // lanes 0-31: lanes 32-63
    }
    {
        const u32x2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        take(a[1], b[1]);   // rows 0 / 2: rows 1 / 3
    }
#define MORNA_DPP_MAX(CTRL)                                                                        \
    take((uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true),                  \
         (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true))
    MORNA_DPP_MAX(0x108);   // row_shl:8
    MORNA_DPP_MAX(0x104);
    MORNA_DPP_MAX(0x102);
    MORNA_DPP_MAX(0x101);
#undef MORNA_DPP_MAX
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
    return ((uint64_t)hi << 32) | lo;
}

__device__ inline uint64_t wave_max_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint64_t o = shfl_xor_u64(v, off);
        v = o > v ? o : v;
    }
    return v;
}

}  // namespace morna
