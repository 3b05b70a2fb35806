// BN254 Fr products in radix 2^29 for the S-box of the matrix-core pipeline (bn_mfma.cuh).
//
// gfx950 multiplies 32 x 32 -> 64 with a 64-bit addend in ONE instruction (v_mad_u64_u32) but has no carry-in: with 32-bit limbs every
// product of a column needs a second instruction to catch the carry out of the 64-bit accumulator (bn_field.cuh: two instructions per
// 32x32 product, 128 products per Montgomery multiplication).  With NINE limbs of 29 bits a product is below 2^58 and a whole column --
// nine products of a x b and nine of m x r -- stays below 2^63: no carry instruction at all.  81 + 81 multiply-adds instead of
// 128 + 128 instructions; a squaring needs 45 + 81 (the doubled operand 2 a_j still fits a 32-bit register); r = 1 mod 2^28 makes
// -1/r mod 2^29 = 2^28 - 1 (the reduction digit: one 32-bit multiplication by a constant and a mask).
//
// The Montgomery radix is 2^261 here, not the 2^256 of the state: M(a, b) = a b / 2^261.  x^5 through two squarings and a product comes
// out as X^5 / 2^1044 for a state value X = x 2^256, i.e. the state form of x^5 times 2^-20.  The constant 2^20 is multiplied into the
// tiles of the linear layer that reads the S-box's output (bn128.hip: mfma_tile's `sboxed` operands), so nothing is paid for it.
// Operands are lazy representatives below 0.9 * 2^256 (eight 32-bit words in, eight out); results are below 2^252 + r.
#pragma once
#include <stdint.h>

namespace bn29 {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u32 MASK = (1u << 29) - 1;
__device__ __forceinline__ u32 r29(int i) {
    constexpr u32 R[9] = { 0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu };
    return R[i];
}

// eight 32-bit words -> nine 29-bit limbs
__device__ __forceinline__ void to29(const u32 x[8], u32 a[9]) {
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, w = bit >> 5, s = bit & 31;
        u64 v = x[w];
        if (w + 1 < 8) v |= (u64)x[w + 1] << 32;
        a[j] = (u32)(v >> s) & MASK;
    }
}
// nine 29-bit limbs (value below 2^256) -> eight 32-bit words
__device__ __forceinline__ void from29(const u32 a[9], u32 x[8]) {
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, j = bit / 29, s = bit - 29 * j;           // word w starts inside limb j at bit s
        u64 v = (u64)a[j] >> s;
        if (j + 1 < 9) v |= (u64)a[j + 1] << (29 - s);
        if (j + 2 < 9 && 58 - s < 32) v |= (u64)a[j + 2] << (58 - s);
        x[w] = (u32)v;
    }
}

#ifndef BN29_DIGIT_MUL
#define BN29_DIGIT_MUL 1
#endif
#ifndef BN29_COLUMNS_C
#define BN29_COLUMNS_C 0
#endif
#if BN29_COLUMNS_C
// (the columns as C templates, round 5: hipcc splits every column into two chains and pays six glue instructions for it -- kept for A/B builds only)
// one column: acc += sum a_i b_(K-i) (or, SQR, sum_{i < K-i} a_i (2a)_(K-i) + a_(K/2)^2) + sum m_i r_(K-i) over the m already known
template <int K, bool SQR>
__device__ __forceinline__ void column(u64 &acc, const u32 a[9], const u32 b[9], const u32 m[9]) {
    constexpr int I0 = K > 8 ? K - 8 : 0, I1 = K < 8 ? K : 8;
    if constexpr (!SQR) {
#pragma unroll
        for (int i = I0; i <= I1; i++) acc = (u64)a[i] * b[K - i] + acc;
    } else {
#pragma unroll
        for (int i = I0; i <= I1; i++) {
            if (2 * i < K) acc = (u64)a[i] * b[K - i] + acc;              // b = 2a: the pair (i, K-i) once, doubled
            else if (2 * i == K) acc = (u64)a[i] * a[i] + acc;
        }
    }
    constexpr int M1 = K < 9 ? K - 1 : 8;                                 // m_i r_(K-i), i < K (i <= 8), K - i <= 8
#pragma unroll
    for (int i = I0; i <= M1; i++) acc = (u64)m[i] * r29(K - i) + acc;
}
template <int K, bool SQR>
__device__ __forceinline__ void columns(u64 &acc, const u32 a[9], const u32 b[9], u32 m[9], u32 out[9]) {
    column<K, SQR>(acc, a, b, m);
    if constexpr (K < 9) {
        const u32 lo = (u32)acc;
        u32 sh = lo << 28, r0 = r29(0);
        asm("" : "+v"(sh), "+s"(r0));                                     // opaque: hipcc would turn the two cheap steps back into a (quarter-rate) multiplication,
        m[K] = (sh - lo) & MASK;                                          //   lo * (2^28 - 1) mod 2^29 = -lo / r mod 2^29, and the product by r_0 = 2^28 + 1 into 64-bit shifts and adds
        acc = (u64)m[K] * r0 + acc;                                       // the low 29 bits are now zero
    } else out[K - 9] = (u32)acc & MASK;
    {   // acc >>= 29 as two full-rate 32-bit steps (the 64-bit shift instruction runs at a quarter of the rate)
        u32 lo = (u32)acc, hi = (u32)(acc >> 32);
        asm("" : "+v"(lo), "+v"(hi));                                     // (opaque halves: hipcc would fuse the steps back into the 64-bit shift)
        u32 nlo = __builtin_amdgcn_alignbit(hi, lo, 29), nhi = hi >> 29;
        asm("" : "+v"(nlo), "+v"(nhi));
        acc = ((u64)nhi << 32) | nlo;
    }
    if constexpr (K < 16) columns<K + 1, SQR>(acc, a, b, m, out);
}
// out = a b / 2^261 mod r (+ r at most).  SQR: b must be the limbs of a, DOUBLED (2 a_j each); the result is a^2 / 2^261.
template <bool SQR>
__device__ __forceinline__ void mont(u32 out[9], const u32 a[9], const u32 b[9]) {
    u32 m[9];
    u64 acc = 0;
    columns<0, SQR>(acc, a, b, m, out);
    out[8] = (u32)acc;
}

#else
// The limbs of r in scalar registers (one constant-bus operand per multiply-add), made opaque once per S-box.
struct RLimbs { u32 r[9]; };
__device__ __forceinline__ RLimbs r_limbs() {
    RLimbs R;
#pragma unroll
    for (int i = 0; i < 9; i++) { R.r[i] = r29(i); asm volatile("" : "+s"(R.r[i])); }
    return R;
}
// the reduction digit of a column: -lo / r mod 2^29 = lo (2^28 - 1) mod 2^29 (r = 1 mod 2^28): one multiplication and a mask (round 6: 27.3 against 27.5 ms
// at 2^20 x 100 for the shift-and-subtract form, three instructions; BN29_DIGIT_MUL=0 keeps it for A/B builds)
__device__ __forceinline__ u32 digit(u32 lo) {
#if BN29_DIGIT_MUL
    u32 k = (1u << 28) - 1;
    asm("" : "+s"(k));
    return (lo * k) & MASK;
#else
    u32 sh = lo << 28;
    asm("" : "+v"(sh));                                                   // opaque: hipcc would turn the two full-rate steps back into a multiplication
    return (sh - lo) & MASK;
#endif
}
// The seventeen columns of a product / a squaring, straight-line (gen_bn29_columns.py): every column ONE chain of multiply-adds on one
// 64-bit accumulator (a x b, then m x r over the digits already known), the digit, and one 64-bit shift.
#include "bn_field29_columns.inc"
template <bool SQR>
__device__ __forceinline__ void mont(u32 out[9], const u32 a[9], const u32 b[9], const RLimbs &R) {
    if constexpr (SQR) mont_sqr_columns(out, a, b, R);
    else mont_mul_columns(out, a, b, R);
}
#endif

// x (eight words, below 0.9 * 2^256) -> x^5 / 2^1044 (eight words, below 2^252 + r)
__device__ __forceinline__ void pow5(u32 x[8]) {
    u32 a[9], d[9], s2[9], s4[9];
    to29(x, a);
#if BN29_COLUMNS_C
#pragma unroll
    for (int j = 0; j < 9; j++) d[j] = a[j] << 1;
    mont<true>(s2, a, d);
#pragma unroll
    for (int j = 0; j < 9; j++) d[j] = s2[j] << 1;
    mont<true>(s4, s2, d);
    mont<false>(s2, s4, a);
#else
    const RLimbs R = r_limbs();
#pragma unroll
    for (int j = 0; j < 9; j++) d[j] = a[j] << 1;
    mont<true>(s2, a, d, R);
#pragma unroll
    for (int j = 0; j < 9; j++) d[j] = s2[j] << 1;
    mont<true>(s4, s2, d, R);
    mont<false>(s2, s4, a, R);
#endif
    from29(s2, x);
}

}  // namespace bn29
