// BN254 scalar field Fr for gfx950: eight 32-bit limbs, Montgomery form with R = 2^256 (the form wasmcurves' F1m keeps
// in memory, so tree.nodes is byte-compatible with merklehash_bn128_p.js).  gfx950 has no 64-bit multiplier: products are
// accumulated COLUMN BY COLUMN (product scanning) into a 64-bit accumulator plus a carry word,
//     v_mad_u64_u32  lo, vcc, x, y, lo        ; lo += x*y, carry-out in vcc
//     v_addc_co_u32  hi, vcc, 0, hi, vcc      ; hi += carry
// two instructions per 32x32 product.  The compiler never uses the multiply-add's carry-out (it re-derives carries with
// 64-bit adds and compares: 4 instructions), hence the inline assembly; measured 10.2 against 12.8-15.3 issue cycles per
// product (tools/bn_microbench.hip), results identical to the operand-scanning forms kept below as fr_mul_os / mac17_os.
#pragma once
#include <stdint.h>

namespace bn {

typedef uint64_t u64;
typedef uint32_t u32;

// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
#define BN_R_LIMBS { 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u }
#define BN_R2_LIMBS { 0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u }
constexpr u32 N0INV = 0xefffffffu;                   // -r^-1 mod 2^32

__device__ __forceinline__ u32 r_limb(int i) { constexpr u32 R[8] = BN_R_LIMBS; return R[i]; }
__device__ __forceinline__ u32 r2_limb(int i) { constexpr u32 R2[8] = BN_R2_LIMBS; return R2[i]; }

// a >= r ?   (n limbs of a against r extended with zeros)
template <int N>
__device__ __forceinline__ bool ge_r(const u32 *a) {
    bool ge = true;                                  // equal so far => ge
#pragma unroll
    for (int i = 0; i < N; i++) {                    // from the least significant limb up: the last difference decides
        const u32 ri = i < 8 ? r_limb(i) : 0u;
        if (a[i] != ri) ge = a[i] > ri;
    }
    return ge;
}
template <int N>
__device__ __forceinline__ void sub_r(u32 *a) {
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const u32 ri = i < 8 ? r_limb(i) : 0u;
        const u64 d = (u64)a[i] - ri - borrow;
        a[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
}

// t (9 limbs, t < 2^288) -> t - r if t >= r, else t, IN PLACE; only limbs 0..7 of the result matter (callers guarantee < 2^256 then).
// One borrow chain through vcc tells whether t >= r; the borrow becomes a mask and r AND the mask is subtracted in place.  Nothing but two
// scratch registers is an output of the first statement: the earlier form wrote the nine differences to early-clobber outputs and selected
// them against the inputs, and hipcc's coalescer, inside a loop that copies the result back over the input, gave a difference the register
// of the limb it was selected against (`v_cndmask_b32 v8, v8, v8`: silently wrong for every value below r -- round 5).
__device__ __forceinline__ void cond_sub_r(u32 t[9]) {
    u32 scr, mask;
    // the limbs of r travel in VGPRs: a literal and the vcc borrow-in would both need the single constant-bus slot
    asm("v_sub_co_u32 %0, vcc, %2, %11\n\tv_subb_co_u32 %0, vcc, %3, %12, vcc\n\tv_subb_co_u32 %0, vcc, %4, %13, vcc\n\tv_subb_co_u32 %0, vcc, %5, %14, vcc\n\t"
        "v_subb_co_u32 %0, vcc, %6, %15, vcc\n\tv_subb_co_u32 %0, vcc, %7, %16, vcc\n\tv_subb_co_u32 %0, vcc, %8, %17, vcc\n\tv_subb_co_u32 %0, vcc, %9, %18, vcc\n\t"
        "v_subbrev_co_u32 %0, vcc, 0, %10, vcc\n\tv_cndmask_b32_e64 %1, -1, 0, vcc"
        : "=&v"(scr), "=&v"(mask)
        : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]), "v"(t[8]),
          "v"(r_limb(0)), "v"(r_limb(1)), "v"(r_limb(2)), "v"(r_limb(3)), "v"(r_limb(4)), "v"(r_limb(5)), "v"(r_limb(6)), "v"(r_limb(7)) : "vcc");
    u32 rm[8];
#pragma unroll
    for (int l = 0; l < 8; l++) rm[l] = r_limb(l) & mask;
    asm("v_sub_co_u32 %0, vcc, %0, %9\n\tv_subb_co_u32 %1, vcc, %1, %10, vcc\n\tv_subb_co_u32 %2, vcc, %2, %11, vcc\n\tv_subb_co_u32 %3, vcc, %3, %12, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %4, %13, vcc\n\tv_subb_co_u32 %5, vcc, %5, %14, vcc\n\tv_subb_co_u32 %6, vcc, %6, %15, vcc\n\tv_subb_co_u32 %7, vcc, %7, %16, vcc\n\t"
        "v_subbrev_co_u32 %8, vcc, 0, %8, vcc"
        : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8])
        : "v"(rm[0]), "v"(rm[1]), "v"(rm[2]), "v"(rm[3]), "v"(rm[4]), "v"(rm[5]), "v"(rm[6]), "v"(rm[7]) : "vcc");
}

// a = a + b mod r   (a, b < r)
__device__ __forceinline__ void fr_add(u32 a[8], const u32 b[8]) {
    u32 t[9];
    asm("v_add_co_u32 %0, vcc, %9, %17\n\t"
        "v_addc_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_addc_co_u32 %8, vcc, 0, 0, vcc"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
    cond_sub_r(t);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = t[i];
}

// (lo, hi) += x * y      /     (lo, hi) += x
__device__ __forceinline__ void acc_mad(u64 &lo, u32 &hi, u32 x, u32 y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ void acc_add(u64 &lo, u32 &hi, u32 x) {
    asm("v_mad_u64_u32 %0, vcc, %2, 1, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x) : "vcc");
}
__device__ __forceinline__ void acc_shift(u64 &lo, u32 &hi) { lo = (lo >> 32) | ((u64)hi << 32); hi = 0; }

// Up to eight products accumulated by ONE asm statement.  gfx950's hazard recogniser gives every instruction that reads a
// register written by an asm statement a wait state (it cannot rule out a partial-register write inside), so a chain of
// one-product statements carries an s_nop per product -- with one wave per SIMD a lost issue slot each; a column's products
// in one statement carry one.
#define BN_MA(X, Y) "v_mad_u64_u32 %0, vcc, %" #X ", %" #Y ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
template <int N>
__device__ __forceinline__ void acc_madn(u64 &lo, u32 &hi, const u32 *x, const u32 *y) {
    static_assert(N >= 0 && N <= 9, "up to nine products");
    if constexpr (N == 1) asm(BN_MA(2, 3) : "+v"(lo), "+v"(hi) : "v"(x[0]), "v"(y[0]) : "vcc");
    if constexpr (N == 2) asm(BN_MA(2, 3) BN_MA(4, 5) : "+v"(lo), "+v"(hi) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]) : "vcc");
    if constexpr (N == 3) asm(BN_MA(2, 3) BN_MA(4, 5) BN_MA(6, 7) : "+v"(lo), "+v"(hi) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]) : "vcc");
    if constexpr (N == 4) asm(BN_MA(2, 3) BN_MA(4, 5) BN_MA(6, 7) BN_MA(8, 9) : "+v"(lo), "+v"(hi)
        : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]) : "vcc");
    if constexpr (N == 5) asm(BN_MA(2, 3) BN_MA(4, 5) BN_MA(6, 7) BN_MA(8, 9) BN_MA(10, 11) : "+v"(lo), "+v"(hi)
        : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]) : "vcc");
    if constexpr (N == 6) asm(BN_MA(2, 3) BN_MA(4, 5) BN_MA(6, 7) BN_MA(8, 9) BN_MA(10, 11) BN_MA(12, 13) : "+v"(lo), "+v"(hi)
        : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]) : "vcc");
    if constexpr (N == 7) asm(BN_MA(2, 3) BN_MA(4, 5) BN_MA(6, 7) BN_MA(8, 9) BN_MA(10, 11) BN_MA(12, 13) BN_MA(14, 15) : "+v"(lo), "+v"(hi)
        : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]) : "vcc");
    if constexpr (N == 8) asm(BN_MA(2, 3) BN_MA(4, 5) BN_MA(6, 7) BN_MA(8, 9) BN_MA(10, 11) BN_MA(12, 13) BN_MA(14, 15) BN_MA(16, 17) : "+v"(lo), "+v"(hi)
        : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]) : "vcc");
    if constexpr (N == 9) asm(BN_MA(2, 3) BN_MA(4, 5) BN_MA(6, 7) BN_MA(8, 9) BN_MA(10, 11) BN_MA(12, 13) BN_MA(14, 15) BN_MA(16, 17) BN_MA(18, 19) : "+v"(lo), "+v"(hi)
        : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]) : "vcc");
}
// column K of x * y (the products x_j * y_(K-j) with both indices in 0..7, j < JMAX), preceded by the word `w` when WITH_W:
// one statement per column
template <int K, int JMAX, bool WITH_W>
__device__ __forceinline__ void acc_column(u64 &lo, u32 &hi, const u32 *x, const u32 *y, u32 w = 0) {
    constexpr int J0 = K > 7 ? K - 7 : 0, J1 = (K < JMAX - 1 ? K : JMAX - 1) < 7 ? (K < JMAX - 1 ? K : JMAX - 1) : 7;
    constexpr int NP = J1 >= J0 ? J1 - J0 + 1 : 0, N = NP + (WITH_W ? 1 : 0);
    if constexpr (N > 0) {
        u32 xs[N], ys[N];
        if constexpr (WITH_W) { xs[0] = w; ys[0] = 1u; }
#pragma unroll
        for (int i = 0; i < NP; i++) { xs[(WITH_W ? 1 : 0) + i] = x[J0 + i]; ys[(WITH_W ? 1 : 0) + i] = y[K - J0 - i]; }
        acc_madn<N>(lo, hi, xs, ys);
    }
}

// out = a * b / 2^256 mod r   (finely integrated product scanning; a < 2^256, b < r => out < r after one subtraction)
template <int I>
__device__ __forceinline__ void fr_mul_step(u64 &lo, u32 &hi, const u32 a[8], const u32 b[8], u32 m[8], const u32 rl[8], u32 t[9]) {
    acc_column<I, 8, false>(lo, hi, a, b);           // a_j b_(I-j)
    acc_column<I, (I < 8 ? I : 8), false>(lo, hi, m, rl);   // m_j r_(I-j), j < I: only the m already known
    if constexpr (I < 8) { m[I] = (u32)lo * N0INV; acc_mad(lo, hi, m[I], rl[0]); }      // low word becomes 0
    else t[I - 8] = (u32)lo;
    acc_shift(lo, hi);
}
template <int I>
__device__ __forceinline__ void fr_mul_col(u64 &lo, u32 &hi, const u32 a[8], const u32 b[8], u32 m[8], const u32 rl[8], u32 t[9]) {
    fr_mul_step<I>(lo, hi, a, b, m, rl, t);
    if constexpr (I < 15) fr_mul_col<I + 1>(lo, hi, a, b, m, rl, t);
}
__device__ __forceinline__ void fr_mul(u32 out[8], const u32 a[8], const u32 b[8]) {
    u32 m[8], t[9], rl[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rl[i] = r_limb(i);
    u64 lo = 0; u32 hi = 0;
    fr_mul_col<0>(lo, hi, a, b, m, rl, t);
    t[8] = (u32)lo;
    cond_sub_r(t);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = t[i];
}

// the same product WITHOUT the final subtraction: out = (ab + m r) / 2^256 < ab / 2^256 + r, which fits eight limbs whenever
// a, b < 0.9 * 2^256 (and is < 2^255 for a, b < 2^255): the lazy representatives between the matrix-core layers (bn_mfma.cuh)
__device__ __forceinline__ void fr_mul_nr(u32 out[8], const u32 a[8], const u32 b[8]) {
    u32 m[8], t[9], rl[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rl[i] = r_limb(i);
    u64 lo = 0; u32 hi = 0;
    fr_mul_col<0>(lo, hi, a, b, m, rl, t);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = t[i];
}

// operand-scanning (CIOS) form of the same product, compiler-generated carries (reference implementation for tests)
__device__ __forceinline__ void fr_mul_os(u32 out[8], const u32 a[8], const u32 b[8]) {
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { const u64 x = (u64)a[j] * b[i] + t[j] + c; t[j] = (u32)x; c = x >> 32; }
        u64 x = (u64)t[8] + c; t[8] = (u32)x; t[9] = (u32)(x >> 32);
        const u32 m = t[0] * N0INV;
        c = ((u64)m * r_limb(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) { const u64 y = (u64)m * r_limb(j) + t[j] + c; t[j - 1] = (u32)y; c = y >> 32; }
        x = (u64)t[8] + c; t[7] = (u32)x; t[8] = t[9] + (u32)(x >> 32);
    }
    if (ge_r<9>(t)) sub_r<9>(t);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = t[i];
}

// acc (17 limbs) += a * b, column by column: column k takes the old limb k and the products a_j * b_(k-j)
template <int K>
__device__ __forceinline__ void mac17_step(u64 &lo, u32 &hi, u32 acc[17], const u32 a[8], const u32 b[8]) {
    acc_column<K, 8, true>(lo, hi, a, b, acc[K]);
    acc[K] = (u32)lo;
    acc_shift(lo, hi);
}
template <int K>
__device__ __forceinline__ void mac17_col(u64 &lo, u32 &hi, u32 acc[17], const u32 a[8], const u32 b[8]) {
    mac17_step<K>(lo, hi, acc, a, b);
    if constexpr (K < 15) mac17_col<K + 1>(lo, hi, acc, a, b);
}
__device__ __forceinline__ void mac17(u32 acc[17], const u32 a[8], const u32 b[8]) {
    u64 lo = 0; u32 hi = 0;
    mac17_col<0>(lo, hi, acc, a, b);
    acc[16] += (u32)lo;
}
// acc += a * b  and  out = c * d / 2^256 mod r  together, column by column in turn: the two chains are independent, so
// the wait state each asm statement owes its successor (see acc_madn) is filled by the other chain's statement
template <int K>
__device__ __forceinline__ void mac17_fr_mul_col(u64 &lo1, u32 &hi1, u32 acc[17], const u32 a[8], const u32 b[8],
                                                 u64 &lo2, u32 &hi2, const u32 c[8], const u32 d[8], u32 m[8], const u32 rl[8], u32 t[9]) {
    mac17_step<K>(lo1, hi1, acc, a, b);
    fr_mul_step<K>(lo2, hi2, c, d, m, rl, t);
    if constexpr (K < 15) mac17_fr_mul_col<K + 1>(lo1, hi1, acc, a, b, lo2, hi2, c, d, m, rl, t);
}
__device__ __forceinline__ void mac17_and_fr_mul(u32 acc[17], const u32 a[8], const u32 b[8], u32 out[8], const u32 c[8], const u32 d[8]) {
    u32 m[8], t[9], rl[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rl[i] = r_limb(i);
    u64 lo1 = 0, lo2 = 0; u32 hi1 = 0, hi2 = 0;
    mac17_fr_mul_col<0>(lo1, hi1, acc, a, b, lo2, hi2, c, d, m, rl, t);
    acc[16] += (u32)lo1;
    t[8] = (u32)lo2;
    cond_sub_r(t);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = t[i];
}
// operand-scanning form (reference implementation for tests)
__device__ __forceinline__ void mac17_os(u32 acc[17], const u32 a[8], const u32 b[8]) {
    u32 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { const u64 x = (u64)a[j] * b[i] + p[i + j] + c; p[i + j] = (u32)x; c = x >> 32; }
        p[i + 8] = (u32)c;
    }
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { c += (u64)acc[i] + p[i]; acc[i] = (u32)c; c >>= 32; }
    acc[16] += (u32)c;
}

// Montgomery reduction of a 17-limb accumulator T < 32 * r^2:  out = T / 2^256 mod r   (product scanning)
template <int K>
__device__ __forceinline__ void redc17_col(u64 &lo, u32 &hi, const u32 acc[17], u32 m[8], const u32 rl[8], u32 t[9]) {
    acc_column<K, (K < 8 ? K : 8), true>(lo, hi, m, rl, acc[K]);      // acc_K + m_j r_(K-j), j < K
    if constexpr (K < 8) { m[K] = (u32)lo * N0INV; acc_mad(lo, hi, m[K], rl[0]); }
    else t[K - 8] = (u32)lo;
    acc_shift(lo, hi);
    if constexpr (K < 15) redc17_col<K + 1>(lo, hi, acc, m, rl, t);
}
__device__ __forceinline__ void redc17(u32 out[8], u32 acc[17]) {
    u32 m[8], t[9], rl[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rl[i] = r_limb(i);
    u64 lo = 0; u32 hi = 0;
    redc17_col<0>(lo, hi, acc, m, rl, t);
    t[8] = (u32)lo + acc[16];                        // T/2^256 + r < 2^261: the ninth limb cannot overflow
    // T < 17 r^2 in every caller (at most 17 products): T/2^256 + r < 17*0.19 r + r < 5 r, so five subtractions at most
#pragma unroll
    for (int k = 0; k < 5; k++) cond_sub_r(t);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = t[i];
}
// operand-scanning form (reference implementation for tests)
__device__ __forceinline__ void redc17_os(u32 out[8], u32 acc[17]) {
    u32 top = 0;                                     // carry out of limb 16 (T + m*r*2^(32i) < 2^544 + ..., one bit)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 m = acc[i] * N0INV;
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { const u64 x = (u64)m * r_limb(j) + acc[i + j] + c; acc[i + j] = (u32)x; c = x >> 32; }
#pragma unroll
        for (int k = i + 8; k < 17; k++) { c += acc[k]; acc[k] = (u32)c; c >>= 32; }
        top += (u32)c;
    }
    (void)top;                                       // T/2^256 + r < 2^261 fits limbs 8..16
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = acc[8 + i];
    for (int k = 0; k < 34 && ge_r<9>(t); k++) sub_r<9>(t);      // < 32*r*(r/2^256) + r < 8r in practice (t <= 17: < 5r)
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = t[i];
}

}  // namespace bn
