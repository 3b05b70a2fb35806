// Goldilocks field (p = 2^64 - 2^32 + 1) and its cubic extension for gfx950 device code.
//
// Arithmetic spec: src/helpers/f3g.js:47-172 of the reference (canonical results).
// gfx950 has no 64-bit integer multiplier: a 64x64->128 product is four
// v_mad_u64_u32, and the reduction uses 2^64 = 2^32 - 1, 2^96 = -1 (mod p), i.e.
// only adds/subs -- no Montgomery form is needed for this prime.
//
// Two value classes are used:
//   canonical  [0, p)      -- everything stored to memory
//   lazy       [0, 2^64)   -- any representative; cheaper to produce, accepted by mul()/add_lazy()
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 P = 0xFFFFFFFF00000001ull;
static constexpr u64 EPS = 0xFFFFFFFFull;           // 2^64 mod p

__device__ __forceinline__ u64 canon(u64 a) { return a >= P ? a - P : a; }

// The forms below are written with the carry/borrow builtins: hipcc then keeps the carry in a lane mask
// instead of re-deriving it with 64-bit compares (measured on MI355X: add 42 -> 25, mul 88 -> 74 issue
// cycles per wave; tools/microbench.hip).

// canonical + canonical -> canonical
__device__ __forceinline__ u64 add(u64 a, u64 b) {
    u64 s, t;
    const bool c1 = __builtin_uaddl_overflow(a, b, &s);      // a,b < p: true sum < 2p < 2^65
    const bool c2 = __builtin_uaddl_overflow(s, EPS, &t);    // t = s - p (mod 2^64); c2 <=> s >= p
    return (c1 | c2) ? t : s;
}
// canonical - canonical -> canonical
__device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u64 d;
    const bool br = __builtin_usubl_overflow(a, b, &d);
    return br ? d - EPS : d;                                  // borrow: add p (== subtract EPS mod 2^64)
}
__device__ __forceinline__ u64 neg(u64 a) { return a ? P - a : 0; }

// lazy + canonical -> lazy   (one wrap at most: s - 2^64 + EPS < 2^64 because b < p)
__device__ __forceinline__ u64 add_lazy_canon(u64 a, u64 b) {
    u64 s;
    const bool c = __builtin_uaddl_overflow(a, b, &s);
    return c ? s + EPS : s;
}
// lazy + lazy -> lazy
__device__ __forceinline__ u64 add_lazy(u64 a, u64 b) {
    u64 s, t;
    if (__builtin_uaddl_overflow(a, b, &s)) {                // wrapped: add 2^64 mod p
        if (__builtin_uaddl_overflow(s, EPS, &t)) t += EPS;  // (second wrap only for non-canonical inputs)
        s = t;
    }
    return s;
}

// 128-bit (hi,lo) -> lazy:  lo - hh + hl*(2^32-1)  with 2^64 = 2^32-1, 2^96 = -1 (mod p)
__device__ __forceinline__ u64 reduce128_lazy(u64 lo, u64 hi) {
    const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0, t2;
    const bool br = __builtin_usubl_overflow(lo, (u64)hh, &t0);
    const u64 t1 = (u64)hl * EPS;                            // one v_mad_u64_u32
    const bool c = __builtin_uaddl_overflow(t0, t1, &t2);
    // borrow => the true value is t - 2^64 = t - EPS; carry => t + 2^64 = t + EPS; neither correction can wrap
    return t2 + ((c ? EPS : 0) - (br ? EPS : 0));
}

// any u64 x any u64 -> lazy.  Four v_mad_u64_u32 build the 128-bit product (no 64-bit multiplier on gfx950).
__device__ __forceinline__ u64 mul_lazy(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);                  // < 2^64: (2^32-1)^2 + 2^32 - 1
    const u64 v = (u64)a1 * b0 + (u32)u;
    const u64 w = (u64)a1 * b1 + (u >> 32) + (v >> 32);      // <= 2^64 - 1
    return reduce128_lazy((v << 32) | (u32)t, w);
}

// The same product with the carries taken where the hardware leaves them -- the multiply-add's carry-out and the
// subtraction's borrow, as lane masks in SGPRs -- instead of re-derived by 64-bit compares: 13 vector instructions
// against the 22 hipcc emits for mul_lazy (tools/sbox_bench.hip: an x^7 chain 331 -> 203 issue cycles per wave, the Poseidon
// permutation 2.28 -> 2.72 G/s).
//   t = a0 b0;  u = a0 b1 + (t >> 32);  v = a1 b0 + u  (the WHOLE u as addend: the 65th bit is the carry-out);
//   w = a1 b1 + (v >> 32) + carry 2^32;     z = w0 (2^32-1) + (v0:t0)  (carry c);  z += c (2^32-1)  (cannot wrap);  r = z - w1.
// The last subtraction borrows with probability ~2^-32 (z < w1 < 2^32): the lane's bit is OR-ed into `bad`, a wave-level
// value, and the CALLER recomputes the flagged work with mul_lazy (a branch on `bad` is uniform for the wave).
// Each asm statement is one instruction (plus the two wait states gfx950 wants between a vector instruction that writes
// an SGPR and one that reads it: hipcc pads its own carry chains the same way, nobody pads an asm string), so register
// allocation and scheduling stay the compiler's.  Measured, not assumed: where independent work surrounds the products
// (the NTT tiles) the plain form schedules better and stays; a branch per product, or folding the borrow back in place
// (three more instructions), both lose the gain.
#ifndef GL_MUL_B
#define GL_MUL_B 12         // which form of the carry-out product: 13 (round 2), 12 (its tail on the carry flags), 11 (no moves at all: fewer instructions, slower)
#endif
#ifndef GL_SGPR_WAIT
#define GL_SGPR_WAIT "s_nop 1\n\t"       // (tools/sbox_bench.hip builds an experiment without the wait states: what they cost)
#endif
__device__ __forceinline__ u64 mul_lazy_b(u64 a, u64 b, u64 &bad) {
#if GL_MUL_B == 12
    // Twelve vector instructions: round 2's product with its tail -- select, multiply-add by 2^32-1, two subtractions -- replaced by
    // three additions / subtractions that take the multiply-add's carry c as their carry-in:
    //   r = z + c (2^32-1) - w1   as   r0 = z0 - w1 - c (borrow b),  r1 = (z1 + c) - b
    // (z1 + c cannot wrap: a wrapped z is below 2^64 - 2^33).  Same number of SGPR-carried carries in a row as before (three), one
    // multiply-add fewer.  The last borrow is the rare case (probability ~2^-32) the caller recomputes.
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v, cy, z, c, bo, br, cx; u32 c01, r0, r1, r1a;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v), "=s"(cy) : "v"(a1), "v"(b0), "v"(u));
    asm(GL_SGPR_WAIT "v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c01) : "s"(cy));
    const u64 w = (u64)a1 * b1 + (((u64)c01 << 32) | (v >> 32));   // <= 2^64 - 1: the full product is < 2^128
    const u64 lo = (v << 32) | (u32)t;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"((u32)w), "v"(lo));
    asm(GL_SGPR_WAIT "v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r0), "=s"(bo) : "v"((u32)z), "v"((u32)(w >> 32)), "s"(c));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r1a), "=s"(cx) : "v"((u32)(z >> 32)), "s"(c));
    asm(GL_SGPR_WAIT "v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r1), "=s"(br) : "v"(r1a), "s"(bo));
    bad |= br;
    return ((u64)r1 << 32) | r0;
#elif GL_MUL_B == 11
    // ELEVEN vector instructions: five multiply-adds and six 32-bit additions / subtractions with carry (round 3; measured 3 %
    // SLOWER than the 13-instruction form in the permutation: five SGPR-carried carries in a row instead of three, and the moves it
    // removes are the cheapest instructions there are -- tools/issue_cost.hip: 2.9 cycles saturated against 4.8-5.0 for a carry
    // operation or a multiply-add.  Kept for the record; the 13-instruction form of round 2 is kept below under GL_MUL_B_13).  In a kernel that
    // saturates the vector issue every instruction costs the same four cycles, moves included, and the 13-instruction form carried
    // THREE moves per product: gfx950 wants 64-bit operands in even-aligned register pairs, and (t >> 32), (v >> 32) and the low
    // word of v are each born in the wrong half of a pair.  Here every value that has to change halves does so inside an
    // addition that is needed anyway:
    //   t = a0 b0;  u = a0 b1;  v = a1 b0 + u            (carry cy; t's high word is NOT added in yet)
    //   l1 = t1 + v0   (carry ca)   -> (t0, l1) is the low 64 bits of the product, in t's own pair
    //   h0 = v1 + ca   (carry cb)   -> (h0, cy|cb) is what a1 b1 still has to take in: cy and cb exclude each other
    //   w  = a1 b1 + h0;  w1 += cy|cb   (as the carry-in of an add: no select, no move)
    //   z  = w0 (2^32-1) + (t0, l1) (carry c);   r = z + c (2^32-1) - w1  as  r0 = z0 - w1 - c (borrow b),  r1 = z1 + c - b
    // (z + c 2^64 = z + c (2^32-1) mod p; z1 + c cannot wrap: a wrapped z is below 2^64 - 2^33).  The last borrow is the rare
    // case (probability ~2^-32) the caller recomputes, as before.
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1;
    u64 v, cy, ca, cb, z, c, bo, br; u32 l1, h0, hc, r0, r1, r1a;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v), "=s"(cy) : "v"(a1), "v"(b0), "v"(u));
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(l1), "=s"(ca) : "v"((u32)(t >> 32)), "v"((u32)v));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(h0), "=s"(cb) : "v"((u32)(v >> 32)), "s"(ca));
    const u64 cc = cy | cb;
    const u64 w = (u64)a1 * b1 + h0;                 // the bit cc belongs at 2^32 of this addend: added to w's high word below
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(hc), "=s"(cb) : "v"((u32)(w >> 32)), "s"(cc));    // cannot wrap: the product is < 2^128
    const u64 lo = ((u64)l1 << 32) | (u32)t;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"((u32)w), "v"(lo));
    asm(GL_SGPR_WAIT "v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r0), "=s"(bo) : "v"((u32)z), "v"(hc), "s"(c));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r1a), "=s"(cb) : "v"((u32)(z >> 32)), "s"(c));
    asm(GL_SGPR_WAIT "v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r1), "=s"(br) : "v"(r1a), "s"(bo));
    bad |= br;
    return ((u64)r1 << 32) | r0;
#else           // GL_MUL_B == 13: round 2's form
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v, cy, z, c, br; u32 c01, c201, r0, r1;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v), "=s"(cy) : "v"(a1), "v"(b0), "v"(u));
    asm(GL_SGPR_WAIT "v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c01) : "s"(cy));
    const u64 w = (u64)a1 * b1 + (((u64)c01 << 32) | (v >> 32));   // <= 2^64 - 1: the full product is < 2^128
    const u64 lo = (v << 32) | (u32)t;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"((u32)w), "v"(lo));
    asm(GL_SGPR_WAIT "v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c201) : "s"(c));
    const u64 z2 = (u64)c201 * 0xFFFFFFFFu + z;
    asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\t" GL_SGPR_WAIT "v_subbrev_co_u32_e64 %1, %2, 0, %4, %2"
        : "=&v"(r0), "=&v"(r1), "=&s"(br) : "v"((u32)z2), "v"((u32)(z2 >> 32)), "v"((u32)(w >> 32)));
    bad |= br;
    return ((u64)r1 << 32) | r0;
#endif
}
// The same carry-out product with the rare borrow folded back in place: exact, no flag for the caller (15 instructions: the
// 12-instruction form above, then a borrow b means the true value is r - 2^64 = r - (2^32 - 1) mod p = (r0 + 1, r1 - 1 + carry)).
__device__ __forceinline__ u64 mul_lazy_x(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v, cy, z, c, bo, br, cx, k; u32 c01, r0, r1, r1a, m, q0, q1;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v), "=s"(cy) : "v"(a1), "v"(b0), "v"(u));
    asm(GL_SGPR_WAIT "v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c01) : "s"(cy));
    const u64 w = (u64)a1 * b1 + (((u64)c01 << 32) | (v >> 32));
    const u64 lo = (v << 32) | (u32)t;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"((u32)w), "v"(lo));
    asm(GL_SGPR_WAIT "v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r0), "=s"(bo) : "v"((u32)z), "v"((u32)(w >> 32)), "s"(c));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r1a), "=s"(cx) : "v"((u32)(z >> 32)), "s"(c));
    asm(GL_SGPR_WAIT "v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r1), "=s"(br) : "v"(r1a), "s"(bo));
    asm(GL_SGPR_WAIT "v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m) : "s"(br));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(q0), "=s"(k) : "v"(r0), "s"(br));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(q1), "=s"(cx) : "v"(r1), "v"(m), "s"(k));
    return ((u64)q1 << 32) | q0;
}
// any x any -> canonical
__device__ __forceinline__ u64 mul(u64 a, u64 b) { return canon(mul_lazy(a, b)); }
__device__ __forceinline__ u64 sqr(u64 a) { return mul(a, a); }

__device__ inline u64 pow(u64 base, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, base); base = mul(base, base); e >>= 1; }
    return r;
}
__device__ inline u64 inv(u64 a) { return pow(a, P - 2); }

// ---- cubic extension, x^3 = x + 1 (f3g.js:94-102) ----
struct E3 { u64 v[3]; };

__device__ __forceinline__ E3 e3_add(const E3 &a, const E3 &b) { return { { add(a.v[0], b.v[0]), add(a.v[1], b.v[1]), add(a.v[2], b.v[2]) } }; }
__device__ __forceinline__ E3 e3_sub(const E3 &a, const E3 &b) { return { { sub(a.v[0], b.v[0]), sub(a.v[1], b.v[1]), sub(a.v[2], b.v[2]) } }; }
__device__ __forceinline__ E3 e3_scale(const E3 &a, u64 s) { return { { mul(a.v[0], s), mul(a.v[1], s), mul(a.v[2], s) } }; }
__device__ __forceinline__ E3 e3_mul(const E3 &a, const E3 &b) {
    u64 A = mul(add(a.v[0], a.v[1]), add(b.v[0], b.v[1]));
    u64 B = mul(add(a.v[0], a.v[2]), add(b.v[0], b.v[2]));
    u64 C = mul(add(a.v[1], a.v[2]), add(b.v[1], b.v[2]));
    u64 D = mul(a.v[0], b.v[0]);
    u64 E = mul(a.v[1], b.v[1]);
    u64 F = mul(a.v[2], b.v[2]);
    u64 G = sub(D, E);
    E3 r;
    r.v[0] = sub(add(C, G), F);
    r.v[1] = sub(sub(sub(add(A, C), E), E), D);
    r.v[2] = sub(B, G);
    return r;
}
// f3g.js:136-172
__device__ inline E3 e3_inv(const E3 &x) {
    u64 a = x.v[0], b = x.v[1], c = x.v[2];
    u64 aa = mul(a, a), ac = mul(a, c), ba = mul(b, a), bb = mul(b, b), bc = mul(b, c), cc = mul(c, c);
    u64 aaa = mul(aa, a), aac = mul(aa, c), abc = mul(ba, c), abb = mul(ba, b);
    u64 acc = mul(ac, c), bbb = mul(bb, b), bcc = mul(bc, c), ccc = mul(cc, c);
    u64 t = neg(aaa);
    t = sub(t, aac); t = sub(t, aac);
    t = add(t, abc); t = add(t, abc); t = add(t, abc);
    t = add(t, abb); t = sub(t, acc); t = sub(t, bbb); t = add(t, bcc); t = sub(t, ccc);
    u64 ti = inv(t);
    u64 i1 = neg(aa);
    i1 = sub(i1, ac); i1 = sub(i1, ac); i1 = add(i1, bc); i1 = add(i1, bb); i1 = sub(i1, cc);
    u64 i2 = sub(ba, cc);
    u64 i3 = add(add(neg(bb), ac), cc);
    return { { mul(i1, ti), mul(i2, ti), mul(i3, ti) } };
}

__device__ __forceinline__ u32 bitrev32(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

}  // namespace gl
