// Small DFTs over Goldilocks computed in the ring Z/(2^96+1)  (gfx950).
//
// p = 2^64 - 2^32 + 1 divides 2^96 + 1, and 2 has order 192 mod p, so the roots of unity the transforms use
// (F.w[c] of the reference, src/helpers/f3g.js:24-34) are powers of two up to order 64:
//     w_2 = 2^96 = -1, w_4 = 2^48, w_8 = 2^24, w_16 = 2^12, w_32 = 2^6, w_64 = 2^3.
// A 2^c-point DFT (c <= 5) held in the registers of one lane therefore needs no multiplications: values are kept
// as signed 128-bit integers (four 32-bit words, two's complement) that are only *congruent* mod 2^96+1 to the field
// element; a butterfly is a 4-word add and a 4-word sub on the carry flags, a twiddle is a constant funnel shift
// followed by one subtraction (2^96 = -1 folds the bits that leave the 96-bit window back in), and a single
// reduction mod p ends the block.  The shift amounts are compile-time constants: they depend on the register index
// and the stage, never on the lane.
//
// Magnitudes: inputs are < 2^64; a shift maps |v| to at most 2^96 + |v|/2 and a butterfly doubles the bound, so
// after five stages |v| < 2^100; to_gl_lazy() accepts |v| < 2^101.
#pragma once
#include "gl_field.cuh"

namespace gl {
namespace fermat {

// FERMAT_TIMING (never defined in the library build): cost-split experiments of tools/lde_cost_split.sh -- bit 0 butterflies without
// carries, bit 1 every twiddle shift a register rotation, bit 2 no final reduction.  The results are WRONG; only the run time means anything.
#ifndef FERMAT_TIMING
#define FERMAT_TIMING 0
#endif

struct f128 { u32 w0, w1, w2, w3; };                 // value = w0 + w1 2^32 + w2 2^64 + (int)w3 2^96

__device__ __forceinline__ f128 from_gl(u64 a) { return { (u32)a, (u32)(a >> 32), 0u, 0u }; }

__device__ __forceinline__ f128 add(const f128 &a, const f128 &b) {
    f128 r;
    if constexpr (FERMAT_TIMING & 1) { r.w0 = a.w0 + b.w0; r.w1 = a.w1 + b.w1; r.w2 = a.w2 + b.w2; r.w3 = a.w3 + b.w3; return r; }
    asm("v_add_co_u32 %0, vcc, %4, %8\n\tv_addc_co_u32 %1, vcc, %5, %9, vcc\n\tv_addc_co_u32 %2, vcc, %6, %10, vcc\n\tv_addc_co_u32 %3, vcc, %7, %11, vcc"
        : "=&v"(r.w0), "=&v"(r.w1), "=&v"(r.w2), "=&v"(r.w3)
        : "v"(a.w0), "v"(a.w1), "v"(a.w2), "v"(a.w3), "v"(b.w0), "v"(b.w1), "v"(b.w2), "v"(b.w3) : "vcc");
    return r;
}
__device__ __forceinline__ f128 sub(const f128 &a, const f128 &b) {
    f128 r;
    if constexpr (FERMAT_TIMING & 1) { r.w0 = a.w0 - b.w0; r.w1 = a.w1 - b.w1; r.w2 = a.w2 - b.w2; r.w3 = a.w3 - b.w3; return r; }
    asm("v_sub_co_u32 %0, vcc, %4, %8\n\tv_subb_co_u32 %1, vcc, %5, %9, vcc\n\tv_subb_co_u32 %2, vcc, %6, %10, vcc\n\tv_subb_co_u32 %3, vcc, %7, %11, vcc"
        : "=&v"(r.w0), "=&v"(r.w1), "=&v"(r.w2), "=&v"(r.w3)
        : "v"(a.w0), "v"(a.w1), "v"(a.w2), "v"(a.w3), "v"(b.w0), "v"(b.w1), "v"(b.w2), "v"(b.w3) : "vcc");
    return r;
}

// v * 2^S mod (2^96+1), 0 < S < 96:   (v << S mod 2^96)  -  (v >> (96-S), arithmetic)
template <int S>
__device__ __forceinline__ f128 shl(const f128 &v) {
    static_assert(S > 0 && S < 96, "shift out of range");
    if constexpr (FERMAT_TIMING & 2) return { v.w1, v.w2, v.w3, v.w0 };
    constexpr int q = S / 32, r = S % 32;
    // t = v << r as five words (t4 carries the sign)
    u32 t0, t1, t2, t3, t4;
    if (r == 0) { t0 = v.w0; t1 = v.w1; t2 = v.w2; t3 = v.w3; t4 = (u32)((int)v.w3 >> 31); }
    else {
        t0 = v.w0 << r;
        t1 = __builtin_amdgcn_alignbit(v.w1, v.w0, 32 - r);
        t2 = __builtin_amdgcn_alignbit(v.w2, v.w1, 32 - r);
        t3 = __builtin_amdgcn_alignbit(v.w3, v.w2, 32 - r);
        t4 = (u32)((int)v.w3 >> (32 - r));
    }
    const u32 sx = (u32)((int)t4 >> 31);
    f128 res;
    if (q == 0) {          // lo = (t2,t1,t0), hi = (sx,sx,t4,t3)
        asm("v_sub_co_u32 %0, vcc, %4, %7\n\tv_subb_co_u32 %1, vcc, %5, %8, vcc\n\tv_subb_co_u32 %2, vcc, %6, %9, vcc\n\tv_subb_co_u32 %3, vcc, 0, %9, vcc"
            : "=&v"(res.w0), "=&v"(res.w1), "=&v"(res.w2), "=&v"(res.w3) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(sx) : "vcc");
    } else if (q == 1) {   // lo = (t1,t0,0), hi = (sx,t4,t3,t2)
        asm("v_sub_co_u32 %0, vcc, 0, %6\n\tv_subb_co_u32 %1, vcc, %4, %7, vcc\n\tv_subb_co_u32 %2, vcc, %5, %8, vcc\n\tv_subb_co_u32 %3, vcc, 0, %9, vcc"
            : "=&v"(res.w0), "=&v"(res.w1), "=&v"(res.w2), "=&v"(res.w3) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(sx) : "vcc");
    } else {               // lo = (t0,0,0), hi = (t4,t3,t2,t1)
        asm("v_sub_co_u32 %0, vcc, 0, %5\n\tv_subb_co_u32 %1, vcc, 0, %6, vcc\n\tv_subb_co_u32 %2, vcc, %4, %7, vcc\n\tv_subb_co_u32 %3, vcc, 0, %8, vcc"
            : "=&v"(res.w0), "=&v"(res.w1), "=&v"(res.w2), "=&v"(res.w3) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4) : "vcc");
    }
    return res;
}

// any representative with |v| < 2^101 -> lazy field element:  (w1:w0) + w2 (2^32-1) - w3, with w3 biased by 32 to
// keep it unsigned and the bias (32 * 2^96 = -32) returned through the multiply-add's addend
__device__ __forceinline__ u64 to_gl_lazy(const f128 &v) {
    if constexpr (FERMAT_TIMING & 4) return (((u64)v.w1 << 32) | v.w0) ^ (((u64)v.w3 << 32) | v.w2);
#ifndef GL_FERMAT_TO_GL_ASM
    const u64 lo = ((u64)v.w1 << 32) | v.w0;
    const u32 hh = v.w3 + 32u;
    u64 t0, t2;
    const bool br = __builtin_usubl_overflow(lo, (u64)hh, &t0);
    const u64 t1 = (u64)v.w2 * EPS + 32u;
    const bool c = __builtin_uaddl_overflow(t0, t1, &t2);
    return t2 + ((c ? EPS : 0) - (br ? EPS : 0));
#else
    // the same on the carry flags (ten instructions; the compiler's form above re-derives each carry with a 64-bit compare) --
    // measured 1 % SLOWER per config-3 interpolate (200.6 vs 198.8 ms, same box): six carries in a row; not the default:
    //   s = (w1:w0) + [w2 (2^32-1) + 32]  (carry c);   r = s + c (2^32-1) - hh   as   r0 = s0 - hh - c (borrow b), r1 = (s1 + c) - b;
    //   a final borrow (the value was negative: it happens, w3 may be positive) adds p: (r0 + 1, r1 - 1 + carry)
    const u32 hh = v.w3 + 32u;
    const u64 t1 = (u64)v.w2 * EPS + 32u;
    u64 c, bo, br, cx, k; u32 s0, s1, r0, r1a, r1, m, q0, q1;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(s0), "=s"(c) : "v"(v.w0), "v"((u32)t1));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s1), "=s"(c) : "v"(v.w1), "v"((u32)(t1 >> 32)), "s"(c));
    asm(GL_SGPR_WAIT "v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r0), "=s"(bo) : "v"(s0), "v"(hh), "s"(c));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r1a), "=s"(cx) : "v"(s1), "s"(c));
    asm(GL_SGPR_WAIT "v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r1), "=s"(br) : "v"(r1a), "s"(bo));
    asm(GL_SGPR_WAIT "v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m) : "s"(br));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(q0), "=s"(k) : "v"(r0), "s"(br));
    asm(GL_SGPR_WAIT "v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(q1), "=s"(cx) : "v"(r1), "v"(m), "s"(k));
    return ((u64)q1 << 32) | q0;
#endif
}

template <int C, bool INV, int H, int BASE, int I>
struct DifStage {
    static __device__ __forceinline__ void run(f128 *x) {
        if constexpr (BASE < (1 << C)) {
            constexpr int e = 96 * I / H;            // w_(2H)^I = 2^e
            const f128 a = x[BASE + I], b = x[BASE + I + H];
            x[BASE + I] = add(a, b);
            if constexpr (e == 0) x[BASE + I + H] = sub(a, b);
            else if constexpr (!INV) x[BASE + I + H] = shl<e>(sub(a, b));
            else x[BASE + I + H] = shl<96 - e>(sub(b, a));           // 2^(-e) = 2^(192-e) = -2^(96-e)
            if constexpr (I + 1 < H) DifStage<C, INV, H, BASE, I + 1>::run(x);
            else DifStage<C, INV, H, BASE + 2 * H, 0>::run(x);
        }
    }
};
template <int C, bool INV, int H, int BASE, int I>
struct DitStage {
    static __device__ __forceinline__ void run(f128 *x) {
        if constexpr (BASE < (1 << C)) {
            constexpr int e = 96 * I / H;
            const f128 a = x[BASE + I];
            if constexpr (e == 0) { const f128 b = x[BASE + I + H]; x[BASE + I] = add(a, b); x[BASE + I + H] = sub(a, b); }
            else if constexpr (!INV) { const f128 b = shl<e>(x[BASE + I + H]); x[BASE + I] = add(a, b); x[BASE + I + H] = sub(a, b); }
            else { const f128 b = shl<96 - e>(x[BASE + I + H]); x[BASE + I] = sub(a, b); x[BASE + I + H] = add(a, b); }
            if constexpr (I + 1 < H) DitStage<C, INV, H, BASE, I + 1>::run(x);
            else DitStage<C, INV, H, BASE + 2 * H, 0>::run(x);
        }
    }
};

// x[r] natural order in, X[q] at x[bitrev(q)] out; X[q] = sum_r x[r] w^(rq), w = F.w[C] (INV: its inverse)
template <int C, bool INV, int H = (1 << C) / 2>
__device__ __forceinline__ void dft_dif(f128 *x) {
    if constexpr (H >= 1) {
        DifStage<C, INV, H, 0, 0>::run(x);
        dft_dif<C, INV, H / 2>(x);
    }
}
// x[bitrev(r)] in, X[q] natural order out
template <int C, bool INV, int H = 1>
__device__ __forceinline__ void dft_dit(f128 *x) {
    if constexpr (H < (1 << C)) {
        DitStage<C, INV, H, 0, 0>::run(x);
        dft_dit<C, INV, H * 2>(x);
    }
}

}  // namespace fermat
}  // namespace gl
