// Goldilocks arithmetic on "lazy" values (any u64 congruent to the element) with the carries taken from the instructions
// that produce them -- the flavour of gl::mul_lazy_b (gl_field.cuh) for additions, subtractions and multiplications by
// powers of two, which is what a radix-16 butterfly block of the NTT consists of (p divides 2^96 + 1, so every 16th root of
// unity is a power of 2^12: gl_fermat.cuh).
//
// Every operation is exact except in a case that has probability about 2^-32 per operation: a SECOND wrap while the
// first one is folded back (2^64 = 2^32 - 1 mod p).  Those cases only OR a lane mask into `bad`; the caller (a whole NTT
// tile) checks `bad` once and recomputes with the exact code path (the Z/(2^96+1) form) when it is set.
//
// gfx950 wants two wait states between a vector instruction that writes an SGPR pair and a vector instruction that
// reads it (hipcc pads its own carry chains with s_nop 1); inside an asm string nobody pads, so the strings carry them.
#pragma once
#include "gl_field.cuh"

namespace gl {
namespace lazy {

// a + b  (lazy + lazy -> lazy)
__device__ __forceinline__ u64 add_b(u64 a, u64 b, u64 &bad) {
    u32 s0, s1, c01; u64 c, c2, r;
    asm("v_add_co_u32_e64 %0, %2, %3, %5\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, %4, %6, %2"
        : "=&v"(s0), "=&v"(s1), "=&s"(c) : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c01) : "s"(c));
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(c2) : "v"(c01), "v"(((u64)s1 << 32) | s0));     // + 2^32-1 where it wrapped
    bad |= c2;                                                   // wrapped again: s >= 2^64 - 2^32 + 1 after a wrap
    return r;
}
// a - b
__device__ __forceinline__ u64 sub_b(u64 a, u64 b, u64 &bad) {
    u32 d0, d1; u64 br, c2, u;
    asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\ts_nop 1\n\tv_subb_co_u32_e64 %1, %2, %4, %6, %2"
        : "=&v"(d0), "=&v"(d1), "=&s"(br) : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)));
    // a borrow means the stored value is 2^64 too large: subtract 2^32 - 1, i.e. (d0 + 1, d1 - 1 + carry of the low word)
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %0, 0, %2" : "+v"(d0), "=&s"(c2) : "s"(br));
    const u64 t = br & ~c2;                                      // lanes whose high word still owes the 1
    asm("s_nop 1\n\tv_subbrev_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(d1), "=&s"(u) : "s"(t));
    bad |= u;                                                    // d < 2^32 - 1 after a borrow
    return ((u64)d1 << 32) | d0;
}

// lo + hi 2^64 with hi < 2^32 (the shift-by-less-than-32 case and the tail of the wider ones)
__device__ __forceinline__ u64 fold_hi32_b(u64 lo, u32 hi, u64 &bad) {
    u64 z, c, c2, r; u32 c01;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"(hi), "v"(lo));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c01) : "s"(c));
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(c2) : "v"(c01), "v"(z));
    bad |= c2;
    return r;
}

// x * 2^E mod p for a compile-time 0 < E < 96
template <int E>
__device__ __forceinline__ u64 mul2e_b(u64 x, u64 &bad) {
    static_assert(E > 0 && E < 96, "exponent out of range");
    if constexpr (E < 32) {
        return fold_hi32_b(x << E, (u32)(x >> 32) >> (32 - E), bad);
    } else if constexpr (E < 64) {                               // x 2^E = lo + hi 2^64, hi < 2^E: the general 128-bit reduction
        const u64 lo = x << E, hi = x >> (64 - E);
        u64 z = fold_hi32_b(lo, (u32)hi, bad);                   // + hl (2^32-1)
        u32 r0, r1; u64 br;                                      // - hh  (2^96 = -1)
        asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\ts_nop 1\n\tv_subbrev_co_u32_e64 %1, %2, 0, %4, %2"
            : "=&v"(r0), "=&v"(r1), "=&s"(br) : "v"((u32)z), "v"((u32)(z >> 32)), "v"((u32)(hi >> 32)));
        bad |= br;                                               // z < hh < 2^32
        return ((u64)r1 << 32) | r0;
    } else {                                                     // y = x 2^(E-64) = (y2, y1, y0);  y 2^64 = y0 (2^32-1) - y1 - y2 2^32
        constexpr int e = E - 64;
        const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
        const u32 y0 = e ? (x0 << e) : x0, y1 = e ? __builtin_amdgcn_alignbit(x1, x0, 32 - e) : x1, y2 = e ? (x1 >> (32 - e)) : 0u;
        const u64 z = (u64)y0 * 0xFFFFFFFFu;
        return sub_b(z, ((u64)y2 << 32) | y1, bad);              // the borrow here is NOT rare (up to 2^-12): sub_b folds it back exactly
    }
}

// 16-point DFT in registers, decimation in time: x[bitrev(r)] in, X[q] natural order out, X[q] = sum_r x[r] w^(rq) with
// w = F.w[4] = 2^12 (INV: its inverse, w^-k = -w^(8-k)).  Same contract as fermat::dft_dit<4, INV> followed by to_gl_lazy.
template <bool INV, int H, int BASE, int I>
struct DitStage16 {
    static __device__ __forceinline__ void run(u64 *x, u64 &bad) {
        if constexpr (BASE < 16) {
            constexpr int e = 96 * I / H;                        // w_(2H)^I = 2^e
            const u64 a = x[BASE + I];
            if constexpr (e == 0) { const u64 b = x[BASE + I + H]; x[BASE + I] = add_b(a, b, bad); x[BASE + I + H] = sub_b(a, b, bad); }
            else if constexpr (!INV) { const u64 b = mul2e_b<e>(x[BASE + I + H], bad); x[BASE + I] = add_b(a, b, bad); x[BASE + I + H] = sub_b(a, b, bad); }
            else { const u64 b = mul2e_b<96 - e>(x[BASE + I + H], bad); x[BASE + I] = sub_b(a, b, bad); x[BASE + I + H] = add_b(a, b, bad); }   // 2^-e = -2^(96-e)
            if constexpr (I + 1 < H) DitStage16<INV, H, BASE, I + 1>::run(x, bad);
            else DitStage16<INV, H, BASE + 2 * H, 0>::run(x, bad);
        }
    }
};
template <bool INV, int H = 1>
__device__ __forceinline__ void dft16_dit(u64 *x, u64 &bad) {
    if constexpr (H < 16) {
        DitStage16<INV, H, 0, 0>::run(x, bad);
        dft16_dit<INV, H * 2>(x, bad);
    }
}
// decimation in frequency: natural in, X[q] at x[bitrev(q)] out
template <bool INV, int H, int BASE, int I>
struct DifStage16 {
    static __device__ __forceinline__ void run(u64 *x, u64 &bad) {
        if constexpr (BASE < 16) {
            constexpr int e = 96 * I / H;
            const u64 a = x[BASE + I], b = x[BASE + I + H];
            x[BASE + I] = add_b(a, b, bad);
            if constexpr (e == 0) x[BASE + I + H] = sub_b(a, b, bad);
            else if constexpr (!INV) x[BASE + I + H] = mul2e_b<e>(sub_b(a, b, bad), bad);
            else x[BASE + I + H] = mul2e_b<96 - e>(sub_b(b, a, bad), bad);
            if constexpr (I + 1 < H) DifStage16<INV, H, BASE, I + 1>::run(x, bad);
            else DifStage16<INV, H, BASE + 2 * H, 0>::run(x, bad);
        }
    }
};
template <bool INV, int H = 8>
__device__ __forceinline__ void dft16_dif(u64 *x, u64 &bad) {
    if constexpr (H >= 1) {
        DifStage16<INV, H, 0, 0>::run(x, bad);
        dft16_dif<INV, H / 2>(x, bad);
    }
}

}  // namespace lazy
}  // namespace gl
