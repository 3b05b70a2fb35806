// BN254 Poseidon linear layers on the gfx950 matrix cores (v_mfma_i32_32x32x32_i8).
//
// A linear layer of the permutation (poseidon.circom:32-43: the dense MDS product; in the sparse form of the partial rounds the
// rows V_k, the columns W_k and the closing matrix) is out_i = sum_j A_ij x_j mod r with CONSTANT A.  On the vector ALU a term is
// 64 32x32 products (two instructions each); here the constant goes into the matrix operand instead:
//     c[i][j][b] = A_ij * 256^b * 2^32 mod r,  b < 32        plain integers < r, written in signed base-256 digits d_k, k < 32
//     S[i][k]    = 2^30 + sum_{j,b} d_k(c[i][j][b]) * (byte_b(x_j) - 128)            an i8 x i8 -> i32 product, one value per byte position
//     V_i        = sum_k 256^k S[i][k]  < 2^280;    out_i = (V_i + m r) / 2^32 + K_i,   m = -V_i / r mod 2^32
// K_i = (128 sum_{j,b} c[i][j][b] - sum_k 2^30 256^k) / 2^32 mod r takes back the two offsets (operand bytes are fed as signed
// u - 128; the accumulators start at 2^30 -- the bit pattern of 2.0f, an inline constant of the instruction -- so that every S is positive).  The reduction mod r
// is in the constants: ONE 32-bit Montgomery step per row is left of the 8-step reduction the vector form pays per row, and no
// 32x32 product of the state at all.  out_i < 2^249 + 2r < 2^255 stays as it is (lazy) until the permutation's output is read.
// Integer model of exactly these steps against the plain statement: tools/bn_mfma_model.py.
//
// Who holds what.  The kernels keep ONE PERMUTATION PER LANE (bn128.hip).  D(32x32) = A(32x32) * B(32x32) serves 32 permutations
// (columns n); the wave's 64 go in two tiles T: permutation p = 32 T + n.
//   * B, lane (n, g = lane/32): 16 bytes = K slots of group g = bytes 16g..16g+15 of x_j of permutation 32T + n.  Lane p owns all
//     eight limbs of ITS x_j: four v_permlane32_swap (limb q of the upper half-wave against limb 4+q of the lower) turn the two
//     owners' registers into the operands of both tiles -- x[0..3] becomes tile 0's, x[4..7] tile 1's.
//   * A, lane (m, g): row m, the 16 digits d_pos(m)(c[i][j][16g + s]), s < 16.  Rows are placed so that result lane (n, h) --
//     which receives rows 8q + 4h + r in register 4q + r -- holds byte positions 16h .. 16h+15 in register order:
//     pos(m) = 16 ((m/4)%2) + 4 (m/8) + m%4.  A tile is 1 KB in lane order; tile (i, j) of a rows x cols layer is at
//     tiles + (i * cols + j) * 64 lanes (uint4 each): a coalesced 16-byte load per lane, served by the L2 (a t = 17 layer is 289 KB).
//   * result: each lane carries its 16 positions into 4 limbs + an overflow word per tile; five more swaps give lane p the low
//     and the high half of ITS row, which it finishes alone.
#pragma once
#include "bn_field.cuh"

namespace bnm {

using bn::u32;
using bn::u64;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef const v4i __attribute__((address_space(1))) *gtile;

constexpr int ACC_BIAS = 1 << 30;         // 2.0f's bit pattern: an inline constant of the matrix instruction (its C operand), no register set-up per accumulator

__device__ __forceinline__ v16i acc_init() {
    v16i a;
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = ACC_BIAS;
    return a;
}

// the operands of both tiles from the lane's own eight limbs
// (the swaps go through the builtin: hipcc pads a vector write against the swap that reads it -- two wait states -- itself;
// their inputs must therefore come from compiler-generated instructions, not straight out of an asm statement)
__device__ __forceinline__ void b_prep(const u32 x[8], v4i &b0, v4i &b1) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const auto r = __builtin_amdgcn_permlane32_swap(x[q] ^ 0x80808080u, x[4 + q] ^ 0x80808080u, false, false);
        b0[q] = (int)r[0];
        b1[q] = (int)r[1];
    }
}

__device__ __forceinline__ v16i mfma(v4i a, v4i b, v16i c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
// The FIRST products of a row's two accumulators, started at the bias as the instruction's inline constant (C = 2.0: the bit pattern 2^30) instead of
// sixteen registers per accumulator set up first (hipcc has no way to say that: 16 v_mov_b64 per row).  One asm statement that keeps the hazards the
// compiler keeps around the builtin: two wait states after the vector writes of the operands (the swaps of b_prep), and the 8-pass result's eleven
// before anything but a matrix instruction accumulating on it may read it.  -DBN_ACC_INLINE=0: the builtin on acc_init() (A/B builds).
#ifndef BN_ACC_INLINE
#define BN_ACC_INLINE 1
#endif
__device__ __forceinline__ void mfma_first(v4i a, v4i b0, v4i b1, v16i &c0, v16i &c1) {
#if BN_ACC_INLINE
    asm("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %2, %3, 2.0\n\tv_mfma_i32_32x32x32_i8 %1, %2, %4, 2.0\n\ts_nop 10"
        : "=&v"(c0), "=&v"(c1) : "v"(a), "v"(b0), "v"(b1));
#else
    c0 = mfma(a, b0, acc_init()); c1 = mfma(a, b1, acc_init());
#endif
}

// ---- carry chains through vcc, one statement each (the compiler's 64-bit emulation costs two to three times the instructions)
// a[0..N) += b[0..N), the carry out of the top limb is dropped (the callers' sums fit)
#define BNM_ADDC(i, j) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #j ", vcc\n\t"
__device__ __forceinline__ void add_chain5(u32 a[5], const u32 b[5]) {
    asm("v_add_co_u32 %0, vcc, %0, %5\n\t" BNM_ADDC(1, 6) BNM_ADDC(2, 7) BNM_ADDC(3, 8) BNM_ADDC(4, 9)
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]) : "vcc");
}
__device__ __forceinline__ void add_chain8(u32 a[8], const u32 b[8]) {
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t" BNM_ADDC(1, 9) BNM_ADDC(2, 10) BNM_ADDC(3, 11) BNM_ADDC(4, 12) BNM_ADDC(5, 13) BNM_ADDC(6, 14) BNM_ADDC(7, 15)
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
}
// a[0..9) += b[0..8) (the ninth limb takes the carry)
__device__ __forceinline__ void add_chain9_8(u32 a[9], const u32 b[8]) {
    asm("v_add_co_u32 %0, vcc, %0, %9\n\t" BNM_ADDC(1, 10) BNM_ADDC(2, 11) BNM_ADDC(3, 12) BNM_ADDC(4, 13) BNM_ADDC(5, 14) BNM_ADDC(6, 15) BNM_ADDC(7, 16)
        "v_addc_co_u32 %8, vcc, 0, %8, vcc"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8])
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
}
// 32 x 32 -> 64 and multiply-add on a 64-bit accumulator, one v_mad_u64_u32 each; the constant operand sits in a scalar register
__device__ __forceinline__ u64 mul_s(u32 x, u32 sc) { u64 d; asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(x), "s"(sc) : "vcc"); return d; }
__device__ __forceinline__ u64 mad_s(u32 x, u32 sc, u64 acc) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "s"(sc) : "vcc"); return acc; }

// sum_{k<16} 256^k a[k] (every a[k] < 2^31) -> five 32-bit words.  Per limb the three shifted terms are ONE 64-bit value (three
// multiply-adds by 2^8, 2^16, 2^24 held in scalar registers, no chain between limbs); the unshifted term and the previous limb's
// overflow (< 2^24) add without carry in 32 bits; one five-limb chain joins them: 12 + 3 + 5 instructions.
struct Sh { u32 s8, s16, s24; u64 zero; };
__device__ __forceinline__ Sh sh_init() {
    Sh s = { 1u << 8, 1u << 16, 1u << 24, 0 };
    asm volatile("" : "+s"(s.s8), "+s"(s.s16), "+s"(s.s24), "+v"(s.zero));   // opaque: x * 2^8 + 0 stays one multiply-add
    return s;
}
__device__ __forceinline__ void carry5(const v16i &a, u32 w[5], const Sh &sh) {
    // The accumulators are read by COMPILER-generated instructions only: hipcc pads a matrix instruction's result against its
    // first reader (the hardware does not interlock it), but not against a reader inside an asm statement.
    u64 L[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        L[l] = (u64)(u32)a[4 * l + 1] * sh.s8 + sh.zero;
        L[l] = (u64)(u32)a[4 * l + 2] * sh.s16 + L[l];
        L[l] = (u64)(u32)a[4 * l + 3] * sh.s24 + L[l];
    }
    u32 e[5];
    e[0] = (u32)a[0];
#pragma unroll
    for (int l = 1; l < 4; l++) e[l] = (u32)a[4 * l] + (u32)(L[l - 1] >> 32);
    e[4] = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) w[l] = (u32)L[l];
    w[4] = (u32)(L[3] >> 32);
    add_chain5(w, e);
}

// Both tiles' accumulators of one row -> ten words: w[0..4] / w[5..9] = this lane's 16 positions of tile 0 / tile 1 as 5-limb numbers
__device__ __forceinline__ void carry_pair(const v16i &a0, const v16i &a1, u32 w[10], const Sh &sh) {
    carry5(a0, w, sh);
    carry5(a1, w + 5, sh);
}
// w += v as two 5-limb numbers (a row gathered in two accumulations, e.g. a partial round's row: the block's part and the cross terms)
__device__ __forceinline__ void add_pair(u32 w[10], const u32 v[10]) {
    add_chain5(w, v);
    add_chain5(w + 5, v + 5);
}
// The ten words of one row -> this lane's (permutation's) out_i, LAZY: some representative below 2^255 (the values between the
// layers are such: the matrix operands take ANY 256-bit representative, and the S-box's reduction-free products of operands below
// 0.9 * 2^256 stay below 2^256 -- fr_mul_nr).
// kc: the row's K_i, 8 limbs, wave-uniform.  V + 2^32 K + m r with m = -V/r mod 2^32, all three additions as carry chains.
__device__ __forceinline__ void finish_words(u32 w[10], const u32 *kc, u32 out[8]) {
    u32 *w0 = w, *w1 = w + 5;
    // afterwards w0 = low half (bits 0..159), w1 = high half (from bit 128) of the lane's own row.  The words come out of asm
    // statements (the carry chains), which hipcc does not pad against the swap: the two wait states are in the statement.
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %5\n\tv_permlane32_swap_b32 %1, %6\n\tv_permlane32_swap_b32 %2, %7\n\t"
                 "v_permlane32_swap_b32 %3, %8\n\tv_permlane32_swap_b32 %4, %9"
                 : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]), "+v"(w0[4]), "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]), "+v"(w1[4]));
    u32 t[9], z5[5] = { w0[4], 0, 0, 0, 0 };
#pragma unroll
    for (int l = 0; l < 4; l++) t[l] = w0[l];
#pragma unroll
    for (int l = 0; l < 5; l++) t[4 + l] = w1[l];
    add_chain5(t + 4, z5);                            // V < 2^280: nine limbs
    u32 k[8];
#pragma unroll
    for (int l = 0; l < 8; l++) k[l] = kc[l];
    const u32 m = t[0] * bn::N0INV;
    add_chain8(t + 1, k);                             // + 2^32 K (does not touch the limb m comes from)
    u32 ul[8], uh[8], rl[8];
#pragma unroll
    for (int l = 0; l < 8; l++) { rl[l] = bn::r_limb(l); asm volatile("" : "+s"(rl[l])); }   // r in scalar registers: one v_mad_u64_u32 per limb, no literal moves
    u64 u[8];
    asm("v_mad_u64_u32 %0, vcc, %8, %9, 0\n\tv_mad_u64_u32 %1, vcc, %8, %10, 0\n\tv_mad_u64_u32 %2, vcc, %8, %11, 0\n\tv_mad_u64_u32 %3, vcc, %8, %12, 0\n\t"
        "v_mad_u64_u32 %4, vcc, %8, %13, 0\n\tv_mad_u64_u32 %5, vcc, %8, %14, 0\n\tv_mad_u64_u32 %6, vcc, %8, %15, 0\n\tv_mad_u64_u32 %7, vcc, %8, %16, 0"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
        : "v"(m), "s"(rl[0]), "s"(rl[1]), "s"(rl[2]), "s"(rl[3]), "s"(rl[4]), "s"(rl[5]), "s"(rl[6]), "s"(rl[7]) : "vcc");
#pragma unroll
    for (int l = 0; l < 8; l++) { ul[l] = (u32)u[l]; uh[l] = (u32)(u[l] >> 32); }
    add_chain9_8(t, ul);                              // limb 0 becomes 0
    add_chain8(t + 1, uh);                            // (V + m r) / 2^32 + K < 2^249 + 2r < 2^255
#pragma unroll
    for (int l = 0; l < 8; l++) out[l] = t[1 + l];
}
__device__ __forceinline__ void finish_row(const v16i &a0, const v16i &a1, const u32 *kc, u32 out[8], const Sh &sh) {
    u32 w[10];
    carry_pair(a0, a1, w, sh);
    finish_words(w, kc, out);
}
// any representative below 2^255 (< 3r) -> the canonical one: two subtractions at most
__device__ __forceinline__ void canon(u32 x[8]) {
    u32 t[9];
#pragma unroll
    for (int l = 0; l < 8; l++) t[l] = x[l];
    t[8] = 0;
    bn::cond_sub_r(t);
    bn::cond_sub_r(t);
#pragma unroll
    for (int l = 0; l < 8; l++) x[l] = t[l];
}

}  // namespace bnm
