// BN254 Poseidon linear layers on the gfx950 matrix cores (v_mfma_i32_32x32x32_i8).
//
// A linear layer of the permutation (poseidon.circom:32-43: the dense MDS product; in the sparse form of the partial rounds the
// rows V_k, the columns W_k and the closing matrix) is out_i = sum_j A_ij x_j mod r with CONSTANT A.  On the vector ALU a term is
// 64 32x32 products (two instructions each); here the constant goes into the matrix operand instead:
//     c[i][j][b] = A_ij * 256^b * 2^32 mod r,  b < 32        plain integers < r, written in signed base-256 digits d_k, k < 32
//     S[i][k]    = 2^25 + sum_{j,b} d_k(c[i][j][b]) * (byte_b(x_j) - 128)            an i8 x i8 -> i32 product, one value per byte position
//     V_i        = sum_k 256^k S[i][k]  < 2^274.01;    out_i = (V_i + m r) / 2^32 + K_i,   m = -V_i / r mod 2^32
// K_i = (128 sum_{j,b} c[i][j][b] - sum_k 2^25 256^k) / 2^32 mod r takes back the two offsets (operand bytes are fed as signed
// u - 128; the accumulators start at 2^25, the instruction's inline constant, so that every S is positive).  The reduction mod r
// is in the constants: ONE 32-bit Montgomery step per row is left of the 8-step reduction the vector form pays per row, and no
// 32x32 product of the state at all.  out_i < 2^242.01 + 2r: two conditional subtractions make it canonical.
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

constexpr int ACC_BIAS = 1 << 25;

__device__ __forceinline__ v16i acc_init() {
    v16i a;
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = ACC_BIAS;
    return a;
}

// the operands of both tiles from the lane's own eight limbs
__device__ __forceinline__ void b_prep(const u32 x[8], v4i &b0, v4i &b1) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const auto r = __builtin_amdgcn_permlane32_swap(x[q] ^ 0x80808080u, x[4 + q] ^ 0x80808080u, false, false);
        b0[q] = (int)r[0];
        b1[q] = (int)r[1];
    }
}

__device__ __forceinline__ v16i mfma(v4i a, v4i b, v16i c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }

// sum_{k<16} 256^k a[k] (every a[k] < 2^26) -> five 32-bit words.  The shifts are multiply-adds by constants kept opaque in
// scalar registers: x * 2^8 + acc stays ONE v_mad_u64_u32 instead of a 64-bit shift and a 64-bit add.
struct Sh { u32 s8, s16, s24; };
__device__ __forceinline__ Sh sh_init() {
    Sh s = { 1u << 8, 1u << 16, 1u << 24 };
    asm volatile("" : "+s"(s.s8), "+s"(s.s16), "+s"(s.s24));
    return s;
}
__device__ __forceinline__ void carry5(const v16i &a, u32 w[5], const Sh &sh) {
    u64 acc = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        acc += (u32)a[4 * l];
        acc = (u64)(u32)a[4 * l + 1] * sh.s8 + acc;
        acc = (u64)(u32)a[4 * l + 2] * sh.s16 + acc;
        acc = (u64)(u32)a[4 * l + 3] * sh.s24 + acc;
        w[l] = (u32)acc;
        acc >>= 32;
    }
    w[4] = (u32)acc;
}

// Both tiles' accumulators of one row -> ten words: w[0..4] / w[5..9] = this lane's 16 positions of tile 0 / tile 1 as 5-limb numbers
__device__ __forceinline__ void carry_pair(const v16i &a0, const v16i &a1, u32 w[10], const Sh &sh) {
    carry5(a0, w, sh);
    carry5(a1, w + 5, sh);
}
// w += v as two 5-limb numbers (a row gathered in two accumulations, e.g. a partial round's row: the block's part and the cross terms)
__device__ __forceinline__ void add_pair(u32 w[10], const u32 v[10]) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        u64 c = 0;
#pragma unroll
        for (int l = 0; l < 5; l++) { c += (u64)w[5 * h + l] + v[5 * h + l]; w[5 * h + l] = (u32)c; c >>= 32; }
    }
}
// the ten words of one row -> this lane's (permutation's) canonical out_i.  kc: the row's K_i, 8 limbs, wave-uniform.
__device__ __forceinline__ void finish_words(u32 w[10], const u32 *kc, u32 out[8]) {
    u32 *w0 = w, *w1 = w + 5;
#pragma unroll
    for (int q = 0; q < 5; q++) {                     // afterwards w0 = low half (bits 0..159), w1 = high half (from bit 128) of the lane's own row
        const auto r = __builtin_amdgcn_permlane32_swap(w0[q], w1[q], false, false);
        w0[q] = r[0];
        w1[q] = r[1];
    }
    u32 t[9];
#pragma unroll
    for (int l = 0; l < 4; l++) t[l] = w0[l];
    u64 c = (u64)w0[4] + w1[0];
    t[4] = (u32)c; c >>= 32;
#pragma unroll
    for (int l = 1; l < 5; l++) { c += w1[l]; t[4 + l] = (u32)c; c >>= 32; }
    // one Montgomery step: (V + m r) / 2^32
    const u32 m = t[0] * bn::N0INV;
    u64 v = (u64)m * bn::r_limb(0) + t[0];
    u32 o[9];
#pragma unroll
    for (int l = 1; l < 8; l++) { v = (u64)m * bn::r_limb(l) + t[l] + (v >> 32); o[l - 1] = (u32)v; }
    o[7] = t[8] + (u32)(v >> 32);                     // < 2^244 + r: eight limbs
    // + K_i, then at most two subtractions of r
    u64 s = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) { s += (u64)o[l] + kc[l]; o[l] = (u32)s; s >>= 32; }
    o[8] = 0;                                         // < 2^255
    bn::cond_sub_r(o);
    bn::cond_sub_r(o);
#pragma unroll
    for (int l = 0; l < 8; l++) out[l] = o[l];
}
__device__ __forceinline__ void finish_row(const v16i &a0, const v16i &a1, const u32 *kc, u32 out[8], const Sh &sh) {
    u32 w[10];
    carry_pair(a0, a1, w, sh);
    finish_words(w, kc, out);
}

}  // namespace bnm
