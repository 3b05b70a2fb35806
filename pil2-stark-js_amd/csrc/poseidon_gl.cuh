// Poseidon-12 permutation over Goldilocks for gfx950, one permutation per lane.
//
// Spec: the un-optimised 30-round form of the reference's WASM kernel,
// src/helpers/glwasm.js:216-426 (round constants :535-627; x^7 on all lanes in rounds 0-3 and
// 26-29, on lane 0 otherwise :377-384; MDS = circ(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,..)
// :428-440).  It equals, mod p, the optimised JS form hash/poseidon/poseidon.js:57-108.  The dense
// 64-bit matrices of the optimised form would cost 23 full multiplications per partial round on
// a machine without a 64-bit multiplier; the circulant form only needs 6-bit constants.
//
// Values stay "lazy" (any u64 representative) between rounds; outputs are canonicalised.
#pragma once
#include "gl_field.cuh"
#include "poseidon_mds_mfma.cuh"

namespace gl {

#define POSEIDON_GL_RC_QUAL static __device__
#include "poseidon_gl_constants.inc"     // POSEIDON_GL_RC[360]: round r, lane i at [12*r+i]
#undef POSEIDON_GL_RC_QUAL

// x^7 with the 13-instruction products; a lane flagged in `bad` (probability ~2^-30) must be recomputed with pow7_lazy
__device__ __forceinline__ u64 pow7_b(u64 x, u64 &bad) {
    u64 x2 = mul_lazy_b(x, x, bad);
    u64 x3 = mul_lazy_b(x2, x, bad);
    u64 x4 = mul_lazy_b(x2, x2, bad);
    return mul_lazy_b(x3, x4, bad);
}
__device__ __forceinline__ u64 pow7_lazy(u64 x) {
    u64 x2 = mul_lazy(x, x);
    u64 x3 = mul_lazy(x2, x);
    u64 x4 = mul_lazy(x2, x2);
    return mul_lazy(x3, x4);
}

// out = M * st with the 6-bit circulant.  The low and high 32-bit halves of the 12 lanes are accumulated
// separately with v_mad_u64_u32 (A = sum lo_j*m, B = sum hi_j*m, both < 2^41); then
//   A + B*2^32 = A + B_hi*2^64 + B_lo*2^32 = (A + B_hi*(2^32-1)) + B_lo*2^32   (mod p)
// where the bracket is < 2^42 (one more mad) and B_lo*2^32 only touches the upper word: one carry to fix.
__device__ __forceinline__ void mds_layer(u64 st[12]) {
    constexpr u32 MC[12] = { 17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20 };
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)st[j]; hi[j] = (u32)(st[j] >> 32); }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u64 A = 0, B = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u32 m = MC[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
            A += (u64)lo[j] * m;
            B += (u64)hi[j] * m;
        }
        const u64 t = (u64)(u32)(B >> 32) * EPS + A;          // < 2^42
        u32 th;
        const bool c = __builtin_uadd_overflow((u32)(t >> 32), (u32)B, &th);
        const u64 r = ((u64)th << 32) | (u32)t;
        st[i] = c ? r + EPS : r;                               // wrapped once: r is small, cannot wrap again
    }
}

#ifndef POSEIDON_POW7
#define POSEIDON_POW7(x, bad) pow7_b(x, bad)        // (tools/sbox_bench.hip builds cost-split experiments with other bodies)
#endif
#ifndef POSEIDON_SBOX_GROUP
#define POSEIDON_SBOX_GROUP 6       // S-boxes per fallback check: the inputs of a group stay live until its check (12: 24 VGPRs spilled in the leaf kernel and 41 GB of scratch traffic per config-3 launch; 6 and 4: none; same rate)
#endif
// S-box layers of the matrix-core form, on the 13-instruction products (pow7_b): a full layer is checked once, and a
// wave in which some lane hit the rare borrow recomputes the layer with pow7_lazy (the branch is wave-uniform: every
// lane recomputes, same values)
// rc = nullptr (a compile-time fact at every call): the addends are already in st[] (mds_layer_lds put them there)
template <int GROUP = POSEIDON_SBOX_GROUP>
__device__ __forceinline__ void sbox_full(u64 st[12], const u64 *__restrict__ rc) {
    static_assert(12 % GROUP == 0, "group must divide the state");
#pragma unroll
    for (int g = 0; g < 12; g += GROUP) {
        u64 bad = 0, in[GROUP];
#pragma unroll
        for (int i = 0; i < GROUP; i++) { in[i] = rc ? add_lazy_canon(st[g + i], rc[g + i]) : st[g + i]; st[g + i] = POSEIDON_POW7(in[i], bad); }
        if (__builtin_expect(bad != 0, 0)) {
#pragma unroll
            for (int i = 0; i < GROUP; i++) st[g + i] = pow7_lazy(in[i]);
        }
    }
}
__device__ __forceinline__ u64 sbox_one(u64 x) {
    u64 bad = 0;
    u64 y = POSEIDON_POW7(x, bad);
    if (__builtin_expect(bad != 0, 0)) y = pow7_lazy(x);
    return y;
}

}  // namespace gl
#include "poseidon_blocks.cuh"
namespace gl {

// in-place permutation; st[] canonical or lazy in, canonical out.
// Matrix-core form (poseidon_mds_mfma.cuh): the whole wave must reach every call (no lane may have left the kernel or
// sit in another branch, because the MDS operands of all 64 lanes feed one MFMA); `m` comes from mds_mfma_init().
// NCANON: how many leading outputs are made canonical (12: all; a sponge between two of its permutations needs none -- the
// next permutation takes any representative --, a digest needs its 4)
// Rounds 4..25 run four to a linear layer (poseidon_blocks.cuh), every layer but the last on biased accumulators whose
// constants the generator folded into POSEIDON_BLK_RCF / POSEIDON_BLK_C0; `m` comes from poseidon_init().
// (forced inline: an outlined permutation passes its state through scratch, at a fifth of the rate)
// NOUT: how many leading outputs the caller reads (4: sponges and tree nodes -- the last layer then computes one row set; st[4..11] are stale afterwards)
template <int NCANON = 12, int NOUT = 12>
__device__ __forceinline__ void poseidon_perm(u64 st[12], const MdsMfma &m) {
    static_assert(NOUT == 4 || NOUT == 8 || NOUT == 12, "row sets of four outputs");
    static_assert(NCANON <= NOUT, "canonical outputs must be computed ones");
    const v4i *__restrict__ A = m.blkA;
    // every S-box's addend but round 0's is added by the layer (or block) before it: POSEIDON_BLK_LC[k] = what layer k adds
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = add_lazy_canon(st[i], POSEIDON_BLK_RCF[i]);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        sbox_full(st, nullptr);
        mds_layer_lds<false>(st, m, &POSEIDON_BLK_LC[r * 12]);
    }
#pragma unroll 1
    for (int b = 0; b < POSEIDON_BLK_N; b++) poseidon_partial_block(st, &POSEIDON_BLK_C0[POSEIDON_BLK_K * b], A, m);
#pragma unroll 1
    for (int r = POSEIDON_BLK_N * POSEIDON_BLK_K; r < 22; r++) {
        st[0] = sbox_one(st[0]);
        mds_layer_lds<false>(st, m, &POSEIDON_BLK_LC[(4 + r - POSEIDON_BLK_N * POSEIDON_BLK_K) * 12]);
    }
#pragma unroll 1
    for (int r = 4; r < 7; r++) {
        sbox_full(st, nullptr);
        mds_layer_lds<false>(st, m, &POSEIDON_BLK_LC[(r + 2) * 12]);
    }
    sbox_full(st, nullptr);
#ifdef PIL2GL_LAST_LAYER_FULL
    mds_layer_lds<true, 3>(st, m);                  // (A/B builds: every permutation's last layer whole)
#else
    mds_layer_lds<true, NOUT / 4>(st, m);
#endif
#pragma unroll
    for (int i = 0; i < NCANON; i++) st[i] = canon(st[i]);
}
// the same with every partial round a layer of its own (round 2's form; kept for tools/sbox_bench.hip's A/B and as a second
// statement of the permutation the parity tests compare with: pil2gl_selftest_perm)
template <int NCANON = 12>
__device__ inline void poseidon_perm_single(u64 st[12], const MdsMfma &m) {
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        sbox_full(st, &POSEIDON_GL_RC[r * 12]);
        mds_layer_mfma(st, m);
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        st[0] = sbox_one(add_lazy_canon(st[0], POSEIDON_GL_PARTIAL_C0[r]));
        mds_layer_mfma(st, m);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
        sbox_full(st, r == 26 ? POSEIDON_GL_RC26F : &POSEIDON_GL_RC[r * 12]);
        mds_layer_mfma(st, m);
    }
#pragma unroll
    for (int i = 0; i < NCANON; i++) st[i] = canon(st[i]);
}

// rounds 4..25 alone, for the parity tests (which = 0 blocked, 1 one layer per round): arbitrary states in, any representatives
// out.  The blocked form expects the constant a biased layer leaves on its input and leaves its own on its output
// (POSEIDON_BLK_ERR_IN / _OUT, from the generator): taken off here so that both forms map field elements to field elements.
__device__ inline void poseidon_partial_rounds(u64 st[12], const MdsMfma &m, int which) {
    if (which == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = add_lazy_canon(st[i], POSEIDON_BLK_ERR_IN[i]);
        st[0] = add_lazy_canon(st[0], POSEIDON_BLK_C0[0]);
#pragma unroll 1
        for (int b = 0; b < POSEIDON_BLK_N; b++) poseidon_partial_block(st, &POSEIDON_BLK_C0[POSEIDON_BLK_K * b], m.blkA, m);
#pragma unroll 1
        for (int r = POSEIDON_BLK_N * POSEIDON_BLK_K; r < 22; r++) {
            st[0] = sbox_one(st[0]);
            mds_layer_lds<false>(st, m, &POSEIDON_BLK_LC[(r == 21 ? 9 : 4 + r - POSEIDON_BLK_N * POSEIDON_BLK_K) * 12]);     // row 9: zeros
        }
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = sub(canon(st[i]), POSEIDON_BLK_ERR_OUT[i]);
    } else {
#pragma unroll 1
        for (int r = 0; r < 22; r++) {
            st[0] = sbox_one(add_lazy_canon(st[0], POSEIDON_GL_PARTIAL_C0[r]));
            mds_layer_mfma(st, m);
        }
    }
}

// every thread of the workgroup, once, before the first permutation: the MDS operands of this lane and the workgroup's copy
// of the blocked rounds' operand table (27 KB of LDS)
// (m.A / m.C, the register operands of mds_layer_mfma(), are NOT set: a kernel that also runs that form calls mds_mfma_init first)
__device__ __forceinline__ void poseidon_init(MdsMfma &m) {
    __shared__ v4i poseidon_blk_table[POSEIDON_BLK_OPERANDS * 64];
    m.sh16 = 65536u;
    asm volatile("" : "+s"(m.sh16));
    m.blkA = poseidon_blk_load(poseidon_blk_table);
}

// vector-ALU form (any subset of lanes): in-place permutation; st[] canonical or lazy in, canonical out.
// Rounds 4..25 use the folded constants (one addition per round, see poseidon_gl_constants.inc).
__device__ inline void poseidon_perm(u64 st[12]) {
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = pow7_lazy(add_lazy_canon(st[i], POSEIDON_GL_RC[r * 12 + i]));
        mds_layer(st);
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        st[0] = pow7_lazy(add_lazy_canon(st[0], POSEIDON_GL_PARTIAL_C0[r]));
        mds_layer(st);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = pow7_lazy(add_lazy_canon(st[i], r == 26 ? POSEIDON_GL_RC26F[i] : POSEIDON_GL_RC[r * 12 + i]));
        mds_layer(st);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = canon(st[i]);
}

}  // namespace gl
