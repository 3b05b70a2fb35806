// Partial rounds of the Goldilocks Poseidon-12 permutation, FOUR rounds per linear layer, on the gfx950 matrix cores.
//
// Spec: rounds 4..25 of src/helpers/glwasm.js:216-426 (x^7 on lane 0 only, then the MDS matrix of :428-440); the
// re-arrangement and its tables are derived -- and checked against the plain permutation and the reference's vectors --
// by gen_poseidon_blocks.py, which see.  In short, with y the state after a block's first S-box and d_j the increments
// S(t_j + c_j) - t_j of its three later S-boxes,
//     t_r   = (M^r y)[0] + sum_{j<r} d_j (M^(r-j))[0][0]           r = 1..3
//     x_out = M^4 y + sum_j d_j M^(4-j) e0
// "M^4 y" is the single layer's byte-plane product (poseidon_mds_mfma.cuh) with four signed base-256 digits per
// coefficient: 7 planes per 32-bit word instead of 4, but one recombination per element per four rounds (12 instead
// of 48) and 54 matrix instructions instead of 72.  Why it pays: the single layer spends more vector-ALU issue on
// putting planes back together than the matrix pipe spends multiplying.
//
// Accumulators start at 2^30 instead of the exact 128*rowsum: a pair sum q0 + 256 q1 then sits at 2^30 +- 2^27 whatever the
// signs of the digits, everything below is unsigned or a difference of two such sums, and what the biases add up to is a
// constant per output that the generator has pushed through the linear maps into the S-box addends (POSEIDON_BLK_C0)
// and into round 26's constants (POSEIDON_BLK_RC26).
//
// Operand A of every (row set, K group) lives in LDS, [operand][lane] (27 KB per workgroup, poseidon_blk_load()).
#pragma once
#include "gl_field.cuh"
#include "poseidon_mds_mfma.cuh"

namespace gl {

#define POSEIDON_GL_RC_QUAL static __device__
#include "poseidon_gl_blocks.inc"
#undef POSEIDON_GL_RC_QUAL

// (tools/sbox_bench.hip builds an experiment with the matrix instructions replaced by two vector operations: how much of the block is matrix-pipe time)
#ifdef PBLK_MFMA
#define PBLK_ASM_CHAIN 0
#else
#define PBLK_MFMA(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0)
#endif

// Accumulator chains with the 2^30 bias as an INLINE CONSTANT of the first matrix instruction (srcC = 2.0, whose bit pattern is
// 0x40000000) instead of sixteen registers holding it: hipcc has no way to say that, so the whole chain pair is one asm statement,
// which then also has to keep the hazards the compiler would: two wait states after the vector writes of B, and the 8-pass
// result's 11 before anything reads it (the numbers hipcc emits around the builtin form).
#ifndef PBLK_ASM_CHAIN
#define PBLK_ASM_CHAIN 1
#endif
__device__ __forceinline__ void blk_chain3(v16i &L, v16i &H, v4i a0, v4i a1, v4i a2, const v4i *Bl, const v4i *Bh) {
#if PBLK_ASM_CHAIN
    asm("s_nop 1\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %2, %5, 2.0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %2, %8, 2.0\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %3, %6, %0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %3, %9, %1\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %4, %7, %0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %4, %10, %1\n\t"
        "s_nop 10"
        : "=&v"(L), "=&v"(H)
        : "v"(a0), "v"(a1), "v"(a2), "v"(Bl[0]), "v"(Bl[1]), "v"(Bl[2]), "v"(Bh[0]), "v"(Bh[1]), "v"(Bh[2]));
#else
    v16i bias;
#pragma unroll
    for (int i = 0; i < 16; i++) bias[i] = 0x40000000;
    L = PBLK_MFMA(a0, Bl[0], bias); H = PBLK_MFMA(a0, Bh[0], bias);
    L = PBLK_MFMA(a1, Bl[1], L);    H = PBLK_MFMA(a1, Bh[1], H);
    L = PBLK_MFMA(a2, Bl[2], L);    H = PBLK_MFMA(a2, Bh[2], H);
#endif
}
__device__ __forceinline__ void blk_chain4(v16i &L, v16i &H, v4i a0, v4i a1, v4i a2, v4i a3, const v4i *Bl, const v4i *Bh, v4i Dl, v4i Dh) {
#if PBLK_ASM_CHAIN
    asm("s_nop 1\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %2, %6, 2.0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %2, %9, 2.0\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %3, %7, %0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %3, %10, %1\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %4, %8, %0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %4, %11, %1\n\t"
        "v_mfma_i32_32x32x32_i8 %0, %5, %12, %0\n\t"
        "v_mfma_i32_32x32x32_i8 %1, %5, %13, %1\n\t"
        "s_nop 10"
        : "=&v"(L), "=&v"(H)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(Bl[0]), "v"(Bl[1]), "v"(Bl[2]), "v"(Bh[0]), "v"(Bh[1]), "v"(Bh[2]), "v"(Dl), "v"(Dh));
#else
    blk_chain3(L, H, a0, a1, a2, Bl, Bh);
    L = PBLK_MFMA(a3, Dl, L); H = PBLK_MFMA(a3, Dh, H);
#endif
}

// PBLK_TIMING (never defined in the library build; wrong results): bit 0 the layers do not add their constants, bit 1 every biased single layer
// runs a fourth K group -- the price and the prize of moving the constants into the matrix product (measured: see DESIGN section 4)
#ifndef PBLK_TIMING
#define PBLK_TIMING 0
#endif
static constexpr int BLK_A_DWORDS = POSEIDON_BLK_OPERANDS * 64 * 4;

// all threads of the workgroup; returns this lane's column of the operand table
__device__ __forceinline__ const v4i *poseidon_blk_load(v4i *lds) {
    const u32 tid = threadIdx.x + threadIdx.y * blockDim.x, nt = blockDim.x * blockDim.y;
    const v4i *src = (const v4i *)POSEIDON_BLK_A;
    for (u32 i = tid; i < POSEIDON_BLK_OPERANDS * 64; i += nt) lds[i] = src[i];
    __syncthreads();
    return lds + (tid & 63);
}

__device__ __forceinline__ u32 blk_pair(int q0, int q1) { return (u32)q0 + ((u32)q1 << 8); }

// X (signed) + 2^32 Y -> a 64-bit representative mod p, exact (the S-box inputs: three per block)
__device__ __forceinline__ u64 blk_combine_exact(u64 X, u64 Y) {
    const u64 tt = (u64)(u32)(Y >> 32) * EPS + X;              // >= 0 and < 2^56 by the generator's bounds
    u32 th;
    const bool c = __builtin_uadd_overflow((u32)(tt >> 32), (u32)Y, &th);
    const u64 r = ((u64)th << 32) | (u32)tt;
    return c ? r + EPS : r;
}

// A single MDS layer with its operands read from the LDS table (nothing of it stays in registers between layers).
// EXACT = false: biased accumulators like the blocks (the constant they leave is folded into the next S-box addends by the
// generator); EXACT = true: accumulators start at 128*rowsum, the outputs are the true field elements (the permutation's last
// layer).  Same values as mds_layer_mfma(), which keeps its operands in registers.
// lc (biased form only): the twelve addends of whatever S-boxes come next (POSEIDON_BLK_LC), added here as two 64-bit additions
// on X and Y -- inside the S-box the same addition is an add, an add of 2^32-1, a compare and two selects.
// NSETS: how many of the three row sets (outputs 0-3, 4-7, 8-11) are computed: a sponge and a tree node keep only the first four outputs of a
// permutation (linearhash.js:29-40, glwasm.js:1220-1254), so its LAST layer needs one set -- six matrix instructions and four recombinations
// instead of eighteen and twelve; outputs 4 * NSETS .. 11 are left as they were.
template <bool EXACT, int NSETS = 3>
__device__ __forceinline__ void mds_layer_lds(u64 st[12], const MdsMfma &m, const u64 *__restrict__ lc = nullptr) {
    const v4i *__restrict__ A = m.blkA + POSEIDON_BLK_LAYER_OPERAND * 64;
    v4i Bl[3], Bh[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            Bl[t][e] = (int)((u32)st[4 * t + e] ^ 0x80808080u);
            Bh[t][e] = (int)((u32)(st[4 * t + e] >> 32) ^ 0x80808080u);
        }
    const v4i a0 = A[0], a1 = A[64], a2 = A[128], a00 = A[192];
    u64 cm[12], any = 0;
#pragma unroll
    for (int s = 0; s < NSETS; s++) {
        v16i L, H;
        // row set s multiplies K group t by the circulant's block (t - s) mod 3; row 0's own block carries the diagonal's 8
        const v4i r0 = s == 0 ? a00 : (s == 1 ? a2 : a1), r1 = s == 0 ? a1 : (s == 1 ? a0 : a2), r2 = s == 0 ? a2 : (s == 1 ? a1 : a0);
        if (EXACT) {
            v16i init;
#pragma unroll
            for (int i = 0; i < 16; i++) init[i] = 128 * 256;
            L = PBLK_MFMA(r0, Bl[0], init); H = PBLK_MFMA(r0, Bh[0], init);
            L = PBLK_MFMA(r1, Bl[1], L);    H = PBLK_MFMA(r1, Bh[1], H);
            L = PBLK_MFMA(r2, Bl[2], L);    H = PBLK_MFMA(r2, Bh[2], H);
        }
#if PBLK_TIMING & 2
        else blk_chain4(L, H, r0, r1, r2, A[(256 + 64 * s)], Bl, Bh, A[320 + 64 * s], A[448 + 64 * s]);
#else
        else blk_chain3(L, H, r0, r1, r2, Bl, Bh);
#endif
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            u32 xa = blk_pair(L[4 * ii], L[4 * ii + 1]), xc = blk_pair(L[4 * ii + 2], L[4 * ii + 3]);
            u32 ya = blk_pair(H[4 * ii], H[4 * ii + 1]), yc = blk_pair(H[4 * ii + 2], H[4 * ii + 3]);
            if (EXACT && s == 0 && ii == 0) {       // row 0's sum is 264, not 256: see mds_layer_mfma
                xa += 1024u * 257u; xc += 1024u * 257u; ya += 1024u * 257u; yc += 1024u * 257u;
            }
            u64 X = (u64)xc * m.sh16 + xa;
            u64 Y = (u64)yc * m.sh16 + ya;
#if !(PBLK_TIMING & 1)
            if (!EXACT) { const u64 c = lc[4 * s + ii]; X += (u32)c; Y += c >> 32; }          // X, Y < 2^49 + 2^32
#endif
            const u64 tt = (u64)(u32)(Y >> 32) * EPS + X;
            u32 th;
            asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(th), "=s"(cm[4 * s + ii]) : "v"((u32)(tt >> 32)), "v"((u32)Y));
            any |= cm[4 * s + ii];
            st[4 * s + ii] = ((u64)th << 32) | (u32)tt;
        }
    }
    if (__builtin_expect(any != 0, 0)) {
#pragma unroll
        for (int i = 0; i < 4 * NSETS; i++) {
            u32 e;
            asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e) : "s"(cm[i]));
            st[i] += e;
        }
    }
}

// one block of four partial rounds; c = the block's four S-box addends, of which c[0] is already in st[0] (whoever produced st[]
// added it), and c[4] = the addend of the S-box that follows the block, added to output 0 here; A = this lane's column of the
// LDS operand table.  st[] any representatives in, any representatives out.  Like mds_layer_mfma: the whole wave must arrive together.
__device__ __forceinline__ void poseidon_partial_block(u64 st[12], const u64 *__restrict__ c, const v4i *__restrict__ A, const MdsMfma &m) {
    st[0] = sbox_one(st[0]);
    v4i Bl[3], Bh[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            Bl[t][e] = (int)((u32)st[4 * t + e] ^ 0x80808080u);
            Bh[t][e] = (int)((u32)(st[4 * t + e] >> 32) ^ 0x80808080u);
        }
    // ---- the three later S-box inputs: row set 6 ----
    u64 d[4];
    {
        v16i L, H;
        blk_chain3(L, H, A[18 * 64], A[19 * 64], A[20 * 64], Bl, Bh);
        // t_1: planes 0..3 at rows 0..3
        u64 X = (u64)blk_pair(L[2], L[3]) * m.sh16 + blk_pair(L[0], L[1]);
        u64 Y = (u64)blk_pair(H[2], H[3]) * m.sh16 + blk_pair(H[0], H[1]);
        u64 t = canon(blk_combine_exact(X, Y));
        d[1] = sub(sbox_one(add_lazy_canon(t, c[1])), t);             // lazy - canonical: one borrow at most
        // t_2: planes 0..4 at rows 4..8, + d_1 (M)[0][0]
        {
            const u32 c0 = blk_pair(L[4], L[5]), c1 = blk_pair(L[6], L[7]), c4 = (u32)H[8];
            const u32 e2 = (u32)L[8] + blk_pair(H[4], H[5]) + c4, c3 = blk_pair(H[6], H[7]);
            X = (u64)c1 * m.sh16 + (u64)(int64_t)(int)(c0 - c4);
            Y = (u64)c3 * m.sh16 + e2;
            X += (u64)(u32)d[1] * POSEIDON_BLK_T1;
            Y += (u64)(u32)(d[1] >> 32) * POSEIDON_BLK_T1;
        }
        t = canon(blk_combine_exact(X, Y));
        d[2] = sub(sbox_one(add_lazy_canon(t, c[2])), t);
        // t_3: planes 0..5 at rows 9..14, + d_1 (M^2)[0][0] + d_2 (M)[0][0]
        {
            const u32 c0 = blk_pair(L[9], L[10]), c1 = blk_pair(L[11], L[12]), c4 = blk_pair(H[13], H[14]);
            const u32 e2 = blk_pair(L[13], L[14]) + blk_pair(H[9], H[10]) + c4, c3 = blk_pair(H[11], H[12]);
            X = (u64)c1 * m.sh16 + (u64)(int64_t)(int)(c0 - c4);
            Y = (u64)c3 * m.sh16 + e2;
            X += (u64)(u32)d[1] * POSEIDON_BLK_T2 + (u64)(u32)d[2] * POSEIDON_BLK_T1;
            Y += (u64)(u32)(d[1] >> 32) * POSEIDON_BLK_T2 + (u64)(u32)(d[2] >> 32) * POSEIDON_BLK_T1;
        }
        t = canon(blk_combine_exact(X, Y));
        d[3] = sub(sbox_one(add_lazy_canon(t, c[3])), t);
    }
    v4i Dl, Dh;
#pragma unroll
    for (int j = 0; j < 3; j++) { Dl[j] = (int)((u32)d[j + 1] ^ 0x80808080u); Dh[j] = (int)((u32)(d[j + 1] >> 32) ^ 0x80808080u); }
    Dl[3] = Dh[3] = 0;
    // ---- the new state: row sets 0..5, two elements each ----
    // (issuing row set s+1's matrix instructions between the vector instructions that put row set s back together -- by source
    // order or forced with sched_group_barrier -- was measured: +1 % / -0.5 %, 29 more registers; the other waves of the SIMD
    // already fill the matrix pipe's shadow, tools/mfma_overlap.hip)
    u64 cm[12], any = 0;
    auto rowset = [&](int s, v16i &L, v16i &H) {
        blk_chain4(L, H, A[(3 * s) * 64], A[(3 * s + 1) * 64], A[(3 * s + 2) * 64], A[(21 + s) * 64], Bl, Bh, Dl, Dh);
    };
    auto recombine = [&](int s, const v16i &L, const v16i &H) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int o = 8 * ii;
            // value = c0 + 2^16 c1 + 2^32 c2 + 2^48 c3 + 2^64 c4 + 2^80 c5, pair sums of the low word at c0..c3, of the high word at c2..c5;
            // 2^64 = 2^32 - 1, 2^80 = 2^48 - 2^16 (mod p):  X = (c0 - c4) + 2^16 (c1 - c5),  Y = (c2 + c4) + 2^16 (c3 + c5)
            const u32 al0 = blk_pair(L[o], L[o + 1]), al1 = blk_pair(L[o + 2], L[o + 3]), al2 = blk_pair(L[o + 4], L[o + 5]);
            const u32 ah0 = blk_pair(H[o], H[o + 1]), ah1 = blk_pair(H[o + 2], H[o + 3]), ah2 = blk_pair(H[o + 4], H[o + 5]);
            const u32 e2 = al2 + ah0 + ah2, e3 = (u32)L[o + 6] + ah1 + (u32)H[o + 6];
            const int d0 = (int)(al0 - ah2), d1 = (int)(al1 - (u32)H[o + 6]);
            u64 X = (u64)((int64_t)d1 * (int)m.sh16 + d0);
            u64 Y = (u64)e3 * m.sh16 + e2;
            if (s == 0 && ii == 0) { X += (u32)c[4]; Y += c[4] >> 32; }
            const u64 tt = (u64)(u32)(Y >> 32) * EPS + X;              // 0 < tt < 2^49: Y_hi (2^32-1) >= 2^46 outweighs |X| < 2^45 (+ 2^32 with the addend)
            u32 th;
            asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(th), "=s"(cm[2 * s + ii]) : "v"((u32)(tt >> 32)), "v"((u32)Y));
            any |= cm[2 * s + ii];
            st[2 * s + ii] = ((u64)th << 32) | (u32)tt;
        }
    };
#pragma unroll
    for (int s = 0; s < 6; s++) {
        v16i L, H;
        rowset(s, L, H);
        recombine(s, L, H);
    }
    if (__builtin_expect(any != 0, 0)) {                       // the last addition wrapped (probability ~2^-15 per element)
#pragma unroll
        for (int i = 0; i < 12; i++) {
            u32 e;
            asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e) : "s"(cm[i]));
            st[i] += e;
        }
    }
}

}  // namespace gl
