// Poseidon-12 MDS layer on the gfx950 matrix cores (v_mfma_i32_32x32x32_i8), one permutation per lane.
//
// The MDS product out_i = sum_j M[i][j]*st_j (glwasm.js:428-440: M = circ(17,15,41,16,2,28,13,13,39,18,34,20) +
// diag(8,0,..)) has 6-bit coefficients, so with st_j cut into bytes, out_i = sum_b 2^(8b) * P_ib with
// P_ib = sum_j M[i][j]*byte_b(st_j) < 2^17: an i8 x i8 -> i32 dot product.  It is integer-issue bound on the
// vector ALU (288 v_mad_u64_u32 per layer), while the matrix pipe is idle; here the matrix pipe does it.
//
// How one MFMA serves 64 independent permutations.  D(32x32) = A(32x32) * B(32x32):
//   * operand B, lane l: column n = l%32, K-group g = l/32, 16 bytes.  A lane feeds its OWN data: four 32-bit
//     state words (4 elements x 4 bytes) = 16 K-slots, no byte shuffling at all.
//   * result D, lane l = (n, h = l/32) receives column n, rows {8q + 4h + r : q,r in 0..3} in 16 VGPRs.
//   * A is built so that a row owned by lane-half h has non-zero coefficients only in K-group h.  Then lane
//     (n, h) gets M applied to ITS OWN 16 bytes and nothing of lane (n, 1-h): both halves of the wave use the
//     same instruction, each on its own data (a block-diagonal A).
//   * the 16 result VGPRs of a lane are 4 output rows x 4 byte planes: row r = 4*ii + b of a lane's set means
//     output element 4s+ii, byte plane b; the coefficient of K-slot (element e, byte b') is M[4s+ii][4t+e] when
//     b' == b and 0 otherwise.  Three chained MFMAs (t = 0,1,2: elements 4t..4t+3) finish 4 output rows x 4 planes;
//     s = 0,1,2 and the low / high state words make 18 MFMAs per layer.
//   * i8 operands are signed: bytes are fed as (u ^ 0x80) = u - 128 and the accumulator starts at
//     128 * rowsum(M) = 128 * 256 (a constant vector; row 0, whose sum is 264, is fixed up in the 32-bit stage), so every P_ib is the exact non-negative sum.
// The circulant makes A depend on (t - s) mod 3 only, except M[0][0]'s +8: 4 constant operands (16 VGPRs).
#pragma once
#include "gl_field.cuh"

namespace gl {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct MdsMfma {
    v4i A[3];       // (t - s) mod 3 = 0, 1, 2
    v4i A00;        // s = 0, t = 0
    v16i C;         // 128 * 256 in every row
    u32 sh16;       // 65536, kept opaque so that x*65536 + y stays ONE v_mad_u64_u32 (not shift + zero-extend + add)
    const v4i *blkA; // this lane's column of the blocked partial rounds' operand table in LDS (poseidon_blocks.cuh)
};

__device__ inline void mds_mfma_init(MdsMfma &m) {
    constexpr u32 MC[12] = { 17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20 };
    const u32 lane = (threadIdx.x + threadIdx.y * blockDim.x) & 63;
    const u32 i = lane & 31, g = lane >> 5;           // A row, K-group
    const bool owned = ((i >> 2) & 1) == g;           // D row i lands in lane-half (i/4)%2
    const u32 ii = i >> 3, b = i & 3;                 // result VGPR 4*ii + b of that half
    for (int d = 0; d < 3; d++)
        for (int e = 0; e < 4; e++) {
            u32 c = 0;
            for (int k = 0; k < 12; k++) if ((u32)k == (4u * d + e + 12u - ii) % 12u) c = MC[k];
            m.A[d][e] = owned ? (int)(c << (8 * b)) : 0;
        }
    m.A00 = m.A[0];
    if (owned && ii == 0) m.A00[0] += (int)(8u << (8 * b));
    for (int r = 0; r < 16; r++) m.C[r] = 128 * 256;
    m.sh16 = 65536u;
    asm volatile("" : "+s"(m.sh16));
}

// st[] lazy in, lazy out; identical values (mod p) to mds_layer()
__device__ __forceinline__ void mds_layer_mfma(u64 st[12], const MdsMfma &m) {
    v4i Bl[3], Bh[3];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            Bl[t][e] = (int)((u32)st[4 * t + e] ^ 0x80808080u);
            Bh[t][e] = (int)((u32)(st[4 * t + e] >> 32) ^ 0x80808080u);
        }
    // the last addition of an element's recombination carries out of 64 bits with probability ~2^-20 (see below): the carry
    // masks (lane masks in SGPRs, straight from the add) are only OR-ed together here and patched after the layer in a
    // branch that is uniform for the wave and rarely taken -- three vector instructions per element less than selecting
    // between r and r + 2^32-1 every time
    u64 cm[12], any = 0;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        v16i L = m.C, H = m.C;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const v4i a = (s == 0 && t == 0) ? m.A00 : m.A[(t - s + 3) % 3];
            L = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bl[t], L, 0, 0, 0);
            H = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bh[t], H, 0, 0, 0);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            // P_b <= 255*264 < 2^17: pairs fit 32 bits, X = sum_{b<4} P_b 2^(8b) and Y (b >= 4) are < 2^43
            u32 xa = (u32)L[4 * ii] + ((u32)L[4 * ii + 1] << 8), xc = (u32)L[4 * ii + 2] + ((u32)L[4 * ii + 3] << 8);
            u32 ya = (u32)H[4 * ii] + ((u32)H[4 * ii + 1] << 8), yc = (u32)H[4 * ii + 2] + ((u32)H[4 * ii + 3] << 8);
            if (s == 0 && ii == 0) {
                // row 0: rowsum is 264, so its accumulators are P_b - 1024 and may be NEGATIVE; the pair sums are exact
                // mod 2^32, and adding the missing 1024*(1 + 2^8) there makes them the true non-negative values
                xa += 1024u * 257u; xc += 1024u * 257u; ya += 1024u * 257u; yc += 1024u * 257u;
            }
            const u64 X = (u64)xc * m.sh16 + xa;
            const u64 Y = (u64)yc * m.sh16 + ya;
            // X + Y*2^32 = X + Y_hi*2^64 + Y_lo*2^32 = (X + Y_hi*(2^32-1)) + Y_lo*2^32   (mod p); the bracket is < 2^44, so the
            // high word of the sum wraps only when Y_lo >= 2^32 - 2^12
            const u64 tt = (u64)(u32)(Y >> 32) * EPS + X;
            u32 th;
            asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(th), "=s"(cm[4 * s + ii]) : "v"((u32)(tt >> 32)), "v"((u32)Y));
            any |= cm[4 * s + ii];
            st[4 * s + ii] = ((u64)th << 32) | (u32)tt;
        }
    }
    if (__builtin_expect(any != 0, 0)) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            u32 e;
            asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e) : "s"(cm[i]));
            st[i] += e;                                        // wrapped once: the value is small, + 2^32-1 cannot wrap again
        }
    }
}

}  // namespace gl
