// EXPERIMENT (tools/mfma_dft16.hip; not part of the library -- profiles/r05_mfma_dft16_microbench.txt says why).
// 16-point DFTs over Goldilocks on the gfx950 matrix cores (v_mfma_i32_32x32x32_i8): the register step of the transform tiles
// (ntt.hip: dif_step<4> / dit_step<4>) as a constant-matrix product, for the whole wave at once.
//
// The vector form (gl_fermat.cuh) keeps a lane's 16 values as 128-bit integers of Z/(2^96+1) and pays 4-word additions and funnel
// shifts: 6.6 instructions per element and stage, and 3 more for the reduction.  Here out_i = sum_j W[i][j] x_j with CONSTANT W goes
// to the matrix pipe, the way bn_mfma.cuh does it for BN254:
//     c[i][j][b'] = W[i][j] * 256^b' mod p        a plain integer, written in eight signed base-256 digits d_b (p - c negated when c > 0x7f7f..7f)
//     D[(i,b)]    = beta_b + sum_{j,b'} d_b(c[i][j][b']) * (byte_b'(x_j) - 128)           i8 x i8 -> i32, K = 16 x 8 = 128, 128 rows (i,b)
//     V_i         = sum_b 256^b D[(i,b)]   =   out_i  -  128 S sum_j W[i][j]  +  sum_b beta_b 256^b      (mod p),   S = sum_b 256^b
// The reduction mod p is in the constants.  Two offsets come back for free: sum_j W[i][j] = 0 for every row of a DFT matrix but the
// all-ones row (there it is 16: K0 = 2048 S is added to that one output), and the accumulators' start values beta_b = 2^21 + ku_b are
// chosen with sum_b beta_b 256^b = 0 mod p (every D is then in [0, 2^22 + 256)).  What the vector ALU still does per element: the
// operand bytes' sign flip, and the recombination of eight 22-bit planes into one lazy 64-bit value (eight instructions).
//
// Who holds what.  One matrix instruction serves 32 sub-transforms (columns n); a wave takes 64 of them in two groups.  Lane
// (n, h = lane / 32) holds values 8h .. 8h+7 of sub-transform n of group 0 and of group 1 -- going in and coming out -- so a tile
// kernel reads and writes LDS with that ownership and nothing crosses lanes.  (One sub-transform per lane, as the vector form has it,
// needs 4 v_permlane32_swap per value to get here and back: measured, they cost more than the whole recombination --
// profiles/r05_mfma_dft16_microbench.txt.)
//   * B, step t < 4, lane (n, g = lane / 32): 16 bytes = values 8g + 2t, 8g + 2t + 1 of sub-transform n (sign bits flipped).
//   * A, tile (mt, t), lane (rho = lane % 32, g): row rho of row tile mt, 16 digits.  Result lane (n, h) receives rows 8q + 4h + r in
//     register 4q + r; rows are placed so that this is output 8h + 2mt + f, plane b at register 8f + b:
//         r = rho % 4, h = (rho / 4) % 2, q = rho / 8;   f = q / 2, b = 4 (q % 2) + r.
//     16 tiles of 1 KB in lane order, in LDS (one table per direction and order).
// Integer statement of exactly these steps against the definition: tools/mfma_dft16.hip (device check) and tests.
#pragma once
#include "gl_field.cuh"
#ifndef DFT16_ABLATE
#define DFT16_ABLATE 0
#endif
#ifndef DFT16_CHAINS
#define DFT16_CHAINS 1          // row tiles whose accumulator chains run side by side (1, 2 or 4: no faster, tools/mfma_dft16.hip)
#endif

namespace gl {
namespace dft16 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr u32 TABLE_WORDS = 16 * 64 * 4;        // u32 words of one operand table (16 KB)
constexpr u32 BIAS0 = 1u << 21;

struct Consts {
    const v4i *A;       // LDS table + lane
    v16i C;             // beta_b at 8f + b
    u64 k0sel;          // K0 on lanes < 32, 0 above
    u32 sh16;           // 65536, opaque (x * 65536 + y stays one v_mad_u64_u32)
#ifdef DFT16_AREG
    v4i Areg[16];
#endif
};

// eight planes -> lazy value; the last addition's carry (probability 2^-16) is left in cm for the caller's rare branch
__device__ __forceinline__ u64 recombine(const v16i &acc, int f, u32 sh16, u64 &cm) {
    const u32 xa = (u32)acc[8 * f] + ((u32)acc[8 * f + 1] << 8), xc = (u32)acc[8 * f + 2] + ((u32)acc[8 * f + 3] << 8);
    const u32 ya = (u32)acc[8 * f + 4] + ((u32)acc[8 * f + 5] << 8), yc = (u32)acc[8 * f + 6] + ((u32)acc[8 * f + 7] << 8);
    const u64 X = (u64)xc * sh16 + xa;                       // < 2^47
    const u64 Y = (u64)yc * sh16 + ya;
    // X + Y 2^32 = (X + Y_hi (2^32-1)) + Y_lo 2^32  (mod p); the bracket is < 2^48
    const u64 tt = (u64)(u32)(Y >> 32) * EPS + X;
    u32 th;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(th), "=s"(cm) : "v"((u32)(tt >> 32)), "v"((u32)Y));
    return ((u64)th << 32) | (u32)tt;
}

// x[0..7]: values 8h..8h+7 of sub-transform n of group 0, x[8..15]: the same of group 1, for lane (n = lane % 32, h = lane / 32); any
// representatives in, lazy out, same places: out[i] = sum_j W[i][j] in[j] for the table's W.
// Every lane of the wave must be here (the operands of all 64 lanes feed each matrix instruction).
__device__ __forceinline__ void run(u64 x[16], const Consts &k) {
    v4i B0[4], B1[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int j = 2 * t + e;
            B0[t][2 * e] = (int)((u32)x[j] ^ 0x80808080u); B0[t][2 * e + 1] = (int)((u32)(x[j] >> 32) ^ 0x80808080u);
            B1[t][2 * e] = (int)((u32)x[8 + j] ^ 0x80808080u); B1[t][2 * e + 1] = (int)((u32)(x[8 + j] >> 32) ^ 0x80808080u);
        }
    u64 cm[16], any = 0;
    // DFT16_CHAINS row tiles at a time: a matrix instruction that accumulates onto the one before it waits for it IN the pipe (the
    // other waves' matrix instructions wait behind it); with 2 row tiles x 2 groups in flight an accumulator comes back every fourth
#pragma unroll
    for (int m0 = 0; m0 < 4; m0 += DFT16_CHAINS) {
        v16i a0[DFT16_CHAINS], a1[DFT16_CHAINS];
#pragma unroll
        for (int c = 0; c < DFT16_CHAINS; c++) a0[c] = a1[c] = k.C;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int c = 0; c < DFT16_CHAINS; c++) {
#ifdef DFT16_AREG                 // (experiment: the operand table in registers)
                const v4i a = k.Areg[(m0 + c) * 4 + t];
#else
                const v4i a = k.A[((m0 + c) * 4 + t) * 64];
#endif
#if DFT16_ABLATE == 3           // (tools/mfma_dft16.hip: the vector work alone, one stand-in instruction per matrix instruction; wrong results)
                a0[c][t] += a[0] ^ B0[t][0]; a1[c][t + 8] += a[1] ^ B1[t][1];
#else
                a0[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0[t], a0[c], 0, 0, 0);
                a1[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1[t], a1[c], 0, 0, 0);
#endif
            }
#pragma unroll
        for (int c = 0; c < DFT16_CHAINS; c++) {
            const int mt = m0 + c;
#if DFT16_ABLATE == 2           // (the matrix instructions alone)
            for (int f = 0; f < 2; f++) { x[2 * mt + f] = (u32)a0[c][8 * f] | ((u64)(u32)a0[c][8 * f + 7] << 32); x[8 + 2 * mt + f] = (u32)a1[c][8 * f] | ((u64)(u32)a1[c][8 * f + 7] << 32); cm[4 * mt + 2 * f] = cm[4 * mt + 2 * f + 1] = 0; }
#else
#pragma unroll
            for (int f = 0; f < 2; f++) {
                x[2 * mt + f] = recombine(a0[c], f, k.sh16, cm[4 * mt + 2 * f]);
                x[8 + 2 * mt + f] = recombine(a1[c], f, k.sh16, cm[4 * mt + 2 * f + 1]);
                any |= cm[4 * mt + 2 * f] | cm[4 * mt + 2 * f + 1];
            }
#endif
        }
    }
    if (__builtin_expect(any != 0, 0)) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            u32 e0, e1;
            asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e0) : "s"(cm[2 * i]));
            asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e1) : "s"(cm[2 * i + 1]));
            x[i] += e0;                                         // wrapped once: the value is below 2^48, + 2^32-1 cannot wrap again
            x[8 + i] += e1;
        }
    }
    x[0] = add_lazy_canon(x[0], k.k0sel);                       // the all-ones row (output 0: lanes < 32 of both groups)
    x[8] = add_lazy_canon(x[8], k.k0sel);
}

// every thread of the workgroup, once: the table into LDS, the lane's constants
__device__ __forceinline__ Consts init(u32 *ldsTable, const u32 *__restrict__ gTable, const u64 *__restrict__ gConsts) {
    const u32 tid = threadIdx.x + threadIdx.y * blockDim.x, nth = blockDim.x * blockDim.y;
    for (u32 i = tid; i < TABLE_WORDS; i += nth) ldsTable[i] = gTable[i];
    __syncthreads();
    Consts k;
    k.A = (const v4i *)ldsTable + (tid & 63);
#pragma unroll
    for (int v = 0; v < 16; v++) k.C[v] = (int)gConsts[v & 7];
    k.k0sel = (tid & 32) ? 0 : gConsts[8];
    k.sh16 = 65536u;
    asm volatile("" : "+s"(k.sh16));
#ifdef DFT16_AREG
    for (int i = 0; i < 16; i++) k.Areg[i] = k.A[i * 64];
#endif
    return k;
}

}  // namespace dft16
}  // namespace gl

// ---- host: the operand table of one 16-point matrix -----------------------------------------------------------------------------
#include <vector>
namespace gl_dft16_host {
typedef unsigned __int128 u128;
static const uint64_t HP = 0xFFFFFFFF00000001ull;
inline uint64_t hmul(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % HP); }
inline uint64_t hpow(uint64_t a, uint64_t e) { uint64_t r = 1; while (e) { if (e & 1) r = hmul(r, a); a = hmul(a, a); e >>= 1; } return r; }
// eight signed base-256 digits of c (mod p): of c itself when it fits, of c - p otherwise
inline void digits(uint64_t c, int d[8]) {
    __int128 v = c <= 0x7F7F7F7F7F7F7F7Full ? (__int128)c : (__int128)c - (__int128)HP;
    for (int b = 0; b < 8; b++) {
        int lo = (int)(v & 255);
        if (lo >= 128) lo -= 256;
        d[b] = lo;
        v = (v - lo) >> 8;
    }
    // v == 0 here: the representative lies in [-0x8080..80, 0x7f7f..7f]
}
// W[i][j] = w^(fq[i] * tm[j]); table: 16 tiles (mt, t) x 64 lanes x 4 words; consts: beta_0..7, K0
inline void build(uint64_t w, const int fq[16], const int tm[16], std::vector<uint32_t> &table, uint64_t consts[9]) {
    table.assign(16 * 64 * 4, 0);
    for (int mt = 0; mt < 4; mt++)
        for (int t = 0; t < 4; t++)
            for (int lane = 0; lane < 64; lane++) {
                const int rho = lane & 31, g = lane >> 5;
                const int r = rho & 3, h = (rho >> 2) & 1, q = rho >> 3, f = q >> 1, b = 4 * (q & 1) + r;
                const int i = 8 * h + 2 * mt + f;
                uint8_t bytes[16];
                for (int e = 0; e < 2; e++)
                    for (int bp = 0; bp < 8; bp++) {
                        const int j = 8 * g + 2 * t + e;
                        const uint64_t c = hmul(hpow(w, (uint64_t)(fq[i] * tm[j])), hpow(256, bp));
                        int d[8]; digits(c, d);
                        bytes[8 * e + bp] = (uint8_t)(int8_t)d[b];
                    }
                uint32_t *dst = &table[((mt * 4 + t) * 64 + lane) * 4];
                for (int wd = 0; wd < 4; wd++) dst[wd] = bytes[4 * wd] | (bytes[4 * wd + 1] << 8) | (bytes[4 * wd + 2] << 16) | ((uint32_t)bytes[4 * wd + 3] << 24);
            }
    const uint64_t S = 0x0101010101010101ull;
    const u128 T0 = (u128)(1u << 21) * S;
    const u128 m = (T0 + HP - 1) / HP;
    const uint64_t R = (uint64_t)(m * HP - T0);                 // < p
    for (int b = 0; b < 8; b++) consts[b] = (1u << 21) + ((R >> (8 * b)) & 255);
    consts[8] = hmul(2048, S % HP);
}
}  // namespace gl_dft16_host
