// Extension-weighted row / column sums of a row-major base-field matrix (gfx950).
//
// Two steps of the prover are, mathematically, matrix-vector products with extension-field weights:
//   * the FRI polynomial (src/pil_info/helpers/polynomials/friPolinomial.js:26-50, evaluated row by row through
//     the op-list in src/stark/stark_gen_helpers.js:325): per opening o,  F_o(r) = sum_j (p_j(r) - ev_j) vf2^(n_o-j)
//     = sum_c M[r][c] * W_o[c]  -  K_o         with one extension constant W_o[c] per base column c;
//   * the evaluations (stark_gen_helpers.js:250-264):  ev[c] = sum_k M[k << b][c] * LEv[k].
// Interpreting them through the generic op-list costs one full extension multiplication (9 modular
// multiplications) per term.  Here every term is base x extension and is accumulated LAZILY: the 64-bit base
// value is split into 32-bit halves, each weight component into 22/22/20-bit limbs, and the six partial sums
//   S[h][l] += half_h * limb_l      (< 2^54 each, one v_mad_u64_u32, no reduction for up to 1024 terms)
// are reduced to a field element once per row (or once per 1024 rows).  Same field values, 6 mads per term
// component instead of a 71-cycle modular multiplication plus a 25-cycle modular addition.
#include "common.h"
#include "gl_field.cuh"
#include <vector>
#include <string.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

using namespace gl;

namespace {

constexpr u32 LIMB_BITS = 22;
constexpr u32 LIMB_MASK = (1u << LIMB_BITS) - 1;

// S[0][0] + S[0][1] 2^22 + S[0][2] 2^44 + S[1][0] 2^32 + S[1][1] 2^54 + S[1][2] 2^76   (mod p), canonical
__device__ __forceinline__ u64 fold6(const u64 S[6]) {
    // constants 2^k mod p are plain shifts while k < 64; 2^76 = 2^12 * 2^64 = 2^12 * (2^32 - 1)
    u64 r = canon(S[0]);
    r = add(r, mul(S[1], 1ull << 22));
    r = add(r, mul(S[2], 1ull << 44));
    r = add(r, mul(S[3], 1ull << 32));
    r = add(r, mul(S[4], 1ull << 54));
    r = add(r, mul(S[5], ((1ull << 32) - 1) << 12));
    return r;
}

struct RowsDotParams {
    const u64 *buf; u64 width; u64 nRows;       // width: columns of THIS launch's window [col0, col0 + width) -- at most 1024 (the unreduced sums hold 1024 terms)
    u64 stride, col0;                           // row length of the matrix, first column of the window
    const u32 *coefLimbs;       // [nOut][width][3 comps][3 limbs]
    u32 nOut;
    u64 *acc;                   // [nRows][nOut][3]
    u32 accumulate;
};

// lanes <-> rows; the 64 x CW tile of a wave is staged through LDS with coalesced row-segment loads
template <int NOUT>
__global__ void __launch_bounds__(256) rows_dot_kernel(RowsDotParams P) {
    constexpr u32 CW = 16, LD = CW + 1;
    __shared__ u64 tile[4][64 * LD];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u64 *T = tile[wave];
    const u64 row0 = ((u64)blockIdx.x * 4 + wave) * 64;
    if (row0 >= P.nRows) return;
    u64 S[NOUT][3][6];
#pragma unroll
    for (int o = 0; o < NOUT; o++)
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 6; i++) S[o][k][i] = 0;
    const u64 myRow = row0 + lane;
    // element e = lane + 64 i of a tile -> (row e >> 4, col e & 15): coalesced 128-byte row segments.  The next tile is
    // fetched into registers while the current one is consumed from LDS (the mads of a tile take far less time than
    // its loads are in flight, so without the overlap a wave mostly waits).
    u64 nxt[CW];
    // row (lane >> 4) + 4 i, column c0 + (lane & 15): the 16 row offsets do not depend on c0 -- an opaque copy of the first
    // one per tile keeps the compiler from holding all of them (and the LDS offsets) in registers across the column loop,
    // which costs the kernel its fourth wave
    const u32 lr = lane >> 4, lc = lane & 15;
    auto fetch = [&](u64 c0) {
        const u32 cw = (u32)min((u64)CW, P.width - c0);
        u64 off = (row0 + lr) * P.stride + P.col0 + c0 + lc;
        asm volatile("" : "+v"(off));
        const u64 step = 4 * P.stride;
#pragma unroll
        for (u32 i = 0; i < CW; i++) {
            const u64 gr = row0 + lr + 4 * i;
            nxt[i] = (lc < cw && gr < P.nRows) ? P.buf[off + i * step] : 0;
        }
    };
    fetch(0);
    for (u64 c0 = 0; c0 < P.width; c0 += CW) {
        const u32 cw = (u32)min((u64)CW, P.width - c0);
        {
            u32 to = lr * LD + lc;
            asm volatile("" : "+v"(to));
#pragma unroll
            for (u32 i = 0; i < CW; i++) T[to + i * 4 * LD] = nxt[i];
        }
        if (c0 + CW < P.width) fetch(c0 + CW);
        // wave-local hand-off through LDS: every lane of this wave wrote, every lane reads; no other wave involved
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        for (u32 c = 0; c < cw; c++) {
            const u64 p = T[lane * LD + c];
            const u32 p0 = (u32)p, p1 = (u32)(p >> 32);
#pragma unroll
            for (int o = 0; o < NOUT; o++) {
                const u32 *L = P.coefLimbs + (((u64)o * P.stride + P.col0 + c0 + c) * 9);   // wave-uniform -> scalar loads
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const u32 w0 = L[3 * k], w1 = L[3 * k + 1], w2 = L[3 * k + 2];
                    S[o][k][0] += (u64)p0 * w0; S[o][k][1] += (u64)p0 * w1; S[o][k][2] += (u64)p0 * w2;
                    S[o][k][3] += (u64)p1 * w0; S[o][k][4] += (u64)p1 * w1; S[o][k][5] += (u64)p1 * w2;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (myRow >= P.nRows) return;
    u64 *out = P.acc + myRow * (3ull * P.nOut);
#pragma unroll
    for (int o = 0; o < NOUT; o++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            u64 v = fold6(S[o][k]);
            if (P.accumulate) v = add(v, out[3 * o + k]);
            out[3 * o + k] = v;
        }
}

// The same sums for matrices whose rows are long (the 100-column stage matrix): a workgroup takes 64 CONSECUTIVE rows -- one
// contiguous piece of memory, 51 KB at 100 columns -- and streams it into LDS with all 256 threads (consecutive lanes read
// consecutive words, whole cache lines, every DRAM page once), instead of each wave collecting 64 separate 128-byte row
// pieces per column tile (which left the kernel at 2.3 TB/s with 1.7x the bytes fetched).  Lanes <-> rows as before, so the
// weights stay wave-uniform; the four waves split the COLUMNS of the tile, their unreduced sums (< 2^64 by the same 1024-term
// bound) are added in LDS (ds_add_u64), and the 3*NOUT folds are dealt over the waves.  Rows wider than CW_MAX columns are taken
// in column chunks of whole-chunk row segments.
constexpr u32 STREAM_CW_MAX = 128;
template <int NOUT>
__global__ void __launch_bounds__(256) rows_dot_stream_kernel(RowsDotParams P, u32 cw, u32 LD) {
    extern __shared__ u64 sm[];                 // tile [64][LD], then the summed partial sums [NOUT*18][64]
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u64 row0 = (u64)blockIdx.x * 64;
    u64 S[NOUT][3][6];
#pragma unroll
    for (int o = 0; o < NOUT; o++)
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 6; i++) S[o][k][i] = 0;
    constexpr u32 NB = 13;                      // loads of a thread in flight together (two batches cover 64 x 100)
    for (u64 c0 = 0; c0 < P.width; c0 += cw) {
        const u32 w = (u32)min((u64)cw, P.width - c0);
        const u32 total = 64 * w, dq = 256 / w, dr = 256 - dq * w;
        u32 r = tid / w, c = tid - r * w;
        for (u32 u0 = 0; u0 < total; u0 += 256 * NB) {
            u64 v[NB]; u32 at[NB];
#pragma unroll
            for (u32 i = 0; i < NB; i++) {
                const bool in = u0 + 256 * i + tid < total;
                at[i] = in ? r * LD + c : 0xFFFFFFFFu;
                v[i] = (in && row0 + r < P.nRows) ? P.buf[(row0 + r) * P.stride + P.col0 + c0 + c] : 0;
                c += dr; r += dq;
                if (c >= w) { c -= w; r++; }
            }
            if (u0 == 0) __syncthreads();       // the previous chunk's tile has been consumed by every wave
#pragma unroll
            for (u32 i = 0; i < NB; i++) if (at[i] != 0xFFFFFFFFu) sm[at[i]] = v[i];
        }
        __syncthreads();
        const u32 cb = w * wave / 4, ce = w * (wave + 1) / 4;
        for (u32 cc = cb; cc < ce; cc++) {
            const u64 p = sm[lane * LD + cc];
            const u32 p0 = (u32)p, p1 = (u32)(p >> 32);
#pragma unroll
            for (int o = 0; o < NOUT; o++) {
                const u32 *L = P.coefLimbs + (((u64)o * P.stride + P.col0 + c0 + cc) * 9);   // wave-uniform -> scalar loads
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const u32 w0 = L[3 * k], w1 = L[3 * k + 1], w2 = L[3 * k + 2];
                    S[o][k][0] += (u64)p0 * w0; S[o][k][1] += (u64)p0 * w1; S[o][k][2] += (u64)p0 * w2;
                    S[o][k][3] += (u64)p1 * w0; S[o][k][4] += (u64)p1 * w1; S[o][k][5] += (u64)p1 * w2;
                }
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int o = 0; o < NOUT; o++)
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int i = 0; i < 6; i++) sm[((o * 3 + k) * 6 + i) * 64 + lane] = S[o][k][i];
    }
    __syncthreads();
    if (wave != 0) {
#pragma unroll
        for (int o = 0; o < NOUT; o++)
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int i = 0; i < 6; i++)
                    __hip_atomic_fetch_add(&sm[((o * 3 + k) * 6 + i) * 64 + lane], S[o][k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const u64 myRow = row0 + lane;
    if (myRow >= P.nRows) return;
    u64 *out = P.acc + myRow * (3ull * P.nOut);
    for (u32 pi = wave; pi < 3 * NOUT; pi += 4) {
        u64 T6[6];
#pragma unroll
        for (int i = 0; i < 6; i++) T6[i] = sm[(pi * 6 + i) * 64 + lane];
        u64 v = fold6(T6);
        if (P.accumulate) v = add(v, out[pi]);
        out[pi] = v;
    }
}


// ---------------------------------------------------------------- the same sums on the matrix cores
// out[r][o] = sum_c M[r][c] * W[c][o] is a matrix product, and a row of M is already the right operand: 8*width bytes, byte i of
// column c having weight 2^(8i).  With every weight written in SIGNED base-256 digits (d_0..d_8 in [-128,127]: nine of them),
//     out[r][o] = sum_p 2^(8p) G[r][o][p],     G[r][o][p] = sum_c sum_i byte_i(M[r][c]) * d_(p-i)(W[c][o])       (p < 16)
// is an i8 x i8 -> i32 product with K = 8*width and 16 planes per output: v_mfma_i32_32x32x32_i8 with the 32 trace rows of a
// row group as the N dimension (operand B: lane (n, g) feeds 16 consecutive bytes of ITS row, straight from the staged
// tile), and the (output, plane) pairs as the M dimension, ordered so that the 16 result registers of lane (n, h) are the 16
// planes of output 2*tile + h for row n (result rows 8q + 4h + r, see poseidon_mds_mfma.cuh).  Bytes are fed as (b ^ 0x80) = b - 128;
// the missing 128 * sum_i 2^(8i) * sum_c W[c][o] is a constant per output and is added with the plane offset at the end.
// One persistent workgroup of 8 waves per CU walks over tiles of 64 consecutive rows (one contiguous piece of memory, streamed
// into LDS by all 512 threads; the NEXT tile's loads are in flight while this one is multiplied).  The digit matrix (1 KB per
// K-step and M tile: 75 KB at 100 columns) sits in LDS beside the tile for the whole kernel; wave (g, t) -- row group g, M
// tile t -- runs the K-steps of ITS 32 rows x 2 outputs as one accumulator chain and recombines the planes from its own
// registers: no sums cross waves.  Per row 1.2 MFMAs instead of 3600 v_mad_u64_u32: what is left is the streaming read
// (profiles/r02_rows_dot_*.txt).
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr u32 MF_WAVES = 8, MF_ROWS = 64, MF_MAXW = 112;
constexpr int MF_PLANE_OFFSET = 1 << 24;        // every plane sum is > -2^24: the accumulators start at 2^24

constexpr u32 MF_MAXSEG = 4;
struct RowsDotMfmaParams {
    // the K dimension is the concatenation of up to MF_MAXSEG matrices with the same rows (a stage's matrices side by side):
    // segment k holds segW[k] (even) columns and starts at 16-byte unit segU0[k] of the staged row
    const u64 *segBuf[MF_MAXSEG]; u32 segW[MF_MAXSEG], segPitch[MF_MAXSEG], segU0[MF_MAXSEG], nSeg;      // segBuf: the window's first column; segPitch: words per matrix row
    u64 nRows; u32 width, nOut;         // width: all segments together
    const v4i *atab;            // [kSteps][NT][64 lanes]: 16 digit bytes per lane
    const u64 *bias;            // [3*nOut] canonical
    u64 *acc; u32 accumulate, accStride;      // acc: this launch's first output word of row 0; accStride: words per row of the caller's array
    u64 nTiles;
    u32 kSteps;
};

// ODD: some segment has an odd number of columns, an odd row pitch or an odd first column.  Its rows then start on 8-byte boundaries only and
// (odd count) its last 16-byte unit is half a unit: the staged row carries a zero word after it (zero digits in the operand table), the loads of that segment are 8-byte aligned
// and the half unit is one 8-byte load.  (The even form keeps its 16-byte aligned loads: config 3's matrices are 2, 100 and 6 wide.)
typedef int v4i_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <int NT, bool ODD = false>
__global__ void __launch_bounds__(512, 1) rows_dot_mfma_kernel(RowsDotMfmaParams P) {
    extern __shared__ u64 sm[];
    constexpr u32 NTH = 64 * MF_WAVES, NG = MF_ROWS / 32;
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 rowBytes = P.width * 8, LDB = rowBytes + 16;          // 16-byte aligned rows, 4-dword skew between rows
    unsigned char *tile = (unsigned char *)sm;
    v4i *Alds = (v4i *)(tile + MF_ROWS * LDB);                       // [kSteps][NT][64]
    const u32 upr = P.width / 2;                                     // 16-byte units per row
    for (u32 i = tid; i < P.kSteps * NT * 64; i += NTH) Alds[i] = P.atab[i];
    const u32 n = lane & 31, gk = lane >> 5;
    const u32 total = MF_ROWS * upr, dq = NTH / upr, dr = NTH - dq * upr;
    // a thread's share of a tile: up to NL 16-byte units, unit tid + 512 i = (row r_i, unit c_i of the row), the same for
    // every tile and stepped through as a recurrence
    constexpr u32 NL = (MF_ROWS * (MF_MAXW / 2) + NTH - 1) / NTH;    // 7
    const u32 r00 = tid / upr, c00 = tid - r00 * upr;
    v4i nxt[NL];
    auto fetch = [&](u64 tileIdx) {
        const u64 row0 = tileIdx * MF_ROWS;
        u32 r = r00, c = c00;
#pragma unroll
        for (u32 i = 0; i < NL; i++) {
            const v4i z = { 0, 0, 0, 0 };
            const u64 *sb = P.segBuf[0]; u32 sw = P.segW[0], sp = P.segPitch[0], su = 0;
#pragma unroll
            for (u32 k = 1; k < MF_MAXSEG; k++) if (k < P.nSeg && c >= P.segU0[k]) { sb = P.segBuf[k]; sw = P.segW[k]; sp = P.segPitch[k]; su = P.segU0[k]; }
            if constexpr (!ODD) nxt[i] = (NTH * i + tid < total && row0 + r < P.nRows) ? *(const v4i *)(sb + (row0 + r) * sp + 2 * (c - su)) : z;
            else {
                nxt[i] = z;
                if (NTH * i + tid < total && row0 + r < P.nRows) {
                    const u64 *q = sb + (row0 + r) * sp + 2 * (c - su);
                    if (2 * (c - su) + 1 < sw) nxt[i] = *(const v4i_a8 *)q;
                    else { const u64 v = *q; nxt[i][0] = (int)(u32)v; nxt[i][1] = (int)(u32)(v >> 32); }      // the row's last word; the staged pad word stays 0
                }
            }
            c += dr; r += dq;
            if (c >= upr) { c -= upr; r++; }
        }
    };
    if (blockIdx.x < P.nTiles) fetch(blockIdx.x);
    const u32 g = wave / NT, t = wave - g * NT;                      // this wave's rows 32 g + n and outputs 2 t + h (waves >= NG*NT only load)
    const unsigned char *brow = tile + (32 * g + n) * LDB + 16 * gk;
    const v4i *arow = Alds + t * 64 + lane;
    for (u64 tileIdx = blockIdx.x; tileIdx < P.nTiles; tileIdx += gridDim.x) {
        const u64 row0 = tileIdx * MF_ROWS;
        __syncthreads();                        // the previous tile has been consumed by every wave (first pass: the digits are in place)
        {
            u32 r = r00, c = c00;
#pragma unroll
            for (u32 i = 0; i < NL; i++) {
                if (NTH * i + tid < total) *(v4i *)(tile + r * LDB + 16 * c) = nxt[i] ^ (int)0x80808080;
                c += dr; r += dq;
                if (c >= upr) { c -= upr; r++; }
            }
        }
        // the next tile's rows travel while this one is multiplied and folded
        if (tileIdx + gridDim.x < P.nTiles) fetch(tileIdx + gridDim.x);
        __syncthreads();
        if (wave < NG * NT) {
            v16i acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = MF_PLANE_OFFSET;
            for (u32 s_ = 0; s_ < P.kSteps; s_++)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(arow[s_ * NT * 64], *(const v4i *)(brow + 32 * s_), acc, 0, 0, 0);
            const u32 o = 2 * t + gk;
            const u64 row = row0 + 32 * g + n;
            if (o < 3 * P.nOut && row < P.nRows) {
                u64 w[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = (u64)(u32)acc[4 * j] + ((u64)(u32)acc[4 * j + 1] << 8) + ((u64)(u32)acc[4 * j + 2] << 16) + ((u64)(u32)acc[4 * j + 3] << 24);   // < 2^50
                // w0 + w1 2^32 + w2 2^64 + w3 2^96 = w0 + w1 2^32 + w2 (2^32 - 1) - w3   (mod p)
                u64 v = add(add(w[0], mul(w[1], 1ull << 32)), mul(w[2], EPS));
                v = add(sub(v, w[3]), P.bias[o]);
                u64 *out = P.acc + row * (u64)P.accStride + o;
                if (P.accumulate) v = add(v, *out);
                *out = v;
            }
        }
    }
}

// f[r] = Horner over openings (vf1) of (acc[r][o] - K_o) * X[r][o]    (friPolinomial.js:38-50); the k-th Horner term is
// opening o = byte k of `order` (the reference walks Object.keys(friExps): not always the order of openingPoints)
__global__ void fri_combine_kernel(const u64 *__restrict__ acc, const u64 *__restrict__ K, E3 vf1, const u64 *__restrict__ xdiv,
                                   u32 nOpen, u32 order, u64 nRows, u64 *__restrict__ f) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nRows) return;
    E3 res = { { 0, 0, 0 } };
    for (u32 k = 0; k < nOpen; k++) {
        const u32 o = (order >> (8 * k)) & 255;
        const u64 *a = acc + (r * nOpen + o) * 3, *x = xdiv + (r * nOpen + o) * 3;
        E3 F = { { sub(a[0], K[3 * o]), sub(a[1], K[3 * o + 1]), sub(a[2], K[3 * o + 2]) } };
        E3 X = { { x[0], x[1], x[2] } };
        E3 t = e3_mul(F, X);
        res = k == 0 ? t : e3_add(e3_mul(vf1, res), t);
    }
    f[3 * r] = res.v[0]; f[3 * r + 1] = res.v[1]; f[3 * r + 2] = res.v[2];
}

constexpr u32 CD_MAXSEG = 8;
struct ColsDotParams {
    // columns of up to CD_MAXSEG matrices with the same rows, side by side: lane c of the launch belongs to the segment whose
    // [segC0, segC0 + segW) holds it (one sweep of the weights for all of them)
    const u64 *segBuf[CD_MAXSEG]; u32 segW[CD_MAXSEG], segC0[CD_MAXSEG], segStride[CD_MAXSEG], nSeg;   // segStride: words per matrix row (>= segW: a column range of a wider matrix)
    u64 width; u64 nRows; u64 rowStep;      // width: all segments together; rows k*rowStep, k < nRows
    const u32 *levLimbs;        // [nLev][nRows][3 comps][3 limbs]
    u32 nLev;
    u64 *partial;               // [nChunks][nLev][width][3]
    u32 rowsPerChunk;           // <= 1024
};

// lanes <-> columns (coalesced rows), each lane accumulates its column over a chunk of rows
template <int NLEV>
__global__ void __launch_bounds__(256) cols_dot_kernel(ColsDotParams P) {
    const u64 c = (u64)blockIdx.y * blockDim.x + threadIdx.x;
    const u64 chunk = blockIdx.x;
    const u64 k0 = chunk * P.rowsPerChunk, k1 = min(P.nRows, k0 + P.rowsPerChunk);
    u64 S[NLEV][3][6];
#pragma unroll
    for (int l = 0; l < NLEV; l++)
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 6; i++) S[l][k][i] = 0;
    const bool valid = c < P.width;
    constexpr int RB = 8;                            // rows whose loads are in flight together
    const u64 *buf = P.segBuf[0]; u64 sw = P.segStride[0], cl = c;
#pragma unroll
    for (u32 k = 1; k < CD_MAXSEG; k++) if (k < P.nSeg && c >= P.segC0[k]) { buf = P.segBuf[k]; sw = P.segStride[k]; cl = c - P.segC0[k]; }
    const u64 stride = P.rowStep * sw;
    for (u64 kb = k0; kb < k1; kb += RB) {
        u64 pv[RB];
#pragma unroll
        for (int j = 0; j < RB; j++) pv[j] = (valid && kb + j < k1) ? buf[(kb + j) * stride + cl] : 0;
#pragma unroll
        for (int j = 0; j < RB; j++) {
            const u64 k = kb + j;
            if (k >= k1) break;
            const u32 p0 = (u32)pv[j], p1 = (u32)(pv[j] >> 32);
#pragma unroll
            for (int l = 0; l < NLEV; l++) {
                const u32 *L = P.levLimbs + (((u64)l * P.nRows + k) * 9);            // uniform
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const u32 w0 = L[3 * q], w1 = L[3 * q + 1], w2 = L[3 * q + 2];
                    S[l][q][0] += (u64)p0 * w0; S[l][q][1] += (u64)p0 * w1; S[l][q][2] += (u64)p0 * w2;
                    S[l][q][3] += (u64)p1 * w0; S[l][q][4] += (u64)p1 * w1; S[l][q][5] += (u64)p1 * w2;
                }
            }
        }
    }
    if (!valid) return;
#pragma unroll
    for (int l = 0; l < NLEV; l++)
#pragma unroll
        for (int q = 0; q < 3; q++)
            P.partial[((chunk * P.nLev + l) * P.width + c) * 3 + q] = fold6(S[l][q]);
}
// sums chunks [blockIdx.y * per, ...) of partial[nChunks][n] into out[blockIdx.y][n]; run twice (nChunks -> <=64 groups -> 1)
// so that the reduction over thousands of chunks is not left to n threads
__global__ void cols_dot_final_kernel(const u64 *__restrict__ partial, u64 nChunks, u64 per, u64 n /* nLev*width*3 */, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 c0 = (u64)blockIdx.y * per, c1 = min(nChunks, c0 + per);
    u64 acc = 0;
    for (u64 ch = c0; ch < c1; ch++) acc = add(acc, partial[ch * n + i]);
    out[(u64)blockIdx.y * n + i] = acc;
}
// canonical u64 -> three 22/22/20-bit limbs
__global__ void limbs_kernel(const u64 *__restrict__ in, u64 n, u32 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 v = in[i];
    out[3 * i] = (u32)v & LIMB_MASK; out[3 * i + 1] = (u32)(v >> LIMB_BITS) & LIMB_MASK; out[3 * i + 2] = (u32)(v >> (2 * LIMB_BITS));
}

inline unsigned nblk(u64 n, u32 t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

using namespace pil2gl;

// host side of rows_dot_mfma_kernel: signed base-256 digits of the weights laid out as the MFMA's A operand, the constant per output.
// Segment k: nRows x widths[k] matrix bufs[k] with weights hostCoefs[k] ([nOut][widths[k]][3]); out = sum over all segments' columns.
// a segment of the staged row: columns [col0, col0 + width) of an nRows x pitch matrix, with the weights of those columns
struct MfSeg { const u64 *buf; u64 pitch, col0, width; const u64 *coef; };     // coef: [nOut][pitch][3], the whole matrix's
static u64 mf_padded(const MfSeg *segs, u32 n) { u64 t = 0; for (u32 k = 0; k < n; k++) t += segs[k].width + (segs[k].width & 1); return t; }
static bool rows_dot_mfma_fits(const MfSeg *segs, u32 nSeg, u32 nOut) {
    const char *sw = getenv("PIL2GL_ROWS_DOT_MFMA");
    if ((sw && sw[0] == '0') || nOut < 1 || nOut > 2 || nSeg < 1 || nSeg > MF_MAXSEG) return false;
    for (u32 k = 0; k < nSeg; k++)
        if (segs[k].width == 0 || ((uintptr_t)segs[k].buf & 7) || (segs[k].pitch >> 31)) return false;
    const u64 total = mf_padded(segs, nSeg);                    // an odd segment is staged with a zero word after each row
    if (total < 32 || total > MF_MAXW) return false;
    // the kernel keeps a 64-row tile and the digit planes in up to ~144 KB of LDS: only where a workgroup may have that much
    // (gfx950: 160 KB); elsewhere the streaming kernel takes the call
    static int ldsMax = -1;
    if (ldsMax < 0) {
        int dev = 0, v = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 0;
        ldsMax = v;
    }
    const u32 kSteps = (u32)((total * 8 + 31) / 32), NT = nOut == 1 ? 2 : 3;
    const size_t need = (size_t)MF_ROWS * (total * 8 + 16) + (size_t)kSteps * NT * 1024;
    return (size_t)ldsMax >= need;
}
static int launch_rows_dot_mfma(const MfSeg *segs, u32 nBufs, u64 nRows, u32 nOut, u64 *acc, u32 accStride, bool accumulate, hipStream_t st) {
    const u64 Pm = 0xFFFFFFFF00000001ull;
    const int NT = nOut == 1 ? 2 : 3;
    const u32 nO = 3 * nOut;
    const u64 winWidth = mf_padded(segs, nBufs);                // columns of the staged row: every segment rounded up to an even count
    bool odd = false;                                           // rows of some segment start on 8-byte boundaries only
    for (u32 k = 0; k < nBufs; k++) odd |= ((segs[k].width | segs[k].pitch | segs[k].col0) & 1) != 0 || ((uintptr_t)segs[k].buf & 15);
    std::vector<signed char> dig((size_t)winWidth * nO * 9, 0);
    std::vector<u64> bias(nO);
    unsigned __int128 k128 = 0, offs = 0;
    for (int i = 0; i < 8; i++) k128 += (unsigned __int128)128 << (8 * i);
    {   // sum_p 2^24 * 2^(8p) mod p, the power kept reduced (2^24 * 2^120 does not fit 128 bits)
        unsigned __int128 pw = 1;
        for (int p_ = 0; p_ < 16; p_++) { offs = (offs + (pw * (u64)MF_PLANE_OFFSET) % Pm) % Pm; pw = (pw * 256) % Pm; }
    }
    for (u32 o = 0; o < nO; o++) {
        unsigned __int128 sumw = 0;
        u64 c = 0;
        for (u32 k = 0; k < nBufs; k++, c += c & 1)             // (an odd segment's pad column keeps its zero digits)
            for (u64 cl = 0; cl < segs[k].width; cl++, c++) {
                u64 w = segs[k].coef[((u64)(o / 3) * segs[k].pitch + segs[k].col0 + cl) * 3 + (o % 3)] % Pm;
                sumw = (sumw + w) % Pm;
                int carry = 0;
                for (int j = 0; j < 9; j++) {
                    int b = (j < 8 ? (int)((w >> (8 * j)) & 255) : 0) + carry;
                    carry = 0;
                    if (b >= 128) { b -= 256; carry = 1; }
                    dig[((size_t)c * nO + o) * 9 + j] = (signed char)b;
                }
            }
        unsigned __int128 b = ((k128 % Pm) * sumw) % Pm;
        bias[o] = (u64)((b + Pm - offs) % Pm);
    }
    const u32 kSteps = (u32)((winWidth * 8 + 31) / 32);
    std::vector<signed char> atab((size_t)kSteps * NT * 64 * 16, 0);
    for (u32 s_ = 0; s_ < kSteps; s_++)
        for (int t = 0; t < NT; t++)
            for (u32 lane = 0; lane < 64; lane++) {
                const u32 m = lane & 31, gk = lane >> 5;
                const u32 h = (m >> 2) & 1, pl = 4 * (m >> 3) + (m & 3), o = 2 * t + h;
                if (o >= nO) continue;
                signed char *dst = &atab[(((size_t)s_ * NT + t) * 64 + lane) * 16];
                for (u32 kk = 0; kk < 16; kk++) {
                    const u64 byteOff = 32ull * s_ + 16 * gk + kk;
                    if (byteOff >= winWidth * 8) continue;              // past the row: zero digits (the lane reads the row's padding)
                    const u64 c = byteOff >> 3; const u32 i = (u32)(byteOff & 7);
                    const int j = (int)pl - (int)i;
                    dst[kk] = (j >= 0 && j <= 8) ? dig[((size_t)c * nO + o) * 9 + j] : 0;
                }
            }
    u64 *d;
    const size_t atWords = (atab.size() + 7) / 8;
    P2_TRY(scratch(7, atWords + nO + 2, &d));
    HIP_TRY(hipMemcpyAsync(d, atab.data(), atab.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d + atWords, bias.data(), nO * 8, hipMemcpyHostToDevice, st));
    RowsDotMfmaParams P;
    u32 u0 = 0;
    for (u32 k = 0; k < MF_MAXSEG; k++) {
        P.segBuf[k] = k < nBufs ? segs[k].buf + segs[k].col0 : nullptr; P.segW[k] = k < nBufs ? (u32)segs[k].width : 0; P.segPitch[k] = k < nBufs ? (u32)segs[k].pitch : 0; P.segU0[k] = u0;
        if (k < nBufs) u0 += (u32)((segs[k].width + 1) / 2);
    }
    P.nSeg = nBufs; P.nRows = nRows; P.width = (u32)winWidth; P.nOut = nOut; P.atab = (const v4i *)d; P.bias = d + atWords;
    P.acc = acc; P.accumulate = (u32)accumulate; P.accStride = accStride; P.nTiles = (nRows + MF_ROWS - 1) / MF_ROWS; P.kSteps = kSteps;
    const size_t lds = (size_t)MF_ROWS * (winWidth * 8 + 16) + atab.size();
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)std::min<u64>(P.nTiles, (u64)cus);
#define RD_LAUNCH(NT_, ODD_) { HIP_TRY(hipFuncSetAttribute((const void *)rows_dot_mfma_kernel<NT_, ODD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                              hipLaunchKernelGGL((rows_dot_mfma_kernel<NT_, ODD_>), dim3(grid), dim3(64 * MF_WAVES), lds, st, P); }
    if (NT == 2) { if (odd) RD_LAUNCH(2, true) else RD_LAUNCH(2, false) }
    else { if (odd) RD_LAUNCH(3, true) else RD_LAUNCH(3, false) }
#undef RD_LAUNCH
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(st));          // the tables are host temporaries in a shared scratch slot
    return PIL2GL_OK;
}

// the vector-ALU kernels (whole-row streaming tiles / column tiles): any shape
static int rows_dot_ext_plain(const uint64_t *buf, uint64_t width, uint64_t nRows, const uint64_t *hostCoef, uint32_t nOut,
                              uint64_t *acc, int accumulate, void *stream) {
    hipStream_t st = as_stream(stream);
    const u64 nC = (u64)nOut * width * 3;
    std::vector<u32> limbs(nC * 3);
    for (u64 i = 0; i < nC; i++) {
        const u64 v = hostCoef[i] % 0xFFFFFFFF00000001ull;
        limbs[3 * i] = (u32)v & LIMB_MASK; limbs[3 * i + 1] = (u32)(v >> LIMB_BITS) & LIMB_MASK; limbs[3 * i + 2] = (u32)(v >> (2 * LIMB_BITS));
    }
    u64 *d;
    P2_TRY(scratch(7, (limbs.size() * 4 + 7) / 8 + 1, &d));
    HIP_TRY(hipMemcpyAsync(d, limbs.data(), limbs.size() * 4, hipMemcpyHostToDevice, st));
    const unsigned blocks = (unsigned)((nRows + 255) / 256);
    // a lane's six partial sums take one term below 2^54 per column and are folded once per launch: windows of at most
    // 1024 columns, the later ones accumulating onto the first one's result
    const char *sw = getenv("PIL2GL_ROWS_DOT_STREAM");
    const bool stream_ok = !(sw && sw[0] == '0');
    for (u64 col0 = 0; col0 < width; col0 += 1024) {
        RowsDotParams P = { buf, std::min<u64>(1024, width - col0), nRows, width, col0, (const u32 *)d, nOut, acc, (u32)(accumulate != 0 || col0 != 0) };
        if (stream_ok && P.width >= 32) {               // long rows: whole-row streaming tiles
            const u32 nch = (u32)((P.width + STREAM_CW_MAX - 1) / STREAM_CW_MAX), cw = (u32)((P.width + nch - 1) / nch), LD = cw | 1;
            const size_t lds = 8 * std::max<size_t>((size_t)64 * LD, (size_t)nOut * 18 * 64);
            const unsigned sb = (unsigned)((nRows + 63) / 64);
#define STREAM_CASE(N_) { if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)rows_dot_stream_kernel<N_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                          hipLaunchKernelGGL((rows_dot_stream_kernel<N_>), dim3(sb), dim3(256), lds, st, P, cw, LD); }
            switch (nOut) {
            case 1: STREAM_CASE(1) break;
            case 2: STREAM_CASE(2) break;
            case 3: STREAM_CASE(3) break;
            default: STREAM_CASE(4) break;
            }
#undef STREAM_CASE
            KERNEL_CHECK();
            continue;
        }
        switch (nOut) {
        case 1: rows_dot_kernel<1><<<blocks, 256, 0, st>>>(P); break;
        case 2: rows_dot_kernel<2><<<blocks, 256, 0, st>>>(P); break;
        case 3: rows_dot_kernel<3><<<blocks, 256, 0, st>>>(P); break;
        default: rows_dot_kernel<4><<<blocks, 256, 0, st>>>(P); break;
        }
        KERNEL_CHECK();
    }
    HIP_TRY(hipStreamSynchronize(st));      // `limbs` is a host temporary and the scratch slot is reused by the next call
    return PIL2GL_OK;
}

// the same sums over several matrices with the same rows (the stage matrices the FRI polynomial reads: friPolinomial.js:26-50
// walks cm1..cmQ and the constants): out[r][o] = sum_k sum_c bufs[k][r][c] * hostCoefs[k][o][c].  One pass over all of them
// when they fit the matrix-core kernel side by side, else one call per matrix accumulating into acc.
// The matrices of a call as launches of the matrix-core kernel: a matrix wider than the kernel's staged row (MF_MAXW columns) is cut into even
// column windows, windows and narrow matrices are packed (largest first) into launches of at most MF_MAXSEG segments and MF_MAXW columns,
// each launch after the first accumulating.  Matrices left in a launch of under 32 columns go to rows_dot_ext_dev one by one.
// -> false: not for the matrix cores (more than two outputs, PIL2GL_ROWS_DOT_MFMA=0, no wide launch at all)
static bool rows_dot_mfma_plan(const uint64_t *const *bufs, const uint64_t *widths, const uint64_t *const *hostCoefs, u32 nBufs, u32 nOut,
                               std::vector<std::vector<MfSeg>> &launches, std::vector<u32> &leftovers) {
    std::vector<MfSeg> wins;
    for (u32 k = 0; k < nBufs; k++) {
        const u64 W = widths[k];
        if (W == 0) continue;
        const u64 n = (W + (W & 1) + MF_MAXW - 1) / MF_MAXW;
        u64 per = (W + n - 1) / n; per += per & 1;
        for (u64 c0 = 0; c0 < W; c0 += per) wins.push_back({ bufs[k], W, c0, std::min<u64>(per, W - c0), hostCoefs[k] });
    }
    std::stable_sort(wins.begin(), wins.end(), [](const MfSeg &a, const MfSeg &b) { return a.width > b.width; });
    std::vector<u64> fill;
    for (const MfSeg &w : wins) {
        const u64 pw = w.width + (w.width & 1);
        size_t b = 0;
        while (b < launches.size() && (launches[b].size() >= MF_MAXSEG || fill[b] + pw > MF_MAXW)) b++;
        if (b == launches.size()) { launches.emplace_back(); fill.push_back(0); }
        launches[b].push_back(w); fill[b] += pw;
    }
    bool any = false;
    for (size_t b = 0; b < launches.size();) {
        if (rows_dot_mfma_fits(launches[b].data(), (u32)launches[b].size(), std::min<u32>(nOut, 2))) { any = true; b++; continue; }
        for (const MfSeg &w : launches[b]) {
            if (w.width != w.pitch) return false;               // (a window of a wide matrix always fits: cannot happen)
            for (u32 k = 0; k < nBufs; k++) if (bufs[k] == w.buf && hostCoefs[k] == w.coef) { leftovers.push_back(k); break; }
        }
        launches.erase(launches.begin() + b); fill.erase(fill.begin() + b);
    }
    return any;
}

extern "C" {

int pil2gl_rows_dot_ext_multi_dev(const uint64_t *const *bufs, const uint64_t *widths, uint32_t nBufs, uint64_t nRows,
                                  const uint64_t *const *hostCoefs, uint32_t nOut, uint64_t *acc, int accumulate, void *stream);
// one matrix: rows of 32 columns and more with one or two outputs go to the matrix cores (in windows when wider than the staged row)
int pil2gl_rows_dot_ext_dev(const uint64_t *buf, uint64_t width, uint64_t nRows, const uint64_t *hostCoef, uint32_t nOut,
                            uint64_t *acc, int accumulate, void *stream) {
    P2_TRY(ensure_init());
    if (!buf || !hostCoef || !acc) return fail(PIL2GL_EINVAL, "null buffer");
    if (nOut < 1 || nOut > 4) return fail(PIL2GL_EINVAL, "nOut must be 1..4");
    if (width == 0 || nRows == 0) return PIL2GL_OK;
    return pil2gl_rows_dot_ext_multi_dev(&buf, &width, 1, nRows, &hostCoef, nOut, acc, accumulate, stream);
}

int pil2gl_rows_dot_ext_multi_dev(const uint64_t *const *bufs, const uint64_t *widths, uint32_t nBufs, uint64_t nRows,
                                  const uint64_t *const *hostCoefs, uint32_t nOut, uint64_t *acc, int accumulate, void *stream) {
    P2_TRY(ensure_init());
    if (!bufs || !widths || !hostCoefs || !acc || nBufs == 0) return fail(PIL2GL_EINVAL, "null buffer");
    for (uint32_t k = 0; k < nBufs; k++) if (!bufs[k] || !hostCoefs[k]) return fail(PIL2GL_EINVAL, "null buffer");
    if (nOut < 1 || nOut > 4) return fail(PIL2GL_EINVAL, "nOut must be 1..4");
    if (nRows == 0) return PIL2GL_OK;
    std::vector<std::vector<MfSeg>> launches;
    std::vector<u32> leftovers;
    bool acc1 = accumulate != 0;
    if (rows_dot_mfma_plan(bufs, widths, hostCoefs, nBufs, nOut, launches, leftovers)) {
        // the kernel produces one or two outputs per launch: three or four opening points take two sweeps of the same launches
        // (outputs 0-1, then 2-3: HBM-bound passes against one vector-ALU-bound pass), writing their own words of the caller's rows
        for (u32 o0 = 0; o0 < nOut; o0 += 2) {
            const u32 no = std::min<u32>(2, nOut - o0);
            bool accg = accumulate != 0;
            for (auto l : launches) {
                for (MfSeg &sg : l) sg.coef += (u64)o0 * sg.pitch * 3;
                P2_TRY(launch_rows_dot_mfma(l.data(), (u32)l.size(), nRows, no, acc + 3 * o0, 3 * nOut, accg, as_stream(stream)));
                accg = true;
            }
        }
        acc1 = true;
        for (u32 k : leftovers) P2_TRY(rows_dot_ext_plain(bufs[k], widths[k], nRows, hostCoefs[k], nOut, acc, 1, stream));
        return PIL2GL_OK;
    }
    for (uint32_t k = 0; k < nBufs; k++) {
        if (widths[k] == 0) continue;
        P2_TRY(rows_dot_ext_plain(bufs[k], widths[k], nRows, hostCoefs[k], nOut, acc, acc1 ? 1 : 0, stream)); acc1 = true;
    }
    return PIL2GL_OK;
}

int pil2gl_fri_combine_dev(const uint64_t *acc, const uint64_t *hostK, const uint64_t vf1[3], const uint64_t *xDivXSubXi,
                           uint32_t nOpen, uint64_t nRows, uint64_t *f, void *stream) {
    const uint32_t ident[4] = { 0, 1, 2, 3 };
    return pil2gl_fri_combine_order_dev(acc, hostK, vf1, xDivXSubXi, nOpen, ident, nRows, f, stream);
}
int pil2gl_fri_combine_order_dev(const uint64_t *acc, const uint64_t *hostK, const uint64_t vf1[3], const uint64_t *xDivXSubXi,
                                 uint32_t nOpen, const uint32_t *order, uint64_t nRows, uint64_t *f, void *stream) {
    P2_TRY(ensure_init());
    if (!acc || !hostK || !vf1 || !xDivXSubXi || !f || !order || nOpen < 1 || nOpen > 4) return fail(PIL2GL_EINVAL, "bad FRI combine arguments");
    u32 ord = 0, seen = 0;
    for (u32 k = 0; k < nOpen; k++) {
        if (order[k] >= nOpen || (seen >> order[k] & 1)) return fail(PIL2GL_EINVAL, "order must be a permutation of the openings");
        seen |= 1u << order[k]; ord |= order[k] << (8 * k);
    }
    hipStream_t st = as_stream(stream);
    std::vector<u64> k(hostK, hostK + 3ull * nOpen);
    u64 *d;
    P2_TRY(scratch(7, 16, &d));
    HIP_TRY(hipMemcpyAsync(d, k.data(), k.size() * 8, hipMemcpyHostToDevice, st));
    E3 v = { { vf1[0], vf1[1], vf1[2] } };
    fri_combine_kernel<<<nblk(nRows), 256, 0, st>>>(acc, d, v, xDivXSubXi, nOpen, ord, nRows, f);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return PIL2GL_OK;
}

// evaluations over several matrices with the same rows in ONE sweep of the weights: hostOuts[k] receives nLev x widths[k] x 3.
// The _range form takes columns [colBegin[k], colBegin[k] + widths[k]) of matrices whose rows are strides[k] words long (a rank of a
// sharded proof evaluates its share of the columns: only those cells are read).
int pil2gl_cols_dot_ext_multi_dev(const uint64_t *const *bufs, const uint64_t *widths, uint32_t nBufs, uint64_t nRows, uint64_t rowStep,
                                  const uint64_t *const *levs, uint32_t nLev, uint64_t *const *hostOuts, void *stream) {
    return pil2gl_cols_dot_ext_range_dev(bufs, widths, nullptr, widths, nBufs, nRows, rowStep, levs, nLev, hostOuts, stream);
}
int pil2gl_cols_dot_ext_range_dev(const uint64_t *const *bufs, const uint64_t *strides, const uint64_t *colBegin, const uint64_t *widths, uint32_t nBufs,
                                  uint64_t nRows, uint64_t rowStep, const uint64_t *const *levs, uint32_t nLev, uint64_t *const *hostOuts, void *stream) {
    P2_TRY(ensure_init());
    if (!bufs || !widths || !strides || !levs || !hostOuts || nBufs == 0) return fail(PIL2GL_EINVAL, "null buffer");
    for (uint32_t k = 0; k < nBufs && k < CD_MAXSEG; k++)
        if ((colBegin ? colBegin[k] : 0) + widths[k] > strides[k] || (strides[k] >> 31)) return fail(PIL2GL_EINVAL, "column range outside the matrix");
    if (nBufs > CD_MAXSEG) return fail(PIL2GL_EINVAL, "at most %u matrices per call", CD_MAXSEG);
    if (nLev < 1 || nLev > 64) return fail(PIL2GL_EINVAL, "nLev must be 1..64");
    u64 width = 0;
    for (uint32_t k = 0; k < nBufs; k++) { if (!bufs[k] || !hostOuts[k]) return fail(PIL2GL_EINVAL, "null buffer"); if (widths[k] >> 31) return fail(PIL2GL_EINVAL, "matrix too wide"); width += widths[k]; }
    for (uint32_t l = 0; l < nLev; l++) if (!levs[l]) return fail(PIL2GL_EINVAL, "null buffer");
    if (width == 0 || nRows == 0) return PIL2GL_OK;
    if (nLev > 4) {                         // a sweep weighs four opening points: more of them take more sweeps (hostOuts[k] is [nLev][widths[k]][3])
        for (uint32_t l0 = 0; l0 < nLev; l0 += 4) {
            std::vector<uint64_t *> outs(nBufs);
            for (uint32_t k = 0; k < nBufs; k++) outs[k] = hostOuts[k] + (u64)l0 * widths[k] * 3;
            P2_TRY(pil2gl_cols_dot_ext_range_dev(bufs, strides, colBegin, widths, nBufs, nRows, rowStep, levs + l0, std::min<u32>(4, nLev - l0), outs.data(), stream));
        }
        return PIL2GL_OK;
    }
    hipStream_t st = as_stream(stream);
    const u32 rpc = 1024;
    const u64 nChunks = (nRows + rpc - 1) / rpc, n = (u64)nLev * width * 3;
    u64 *d;
    const u64 nGroups = std::min<u64>(64, nChunks), per = (nChunks + nGroups - 1) / nGroups;
    P2_TRY(scratch(7, ((u64)nLev * nRows * 9 * 4 + 7) / 8 + nChunks * n + nGroups * n + n + 1, &d));
    u32 *limbs = (u32 *)d;
    u64 *partial = d + ((u64)nLev * nRows * 9 * 4 + 7) / 8, *part2 = partial + nChunks * n, *res = part2 + nGroups * n;
    for (u32 l = 0; l < nLev; l++) limbs_kernel<<<nblk(nRows * 3), 256, 0, st>>>(levs[l], nRows * 3, limbs + (u64)l * nRows * 9);
    ColsDotParams P;
    u32 c0 = 0;
    for (u32 k = 0; k < CD_MAXSEG; k++) {
        P.segBuf[k] = k < nBufs ? bufs[k] + (colBegin ? colBegin[k] : 0) : nullptr; P.segW[k] = k < nBufs ? (u32)widths[k] : 0; P.segStride[k] = k < nBufs ? (u32)strides[k] : 0;
        P.segC0[k] = c0; if (k < nBufs) c0 += (u32)widths[k];
    }
    P.nSeg = nBufs; P.width = width; P.nRows = nRows; P.rowStep = rowStep; P.levLimbs = limbs; P.nLev = nLev; P.partial = partial; P.rowsPerChunk = rpc;
    const u32 threads = (u32)std::min<u64>(256, (width + 63) / 64 * 64);
    dim3 grid((unsigned)nChunks, nblk(width, threads));
    switch (nLev) {
    case 1: cols_dot_kernel<1><<<grid, threads, 0, st>>>(P); break;
    case 2: cols_dot_kernel<2><<<grid, threads, 0, st>>>(P); break;
    case 3: cols_dot_kernel<3><<<grid, threads, 0, st>>>(P); break;
    default: cols_dot_kernel<4><<<grid, threads, 0, st>>>(P); break;
    }
    cols_dot_final_kernel<<<dim3(nblk(n), (unsigned)nGroups), 256, 0, st>>>(partial, nChunks, per, n, part2);
    cols_dot_final_kernel<<<dim3(nblk(n), 1), 256, 0, st>>>(part2, (nChunks + per - 1) / per, nGroups, n, res);
    KERNEL_CHECK();
    std::vector<u64> host(n);
    HIP_TRY(hipMemcpyAsync(host.data(), res, n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // res is [nLev][all columns][3]; hostOuts[k] is [nLev][widths[k]][3]
    u64 cbeg = 0;
    for (uint32_t k = 0; k < nBufs; k++) {
        for (u32 l = 0; l < nLev; l++)
            memcpy(hostOuts[k] + ((u64)l * widths[k]) * 3, host.data() + ((u64)l * width + cbeg) * 3, widths[k] * 3 * 8);
        cbeg += widths[k];
    }
    return PIL2GL_OK;
}

int pil2gl_cols_dot_ext_dev(const uint64_t *buf, uint64_t width, uint64_t nRows, uint64_t rowStep, const uint64_t *const *levs,
                            uint32_t nLev, uint64_t *hostOut, void *stream) {
    if (!buf || !hostOut) { P2_TRY(ensure_init()); return fail(PIL2GL_EINVAL, "null buffer"); }
    return pil2gl_cols_dot_ext_multi_dev(&buf, &width, 1, nRows, rowStep, levs, nLev, &hostOut, stream);
}

}  // extern "C"
