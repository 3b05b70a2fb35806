// Multi-column Goldilocks NTT / iNTT / LDE on a row-major rows x C matrix (gfx950).
//
// Replaces src/helpers/fft/fft_p.js:178-302 (interpolate / fft / ifft) of the reference, whose
// result is, per column, F.fft / F.ifft / extendPol (test/fft_p.test.js:47-229).
//
// Design (not the reference's block/transposition scheme):
//  * A transform of 2^n rows is cut into passes of k index bits (8, or 7 for matrices whose rows are
//    shorter than 128 bytes; at most 10).  One workgroup owns a tile of 2^k rows (stride 2^lo rows)
//    x S column slots in LDS, runs the k stages there as register steps of up to four stages (16-row
//    sub-transforms in Z/(2^96+1), see below), and applies the inter-pass twiddle w_(2^(lo+k))^(b*j)
//    from a per-tile LDS table that is shared by all columns of the tile.  All passes are in place.
//    Workgroups are numbered so that the column chunks of one row block run on one XCD (one L2).
//  * Forward/inverse transforms run decimation-in-frequency (natural in -> bit-reversed out);
//    the last pass scatters rows to their bit-reversed index so the caller sees natural order.
//  * interpolate (LDE) never reorders anything: iNTT as DIF leaves coefficient m at row
//    bitrev(m); the "mid" kernel finishes the iNTT on a contiguous tile, multiplies coefficient m
//    by (7*w_E^j)^m / N for each of the 2^b cosets j and immediately runs the first k stages of a
//    decimation-in-time NTT (bit-reversed in -> natural out), writing coset j's tile to rows
//    (pos << b) + j of dst.  dst is then just an N x (C*2^b) matrix on which the remaining DIT
//    passes run in place.  Extended row i = 8k+j is coset point 7*w_E^(8k+j): natural order.
//  * Rows are contiguous in memory, so a tile row is one contiguous segment of Wc*8 bytes
//    (or several adjacent rows when C is small): every global access is a coalesced segment.
#include "common.h"
#include "gl_field.cuh"
#include "gl_fermat.cuh"
#include <stdlib.h>
#include <algorithm>

#ifndef NTT_MUL
#define NTT_MUL(a, b) mul_lazy_x(a, b)   // the twiddle products of the tile kernels: the 16-instruction exact carry-out form (A/B: -DNTT_MUL=mul_lazy, 22 compiler instructions: 208.5 vs 201.5 ms per config-3 interpolate)
#endif
// experiments only (tools/lde_two_sweep.sh): larger workgroups for the any-geometry instances, so that 10-stage tiles (1024 rows x 16
// slots, 128+ KB of LDS, one workgroup per CU) still put four waves on a SIMD
#ifndef NTT_MAXTHREADS
#define NTT_MAXTHREADS 256
#endif
#ifndef LDE_MAXTHREADS
#define LDE_MAXTHREADS 512
#endif
using namespace gl;

namespace {

__device__ __forceinline__ u64 pow256(const u64 *__restrict__ T, u32 e) {
    u64 r = T[e & 255];
    r = mul_lazy(r, T[256 + ((e >> 8) & 255)]);
    r = mul_lazy(r, T[512 + ((e >> 16) & 255)]);
    return mul(r, T[768 + (e >> 24)]);
}
// w_(2^logM)^e for the root whose pow256 table is T (1 <= logM <= 32, e < 2^logM)
__device__ __forceinline__ u64 root_pow(const u64 *__restrict__ T, u32 logM, u32 e) { return pow256(T, e << (32 - logM)); }

// k stages on tile[2^k][S] (column x).  The stages are grouped into register steps of c <= 4 stages: a lane takes the
// 2^c rows of one sub-transform into registers, runs it in Z/(2^96+1) where its twiddles are constant shifts
// (gl_fermat.cuh), reduces once, multiplies by the twiddle that joins it to the next group and writes it back: one LDS
// round trip and one modular multiplication per element per c stages.  TW[j] = w_(2^k)^j for all j < 2^k.
__host__ __device__ constexpr u32 next_chunk(u32 rem) { return rem <= 4 ? rem : (rem == 5 || rem == 6 || rem == 9) ? 3 : 4; }
__host__ __device__ constexpr u32 brev_c(u32 i, int c) { u32 r = 0; for (int b = 0; b < c; b++) r |= ((i >> b) & 1u) << (c - 1 - b); return r; }

// Gentleman-Sande group on blocks of 2^lm rows: natural in -> bit-reversed out (within the group's c index bits)
// PAD (fixed-geometry kernels: k = 8, C = 4): tile row t lives at row t + (t >> 4), one spare row per 16, so that the four
// sub-transforms a wave reads together (rows 256 words apart otherwise: the same LDS banks) start in different banks
template <int C, bool INV, bool PAD = false>
__device__ __forceinline__ void dif_step(u64 *tile, const u64 *TW, u32 k, u32 lm, u32 S, u32 x, u32 y, u32 by) {
    constexpr u32 R = 1u << C;
    const u32 ls = lm - C, nD = 1u << (k - C), st = PAD && ls == 4 ? (S << ls) + S : S << ls;
    const bool last = ls == 0;
    for (u32 d = y; d < nD; d += by) {
        const u32 np = d & ((1u << ls) - 1);
        u64 *col = tile + (size_t)(((d >> ls) << lm) + np + (PAD && ls == 0 ? d : 0)) * S + x;
        fermat::f128 v[R];
#pragma unroll
        for (u32 r = 0; r < R; r++) v[r] = fermat::from_gl(col[r * st]);
        fermat::dft_dif<C, INV>(v);
        const u32 e1 = np << (k - lm);                         // w_(2^lm)^np = TW[e1]
#pragma unroll
        for (u32 i = 0; i < R; i++) {
            const u32 q = brev_c(i, C);
            u64 o = fermat::to_gl_lazy(v[i]);
            if (!last && q) o = NTT_MUL(o, TW[e1 * q]);            // the tile stays lazy: whoever stores a FINAL result canonicalises it
            col[i * st] = o;
        }
    }
    __syncthreads();
}
// Cooley-Tukey group joining 2^C finished blocks of 2^lp rows: bit-reversed in -> natural out.  The inputs already carry
// this group's twiddles (applied when they were stored); the outputs get the next group's (cn = its stage count, 0: none).
template <int C, bool INV, bool PAD = false>
__device__ __forceinline__ void dit_step(u64 *tile, const u64 *TW, u32 k, u32 lp, u32 cn, u32 S, u32 x, u32 y, u32 by) {
    constexpr u32 R = 1u << C;
    const u32 lm = lp + C, nD = 1u << (k - C), st = PAD && lp == 4 ? (S << lp) + S : S << lp;
    for (u32 d = y; d < nD; d += by) {
        const u32 np = d & ((1u << lp) - 1), blk = d >> lp;
        u64 *col = tile + (size_t)((blk << lm) + np + (PAD && lp == 0 ? blk : 0)) * S + x;
        fermat::f128 v[R];
#pragma unroll
        for (u32 r = 0; r < R; r++) v[r] = fermat::from_gl(col[r * st]);
        fermat::dft_dit<C, INV>(v);
        if (cn) {
            const u32 rho = bitrev32(blk & ((1u << cn) - 1), cn), sh = k - lm - cn;
#pragma unroll
            for (u32 q = 0; q < R; q++) {
                // rho = 0 multiplies by TW[0] = 1: cheaper than a lane-dependent branch around every product
                col[q * st] = NTT_MUL(fermat::to_gl_lazy(v[q]), TW[(rho * (np + (q << lp))) << sh]);
            }
        } else {
#pragma unroll
            for (u32 q = 0; q < R; q++) col[q * st] = fermat::to_gl_lazy(v[q]);     // lazy, see dif_step
        }
    }
    __syncthreads();
}
template <bool INV>
__device__ __forceinline__ void dif_stages(u64 *tile, const u64 *TW, u32 k, u32 S, u32 x, u32 y, u32 by) {
    for (u32 lm = k; lm > 0;) {
        const u32 c = next_chunk(lm);
        switch (c) {
        case 4: dif_step<4, INV>(tile, TW, k, lm, S, x, y, by); break;
        case 3: dif_step<3, INV>(tile, TW, k, lm, S, x, y, by); break;
        case 2: dif_step<2, INV>(tile, TW, k, lm, S, x, y, by); break;
        default: dif_step<1, INV>(tile, TW, k, lm, S, x, y, by); break;
        }
        lm -= c;
    }
}
template <bool INV>
__device__ __forceinline__ void dit_stages(u64 *tile, const u64 *TW, u32 k, u32 S, u32 x, u32 y, u32 by) {
    for (u32 lp = 0; lp < k;) {
        const u32 c = next_chunk(k - lp), cn = next_chunk(k - lp - c);
        switch (c) {
        case 4: dit_step<4, INV>(tile, TW, k, lp, cn, S, x, y, by); break;
        case 3: dit_step<3, INV>(tile, TW, k, lp, cn, S, x, y, by); break;
        case 2: dit_step<2, INV>(tile, TW, k, lp, cn, S, x, y, by); break;
        default: dit_step<1, INV>(tile, TW, k, lp, cn, S, x, y, by); break;
        }
        lp += c;
    }
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one, each XCD has its own L2).  Tiles that are
// neighbours in memory -- the column chunks of the same rows, whose row segments share cache lines when the row length is
// not a multiple of 128 bytes -- are consecutive logical indices; this maps consecutive logical indices to one XCD, so the
// second half of a straddled line is found (reads) or completed (writes) in the same L2.
__device__ __forceinline__ u32 xcd_local_block() {
    const u32 b = blockIdx.x, per = gridDim.x >> 3;
    return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
}

struct PassParams {
    const u64 *src; u64 *dst;
    const u64 *tw;                  // pow256 table of the transform's 2^32-th root (forward or inverse)
    const u64 *twK;                 // twK[j] = w_1024^j of the same direction (tile twiddles: w_(2^k)^j = twK[j << (10-k)])
    u64 C;                          // matrix columns
    u64 tStride, gStride, hiStride; // words between tile rows / slot groups / tiles along the outer index
    u64 scale;                      // 0: none; else every output is multiplied by it (1/N of an inverse transform)
    u32 k, logM, hasTw, dit;
    u32 Wc, nbT;                    // tile slots S = nbT*Wc : nbT adjacent groups x Wc columns
    u32 nColChunks, nGroupTiles;
    u32 n, scatter;                 // scatter: store row bitrev_n(g*2^k + t) (lo = 0 pass of a natural-order transform)
    u32 canonOut;                   // 1: this pass writes a transform's result (canonical values); 0: the next pass takes any representative
};

// KC = 0: any geometry; KC = 8: the geometry of the wide matrices (8 stages, SC = 15 or 16 column slots x 16 sub-transform lanes,
// one slot group: 16 slots for rows of 128 bytes and more in 16-column chunks, 15 for the 100-column matrices whose seven
// chunks are 15 wide), with every stride, LDS offset and twiddle index a compile-time constant
template <bool INV, bool DIT, int KC, int SC = 16>
__global__ void __launch_bounds__(KC ? 256 : NTT_MAXTHREADS) ntt_pass_kernel(PassParams P) {
    extern __shared__ u64 lds[];
    const u32 k = KC ? KC : P.k, K = 1u << k, S = KC ? SC : blockDim.x, by = KC ? 16 : blockDim.y;
    const u32 x = threadIdx.x, y = threadIdx.y, tid = y * S + x, nth = S * by;
    constexpr bool PAD = KC != 0;                   // one spare tile row per 16 (see dif_step)
    const u32 tileRows = PAD ? K + (K >> 4) : K;
    u64 *tile = lds, *TW = tile + (size_t)S * tileRows, *TWO = TW + K;
#define TROW(t_) (PAD ? (t_) + ((t_) >> 4) : (t_))

    u32 bid = xcd_local_block();
    const u32 cc = bid % P.nColChunks; bid /= P.nColChunks;
    const u32 gt = bid % P.nGroupTiles;
    const u32 hi = bid / P.nGroupTiles;
    const u32 gi = KC ? 0 : x / P.Wc, ci = x - gi * P.Wc;
    const u64 c = (u64)cc * P.Wc + ci;
    const bool valid = c < P.C;
    // slot groups: adjacent blocks; in the scattering pass blocks nGroupTiles apart, whose bit-reversed rows are adjacent
    const u64 g = P.scatter ? (u64)gi * P.nGroupTiles + gt : (u64)gt * P.nbT + gi;
    const u64 base = (u64)hi * P.hiStride + g * P.gStride + c;

    // the first LOADB rows of this lane are requested before the tables are built, so that their latency overlaps it;
    // all loads of a batch are in flight together (one load per iteration would serialise the HBM latency)
    constexpr u32 LOADB = 16;
    u64 vin[LOADB];
#pragma unroll
    for (u32 i = 0; i < LOADB; i++) { const u32 t = y + i * by; vin[i] = (valid && t < K) ? P.src[base + (u64)t * P.tStride] : 0; }
    for (u32 j = tid; j < K; j += nth) TW[j] = P.twK[j << (10 - k)];
    if (P.hasTw) {
        for (u32 idx = tid; idx < (KC ? 1 : P.nbT) * K; idx += nth) {
            u32 b = gt * (KC ? 1 : P.nbT) + (idx >> k);
            u64 v = root_pow(P.tw, P.logM, b * bitrev32(idx & (K - 1), k));
            if (P.scale) v = mul(v, P.scale);
            TWO[idx] = v;
        }
    }
    __syncthreads();
    for (u32 t0 = y;;) {
#pragma unroll
        for (u32 i = 0; i < LOADB; i++) {
            const u32 t = t0 + i * by;
            if (t < K) {
                u64 v = vin[i];
                if (DIT && P.hasTw) v = NTT_MUL(v, TWO[gi * K + t]);
                tile[TROW(t) * S + x] = v;
            }
        }
        t0 += LOADB * by;
        if (t0 >= K) break;
#pragma unroll
        for (u32 i = 0; i < LOADB; i++) { const u32 t = t0 + i * by; vin[i] = (valid && t < K) ? P.src[base + (u64)t * P.tStride] : 0; }
    }
    __syncthreads();
    if constexpr (KC == 8) {
        if (DIT) { dit_step<4, INV, true>(tile, TW, 8, 0, 4, SC, x, y, 16); dit_step<4, INV, true>(tile, TW, 8, 4, 0, SC, x, y, 16); }
        else { dif_step<4, INV, true>(tile, TW, 8, 8, SC, x, y, 16); dif_step<4, INV, true>(tile, TW, 8, 4, SC, x, y, 16); }
    } else {
        if (DIT) dit_stages<INV>(tile, TW, k, S, x, y, by); else dif_stages<INV>(tile, TW, k, S, x, y, by);
    }
    if (!valid) return;
    for (u32 t = y; t < K; t += by) {
        u64 v = tile[TROW(t) * S + x];
        if (P.hasTw && !DIT) v = P.canonOut ? mul(v, TWO[gi * K + t]) : NTT_MUL(v, TWO[gi * K + t]);
        else if (!P.hasTw && P.scale) v = mul(v, P.scale);
        else if (P.canonOut) v = canon(v);
        u64 addr = P.scatter ? (u64)bitrev32((u32)(g * K + t), P.n) * P.C + c : base + (u64)t * P.tStride;
        P.dst[addr] = v;
    }
#undef TROW
}

struct LdeParams {
    const u64 *src; u64 *dst;
    const u64 *twi, *twf, *pow7;    // pow256 tables: inverse root, forward root, coset shift 7
    const u64 *twKi, *twKf;         // w_1024^-j, w_1024^j (tile twiddles)
    u64 C, ninv;
    u32 n, k, extBits;
    u32 canonOut;                   // 1: no pass follows (n <= k): the stored values are the result
    u32 cosetBegin, cosetCount;     // this call produces cosets [cosetBegin, cosetBegin+cosetCount) of the 2^extBits (multi-GPU: one slice per rank)
    u32 coefIn;                     // 1: src already holds COEFFICIENTS, coefficient m at row bitrev_n(m) (what the inverse passes + the stages below leave): no inverse stages here
    u32 Wc, G, nColChunks;
};

// Finishes the iNTT on bits [0,k), scales by the coset factors and starts the forward NTT (see header).
// SC = 0: any geometry; SC > 0: 8 stages, SC column slots x 16 sub-transform lanes, one slot group (EPT = 16) -- the geometry
// of the wide matrices, with compile-time strides
template <int EPT, int SC>
__global__ void __launch_bounds__(SC ? 512 : LDE_MAXTHREADS) lde_mid_kernel(LdeParams P) {
    extern __shared__ u64 lds[];
    const u32 k = SC ? 8 : P.k, K = 1u << k, S = SC ? SC : blockDim.x, by = SC ? 16 : blockDim.y;
    const u32 x = threadIdx.x, y = threadIdx.y, tid = y * S + x, nth = S * by;
    constexpr bool PAD = SC != 0;                   // one spare tile row per 16 (see dif_step)
    const u32 tileRows = PAD ? K + (K >> 4) : K, rowStep = PAD ? by + 1 : by;      // rows y + i*by -> y + i*(by+1) when padded (by = 16)
    u64 *tile = lds, *TWi = tile + (size_t)S * tileRows, *TWf = TWi + K, *Sc = TWf + K, *Uc = Sc + (size_t)P.G * K;

    u32 bid = xcd_local_block();
    const u32 cc = bid % P.nColChunks;
    const u32 gt = bid / P.nColChunks;
    const u32 gi = SC ? 0 : x / P.Wc, ci = x - gi * P.Wc;
    const u64 c = (u64)cc * P.Wc + ci;
    const bool valid = c < P.C;
    const u64 g = (u64)gt * (SC ? 1 : P.G) + gi;    // index of this slot's 2^k-row block
    const u64 base = g * K * P.C + c;

    u64 coef[EPT];                                  // EPT >= K / by rows per lane: all loads in flight while the tables are built
#pragma unroll
    for (int i = 0; i < EPT; i++) { const u32 t = y + i * by; coef[i] = (valid && t < K) ? P.src[base + (u64)t * P.C] : 0; }
    for (u32 j = tid; j < K; j += nth) { TWi[j] = P.twKi[j << (10 - k)]; TWf[j] = P.twKf[j << (10 - k)]; }
    for (u32 idx = tid; idx < P.G * K; idx += nth) {
        u32 pos = (gt * P.G + (idx >> k)) * K + (idx & (K - 1));
        u32 m = bitrev32(pos, P.n);                 // this row holds coefficient m of the column polynomial
        Uc[idx] = root_pow(P.twf, P.n + P.extBits, m);   // w_E^m : step from coset j to j+1
        u64 s0 = P.pow7 ? mul(P.ninv, pow256(P.pow7, m)) : P.ninv;    // shift^m / N  (coset j = 0; shift 7, or 1 when pow7 is null)
        if (P.cosetBegin) s0 = mul(s0, root_pow(P.twf, P.n + P.extBits, m * P.cosetBegin));   // (w_E^m)^cosetBegin, m*cb < 2^(n+b)
        Sc[idx] = s0;
    }
    if (!P.coefIn) {                                // (uniform for the launch)
#pragma unroll
        for (int i = 0; i < EPT; i++) { const u32 t = y + i * by; if (t < K) tile[(y + i * rowStep) * S + x] = coef[i]; }
        __syncthreads();
        if constexpr (SC != 0) { dif_step<4, true, true>(tile, TWi, 8, 8, SC, x, y, 16); dif_step<4, true, true>(tile, TWi, 8, 4, SC, x, y, 16); }
        else dif_stages<true>(tile, TWi, k, S, x, y, by);
#pragma unroll
        for (int i = 0; i < EPT; i++) { u32 t = y + i * by; coef[i] = t < K ? tile[(y + i * rowStep) * S + x] : 0; }
    }
    __syncthreads();
    const u32 nCosets = P.cosetCount;
    for (u32 j = 0; j < nCosets; j++) {
        // the per-row offsets do not depend on j: left alone the compiler computes them all once and keeps ~4 registers per
        // row alive across the loop; the opaque copies make it redo the few additions in every coset instead
        u64 dstOff = ((g * K + y) * P.cosetCount + j) * P.C + c;
        u32 tileOff = y * S + x, scOff = gi * K + y;
        asm volatile("" : "+v"(dstOff), "+v"(tileOff), "+v"(scOff));
        const u64 dstStep = (u64)by * P.cosetCount * P.C;
#pragma unroll
        for (int i = 0; i < EPT; i++) { u32 t = y + i * by; if (t < K) tile[tileOff + i * rowStep * S] = NTT_MUL(coef[i], Sc[scOff + i * by]); }
        __syncthreads();
        if constexpr (SC != 0) { dit_step<4, false, true>(tile, TWf, 8, 0, 4, SC, x, y, 16); dit_step<4, false, true>(tile, TWf, 8, 4, 0, SC, x, y, 16); }
        else dit_stages<false>(tile, TWf, k, S, x, y, by);
        asm volatile("" : "+v"(dstOff), "+v"(tileOff));
        if (valid) {
#pragma unroll
            for (int i = 0; i < EPT; i++) {
                u32 t = y + i * by;
                if (t < K) { const u64 v = tile[tileOff + i * rowStep * S]; P.dst[dstOff + i * dstStep] = P.canonOut ? canon(v) : v; }
            }
        }
        for (u32 idx = tid; idx < P.G * K; idx += nth) Sc[idx] = mul(Sc[idx], Uc[idx]);
        __syncthreads();
    }
}

__global__ void copy_rows_kernel(const u64 *src, u64 *dst, u64 nWords) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nWords) dst[i] = src[i];
}
__global__ void broadcast_row_kernel(const u64 *src, u64 *dst, u64 C, u64 rows) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) dst[i] = src[i % C];
}

// ---------------------------------------------------------------- host-side planning
using namespace pil2gl;

u32 env_u32(const char *name, u32 dflt) { const char *s = getenv(name); return s ? (u32)atoi(s) : dflt; }
u32 pow2floor(u64 v) { u32 r = 1; while ((u64)r * 2 <= v) r *= 2; return r; }

struct Geom { u32 Wc, nbT, nColChunks, S, by; };
// tile slots for 2^k rows: at most maxElems u64 in LDS, S*by = up to nThreads threads
Geom make_geom(u32 k, u64 C, u64 totalGroups, u32 maxElems, u32 nThreads) {
    Geom g;
    u32 smax = std::min<u32>(256, std::max<u32>(1, maxElems >> k));
    if (C <= smax) {
        g.Wc = (u32)C;
        g.nbT = (u32)std::min<u64>(pow2floor(smax / C), totalGroups);
        g.nColChunks = 1;
    } else {
        u32 chunks = (u32)((C + smax - 1) / smax);
        g.Wc = (u32)((C + chunks - 1) / chunks);
        // 8-stage tiles have fixed-geometry kernels for 15 and 16 slots only: chunks of 16 with a ragged last chunk beat evenly cut chunks
        // on the any-geometry kernels (81 columns: 16 x 5 + 1 instead of 14 x 5 + 11: -11 %), and 16 beats 15 where both give the same
        // number of chunks (100 columns, 16 x 6 + 4 against 15 x 6 + 10: 194.7 against 198.1 ms per config-3 interpolate, four same-box
        // pairs; other widths +-1 %: profiles/r05_lde_chunks_15_16.txt).  PIL2GL_NTT_WC15=1: 15 where it fits, as rounds 2-4 had it.
        if (k == 8 && smax == 16 && !env_u32("PIL2GL_NTT_EVEN_CHUNKS", 0)) g.Wc = ((C + 14) / 15 == chunks && env_u32("PIL2GL_NTT_WC15", 0)) ? 15 : 16;
        g.nbT = 1;
        g.nColChunks = (u32)((C + g.Wc - 1) / g.Wc);
    }
    g.S = g.Wc * g.nbT;
    u32 nSub = 1u << (k - next_chunk(k));          // sub-transforms per column in the widest register step
    g.by = std::max<u32>(1, std::min<u32>(nThreads / g.S, nSub));
    return g;
}

// stages per pass: a tile row is S*8 contiguous bytes and S <= 256 / 2^(k-4), so 8 stages keep 128-byte segments for
// wide matrices; narrow ones (rows under 128 bytes) take 7 so that adjacent rows fill the segment
u32 pick_kmax(u64 C) {
    u32 dflt = C * 8 >= 128 ? 8 : 7;
    return std::min<u32>(10, std::max<u32>(1, env_u32("PIL2GL_NTT_KMAX", dflt)));
}

int set_lds(const void *fn, size_t bytes) {
    if (bytes > 160 * 1024) return fail(PIL2GL_EINVAL, "tile needs %zu bytes of LDS (>160 KiB)", bytes);
    if (bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PIL2GL_OK;
}

// One pass over index bits [lo, lo+k) of a 2^n x C matrix.
int launch_pass(const u64 *src, u64 *dst, u64 C, u32 n, u32 lo, u32 k, bool dit, bool inverse, u64 scale, bool scatter, bool canonOut, hipStream_t st) {
    PassParams P;
    P.src = src; P.dst = dst; P.canonOut = canonOut;
    P.tw = inverse ? tables().powWi : tables().powW;
    P.twK = inverse ? tables().tw1024i : tables().tw1024;
    P.C = C; P.scale = scale; P.k = k; P.logM = lo + k; P.hasTw = lo > 0; P.dit = dit; P.n = n; P.scatter = scatter;
    u64 totalGroups;
    u32 nHi;
    if (lo > 0) { P.tStride = (C << lo); P.gStride = C; P.hiStride = (C << (lo + k)); totalGroups = 1ull << lo; nHi = 1u << (n - lo - k); }
    else { P.tStride = C; P.gStride = (C << k); P.hiStride = 0; totalGroups = 1ull << (n - k); nHi = 1; }
    Geom g = make_geom(k, C, totalGroups, env_u32("PIL2GL_NTT_TILE", 4096), std::min<u32>(NTT_MAXTHREADS, env_u32("PIL2GL_NTT_THREADS", 256)));
    P.Wc = g.Wc; P.nbT = g.nbT; P.nColChunks = g.nColChunks; P.nGroupTiles = (u32)(totalGroups / g.nbT);
    u64 K = 1ull << k;
    const bool fixedGeom = k == 8 && (g.S == 16 || g.S == 15) && g.by == 16 && g.nbT == 1 && g.Wc == g.S && !env_u32("PIL2GL_NTT_GENERIC", 0);
    size_t ldsBytes = 8 * ((size_t)g.S * (fixedGeom ? K + K / 16 : K) + K + (P.hasTw ? (size_t)g.nbT * K : 0));
    u64 blocks = (u64)nHi * P.nGroupTiles * P.nColChunks;
    if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");
    if (dit && inverse) return fail(PIL2GL_EINVAL, "no inverse decimation-in-time pass");
#define PASS_CASE(INV_, DIT_)                                                                                     \
    { if (fixedGeom && g.S == 16) {                                                                               \
          P2_TRY(set_lds((const void *)ntt_pass_kernel<INV_, DIT_, 8, 16>, ldsBytes));                               \
          hipLaunchKernelGGL((ntt_pass_kernel<INV_, DIT_, 8, 16>), dim3((unsigned)blocks), dim3(16, 16), ldsBytes, st, P); \
      } else if (fixedGeom) {                                                                                      \
          P2_TRY(set_lds((const void *)ntt_pass_kernel<INV_, DIT_, 8, 15>, ldsBytes));                               \
          hipLaunchKernelGGL((ntt_pass_kernel<INV_, DIT_, 8, 15>), dim3((unsigned)blocks), dim3(15, 16), ldsBytes, st, P); \
      } else {                                                                                                     \
          P2_TRY(set_lds((const void *)ntt_pass_kernel<INV_, DIT_, 0>, ldsBytes));                                   \
          hipLaunchKernelGGL((ntt_pass_kernel<INV_, DIT_, 0>), dim3((unsigned)blocks), dim3(g.S, g.by), ldsBytes, st, P); } }
    if (dit) PASS_CASE(false, true) else if (inverse) PASS_CASE(true, false) else PASS_CASE(false, false)
#undef PASS_CASE
    KERNEL_CHECK();
    return PIL2GL_OK;
}

// split `bits` into ceil(bits/kmax) nearly equal passes
int split_bits(u32 bits, u32 kmax, u32 *ks) {
    if (bits == 0) return 0;
    if (const char *e = getenv("PIL2GL_NTT_SPLIT")) {          // experiments: "8,8,2" is taken when it adds up to `bits`
        u32 tot = 0; int np = 0; u32 tmp[32];
        for (const char *q = e; *q && np < 32;) { tmp[np] = (u32)atoi(q); tot += tmp[np++]; while (*q && *q != ',') q++; if (*q) q++; }
        if (tot == bits) { for (int i = 0; i < np; i++) ks[i] = tmp[i]; return np; }
    }
    u32 np = (bits + kmax - 1) / kmax, base = bits / np, extra = bits % np;
    for (u32 i = 0; i < np; i++) ks[i] = base + (i < extra ? 1 : 0);
    return (int)np;
}

// passes of a standalone transform.  Where the 8-bit pass has its fixed-geometry kernel (tiles of 15 / 16 slots) as many passes as possible take
// 8 bits and ONE takes the remainder, if that is at least 4 bits: 2^20 x 100 in 8,8,4 instead of 7,7,6 -17 %, 2^22 in 8,8,6 instead of 8,7,7 -5 %;
// a remainder of 1..3 bits (a whole sweep for almost no arithmetic) stays balanced: 2^26 in 8,8,8,2 loses 8 % (profiles/r05_lde_mid8_planner.txt)
int plan_passes(u32 n, u32 kmax, u64 C, u32 *ks) {
    const u32 r = n & 7;
    if (kmax == 8 && n > 16 && (r == 0 || r >= 4) && !getenv("PIL2GL_NTT_SPLIT") && !env_u32("PIL2GL_NTT_GENERIC", 0) && !env_u32("PIL2GL_NTT_BALANCED", 0)) {
        const Geom g = make_geom(8, C, 1ull << (n - 8), env_u32("PIL2GL_NTT_TILE", 4096), std::min<u32>(NTT_MAXTHREADS, env_u32("PIL2GL_NTT_THREADS", 256)));
        if ((g.S == 15 || g.S == 16) && g.by == 16 && g.nbT == 1 && g.Wc == g.S) {
            int np = 0;
            for (u32 i = 0; i < n / 8; i++) ks[np++] = 8;
            if (r) ks[np++] = r;
            return np;
        }
    }
    return split_bits(n, kmax, ks);
}

}  // namespace

namespace pil2gl {

int ntt_launch(const u64 *src, u64 C, u32 n, u64 *dst, bool inverse, hipStream_t st) {
    if (C == 0) return PIL2GL_OK;
    u64 N = 1ull << n;
    if (n == 0) {
        if (src != dst) HIP_TRY(hipMemcpyAsync(dst, src, C * 8, hipMemcpyDeviceToDevice, st));
        return PIL2GL_OK;
    }
    u32 kmax = pick_kmax(C);
    u32 ks[32];
    int np = plan_passes(n, kmax, C, ks);
    u64 scale = inverse ? h_inv(N % 0xFFFFFFFF00000001ull) : 0;
    if (np == 1) return launch_pass(src, dst, C, n, 0, n, false, inverse, scale, true, true, st);
    u64 *tmp;
    P2_TRY(scratch(0, N * C, &tmp));
    u32 lo = n;
    for (int i = 0; i < np; i++) {
        lo -= ks[i];
        const u64 *in = i == 0 ? src : tmp;
        u64 *out = i == np - 1 ? dst : tmp;
        P2_TRY(launch_pass(in, out, C, n, lo, ks[i], false, inverse, i == 0 ? scale : 0, i == np - 1, i == np - 1, st));
    }
    return PIL2GL_OK;
}

int lde_launch(const u64 *src, u64 C, u32 n, u64 *dst, u32 nExt, hipStream_t st, u32 cosetBegin, u32 cosetCount, u64 *work, bool unitShift, bool coefIn) {
    if (C == 0) return PIL2GL_OK;
    u32 eb = nExt - n;
    if (cosetCount == 0) { cosetBegin = 0; cosetCount = 1u << eb; }
    u64 N = 1ull << n, E = 1ull << nExt;
    if (n == 0) {                       // constant polynomial: every coset point evaluates to it
        (void)E;
        broadcast_row_kernel<<<(unsigned)(((u64)cosetCount * C + 255) / 256), 256, 0, st>>>(src, dst, C, cosetCount);
        KERNEL_CHECK();
        return PIL2GL_OK;
    }
    // the forward passes run on dst viewed as N x (C * cosets): a narrow matrix (rows under 128 bytes) that becomes wide there takes
    // 8-stage passes on that side (PIL2GL_LDE_WIDEFWD=0: the narrow matrix's 7 everywhere, as before)
    u32 kmax = pick_kmax(C);
    const u32 kmaxF = env_u32("PIL2GL_LDE_WIDEFWD", 1) ? std::max(kmax, pick_kmax(C * cosetCount)) : kmax;
    u32 nfp = (n + kmaxF - 1) / kmaxF;
    u32 kf = (n + nfp - 1) / nfp;       // bits done by the mid kernel (both directions)
    // The balanced split gives the mid kernel 6 or 7 bits whenever n is not 8 * passes (n = 17..21, 25..29): its fixed-geometry form
    // (8 bits, 15 / 16 slots, coefficients in registers over the cosets) is worth more than balance -- 8-17 % of an interpolate at
    // 2^17..2^28 rows x 16 / 32 / 64 / 100 columns, full extensions and single cosets alike (profiles/r05_lde_mid8_planner.txt);
    // widths whose tiles are not 15 or 16 slots keep the balanced split (20 columns lose 2-4 % with 8).  PIL2GL_LDE_KF: experiments.
    if (kf < 8 && n > 8 && kmaxF == 8 && !env_u32("PIL2GL_NTT_GENERIC", 0)) {
        const Geom g8 = make_geom(8, C, 1ull << (n - 8), env_u32("PIL2GL_LDE_TILE", 4096), std::min<u32>(LDE_MAXTHREADS, env_u32("PIL2GL_LDE_THREADS", 512)));
        if ((g8.S == 15 || g8.S == 16) && g8.by == 16 && g8.nbT == 1) kf = 8;
    }
    if (const char *e = getenv("PIL2GL_LDE_KF")) { const u32 f = (u32)atoi(e); kf = f ? std::min(f, n) : (n + nfp - 1) / nfp; }      // 0: the balanced split
    // 1. iNTT, decimation in frequency, bits [kf, n) from the top down: src -> tmp, then in place
    const u64 *coef = src;
    if (n > kf && !coefIn) {
        u32 ks[32];
        int np = split_bits(n - kf, kmax, ks);
        u64 *tmp = work;                            // caller's workspace (may be src itself: every pass is in place)
        if (!tmp) P2_TRY(scratch(0, N * C, &tmp));
        u32 lo = n;
        for (int i = 0; i < np; i++) {
            lo -= ks[i];
            P2_TRY(launch_pass(i == 0 ? src : tmp, tmp, C, n, lo, ks[i], false, true, 0, false, false, st));
        }
        coef = tmp;
    }
    // 2. mid kernel: last kf iNTT stages + coset scaling + first kf NTT stages, tmp -> dst
    {
        LdeParams P;
        P.src = coef; P.dst = dst; P.twi = tables().powWi; P.twf = tables().powW; P.twKi = tables().tw1024i; P.twKf = tables().tw1024; P.pow7 = unitShift ? nullptr : tables().pow7;
        P.C = C; P.ninv = coefIn ? 1 : h_inv(N % 0xFFFFFFFF00000001ull); P.n = n; P.k = kf; P.extBits = eb; P.coefIn = coefIn ? 1 : 0;
        P.cosetBegin = cosetBegin; P.cosetCount = cosetCount; P.canonOut = n > kf ? 0 : 1;
        u64 totalGroups = 1ull << (n - kf);
        u32 nThreads = std::min<u32>(LDE_MAXTHREADS, env_u32("PIL2GL_LDE_THREADS", 512));
        // LDS = tile (S*K) + two local twiddle tables (K) + coset scale tables (2*G*K); narrow matrices
        // (small C => many row groups per tile) are dominated by the scale tables, so shrink until it fits
        u64 K = 1ull << kf;
        u32 maxElems = env_u32("PIL2GL_LDE_TILE", 4096);
        Geom g = make_geom(kf, C, totalGroups, maxElems, nThreads);
        size_t ldsBytes = 8 * ((size_t)g.S * K + 2 * K + 2 * (size_t)g.nbT * K);
        while (ldsBytes > 96 * 1024 && g.nbT > 1) {
            maxElems /= 2;
            g = make_geom(kf, C, totalGroups, maxElems, nThreads);
            ldsBytes = 8 * ((size_t)g.S * K + 2 * K + 2 * (size_t)g.nbT * K);
        }
        P.Wc = g.Wc; P.G = g.nbT; P.nColChunks = g.nColChunks;
        const bool fixedGeom = kf == 8 && (g.S == 15 || g.S == 16) && g.by == 16 && g.nbT == 1 && !env_u32("PIL2GL_NTT_GENERIC", 0);
        if (fixedGeom) ldsBytes += 8 * (size_t)g.S * (K / 16);
        u32 need = (u32)((K + g.by - 1) / g.by);
        u64 blocks = (totalGroups / g.nbT) * g.nColChunks;
        if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");
        dim3 grid((unsigned)blocks), block(g.S, g.by);
#define LDE_CASE(E_)                                                                              \
        if (need <= E_) {                                                                         \
            P2_TRY(set_lds((const void *)lde_mid_kernel<E_, 0>, ldsBytes));                        \
            hipLaunchKernelGGL((lde_mid_kernel<E_, 0>), grid, block, ldsBytes, st, P);             \
        } else
#define LDE_FIXED(S_)                                                                             \
        if (fixedGeom && g.S == S_) {                                                             \
            P2_TRY(set_lds((const void *)lde_mid_kernel<16, S_>, ldsBytes));                       \
            hipLaunchKernelGGL((lde_mid_kernel<16, S_>), grid, block, ldsBytes, st, P);            \
        } else
        LDE_FIXED(15) LDE_FIXED(16)
        LDE_CASE(1) LDE_CASE(2) LDE_CASE(4) LDE_CASE(8) LDE_CASE(16) LDE_CASE(32) LDE_CASE(64)
        { return fail(PIL2GL_EINVAL, "lde tile too tall for the thread block (need %u rows per thread)", need); }
#undef LDE_CASE
#undef LDE_FIXED
        KERNEL_CHECK();
    }
    // 3. remaining forward stages, decimation in time, bits [kf, n) from the bottom up, in place on
    //    dst viewed as N x (C * 2^eb)
    if (n > kf) {
        u32 ks[32];
        int np = split_bits(n - kf, kmaxF, ks);
        u32 lo = kf;
        for (int i = 0; i < np; i++) {
            P2_TRY(launch_pass(dst, dst, C * cosetCount, n, lo, ks[i], true, false, 0, false, i == np - 1, st));
            lo += ks[i];
        }
    }
    return PIL2GL_OK;
}

}  // namespace pil2gl

using namespace pil2gl;

static int check_ntt_args(const void *src, const void *dst, uint32_t nBits, uint32_t nBitsExt) {
    if (!src || !dst) return fail(PIL2GL_EINVAL, "null buffer");
    // the reference's transforms take any nBits <= 32 (f3g.js:40, fft.js:39-50); here row indices, twiddle exponents and grid sizes are
    // 32-bit quantities that have been checked to 2^30 rows (four passes of 8/8/7/7 stages; 2^30 x 1 column is 8.6 GB each way)
    if (nBits > PIL2GL_MAX_NTT_BITS || nBitsExt > PIL2GL_MAX_NTT_BITS) return fail(PIL2GL_EINVAL, "domain of 2^%u rows is not supported (max 2^%d)", nBits > nBitsExt ? nBits : nBitsExt, PIL2GL_MAX_NTT_BITS);
    if (nBitsExt < nBits) return fail(PIL2GL_EINVAL, "nBitsExt (%u) < nBits (%u)", nBitsExt, nBits);
    return PIL2GL_OK;
}

// a coset slice only ever indexes 2^nBits * cosetCount local rows: the full extension may be larger than one device holds
// ---- the reference's worker-level operators (fft_worker.js:6-67), for a caller that keeps fft_p.js's own block loop ----------------
// The product path never runs these (pil2gl_fft / _interpolate replace the whole loop: bit reversal, blocks and transposes); they
// exist so that `pool.exec("fft_block", ...)` / `("interpolatePrepareBlock", ...)` have a device twin with the same arguments.
// One launch per stage, twiddles straight from the pow256 table.
//
// _fft_block (fft_worker.js:21-60) on a buffer of 2^blockBits rows that sits at row start_pos of an n-row transform: the recursion
// first cuts the block into pieces of 2^layers rows (:30-34), then runs a decimation-in-time transform of `layers` stages on each
// piece (:35-38); stage l = 1..layers pairs rows 2^(l-1) apart inside sub-blocks of 2^l rows and is the recursion level whose
// (s, blockBits, layers) are (s - layers + l, l, l).  Its twiddle for pair i of the sub-block at absolute row sp (:40-58):
//   w = w0 * F.w[l]^i,   w0 = F.w[s - layers + l]^p  if s > layers, else 1,   width = 2^(s - layers), height = n / width,
//   p = (sp % height) * width + floor(sp / height)
// (s > layers and width are the same at every level: s and the level's block size fall together.)
namespace {
__global__ void fft_block_stage_kernel(u64 *__restrict__ buf, u64 startPos, u64 nPols, u32 nBits, u32 s, u32 layers, u32 l, u64 nPairs,
                                       const u64 *__restrict__ powW) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nPairs * nPols) return;
    const u64 g = idx / nPols, c = idx - g * nPols;
    const u64 half = 1ull << (l - 1), i = g & (half - 1), rel = (g >> (l - 1)) << l;
    u64 w = root_pow(powW, l, (u32)i);
    if (s > layers) {
        const u32 sl = s - layers + l, wBits = s - layers;               // sl <= s <= 32
        const u64 sp = startPos + rel, hBits = nBits - wBits;             // height = 2^hBits rows
        const u64 pe = ((sp & ((1ull << hBits) - 1)) << wBits) + (sp >> hBits);
        w = mul(w, root_pow(powW, sl, (u32)(pe & ((1ull << sl) - 1))));   // the root has order 2^sl
    }
    u64 *a = buf + (rel + i) * nPols + c, *b = buf + (rel + half + i) * nPols + c;
    const u64 u = canon(*a), t = mul(w, *b);
    *a = add(u, t);
    *b = sub(u, t);
}
// interpolatePrepareBlock (fft_worker.js:6-19): row i of the block times start * inc^i
__global__ void interpolate_prepare_block_kernel(u64 *__restrict__ buf, u64 width, u64 height, u64 start, u64 inc) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height) return;
    buf[idx] = mul(buf[idx], mul(start, pow(inc, idx / width)));
}
}  // namespace

static int check_coset_args(const void *src, const void *dst, uint32_t nBits, uint32_t nBitsExt, uint32_t cosetBegin, uint32_t cosetCount) {
    if (!src || !dst) return fail(PIL2GL_EINVAL, "null buffer");
    if (nBitsExt < nBits) return fail(PIL2GL_EINVAL, "nBitsExt (%u) < nBits (%u)", nBitsExt, nBits);
    if (nBits > PIL2GL_MAX_NTT_BITS || nBitsExt > 31) return fail(PIL2GL_EINVAL, "domain of 2^%u rows (extended 2^%u) is not supported", nBits, nBitsExt);
    if (cosetCount == 0 || (uint64_t)cosetBegin + cosetCount > (1ull << (nBitsExt - nBits))) return fail(PIL2GL_EINVAL, "coset range [%u,%u) outside 2^%u", cosetBegin, cosetBegin + cosetCount, nBitsExt - nBits);
    if (((uint64_t)cosetCount << nBits) > (1ull << PIL2GL_MAX_NTT_BITS)) return fail(PIL2GL_EINVAL, "slice of %u cosets x 2^%u rows exceeds 2^%d local rows", cosetCount, nBits, PIL2GL_MAX_NTT_BITS);
    return PIL2GL_OK;
}

extern "C" {

int pil2gl_interpolate_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_ntt_args(src, dst, nBits, nBitsExt));
    return lde_launch(src, nPols, nBits, dst, nBitsExt, as_stream(stream), 0, 0, nullptr, false);
}
int pil2gl_interpolate_cosets_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                  uint32_t cosetBegin, uint32_t cosetCount, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_coset_args(src, dst, nBits, nBitsExt, cosetBegin, cosetCount));
    return lde_launch(src, nPols, nBits, dst, nBitsExt, as_stream(stream), cosetBegin, cosetCount, nullptr, false);
}
// the same slice of the PLAIN extension (evaluations on w_E^j <w_N>, no coset shift): rows (pos << b) + j of fft_E applied to
// the zero-padded coefficients of the columns -- how computeQStark extends its split quotient (stark_gen_helpers.js:192)
int pil2gl_extend_cosets_unshifted_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                       uint32_t cosetBegin, uint32_t cosetCount, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_coset_args(src, dst, nBits, nBitsExt, cosetBegin, cosetCount));
    return lde_launch(src, nPols, nBits, dst, nBitsExt, as_stream(stream), cosetBegin, cosetCount, nullptr, true);
}
// The evaluations, on the UNSHIFTED domain of 2^nBitsExt points in natural order, of polynomials handed over as COEFFICIENTS:
// coefBrev is a 2^nBits x nPols matrix whose row bitrev(m) holds coefficient m (the order the inverse passes leave, and the one
// pil2gl_compute_q_split_brev_dev writes) -- fft of the zero-padded coefficient matrix (stark_gen_helpers.js:192) without the
// 2^nBitsExt-row padded input and without its first nBitsExt - nBits stages.
int pil2gl_extend_coefs_brev_dev(const uint64_t *coefBrev, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_ntt_args(coefBrev, dst, nBits, nBitsExt));
    return lde_launch(coefBrev, nPols, nBits, dst, nBitsExt, as_stream(stream), 0, 0, nullptr, true, true);
}
int pil2gl_extend_coefs_brev_cosets_dev(const uint64_t *coefBrev, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                        uint32_t cosetBegin, uint32_t cosetCount, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_coset_args(coefBrev, dst, nBits, nBitsExt, cosetBegin, cosetCount));
    return lde_launch(coefBrev, nPols, nBits, dst, nBitsExt, as_stream(stream), cosetBegin, cosetCount, nullptr, true, true);
}
int pil2gl_interpolate_cosets_ws_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                     uint32_t cosetBegin, uint32_t cosetCount, uint64_t *workspace, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_coset_args(src, dst, nBits, nBitsExt, cosetBegin, cosetCount));
    if (!workspace) return fail(PIL2GL_EINVAL, "null workspace");
    return lde_launch(src, nPols, nBits, dst, nBitsExt, as_stream(stream), cosetBegin, cosetCount, workspace, false);
}
int pil2gl_fft_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_ntt_args(src, dst, nBits, nBits));
    return ntt_launch(src, nPols, nBits, dst, false, as_stream(stream));
}
int pil2gl_ifft_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, void *stream) {
    P2_TRY(ensure_init());
    P2_TRY(check_ntt_args(src, dst, nBits, nBits));
    return ntt_launch(src, nPols, nBits, dst, true, as_stream(stream));
}

static int host_wrap(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsOut, int mode) {
    P2_TRY(ensure_init());
    P2_TRY(check_ntt_args(src, dst, nBits, nBitsOut));
    uint64_t nIn = (nPols << nBits), nOut = (nPols << nBitsOut);
    if (nIn == 0) return PIL2GL_OK;
    // device copies: up to 1 GiB each they live in persistent slots (a caller looping over BigBuffers of one size -- the
    // reference's extendAndMerkelize does -- pays the allocations once); larger ones are allocated and freed per call
    const uint64_t keepWords = 1ull << 27;
    const bool keepIn = nIn <= keepWords, keepOut = nOut <= keepWords;
    uint64_t *dIn = nullptr, *dOut = nullptr;
    if (keepIn) P2_TRY(scratch(12, nIn, &dIn)); else HIP_TRY(hipMalloc((void **)&dIn, nIn * 8));
    int rc = PIL2GL_OK;
    if (keepOut) rc = scratch(13, nOut, &dOut);
    else { hipError_t e = hipMalloc((void **)&dOut, nOut * 8); if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(dst)"); }
    hipError_t e = hipSuccess;
    if (rc == PIL2GL_OK) { e = hipMemcpy(dIn, src, nIn * 8, hipMemcpyHostToDevice); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D"); }
    if (rc == PIL2GL_OK) rc = mode == 0 ? lde_launch(dIn, nPols, nBits, dOut, nBitsOut, 0, 0, 0, nullptr, false) : ntt_launch(dIn, nPols, nBits, dOut, mode == 2, 0);
    if (rc == PIL2GL_OK) { e = hipMemcpy(dst, dOut, nOut * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H"); }
    if (!keepIn) (void)hipFree(dIn);
    if (!keepOut && dOut) (void)hipFree(dOut);
    return rc;
}
int pil2gl_interpolate(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt) { return host_wrap(src, nPols, nBits, dst, nBitsExt, 0); }
int pil2gl_fft(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst) { return host_wrap(src, nPols, nBits, dst, nBits, 1); }
int pil2gl_ifft(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst) { return host_wrap(src, nPols, nBits, dst, nBits, 2); }

// fft_block(buff, start_pos, nPols, nBits, s, blockBits, layers)  fft_worker.js:62-67: buf = the 2^blockBits x nPols block, in place
int pil2gl_fft_block_dev(uint64_t *buf, uint64_t start_pos, uint64_t nPols, uint32_t nBits, uint32_t s, uint32_t blockBits, uint32_t layers, void *stream) {
    P2_TRY(ensure_init());
    // (the reference checks nothing; what is refused here has no meaning there either: F.w has 33 entries, :30-38 never end for
    // layers > blockBits, and n / width is not an integer for s - layers > nBits)
    if (nBits > 32 || s > 32 || blockBits > 32) return fail(PIL2GL_EINVAL, "fft_block: nBits %u, s %u, blockBits %u", nBits, s, blockBits);
    if (layers > blockBits) return fail(PIL2GL_EINVAL, "fft_block: %u layers in a block of 2^%u rows", layers, blockBits);
    if (s > layers && s - layers > nBits) return fail(PIL2GL_EINVAL, "fft_block: stage %u with %u layers in a transform of 2^%u rows", s, layers, nBits);
    if (layers == 0 || nPols == 0) return PIL2GL_OK;                      // (:30-34 cut the block down to single rows: nothing to pair)
    if (!buf) return fail(PIL2GL_EINVAL, "null buffer");
    const u64 nPairs = 1ull << (blockBits - 1);
    if (nPols > (1ull << 40) / nPairs) return fail(PIL2GL_EINVAL, "fft_block: block too large");
    const u64 blocks = (nPairs * nPols + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "fft_block: grid too large");
    hipStream_t st = as_stream(stream);
    for (u32 l = 1; l <= layers; l++) {
        fft_block_stage_kernel<<<(unsigned)blocks, 256, 0, st>>>(buf, start_pos, nPols, nBits, s, layers, l, nPairs, tables().powW);
        KERNEL_CHECK();
    }
    return PIL2GL_OK;
}
// interpolatePrepareBlock(buff, width, start, inc, st_i, st_n)  fft_worker.js:6-19: buf = height x width, in place
int pil2gl_interpolate_prepare_block_dev(uint64_t *buf, uint64_t width, uint64_t height, uint64_t start, uint64_t inc, void *stream) {
    P2_TRY(ensure_init());
    if (width == 0 || height == 0) return PIL2GL_OK;
    if (!buf) return fail(PIL2GL_EINVAL, "null buffer");
    if (height > (1ull << 40) / width) return fail(PIL2GL_EINVAL, "interpolatePrepareBlock: block too large");
    const u64 blocks = (width * height + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "interpolatePrepareBlock: grid too large");
    interpolate_prepare_block_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(buf, width, height, start % P, inc % P);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

}  // extern "C"
