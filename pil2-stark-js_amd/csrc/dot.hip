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

// f[r] = Horner over openings (vf1) of (acc[r][o] - K_o) * X[r][o]    (friPolinomial.js:38-50)
__global__ void fri_combine_kernel(const u64 *__restrict__ acc, const u64 *__restrict__ K, E3 vf1, const u64 *__restrict__ xdiv,
                                   u32 nOpen, u64 nRows, u64 *__restrict__ f) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nRows) return;
    E3 res = { { 0, 0, 0 } };
    for (u32 o = 0; o < nOpen; o++) {
        const u64 *a = acc + (r * nOpen + o) * 3, *x = xdiv + (r * nOpen + o) * 3;
        E3 F = { { sub(a[0], K[3 * o]), sub(a[1], K[3 * o + 1]), sub(a[2], K[3 * o + 2]) } };
        E3 X = { { x[0], x[1], x[2] } };
        E3 t = e3_mul(F, X);
        res = o == 0 ? t : e3_add(e3_mul(vf1, res), t);
    }
    f[3 * r] = res.v[0]; f[3 * r + 1] = res.v[1]; f[3 * r + 2] = res.v[2];
}

struct ColsDotParams {
    const u64 *buf; u64 width; u64 nRows; u64 rowStep;      // rows k*rowStep, k < nRows
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
    const u64 stride = P.rowStep * P.width;
    for (u64 kb = k0; kb < k1; kb += RB) {
        u64 pv[RB];
#pragma unroll
        for (int j = 0; j < RB; j++) pv[j] = (valid && kb + j < k1) ? P.buf[(kb + j) * stride + c] : 0;
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

extern "C" {

int pil2gl_rows_dot_ext_dev(const uint64_t *buf, uint64_t width, uint64_t nRows, const uint64_t *hostCoef, uint32_t nOut,
                            uint64_t *acc, int accumulate, void *stream) {
    P2_TRY(ensure_init());
    if (!buf || !hostCoef || !acc) return fail(PIL2GL_EINVAL, "null buffer");
    if (nOut < 1 || nOut > 4) return fail(PIL2GL_EINVAL, "nOut must be 1..4");
    if (width == 0 || nRows == 0) return PIL2GL_OK;
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
    for (u64 col0 = 0; col0 < width; col0 += 1024) {
        RowsDotParams P = { buf, std::min<u64>(1024, width - col0), nRows, width, col0, (const u32 *)d, nOut, acc, (u32)(accumulate != 0 || col0 != 0) };
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

int pil2gl_fri_combine_dev(const uint64_t *acc, const uint64_t *hostK, const uint64_t vf1[3], const uint64_t *xDivXSubXi,
                           uint32_t nOpen, uint64_t nRows, uint64_t *f, void *stream) {
    P2_TRY(ensure_init());
    if (!acc || !hostK || !vf1 || !xDivXSubXi || !f || nOpen < 1 || nOpen > 4) return fail(PIL2GL_EINVAL, "bad FRI combine arguments");
    hipStream_t st = as_stream(stream);
    std::vector<u64> k(hostK, hostK + 3ull * nOpen);
    u64 *d;
    P2_TRY(scratch(7, 16, &d));
    HIP_TRY(hipMemcpyAsync(d, k.data(), k.size() * 8, hipMemcpyHostToDevice, st));
    E3 v = { { vf1[0], vf1[1], vf1[2] } };
    fri_combine_kernel<<<nblk(nRows), 256, 0, st>>>(acc, d, v, xDivXSubXi, nOpen, nRows, f);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return PIL2GL_OK;
}

int pil2gl_cols_dot_ext_dev(const uint64_t *buf, uint64_t width, uint64_t nRows, uint64_t rowStep, const uint64_t *const *levs,
                            uint32_t nLev, uint64_t *hostOut, void *stream) {
    P2_TRY(ensure_init());
    if (!buf || !levs || !hostOut) return fail(PIL2GL_EINVAL, "null buffer");
    if (nLev < 1 || nLev > 4) return fail(PIL2GL_EINVAL, "nLev must be 1..4");
    if (width == 0 || nRows == 0) return PIL2GL_OK;
    hipStream_t st = as_stream(stream);
    const u32 rpc = 1024;
    const u64 nChunks = (nRows + rpc - 1) / rpc, n = (u64)nLev * width * 3;
    u64 *d;
    const u64 nGroups = std::min<u64>(64, nChunks), per = (nChunks + nGroups - 1) / nGroups;
    P2_TRY(scratch(7, ((u64)nLev * nRows * 9 * 4 + 7) / 8 + nChunks * n + nGroups * n + n + 1, &d));
    u32 *limbs = (u32 *)d;
    u64 *partial = d + ((u64)nLev * nRows * 9 * 4 + 7) / 8, *part2 = partial + nChunks * n, *res = part2 + nGroups * n;
    for (u32 l = 0; l < nLev; l++) limbs_kernel<<<nblk(nRows * 3), 256, 0, st>>>(levs[l], nRows * 3, limbs + (u64)l * nRows * 9);
    ColsDotParams P = { buf, width, nRows, rowStep, limbs, nLev, partial, rpc };
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
    HIP_TRY(hipMemcpyAsync(hostOut, res, n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PIL2GL_OK;
}

}  // extern "C"
