// FRI folding and the element-wise STARK step helpers (Goldilocks cubic extension), gfx950.
//
// Replaces src/stark/fri.js:22-81,187-202 (fold, getTransposedBuffer) and the serial BigInt loops
// of src/stark/stark_gen_helpers.js:104-160,179-190,216-264,293-322 and src/helpers/polutils.js:39-102.
#include "common.h"
#include "gl_field.cuh"
#include <vector>

using namespace gl;

namespace {

__device__ __forceinline__ u64 pow256(const u64 *__restrict__ T, u32 e) {
    u64 r = T[e & 255];
    r = mul_lazy(r, T[256 + ((e >> 8) & 255)]);
    r = mul_lazy(r, T[512 + ((e >> 16) & 255)]);
    return mul(r, T[768 + (e >> 24)]);
}
__device__ __forceinline__ u64 root_pow(const u64 *__restrict__ T, u32 logM, u32 e) { return logM ? pow256(T, e << (32 - logM)) : 1; }

__device__ __forceinline__ E3 ld3(const u64 *p) { return { { p[0], p[1], p[2] } }; }
__device__ __forceinline__ void st3(u64 *p, const E3 &v) { p[0] = v.v[0]; p[1] = v.v[1]; p[2] = v.v[2]; }

// fri.js:45-60 after the group iNTT: out[g] = sum_i coef[i][g] * (sinv_g * challenge)^i, sinv_g = shiftInv * wi^g.
// coef is the nX x (pol2N*3) matrix of iNTT'd groups (row i, column g): lanes walk consecutive g.
// With sinvOf (the verifier, fri.js:125-127: one opened group per query, each with its own position) the pol2N columns
// are unrelated groups and sinv_g = sinvOf[g]; pol2N need not be a power of two then.
__global__ void fri_horner_kernel(const u64 *__restrict__ coef, u32 polBits, u64 pol2N, u64 nX, u64 shiftInv, E3 challenge,
                                  const u64 *__restrict__ powWi, const u64 *__restrict__ sinvOf, u64 *__restrict__ out) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pol2N) return;
    const u64 sinv = sinvOf ? sinvOf[g] : mul(shiftInv, root_pow(powWi, polBits, (u32)g));
    const E3 Y = e3_scale(challenge, sinv);
    E3 acc = ld3(coef + ((nX - 1) * pol2N + g) * 3);                 // polutils.js:9-16 evalPol
    for (u64 i = nX - 1; i-- > 0;) acc = e3_add(e3_mul(acc, Y), ld3(coef + (i * pol2N + g) * 3));
    st3(out + 3 * g, acc);
}

// fri.js:187-202: out[(i*h + j)] = pol[j*w + i]
__global__ void fri_transpose_kernel(const u64 *__restrict__ pol, u32 polBits, u32 tBits, u64 *__restrict__ out) {
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;       // output element index
    const u64 n = 1ull << polBits;
    if (o >= n) return;
    const u64 h = n >> tBits;
    const u64 i = o / h, j = o - i * h;
    st3(out + 3 * o, ld3(pol + 3 * ((j << tBits) + i)));
}

__global__ void build_x_kernel(u32 nBits, u64 shift, const u64 *__restrict__ powW, u64 *__restrict__ x) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ull << nBits)) return;
    x[i] = mul(shift, root_pow(powW, nBits, (u32)i));
}
// out[i] = first * ratio^i: ratio^(2^b) by value, one product per set bit of i
struct GeomPow { u64 p[40]; };
__global__ void geometric_kernel(u64 first, GeomPow R, u64 n, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 acc = first;
    for (u32 b = 0; (i >> b) != 0; b++) if ((i >> b) & 1) acc = mul(acc, R.p[b]);
    out[i] = acc;
}
__global__ void periodic_kernel(const u64 *__restrict__ tab, u64 period, u64 n, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tab[i % period];
}

// Montgomery batch inversion of CH consecutive values per thread
template <int CH>
__device__ __forceinline__ void batch_inv(u64 v[CH]) {
    u64 pre[CH];
    u64 acc = 1;
#pragma unroll
    for (int i = 0; i < CH; i++) { pre[i] = acc; acc = mul(acc, v[i]); }
    u64 z = inv(acc);
#pragma unroll
    for (int i = CH - 1; i >= 0; i--) { u64 t = mul(z, pre[i]); z = mul(z, v[i]); v[i] = t; }
}

// polutils.js:57-71: out[i] = 1 / ((x_i - root) * ZhInv[i]) = zh[i % ext] / (x_i - root), zh = 1/ZhInv
__global__ void one_row_zerofier_kernel(u32 nBitsExt, u64 root, const u64 *__restrict__ zh, u64 ext,
                                        const u64 *__restrict__ powW, u64 *__restrict__ out) {
    constexpr int CH = 8;
    const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * CH;
    const u64 n = 1ull << nBitsExt;
    if (i0 >= n) return;
    u64 v[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) { u64 i = i0 + k < n ? i0 + k : i0; v[k] = sub(mul(7, root_pow(powW, nBitsExt, (u32)i)), root); }
    batch_inv<CH>(v);
#pragma unroll
    for (int k = 0; k < CH; k++) if (i0 + k < n) out[i0 + k] = mul(v[k], zh[(i0 + k) % ext]);
}
// polutils.js:74-102: out[i] = prod_j (x_i - roots[j])
__global__ void frame_zerofier_kernel(u32 nBitsExt, const u64 *__restrict__ roots, u32 nRoots, const u64 *__restrict__ powW, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ull << nBitsExt)) return;
    const u64 x = mul(7, root_pow(powW, nBitsExt, (u32)i));
    u64 zi = 1;
    for (u32 j = 0; j < nRoots; j++) zi = mul(zi, sub(x, roots[j]));
    out[i] = zi;
}
// stark_gen_helpers.js:179-190
__global__ void q_split_kernel(const u64 *__restrict__ qq1, u32 nBits, u32 nBitsExt, u32 qDim, u32 qDeg,
                               const u64 *__restrict__ sPow /* (7^-N)^p */, u64 *__restrict__ qq2) {
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 W = (u64)qDim * qDeg, N = 1ull << nBits;
    if (o >= (W << nBitsExt)) return;
    const u64 i = o / W, r = o - i * W;
    if (i >= N) { qq2[o] = 0; return; }
    const u64 p = r / qDim, k = r - p * qDim;
    qq2[o] = mul(qq1[(p * N + i) * qDim + k], sPow[p]);
}
// the same pieces as an N-row COEFFICIENT matrix in bit-reversed row order (row bitrev(i) = coefficient i of every piece): the input of
// pil2gl_extend_coefs_brev_dev, which then replaces the plain transform of the zero-padded matrix (stark_gen_helpers.js:192)
__global__ void q_split_brev_kernel(const u64 *__restrict__ qq1, u32 nBits, u32 qDim, u32 qDeg, const u64 *__restrict__ sPow, u64 *__restrict__ out) {
    const u64 o = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 W = (u64)qDim * qDeg, N = 1ull << nBits;
    if (o >= W * N) return;
    const u64 i = o / W, r = o - i * W;
    const u64 p = r / qDim, k = r - p * qDim;
    out[(u64)bitrev32((u32)i, nBits) * W + r] = mul(qq1[(p * N + i) * qDim + k], sPow[p]);
}
// stark_gen_helpers.js:302-322: (x_k - xi)^-1 * x_k with F.sub(scalar, triple) (f3g.js:66).
// The denominator (a, b, c) = (x_k - xi_0, -xi_1, -xi_2) varies in its first component only, so the extension inverse
// (f3g.js:136-172: adjugate (i1, i2, i3) over the norm t) is a polynomial in a with per-call constants:
//     u = a (a + 2c);  i1 = m1 - u;  t = a (i1 + 2bc) + k0;  i2 = b a - cc;  i3 = c a + (cc - bb)
// with m1 = bc + bb - cc, k0 = -b^3 + b c^2 - c^3: four multiplications, and the one inversion per row is of the BASE
// field element t.  A lane takes XD_BATCH rows k, k+T, k+2T, ... (T = threads in the grid: neighbouring lanes stay on
// neighbouring rows) and inverts their norms with one field inversion (Montgomery's trick, the device form of the
// reference's F.batchInverse :316); x advances by the constant w_E^T from row to row.
// Coset slices (one rank's rows of a sharded proof): local row l = pos * cc + jl stands for extended row (pos << eb) + cb + jl; T is
// a multiple of cc, so a lane's rows keep their coset and x still advances by one constant, w_N^(T / cc).
constexpr int XD_BATCH = 16;
struct XDivConst { u64 xi0, b, c, c2, bc2, m1, k0, cc, ccbb; };
__global__ void __launch_bounds__(256) x_div_x_sub_xi_kernel(u32 nBitsExt, XDivConst K, u64 nOpen, u64 iOpen, const u64 *__restrict__ powW, u64 wStep, u64 *__restrict__ out,
                                                             u64 E /* rows written */, u32 extBits, u32 cosetBegin, u32 ccLog) {
    const u64 T = (u64)gridDim.x * blockDim.x;
    const u64 k0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k0 >= E) return;
    u64 xs[XD_BATCH], as[XD_BATCH], ts[XD_BATCH], pre[XD_BATCH];
    const u64 g0 = ((k0 >> ccLog) << extBits) + cosetBegin + (k0 & ((1ull << ccLog) - 1));      // extended row of local row k0
    u64 x = mul(7, root_pow(powW, nBitsExt, (u32)g0));
    int n = 0;
#pragma unroll
    for (int i = 0; i < XD_BATCH; i++) {
        if (k0 + (u64)i * T < E) {
            const u64 a = sub(x, K.xi0);
            const u64 u = mul(a, add(a, K.c2));
            const u64 t = add(mul(a, add(sub(K.m1, u), K.bc2)), K.k0);
            xs[i] = x; as[i] = a; ts[i] = t;
            pre[i] = i ? mul(pre[i - 1], t) : t;
            x = mul(x, wStep);
            n = i + 1;
        }
    }
    u64 tinv = inv(pre[n - 1]);
#pragma unroll
    for (int i = XD_BATCH - 1; i >= 0; i--) {
        if (i < n) {
            const u64 ti = i ? mul(tinv, pre[i - 1]) : tinv;          // 1 / t_i
            if (i) tinv = mul(tinv, ts[i]);
            const u64 a = as[i], s = mul(ti, xs[i]);                  // x / norm
            const u64 i1 = sub(K.m1, mul(a, add(a, K.c2)));
            const u64 i2 = sub(mul(K.b, a), K.cc);
            const u64 i3 = add(mul(K.c, a), K.ccbb);
            u64 *o = out + 3 * ((k0 + (u64)i * T) * nOpen + iOpen);
            o[0] = mul(i1, s); o[1] = mul(i2, s); o[2] = mul(i3, s);
        }
    }
}
// stark_gen_helpers.js:216-229: lev[k] = xi^k; xiPow[b] = xi^(2^b)
__global__ void lev_pow_kernel(u32 nBits, const u64 *__restrict__ xiPow, u64 *__restrict__ lev) {
    const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1ull << nBits)) return;
    E3 acc = { { 1, 0, 0 } };
    for (u32 b = 0; b < nBits; b++) if ((k >> b) & 1) acc = e3_mul(acc, ld3(xiPow + 3 * b));
    st3(lev + 3 * k, acc);
}
// stark_gen_helpers.js:250-264: partial[e][blk] = sum over this block's k of v_e[k << eb] * lev[k]
__global__ void evals_partial_kernel(pil2gl_eval_desc d, const u64 *__restrict__ lev, u32 nBits, u32 eb, u64 *__restrict__ partial) {
    __shared__ u64 red[256 * 3];
    const u64 N = 1ull << nBits;
    E3 acc = { { 0, 0, 0 } };
    for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (u64)gridDim.x * blockDim.x) {
        const u64 *v = d.buf + (k << eb) * d.width + d.offset;
        const E3 l = ld3(lev + 3 * k);
        acc = e3_add(acc, d.dim == 1 ? e3_scale(l, v[0]) : e3_mul(ld3(v), l));
    }
    red[threadIdx.x * 3] = acc.v[0]; red[threadIdx.x * 3 + 1] = acc.v[1]; red[threadIdx.x * 3 + 2] = acc.v[2];
    __syncthreads();
    for (u32 s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) for (int c = 0; c < 3; c++) red[threadIdx.x * 3 + c] = add(red[threadIdx.x * 3 + c], red[(threadIdx.x + s) * 3 + c]);
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[(u64)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x];
}
__global__ void evals_final_kernel(const u64 *__restrict__ partial, u32 nBlocks, u32 nEvals, u64 *__restrict__ out) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nEvals * 3) return;
    const u32 ev = e / 3, c = e % 3;
    u64 acc = 0;
    for (u32 b = 0; b < nBlocks; b++) acc = add(acc, partial[((u64)ev * nBlocks + b) * 3 + c]);
    out[e] = acc;
}

// Witness of K independent Fibonacci machines (test/state_machines/sm_fibonacci/sm_fibonacci.js:12-23):
// l2' = l1, l1' = l1^2 + l2^2.  Sequential in the row index, so one lane per machine; bench/test support only.
__global__ void synth_fibonacci_kernel(u32 nBits, u32 nPairs, const u64 *__restrict__ init, u64 *__restrict__ cm) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nPairs) return;
    u64 l1 = canon(init[2 * k]), l2 = canon(init[2 * k + 1]);
    const u64 N = 1ull << nBits, W = 2ull * nPairs;
    for (u64 i = 0; i < N; i++) {
        cm[i * W + 2 * k] = l1; cm[i * W + 2 * k + 1] = l2;
        const u64 n1 = add(mul(l1, l1), mul(l2, l2));
        l2 = l1; l1 = n1;
    }
}

inline unsigned nblk(u64 n, u32 t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

using namespace pil2gl;

extern "C" {

int pil2gl_fri_fold_dev(const uint64_t *pol, uint32_t polBits, uint32_t outBits, uint64_t shiftInv,
                        const uint64_t challenge[3], uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (!pol || !out || !challenge) return fail(PIL2GL_EINVAL, "null buffer");
    if (outBits > polBits || polBits > PIL2GL_MAX_NTT_BITS) return fail(PIL2GL_EINVAL, "Invalid polynomial size");
    hipStream_t st = as_stream(stream);
    const u64 n = 1ull << polBits;
    u64 *coef;
    P2_TRY(scratch(1, 3 * n, &coef));
    // group iNTT (fri.js:51-55): pol is an nX x (pol2N*3) row-major matrix, transform along its rows' index
    P2_TRY(ntt_launch(pol, 3ull << outBits, polBits - outBits, coef, true, st));
    E3 ch = { { challenge[0], challenge[1], challenge[2] } };
    fri_horner_kernel<<<nblk(1ull << outBits), 256, 0, st>>>(coef, polBits, 1ull << outBits, 1ull << (polBits - outBits), shiftInv, ch, tables().powWi, nullptr, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

// FRI.verify's inner step for all queries of a layer (fri.js:121-127): groups = 2^foldBits x (nQueries*3) row-major (row i =
// element i of every query's opened group), sinv[q] = 1 / (shift * w_polBits^idx_q); out[q] = evalPol(ifft(group_q), challenge * sinv[q])
int pil2gl_fri_verify_fold_dev(const uint64_t *groups, uint32_t foldBits, uint32_t nQueries, const uint64_t *sinv,
                               const uint64_t challenge[3], uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (!groups || !sinv || !out || !challenge) return fail(PIL2GL_EINVAL, "null buffer");
    if (foldBits > 20 || nQueries == 0) return fail(PIL2GL_EINVAL, "Invalid group size or query count");
    hipStream_t st = as_stream(stream);
    const u64 nX = 1ull << foldBits;
    const u64 *coef = groups;
    if (foldBits > 0) {
        u64 *c;
        P2_TRY(scratch(1, 3 * nX * nQueries, &c));
        P2_TRY(ntt_launch(groups, 3ull * nQueries, foldBits, c, true, st));
        coef = c;
    }
    E3 ch = { { challenge[0], challenge[1], challenge[2] } };
    fri_horner_kernel<<<nblk(nQueries), 256, 0, st>>>(coef, 0, nQueries, nX, 0, ch, tables().powWi, sinv, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

int pil2gl_fri_transpose_dev(const uint64_t *pol, uint32_t polBits, uint32_t transposeBits, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (!pol || !out) return fail(PIL2GL_EINVAL, "null buffer");
    if (transposeBits > polBits || polBits > 31) return fail(PIL2GL_EINVAL, "Invalid polynomial size");
    fri_transpose_kernel<<<nblk(1ull << polBits), 256, 0, as_stream(stream)>>>(pol, polBits, transposeBits, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

static int host3(const uint64_t *in, u64 nIn, uint64_t *out, u64 nOut, int (*fn)(const u64 *, u64 *, void *), void *arg) {
    P2_TRY(ensure_init());
    u64 *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (nIn + nOut + 1) * 8));
    int rc = PIL2GL_OK;
    hipError_t e = hipMemcpy(d, in, nIn * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) rc = fn(d, d + nIn, arg);
    if (rc == PIL2GL_OK) { e = hipMemcpy(out, d + nIn, nOut * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H"); }
    (void)hipFree(d);
    return rc;
}
struct FoldArgs { u32 polBits, outBits; u64 shiftInv; const u64 *ch; };
int pil2gl_fri_fold(const uint64_t *pol, uint32_t polBits, uint32_t outBits, uint64_t shiftInv, const uint64_t challenge[3], uint64_t *out) {
    if (outBits > polBits || polBits > PIL2GL_MAX_NTT_BITS) return fail(PIL2GL_EINVAL, "Invalid polynomial size");
    FoldArgs a = { polBits, outBits, shiftInv, challenge };
    return host3(pol, 3ull << polBits, out, 3ull << outBits,
                 [](const u64 *i, u64 *o, void *p) { FoldArgs *a = (FoldArgs *)p; return pil2gl_fri_fold_dev(i, a->polBits, a->outBits, a->shiftInv, a->ch, o, nullptr); }, &a);
}
int pil2gl_fri_verify_fold(const uint64_t *groups, uint32_t foldBits, uint32_t nQueries, const uint64_t *sinv,
                           const uint64_t challenge[3], uint64_t *out) {
    P2_TRY(ensure_init());
    if (!groups || !sinv || !out || !challenge) return fail(PIL2GL_EINVAL, "null buffer");
    if (foldBits > 20 || nQueries == 0) return fail(PIL2GL_EINVAL, "Invalid group size or query count");
    const u64 nG = (3ull << foldBits) * nQueries;
    u64 *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (nG + 4ull * nQueries) * 8));
    int rc = PIL2GL_OK;
    hipError_t e = hipMemcpy(d, groups, nG * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + nG, sinv, nQueries * 8ull, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) rc = pil2gl_fri_verify_fold_dev(d, foldBits, nQueries, d + nG, challenge, d + nG + nQueries, nullptr);
    if (rc == PIL2GL_OK) { e = hipMemcpy(out, d + nG + nQueries, 3ull * nQueries * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H"); }
    (void)hipFree(d);
    return rc;
}
int pil2gl_fri_transpose(const uint64_t *pol, uint32_t polBits, uint32_t transposeBits, uint64_t *out) {
    if (transposeBits > polBits || polBits > 31) return fail(PIL2GL_EINVAL, "Invalid polynomial size");
    u32 a[2] = { polBits, transposeBits };
    return host3(pol, 3ull << polBits, out, 3ull << polBits,
                 [](const u64 *i, u64 *o, void *p) { u32 *a = (u32 *)p; return pil2gl_fri_transpose_dev(i, a[0], a[1], o, nullptr); }, a);
}

// ---- STARK step helpers ----
int pil2gl_build_x_dev(uint32_t nBits, uint64_t shift, uint64_t *x, void *stream) {
    P2_TRY(ensure_init());
    if (nBits > 31) return fail(PIL2GL_EINVAL, "nBits too large");
    if (!x) return fail(PIL2GL_EINVAL, "null buffer");
    build_x_kernel<<<nblk(1ull << nBits), 256, 0, as_stream(stream)>>>(nBits, shift, tables().powW, x);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

// A geometric sequence out[i] = first * ratio^i, i < n: the x table of ONE coset (first = 7 w_E^j, ratio = w_N: rows (pos << b) + j of
// x_ext, stark_gen_helpers.js:139-144) or the twiddles w_E^(-i j) of the coset-wise quotient transform, without the 2^nBitsExt-row
// table they would otherwise be gathered from (a rank of a coset-sharded proof never holds a whole extended column)
int pil2gl_geometric_dev(uint64_t first, uint64_t ratio, uint64_t n, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (n == 0) return PIL2GL_OK;
    if (!out || n > (1ull << 40)) return fail(PIL2GL_EINVAL, "bad geometric-sequence arguments");
    if ((n + 255) / 256 > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");      // 2^39 elements and more: beyond one launch
    const u64 p = 0xFFFFFFFF00000001ull;
    GeomPow R;
    u64 cur = ratio % p;
    for (int b = 0; b < 40; b++) { R.p[b] = cur; cur = h_mul(cur, cur); }
    geometric_kernel<<<nblk(n), 256, 0, as_stream(stream)>>>(first % p, R, n, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

// zh[i] = 7^N * w_ext^i - 1 for i < 2^eb (polutils.js:44-51); inv = its inverse
static void zh_table(u32 nBits, u32 nBitsExt, std::vector<u64> &zh, bool inverted) {
    u32 eb = nBitsExt - nBits;
    u64 sn = 7;
    for (u32 i = 0; i < nBits; i++) sn = h_mul(sn, sn);
    u64 w = 1, we = h_root(eb);
    zh.resize(1ull << eb);
    for (u64 i = 0; i < zh.size(); i++) { u64 z = h_sub(h_mul(sn, w), 1); zh[i] = inverted ? h_inv(z) : z; w = h_mul(w, we); }
}
static int upload_small(const std::vector<u64> &h, u32 slot, u64 **d, hipStream_t st) {
    P2_TRY(scratch(slot, h.size() ? h.size() : 1, d));
    // pageable-host async copies are staged by the runtime before returning, so `h` may go out of scope
    HIP_TRY(hipMemcpyAsync(*d, h.data(), h.size() * 8, hipMemcpyHostToDevice, st));
    return PIL2GL_OK;
}

int pil2gl_build_zhinv_dev(uint32_t nBits, uint32_t nBitsExt, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (nBitsExt < nBits || nBitsExt > 31 || nBitsExt - nBits > 20) return fail(PIL2GL_EINVAL, "bad domain sizes");
    if (!out) return fail(PIL2GL_EINVAL, "null buffer");
    std::vector<u64> zh; zh_table(nBits, nBitsExt, zh, true);
    u64 *d; P2_TRY(upload_small(zh, 2, &d, as_stream(stream)));
    periodic_kernel<<<nblk(1ull << nBitsExt), 256, 0, as_stream(stream)>>>(d, zh.size(), 1ull << nBitsExt, out);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));       // scratch slot 2 is reused by the next helper call
    return PIL2GL_OK;
}
int pil2gl_build_one_row_zerofier_inv_dev(uint32_t nBits, uint32_t nBitsExt, uint64_t rowIndex, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (nBitsExt < nBits || nBitsExt > 31 || nBitsExt - nBits > 20) return fail(PIL2GL_EINVAL, "bad domain sizes");
    if (!out) return fail(PIL2GL_EINVAL, "null buffer");
    std::vector<u64> zh; zh_table(nBits, nBitsExt, zh, false);
    u64 *d; P2_TRY(upload_small(zh, 2, &d, as_stream(stream)));
    u64 root = h_pow(h_root(nBits), rowIndex);
    one_row_zerofier_kernel<<<nblk(((1ull << nBitsExt) + 7) / 8), 256, 0, as_stream(stream)>>>(nBitsExt, root, d, zh.size(), tables().powW, out);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return PIL2GL_OK;
}
int pil2gl_build_frame_zerofier_dev(uint32_t nBits, uint32_t nBitsExt, uint64_t offsetMin, uint64_t offsetMax, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (nBitsExt < nBits || nBitsExt > 31 || offsetMin + offsetMax > 4096 || !out) return fail(PIL2GL_EINVAL, "bad frame zerofier arguments");
    std::vector<u64> roots;
    u64 w = h_root(nBits), N = 1ull << nBits;
    for (u64 i = 0; i < offsetMin; i++) roots.push_back(h_pow(w, i));
    for (u64 i = 0; i < offsetMax; i++) roots.push_back(h_pow(w, N - i - 1));
    u64 *d; P2_TRY(upload_small(roots, 2, &d, as_stream(stream)));
    frame_zerofier_kernel<<<nblk(1ull << nBitsExt), 256, 0, as_stream(stream)>>>(nBitsExt, d, (u32)roots.size(), tables().powW, out);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return PIL2GL_OK;
}
int pil2gl_compute_q_split_dev(const uint64_t *qq1, uint32_t nBits, uint32_t nBitsExt, uint32_t qDim, uint32_t qDeg, uint64_t *qq2, void *stream) {
    P2_TRY(ensure_init());
    if (!qq1 || !qq2) return fail(PIL2GL_EINVAL, "null buffer");
    if (nBitsExt < nBits || nBitsExt > 31 || !qDim || !qDeg || ((u64)qDeg << nBits) > (1ull << nBitsExt)) return fail(PIL2GL_EINVAL, "bad q split arguments");
    std::vector<u64> sp(qDeg);
    u64 shiftIn = h_pow(h_inv(7), 1ull << nBits), cur = 1;
    for (u32 p = 0; p < qDeg; p++) { sp[p] = cur; cur = h_mul(cur, shiftIn); }
    u64 *d; P2_TRY(upload_small(sp, 2, &d, as_stream(stream)));
    q_split_kernel<<<nblk(((u64)qDim * qDeg) << nBitsExt), 256, 0, as_stream(stream)>>>(qq1, nBits, nBitsExt, qDim, qDeg, d, qq2);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return PIL2GL_OK;
}
int pil2gl_compute_q_split_brev_dev(const uint64_t *qq1, uint32_t nBits, uint32_t nBitsExt, uint32_t qDim, uint32_t qDeg, uint64_t *coefBrev, void *stream) {
    P2_TRY(ensure_init());
    if (!qq1 || !coefBrev) return fail(PIL2GL_EINVAL, "null buffer");
    if (nBitsExt < nBits || nBitsExt > 31 || !qDim || !qDeg || ((u64)qDeg << nBits) > (1ull << nBitsExt)) return fail(PIL2GL_EINVAL, "bad q split arguments");
    std::vector<u64> sp(qDeg);
    u64 shiftIn = h_pow(h_inv(7), 1ull << nBits), cur = 1;
    for (u32 p = 0; p < qDeg; p++) { sp[p] = cur; cur = h_mul(cur, shiftIn); }
    u64 *d; P2_TRY(upload_small(sp, 2, &d, as_stream(stream)));
    q_split_brev_kernel<<<nblk(((u64)qDim * qDeg) << nBits), 256, 0, as_stream(stream)>>>(qq1, nBits, qDim, qDeg, d, coefBrev);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return PIL2GL_OK;
}
int pil2gl_x_div_x_sub_xi_dev(uint32_t nBitsExt, const uint64_t xi[3], uint64_t nOpen, uint64_t iOpen, uint64_t *out, void *stream) {
    return pil2gl_x_div_x_sub_xi_cosets_dev(nBitsExt, 0, xi, nOpen, iOpen, 0, 1, out, stream);
}
// rows of cosets [cosetBegin, cosetBegin + cosetCount) of the 2^extBits only, in slice order (row pos * cosetCount + jl); cosetCount a
// power of two; extBits = 0: the whole table
int pil2gl_x_div_x_sub_xi_cosets_dev(uint32_t nBitsExt, uint32_t extBits, const uint64_t xi[3], uint64_t nOpen, uint64_t iOpen,
                                     uint32_t cosetBegin, uint32_t cosetCount, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (!xi || !out || iOpen >= nOpen || nBitsExt > 31 || extBits > nBitsExt) return fail(PIL2GL_EINVAL, "bad xDivXSubXi arguments");
    if (cosetCount == 0 || (cosetCount & (cosetCount - 1)) || (uint64_t)cosetBegin + cosetCount > (1ull << extBits))
        return fail(PIL2GL_EINVAL, "coset range [%u,%u) outside 2^%u (the count must be a power of two)", cosetBegin, cosetBegin + cosetCount, extBits);
    // a lane's rows are T = 256*blocks apart and must stay in one coset for its constant step: T a multiple of cosetCount
    if (cosetCount > 256) return fail(PIL2GL_EINVAL, "at most 256 cosets per call (%u asked)", cosetCount);
    u32 ccLog = 0; while ((1u << ccLog) < cosetCount) ccLog++;
    const u64 GP = 0xFFFFFFFF00000001ull;
    const u64 xi0 = xi[0] % GP, b = h_sub(0, xi[1] % GP), c = h_sub(0, xi[2] % GP);
    const u64 bb = h_mul(b, b), cc = h_mul(c, c), bc = h_mul(b, c);
    XDivConst K;
    K.xi0 = xi0; K.b = b; K.c = c; K.c2 = h_add(c, c); K.bc2 = h_add(bc, bc);
    K.m1 = h_sub(h_add(bc, bb), cc);
    K.k0 = h_sub(h_sub(h_mul(b, cc), h_mul(bb, b)), h_mul(cc, c));
    K.cc = cc; K.ccbb = h_sub(cc, bb);
    const u64 E = (1ull << (nBitsExt - extBits)) << ccLog;                          // rows of the slice
    const unsigned blocks = nblk((E + XD_BATCH - 1) / XD_BATCH);
    const u64 T = (u64)blocks * 256;                                                // a multiple of cosetCount (<= 256)
    const u64 wStep = h_pow(h_root(nBitsExt), (T >> ccLog) << extBits);             // w_E^(extended rows between a lane's rows)
    x_div_x_sub_xi_kernel<<<blocks, 256, 0, as_stream(stream)>>>(nBitsExt, K, nOpen, iOpen, tables().powW, wStep, out, E, extBits, cosetBegin, ccLog);
    KERNEL_CHECK();
    return PIL2GL_OK;
}
// LEv = F.ifft of (xi^k)_{k<N} (stark_gen_helpers.js:216-231).  The inverse transform of a geometric sequence has a closed form,
//     LEv[j] = (1/N) sum_k (xi w^-j)^k = (1 - xi^N) / N * w^j / (w^j - xi) = (1 - xi^N) / N * x_j / (x_j - 7 xi),   x_j = 7 w^j,
// i.e. the batched-inversion kernel of the FRI table at the point 7 xi over the N rows, times one extension constant: one sweep that
// writes N triples instead of an N x 3 transform (2^24 rows: 0.5 ms against 2.8).  Same values, exactly (field arithmetic).
// PIL2GL_LEV_NTT=1 keeps the transform (A/B runs, tests).
__global__ void e3_scale_kernel(u64 *__restrict__ v, u64 n, E3 c) {
    const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) st3(v + 3 * k, e3_mul(ld3(v + 3 * k), c));
}
int pil2gl_build_lev_dev(uint32_t nBits, const uint64_t xi[3], uint64_t *lev, void *stream) {
    P2_TRY(ensure_init());
    if (!xi || !lev || nBits > PIL2GL_MAX_NTT_BITS) return fail(PIL2GL_EINVAL, "bad LEv arguments");
    static const bool viaNtt = getenv("PIL2GL_LEV_NTT") && atoi(getenv("PIL2GL_LEV_NTT"));
    if (!viaNtt && nBits > 0) {
        const u64 GP = 0xFFFFFFFF00000001ull;
        u64 z[3] = { xi[0] % GP, xi[1] % GP, xi[2] % GP }, zN[3] = { z[0], z[1], z[2] };
        for (u32 b = 0; b < nBits; b++) h_e3_mul(zN, zN, zN);                                  // xi^N
        const u64 invN = h_inv((1ull << nBits) % GP);
        E3 c = { { h_mul(h_sub(1, zN[0]), invN), h_mul(h_sub(0, zN[1]), invN), h_mul(h_sub(0, zN[2]), invN) } };    // (1 - xi^N) / N
        const u64 z7[3] = { h_mul(7, z[0]), h_mul(7, z[1]), h_mul(7, z[2]) };
        P2_TRY(pil2gl_x_div_x_sub_xi_cosets_dev(nBits, 0, z7, 1, 0, 0, 1, lev, stream));       // x_j / (x_j - 7 xi), rows j < N
        e3_scale_kernel<<<nblk(1ull << nBits), 256, 0, as_stream(stream)>>>(lev, 1ull << nBits, c);
        KERNEL_CHECK();
        HIP_TRY(hipStreamSynchronize(as_stream(stream)));
        return PIL2GL_OK;
    }
    std::vector<u64> xp(3 * (nBits ? nBits : 1));
    u64 cur[3] = { xi[0], xi[1], xi[2] };
    for (u32 b = 0; b < nBits; b++) { xp[3 * b] = cur[0]; xp[3 * b + 1] = cur[1]; xp[3 * b + 2] = cur[2]; h_e3_mul(cur, cur, cur); }
    u64 *d; P2_TRY(upload_small(xp, 2, &d, as_stream(stream)));
    lev_pow_kernel<<<nblk(1ull << nBits), 256, 0, as_stream(stream)>>>(nBits, d, lev);
    KERNEL_CHECK();
    P2_TRY(ntt_launch(lev, 3, nBits, lev, true, as_stream(stream)));       // F.ifft on triples: component-wise
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return PIL2GL_OK;
}
int pil2gl_compute_evals_dev(const pil2gl_eval_desc *descs, uint32_t nEvals, uint32_t nBits, uint32_t extendBits,
                             const uint64_t *const *levs, uint32_t nLevs, uint64_t *hostEvals, void *stream) {
    P2_TRY(ensure_init());
    if (!nEvals) return PIL2GL_OK;
    if (!descs || !levs || !hostEvals) return fail(PIL2GL_EINVAL, "null buffer");
    hipStream_t st = as_stream(stream);
    const u32 nBlocks = (u32)std::min<u64>(256, ((1ull << nBits) + 255) / 256);
    u64 *partial, *res;
    P2_TRY(scratch(3, (u64)nEvals * nBlocks * 3 + (u64)nEvals * 3, &partial));
    res = partial + (u64)nEvals * nBlocks * 3;
    for (u32 e = 0; e < nEvals; e++) {
        if (descs[e].levIndex >= nLevs || (descs[e].dim != 1 && descs[e].dim != 3)) return fail(PIL2GL_EINVAL, "bad eval descriptor %u", e);
        evals_partial_kernel<<<nBlocks, 256, 0, st>>>(descs[e], levs[descs[e].levIndex], nBits, extendBits, partial + (u64)e * nBlocks * 3);
    }
    KERNEL_CHECK();
    evals_final_kernel<<<nblk(nEvals * 3), 256, 0, st>>>(partial, nBlocks, nEvals, res);
    KERNEL_CHECK();
    HIP_TRY(hipMemcpyAsync(hostEvals, res, (u64)nEvals * 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PIL2GL_OK;
}

int pil2gl_synth_fibonacci_dev(uint32_t nBits, uint32_t nPairs, const uint64_t *hostInit, uint64_t *cm, void *stream) {
    P2_TRY(ensure_init());
    if (!hostInit || !cm || nBits > 31 || nPairs == 0) return fail(PIL2GL_EINVAL, "bad synthetic trace arguments");
    std::vector<u64> h(hostInit, hostInit + 2ull * nPairs);
    u64 *d; P2_TRY(upload_small(h, 2, &d, as_stream(stream)));
    synth_fibonacci_kernel<<<nblk(nPairs, 64), 64, 0, as_stream(stream)>>>(nBits, nPairs, d, cm);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return PIL2GL_OK;
}

}  // extern "C"
