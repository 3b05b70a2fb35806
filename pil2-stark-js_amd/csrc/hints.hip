// Stage-2 witness hints on the device: the grand product / grand sum columns
//   calculateZ(F,num,den)  src/helpers/polutils.js:128-143   z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1]
//   calculateS(F,num,den)  src/helpers/polutils.js:145-164   s[i] = s[i-1] + num / den[i]        (num: one element)
// called from resolveHint (src/prover/hints_helpers.js:91-114).  The reference walks the column serially with BigInt
// arithmetic after one batch inversion; here a thread inverts its eight denominators together (one field inversion per
// eight rows) and the running product / sum is a three-kernel scan: per-block inclusive scan + block totals, scan of the
// totals, then the fix-up (for the product: shifted by one row, z[0] = 1).  Operands may be base (dim 1) or cubic
// extension (dim 3) columns, row-major; the result is dim 3 if either operand is, else dim 1.
#include "common.h"
#include "gl_field.cuh"

using namespace gl;
using namespace pil2gl;

namespace {

constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ E3 ld_dim(const u64 *p, u64 i, u32 dim) { return dim == 3 ? E3{ { p[3 * i], p[3 * i + 1], p[3 * i + 2] } } : E3{ { p[i], 0, 0 } }; }
__device__ __forceinline__ void st3(u64 *p, u64 i, const E3 &v) { p[3 * i] = v.v[0]; p[3 * i + 1] = v.v[1]; p[3 * i + 2] = v.v[2]; }
template <bool PROD> __device__ __forceinline__ E3 comb(const E3 &a, const E3 &b) { return PROD ? e3_mul(a, b) : e3_add(a, b); }
template <bool PROD> __device__ __forceinline__ E3 ident() { return PROD ? E3{ { 1, 0, 0 } } : E3{ { 0, 0, 0 } }; }

// inclusive scan of the 256 per-thread totals in LDS (Hillis-Steele); returns this thread's exclusive prefix
template <bool PROD>
__device__ E3 block_exclusive(E3 total, E3 *sh, E3 *blockTotal) {
    const u32 t = threadIdx.x;
    sh[t] = total;
    __syncthreads();
    for (u32 d = 1; d < SCAN_THREADS; d <<= 1) {
        E3 v = sh[t];
        if (t >= d) v = comb<PROD>(sh[t - d], v);
        __syncthreads();
        sh[t] = v;
        __syncthreads();
    }
    const E3 ex = t ? sh[t - 1] : ident<PROD>();
    *blockTotal = sh[SCAN_THREADS - 1];
    __syncthreads();
    return ex;
}

// pass 1: ratios, block-local inclusive scan into tmp (n x 3), block totals
template <bool PROD>
__global__ void __launch_bounds__(SCAN_THREADS) hint_scan1(const u64 *__restrict__ num, u32 dimNum, const u64 *__restrict__ den, u32 dimDen, u64 n,
                                                          u64 *__restrict__ tmp, u64 *__restrict__ totals) {
    __shared__ E3 sh[SCAN_THREADS];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_ITEMS;
    E3 loc[SCAN_ITEMS];
    E3 run = ident<PROD>();
    const E3 numS = PROD ? ident<true>() : ld_dim(num, 0, dimNum);
    // the thread's SCAN_ITEMS denominators are inverted together (Montgomery's trick, what the reference's F.batchInverse does for the whole
    // column, polutils.js:134): ONE field inversion -- an exponentiation, ~130 products -- and three extension products per row instead
    // of an inversion per row.  A zero denominator inverts to zero, as it does alone (0^(p-2)): it is kept out of the running product.
    const E3 one = ident<true>();
    u32 zero = 0;
    E3 acc = one;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const u64 i = base + k;
        E3 d = i < n ? ld_dim(den, i, dimDen) : one;
        if (!(d.v[0] | d.v[1] | d.v[2])) { zero |= 1u << k; d = one; }
        loc[k] = acc;                                   // the product of the denominators before this one
        acc = e3_mul(acc, d);
    }
    E3 ai = (acc.v[1] | acc.v[2]) ? e3_inv(acc) : E3{ { inv(acc.v[0]), 0, 0 } };
#pragma unroll
    for (int k = SCAN_ITEMS - 1; k >= 0; k--) {
        const u64 i = base + k;
        E3 d = i < n ? ld_dim(den, i, dimDen) : one;
        if ((zero >> k) & 1) d = one;
        const E3 di = e3_mul(ai, loc[k]);               // 1 / d_k
        ai = e3_mul(ai, d);
        loc[k] = ((zero >> k) & 1) ? E3{ { 0, 0, 0 } } : di;
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const u64 i = base + k;
        if (i < n) {
            const E3 r = e3_mul(PROD ? ld_dim(num, i, dimNum) : numS, loc[k]);
            run = comb<PROD>(run, r);
        }
        loc[k] = run;
    }
    E3 bt;
    const E3 ex = block_exclusive<PROD>(run, sh, &bt);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { const u64 i = base + k; if (i < n) st3(tmp, i, comb<PROD>(ex, loc[k])); }
    if (threadIdx.x == 0) st3(totals, blockIdx.x, bt);
}
// pass 2: exclusive scan of the block totals, one block (each thread walks a contiguous run)
template <bool PROD>
__global__ void __launch_bounds__(SCAN_THREADS) hint_scan2(u64 *__restrict__ totals, u64 nb) {
    __shared__ E3 sh[SCAN_THREADS];
    const u64 per = (nb + SCAN_THREADS - 1) / SCAN_THREADS, b0 = (u64)threadIdx.x * per;
    E3 run = ident<PROD>();
    for (u64 k = 0; k < per; k++) { const u64 i = b0 + k; if (i < nb) run = comb<PROD>(run, ld_dim(totals, i, 3)); }
    E3 bt;
    E3 ex = block_exclusive<PROD>(run, sh, &bt);
    for (u64 k = 0; k < per; k++) {
        const u64 i = b0 + k;
        if (i < nb) { const E3 v = ld_dim(totals, i, 3); st3(totals, i, ex); ex = comb<PROD>(ex, v); }
    }
}
// pass 3: add the block prefix; the product column is the EXCLUSIVE scan (z[0] = 1), the sum column the inclusive one
template <bool PROD>
__global__ void hint_scan3(const u64 *__restrict__ tmp, const u64 *__restrict__ totals, u64 n, u32 dimOut, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    E3 v;
    if (PROD) {
        if (i == 0) v = ident<true>();
        else { const u64 j = i - 1; v = comb<true>(ld_dim(totals, j / SCAN_CHUNK, 3), ld_dim(tmp, j, 3)); }
    } else v = comb<false>(ld_dim(totals, i / SCAN_CHUNK, 3), ld_dim(tmp, i, 3));
    if (dimOut == 3) st3(out, i, v); else out[i] = v.v[0];
}

template <bool PROD>
int run_hint(const u64 *num, u32 dimNum, const u64 *den, u32 dimDen, u64 n, u64 *out, hipStream_t st) {
    if (n == 0) return PIL2GL_OK;
    if (!num || !den || !out) return fail(PIL2GL_EINVAL, "null buffer");
    if ((dimNum != 1 && dimNum != 3) || (dimDen != 1 && dimDen != 3)) return fail(PIL2GL_EINVAL, "dimensions must be 1 or 3");
    const u64 nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (nb > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");
    u64 *tmp, *totals;
    P2_TRY(scratch(8, n * 3, &tmp));
    P2_TRY(scratch(9, nb * 3, &totals));
    hint_scan1<PROD><<<(unsigned)nb, SCAN_THREADS, 0, st>>>(num, dimNum, den, dimDen, n, tmp, totals);
    KERNEL_CHECK();
    hint_scan2<PROD><<<1, SCAN_THREADS, 0, st>>>(totals, nb);
    KERNEL_CHECK();
    hint_scan3<PROD><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(tmp, totals, n, (dimNum == 3 || dimDen == 3) ? 3u : 1u, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

}  // namespace

extern "C" {

int pil2gl_gprod_dev(const uint64_t *num, uint32_t dimNum, const uint64_t *den, uint32_t dimDen, uint64_t n, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    return run_hint<true>(num, dimNum, den, dimDen, n, out, as_stream(stream));
}
int pil2gl_gsum_dev(const uint64_t *num, uint32_t dimNum, const uint64_t *den, uint32_t dimDen, uint64_t n, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    return run_hint<false>(num, dimNum, den, dimDen, n, out, as_stream(stream));
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// calculateH1H2(F, f, t)  src/helpers/polutils.js:105-126 (hint "h1h2", hints_helpers.js:115-121).
// The reference builds s = [(t[i], i)] ++ [(f[j], idx_t[f[j]])] with idx_t[v] = LAST i with t[i] = v, sorts it stably by
// the index and reads h1[i] = s[2i], h2[i] = s[2i+1].  All entries with the same index carry the same VALUE, so the
// sorted sequence is "t[i] repeated 1 + cnt[i] times, for i = 0..n-1", with cnt[i] = #{j : f[j] = t[i]} when i is the last
// occurrence of its value and 0 otherwise.  No sort is needed: a hash table value -> last index, the counts, an
// exclusive scan of (1 + cnt) for the group starts, and one binary search per output position.
namespace {

constexpr u64 H_EMPTY = ~0ull;

__device__ __forceinline__ u64 key_hash(const u64 *p, u64 i, u32 dim) {
    u64 h = 0x9E3779B97F4A7C15ull;
    for (u32 k = 0; k < dim; k++) { h ^= p[i * dim + k] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 31; }
    return h;
}
__device__ __forceinline__ bool key_eq(const u64 *a, u64 i, const u64 *b, u64 j, u32 dim) {
    for (u32 k = 0; k < dim; k++) if (a[i * dim + k] != b[j * dim + k]) return false;
    return true;
}
__global__ void h1h2_insert(const u64 *__restrict__ t, u64 n, u32 dim, unsigned long long *__restrict__ table, u64 mask) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s = key_hash(t, i, dim) & mask;
    for (;;) {
        unsigned long long cur = table[s];
        if (cur == H_EMPTY) {
            cur = atomicCAS(&table[s], (unsigned long long)H_EMPTY, (unsigned long long)i);
            if (cur == H_EMPTY) return;                          // claimed the slot
        }
        if (key_eq(t, i, t, cur, dim)) { atomicMax(&table[s], (unsigned long long)i); return; }   // same value: keep the last index
        s = (s + 1) & mask;
    }
}
__global__ void h1h2_count(const u64 *__restrict__ f, const u64 *__restrict__ t, u64 n, u32 dim, const unsigned long long *__restrict__ table, u64 mask,
                           unsigned long long *__restrict__ cnt, unsigned long long *__restrict__ missing) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    u64 s = key_hash(f, j, dim) & mask;
    for (;;) {
        const unsigned long long cur = table[s];
        if (cur == H_EMPTY) { atomicMin(missing, (unsigned long long)j); return; }      // "Number not included" (polutils.js:115)
        if (key_eq(f, j, t, cur, dim)) { atomicAdd(&cnt[cur], 1ull); return; }
        s = (s + 1) & mask;
    }
}
// in: cnt[i];  out: start[i] = sum_{k<i} (1 + cnt[k])  -- block-local exclusive scan + block totals (u64 adds)
__global__ void __launch_bounds__(SCAN_THREADS) h1h2_scan1(const unsigned long long *__restrict__ cnt, u64 n, u64 *__restrict__ start, u64 *__restrict__ totals) {
    __shared__ u64 sh[SCAN_THREADS];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_ITEMS;
    u64 loc[SCAN_ITEMS], run = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { const u64 i = base + k; loc[k] = run; if (i < n) run += 1 + cnt[i]; }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (u32 d = 1; d < SCAN_THREADS; d <<= 1) {
        u64 v = sh[threadIdx.x];
        if (threadIdx.x >= d) v += sh[threadIdx.x - d];
        __syncthreads();
        sh[threadIdx.x] = v;
        __syncthreads();
    }
    const u64 ex = threadIdx.x ? sh[threadIdx.x - 1] : 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { const u64 i = base + k; if (i < n) start[i] = ex + loc[k]; }
    if (threadIdx.x == SCAN_THREADS - 1) totals[blockIdx.x] = sh[SCAN_THREADS - 1];
}
__global__ void h1h2_scan2(u64 *__restrict__ totals, u64 nb) {           // tiny: one thread
    if (blockIdx.x || threadIdx.x) return;
    u64 run = 0;
    for (u64 b = 0; b < nb; b++) { const u64 v = totals[b]; totals[b] = run; run += v; }
}
__global__ void h1h2_scan3(u64 *__restrict__ start, const u64 *__restrict__ totals, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) start[i] += totals[i / SCAN_CHUNK];
}
// output position p in [0, 2n): its group is the last i with start[i] <= p
__global__ void h1h2_expand(const u64 *__restrict__ t, const u64 *__restrict__ start, u64 n, u32 dim, u64 *__restrict__ h1, u64 *__restrict__ h2) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= 2 * n) return;
    u64 lo = 0, hi = n - 1;
    while (lo < hi) { const u64 mid = (lo + hi + 1) >> 1; if (start[mid] <= p) lo = mid; else hi = mid - 1; }
    u64 *o = (p & 1) ? h2 : h1;
    for (u32 k = 0; k < dim; k++) o[(p >> 1) * dim + k] = t[lo * dim + k];
}

}  // namespace

extern "C" int pil2gl_h1h2_dev(const uint64_t *f, const uint64_t *t, uint64_t n, uint32_t dim, uint64_t *h1, uint64_t *h2, void *stream) {
    P2_TRY(ensure_init());
    if (n == 0) return PIL2GL_OK;
    if (!f || !t || !h1 || !h2) return fail(PIL2GL_EINVAL, "null buffer");
    if (dim != 1 && dim != 3) return fail(PIL2GL_EINVAL, "dimension must be 1 or 3");
    if (n > (1ull << 30)) return fail(PIL2GL_EINVAL, "column too long");
    hipStream_t st = as_stream(stream);
    u64 cap = 2; while (cap < 2 * n) cap <<= 1;
    const u64 nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    u64 *ws;                                             // table[cap] | cnt[n] | start[n] | totals[nb] | missing[1]
    P2_TRY(scratch(10, cap + 2 * n + nb + 1, &ws));
    unsigned long long *table = (unsigned long long *)ws, *cnt = table + cap;
    u64 *start = ws + cap + n, *totals = start + n;
    unsigned long long *missing = (unsigned long long *)(totals + nb);
    HIP_TRY(hipMemsetAsync(table, 0xFF, cap * 8, st));
    HIP_TRY(hipMemsetAsync(cnt, 0, n * 8, st));
    HIP_TRY(hipMemsetAsync(missing, 0xFF, 8, st));
    const unsigned g = (unsigned)((n + 255) / 256);
    h1h2_insert<<<g, 256, 0, st>>>(t, n, dim, table, cap - 1);
    KERNEL_CHECK();
    h1h2_count<<<g, 256, 0, st>>>(f, t, n, dim, table, cap - 1, cnt, missing);
    KERNEL_CHECK();
    h1h2_scan1<<<(unsigned)nb, SCAN_THREADS, 0, st>>>(cnt, n, start, totals);
    KERNEL_CHECK();
    h1h2_scan2<<<1, 1, 0, st>>>(totals, nb);
    KERNEL_CHECK();
    h1h2_scan3<<<g, 256, 0, st>>>(start, totals, n);
    KERNEL_CHECK();
    h1h2_expand<<<(unsigned)((2 * n + 255) / 256), 256, 0, st>>>(t, start, n, dim, h1, h2);
    KERNEL_CHECK();
    unsigned long long miss = 0;
    HIP_TRY(hipMemcpyAsync(&miss, missing, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (miss != H_EMPTY) return fail(PIL2GL_EINVAL, "Number not included: w:%llu", miss);     // polutils.js:115
    return PIL2GL_OK;
}
