// Prototype + microbenchmark: 2^c-point DFTs held in registers, computed in Z/(2^96+1) where every root of unity of
// order <= 64 is a power of two (p = 2^64-2^32+1 divides 2^96+1, w_64 = 8, w_32 = 64, w_16 = 2^12, ...), so the
// butterflies are 128-bit adds and constant shifts; one reduction mod p at the end.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc tools/fermat_dft.hip -o tools/fermat_dft
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "gl_field.cuh"
#include "gl_fermat.cuh"
using namespace gl;

// plain field version of the same 16-point DFT (what the radix-4 tile code does today), twiddles as run-time values
template <bool INV>
__device__ __forceinline__ void dft16_dif_gl(u64 x[16], const u64 *w16 /* w16[i] = w_16^(+-i), i<8 */) {
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
        for (int base = 0; base < 16; base += 2 * h) {
#pragma unroll
            for (int i = 0; i < h; i++) {
                u64 a = x[base + i], b = x[base + i + h];
                x[base + i] = add(a, b);
                u64 d = sub(a, b);
                x[base + i + h] = (i == 0) ? d : mul(d, w16[i * (8 / h)]);
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_dft(u64 *out, int iters, u64 seed, const u64 *w16) {
    u64 x[16];
    u64 tw = seed * 77 + threadIdx.x * 0x9E3779B97F4A7C15ull;
    tw = canon(tw);
    for (int i = 0; i < 16; i++) x[i] = canon((seed + i) * 0x9E3779B97F4A7C15ull + (threadIdx.x + blockIdx.x * 256) * 0x123456789ull);
    u64 w[8];
    for (int i = 0; i < 8; i++) w[i] = w16[i];
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            dft16_dif_gl<false>(x, w);
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = mul(x[i], tw);
        } else if (MODE == 1) {
            fermat::f128 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = fermat::from_gl(x[i]);
            fermat::dft_dif<4, false>(v);
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = mul(fermat::to_gl_lazy(v[i]), tw);
        } else if (MODE == 2) {                        // inverse direction, DIT order
            fermat::f128 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = fermat::from_gl(x[i]);
            fermat::dft_dit<4, true>(v);
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = mul(fermat::to_gl_lazy(v[i]), tw);
        } else if (MODE == 3) {                        // 8-point
            fermat::f128 v[8];
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = fermat::from_gl(x[half * 8 + i]);
                fermat::dft_dif<3, false>(v);
#pragma unroll
                for (int i = 0; i < 8; i++) x[half * 8 + i] = mul(fermat::to_gl_lazy(v[i]), tw);
            }
        }
        else if (MODE == 4) {                        // DFT + reduction, no twiddle
            fermat::f128 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = fermat::from_gl(x[i]);
            fermat::dft_dif<4, false>(v);
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = fermat::to_gl_lazy(v[i]);
        } else if (MODE == 5) {                        // reduction + twiddle only
#pragma unroll
            for (int i = 0; i < 16; i++) { fermat::f128 v = fermat::from_gl(x[i]); v.w2 = (u32)it; v.w3 = (u32)(it & 7) - 3; x[i] = mul(fermat::to_gl_lazy(v), tw); }
        } else if (MODE == 6) {                        // twiddle only
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = mul(x[i], tw);
        } else if (MODE == 7) {                        // twiddle only, lazy
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = mul_lazy(x[i], tw);
        }
        tw = add(tw, 12345);
    }
    u64 s = 0;
    for (int i = 0; i < 16; i++) s = add(mul(s, 3), x[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// correctness: one DFT per thread against the O(R^2) definition, all radices, both directions and orders
template <int C, bool INV, bool DIT>
__global__ void k_check(const u64 *in, u64 *got, u64 *want, u64 wR) {
    constexpr int R = 1 << C;
    u64 x[R];
    fermat::f128 v[R];
    const u64 *src = in + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * R;
    for (int i = 0; i < R; i++) x[i] = src[i];
    // definition: X[q] = sum_r x[r] w^(rq); DIF: natural in, out[bitrev(q)]; DIT: in[bitrev(r)], natural out
    for (int q = 0; q < R; q++) {
        u64 acc = 0, wq = gl::pow(wR, (u64)q), cur = 1;
        for (int r = 0; r < R; r++) {
            u64 xr = DIT ? x[bitrev32(r, C)] : x[r];
            acc = add(acc, mul(xr, cur));
            cur = mul(cur, wq);
        }
        want[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * R + (DIT ? q : bitrev32(q, C))] = acc;
    }
#pragma unroll
    for (int i = 0; i < R; i++) v[i] = fermat::from_gl(x[i]);
    if (DIT) fermat::dft_dit<C, INV>(v); else fermat::dft_dif<C, INV>(v);
#pragma unroll
    for (int i = 0; i < R; i++) got[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * R + i] = canon(fermat::to_gl_lazy(v[i]));
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static const u64 HP = 0xFFFFFFFF00000001ull;
static u64 hmul(u64 a, u64 b) { return (u64)((unsigned __int128)a * b % HP); }
static u64 hpow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = hmul(r, a); a = hmul(a, a); e >>= 1; } return r; }

template <int C, bool INV, bool DIT>
int check(const u64 *din, u64 *dgot, u64 *dwant, int nThreads) {
    constexpr int R = 1 << C;
    u64 wR = hpow(2, 192 / R);
    if (INV) wR = hpow(wR, HP - 2);
    hipLaunchKernelGGL((k_check<C, INV, DIT>), dim3(nThreads / 64), dim3(64), 0, 0, din, dgot, dwant, wR);
    size_t n = (size_t)nThreads * R;
    u64 *g = (u64 *)malloc(8 * n), *w = (u64 *)malloc(8 * n);
    CHECK(hipMemcpy(g, dgot, 8 * n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(w, dwant, 8 * n, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < n; i++) bad += g[i] != w[i];
    printf("R=%2d %s %s: %s (%zu/%zu differ)\n", R, INV ? "inverse" : "forward", DIT ? "DIT" : "DIF", bad ? "MISMATCH" : "ok", bad, n);
    free(g); free(w);
    return bad != 0;
}

template <typename F>
float timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}

int main() {
    const int nT = 4096;
    u64 *din, *dgot, *dwant, *out, *w16;
    CHECK(hipMalloc((void **)&din, 8ull * nT * 64)); CHECK(hipMalloc((void **)&dgot, 8ull * nT * 64)); CHECK(hipMalloc((void **)&dwant, 8ull * nT * 64));
    CHECK(hipMalloc((void **)&out, 8ull * 256 * 4096)); CHECK(hipMalloc((void **)&w16, 64));
    u64 *h = (u64 *)malloc(8ull * nT * 64);
    u64 s = 88172645463325252ull;
    for (size_t i = 0; i < (size_t)nT * 64; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        u64 v = s % HP;
        if (i % 97 == 0) v = HP - 1; if (i % 101 == 0) v = 0; if (i % 103 == 0) v = 0xFFFFFFFFull; if (i % 107 == 0) v = 0xFFFFFFFF00000000ull;
        h[i] = v;
    }
    for (size_t i = 0; i < 64; i++) h[i] = HP - 1;          // an all-(p-1) DFT: the largest magnitudes
    CHECK(hipMemcpy(din, h, 8ull * nT * 64, hipMemcpyHostToDevice));
    u64 hw[8]; for (int i = 0; i < 8; i++) hw[i] = hpow(2, 12 * i);
    CHECK(hipMemcpy(w16, hw, 64, hipMemcpyHostToDevice));
    int bad = 0;
    bad += check<1, false, false>(din, dgot, dwant, nT); bad += check<1, true, true>(din, dgot, dwant, nT);
    bad += check<2, false, false>(din, dgot, dwant, nT); bad += check<2, true, false>(din, dgot, dwant, nT);
    bad += check<2, false, true>(din, dgot, dwant, nT);  bad += check<2, true, true>(din, dgot, dwant, nT);
    bad += check<3, false, false>(din, dgot, dwant, nT); bad += check<3, true, false>(din, dgot, dwant, nT);
    bad += check<3, false, true>(din, dgot, dwant, nT);  bad += check<3, true, true>(din, dgot, dwant, nT);
    bad += check<4, false, false>(din, dgot, dwant, nT); bad += check<4, true, false>(din, dgot, dwant, nT);
    bad += check<4, false, true>(din, dgot, dwant, nT);  bad += check<4, true, true>(din, dgot, dwant, nT);
    bad += check<5, false, false>(din, dgot, dwant, nT); bad += check<5, true, true>(din, dgot, dwant, nT);
    const int blocks = 256 * 8, iters = 200;
    const char *names[] = { "dft16 field ops + twiddle", "dft16 fermat DIF + twiddle", "dft16 fermat DIT inverse + twiddle", "2 x dft8 fermat + twiddle", "dft16 fermat + reduce", "reduce + twiddle", "twiddle (canonical mul)", "twiddle (lazy mul)" };
#define RUN(M) { float ms = timeit([&] { hipLaunchKernelGGL(k_dft<M>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ull, w16); }); \
    double el = (double)blocks * 256 * iters * 16; printf("%-36s %8.3f ms  %.1f cyc per wave-element (%.1f per element-stage)\n", names[M], ms, 2.4e9 * 1024 * 64 / (el / (ms * 1e-3)), 2.4e9 * 1024 * 64 / (el / (ms * 1e-3)) / (M == 3 ? 3 : 4)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    printf(bad ? "FAILED\n" : "all DFT checks passed\n");
    return bad;
}
