// S-box (x^7) variants and the vcc / SGPR carry hazard on gfx950: correctness against pow7_lazy and issue rates.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc tools/sbox_bench.hip -o tools/sbox_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef PBLK_NO_MFMA
typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef int v16i_ __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16i_ fake_mfma(v4i_ a, v4i_ b, v16i_ c) { c[0] += a[0] ^ b[1]; c[5] += a[2] ^ b[3]; c[9] ^= a[1] + b[0]; c[14] ^= a[3] + b[2]; return c; }
#define PBLK_MFMA(a, b, c) fake_mfma(a, b, c)
#endif
#include "poseidon_gl.cuh"
using namespace gl;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; return x; }

// ---- hazard stress: a 4-word add through a carry chain, with and without the two wait states hipcc pads ----
template <int V>
__device__ __forceinline__ void add128(u32 a[4], const u32 b[4]) {
    if (V == 0)           // no padding, vcc
        asm("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %5, vcc\n\tv_addc_co_u32 %2, vcc, %2, %6, vcc\n\tv_addc_co_u32 %3, vcc, %3, %7, vcc"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
    if (V == 1)           // padded, vcc
        asm("v_add_co_u32 %0, vcc, %0, %4\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %5, vcc\n\ts_nop 1\n\tv_addc_co_u32 %2, vcc, %2, %6, vcc\n\ts_nop 1\n\tv_addc_co_u32 %3, vcc, %3, %7, vcc"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "vcc");
    if (V == 2) {         // no padding, SGPR pair (VOP3)
        u64 c;
        asm("v_add_co_u32_e64 %0, %8, %0, %4\n\tv_addc_co_u32_e64 %1, %8, %1, %5, %8\n\tv_addc_co_u32_e64 %2, %8, %2, %6, %8\n\tv_addc_co_u32_e64 %3, %8, %3, %7, %8"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "s"(c = 0));
    }
    if (V == 3) {         // mad carry-out consumed by the next instruction, no padding (the BN254 accumulate form)
        u64 lo = ((u64)a[1] << 32) | a[0]; u32 hi = a[2];
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(b[0]), "v"(b[1]) : "vcc");
        a[0] = (u32)lo; a[1] = (u32)(lo >> 32); a[2] = hi;
    }
    if (V == 4) {         // the same, padded
        u64 lo = ((u64)a[1] << 32) | a[0]; u32 hi = a[2];
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(b[0]), "v"(b[1]) : "vcc");
        a[0] = (u32)lo; a[1] = (u32)(lo >> 32); a[2] = hi;
    }
}
template <int V>
__global__ void __launch_bounds__(256) k_hazard(u64 *out, int iters, u64 seed) {
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 a[4], b[4];
    u64 s = mix(seed + id);
    for (int i = 0; i < 4; i++) { s = mix(s + i); a[i] = (u32)s; b[i] = (u32)(s >> 32) | 0x80000000u; }
    for (int i = 0; i < iters; i++) {
        add128<V>(a, b);
        b[0] = b[0] * 0x9E3779B1u + a[3]; b[1] ^= a[0]; b[2] += a[1] | 0xC0000000u; b[3] = (b[3] ^ a[2]) | 0x80000000u;
    }
    out[id] = ((u64)(a[0] ^ a[2]) << 32) | (a[1] ^ a[3]);
}

// ---- a Goldilocks product from SMALL asm statements (registers stay the compiler's): the carries hipcc re-derives with
// 64-bit compares come out of the multiply-adds as lane masks
__device__ __forceinline__ u64 mul_s(u64 a, u64 b, u64 &bad) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v, cy; u32 c01;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v), "=s"(cy) : "v"(a1), "v"(b0), "v"(u));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c01) : "s"(cy));
    const u64 w = (u64)a1 * b1 + (((u64)c01 << 32) | (v >> 32));
    const u64 lo = (v << 32) | (u32)t;
    u64 z, c2; u32 c201;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c2) : "v"((u32)w), "v"(lo));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c201) : "s"(c2));
    const u64 z2 = (u64)c201 * 0xFFFFFFFFu + z;
    u32 r0, r1; u64 br;
    asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\ts_nop 1\n\tv_subbrev_co_u32_e64 %1, %2, 0, %4, %2" : "=&v"(r0), "=&v"(r1), "=&s"(br) : "v"((u32)z2), "v"((u32)(z2 >> 32)), "v"((u32)(w >> 32)));
    bad |= br;
    return ((u64)r1 << 32) | r0;
}
__device__ __forceinline__ u64 pow7_s(u64 x, u64 &bad) { return pow7_b(x, bad); }      // the library's product (mul_s above: round 2's 13-instruction form, kept for reference)
// exact variant: the borrow is folded back in place (three more instructions, no fallback)
__device__ __forceinline__ u64 mul_e(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v, cy; u32 c01;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v), "=s"(cy) : "v"(a1), "v"(b0), "v"(u));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c01) : "s"(cy));
    const u64 w = (u64)a1 * b1 + (((u64)c01 << 32) | (v >> 32));
    const u64 lo = (v << 32) | (u32)t;
    u64 z, c2; u32 c201;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c2) : "v"((u32)w), "v"(lo));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(c201) : "s"(c2));
    const u64 z2 = (u64)c201 * 0xFFFFFFFFu + z;
    u32 r0, r1, m;
    // r = z2 - w1; on a borrow the true value is r - (2^32 - 1) = (r0 + 1, r1 - 1 + carry)
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\ts_nop 1\n\tv_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\ts_nop 1\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(m) : "v"((u32)z2), "v"((u32)(z2 >> 32)), "v"((u32)(w >> 32)) : "vcc");
    return ((u64)r1 << 32) | r0;
}
__device__ __forceinline__ u64 pow7_e(u64 x) { u64 x2 = mul_e(x, x), x3 = mul_e(x2, x), x4 = mul_e(x2, x2); return mul_e(x3, x4); }

__device__ __forceinline__ u64 pow7_c(u64 x) { return pow7_lazy(x); }

// ---- S-box variants ----
template <int V>
__global__ void __launch_bounds__(256) k_sbox(u64 *out, int iters, u64 seed, int dump) {
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x[12];
    for (int i = 0; i < 12; i++) x[i] = mix(seed + id * 12 + i);
    if (dump == 2) for (int i = 0; i < 12; i++) x[i] = (i & 1) ? ~0ull - id * i : (u64)(id * i) << (i * 5);      // edge values: tiny, all-ones
    u64 bad = 0, nbad = 0;
    for (int it = 0; it < iters; it++) {
        if (V == 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) x[i] = pow7_c(x[i]);
        }
        if (V == 3) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const u64 in = x[i];
                x[i] = pow7_s(in, bad);
                if (__builtin_expect(bad != 0, 0)) { x[i] = pow7_lazy(in); bad = 0; nbad++; }
            }
        }
        if (V == 4) {
#pragma unroll
            for (int i = 0; i < 12; i++) x[i] = pow7_e(x[i]);
        }
        // a cheap lane-local shuffle between rounds so that the twelve values do not stay in a fixed orbit
        const u64 t = x[0];
#pragma unroll
        for (int i = 0; i < 11; i++) x[i] = x[i + 1] + (u64)i;
        x[11] = t ^ 0x5555;
    }
    if (dump) { for (int i = 0; i < 12; i++) out[id * 12 + i] = canon(x[i]); return; }
    u64 s = nbad; for (int i = 0; i < 12; i++) s += canon(x[i]) * (i + 1);
    out[id] = s;
}

// the whole permutation, matrix-core MDS, with the S-box variant V
template <int V>
__device__ inline void perm_v(u64 st[12], const MdsMfma &m) {
    u64 bad = 0;
    auto full = [&](const u64 *rc) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = add_lazy_canon(st[i], rc[i]);
        if (V == 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) st[i] = pow7_lazy(st[i]);
        } else if (V == 4) {
#pragma unroll
            for (int i = 0; i < 12; i++) { const u64 in = st[i]; st[i] = pow7_b(in, bad); if (__builtin_expect(bad != 0, 0)) { st[i] = pow7_lazy(in); bad = 0; } }
        } else if (V == 5) {
            u64 in[12];
#pragma unroll
            for (int i = 0; i < 12; i++) { in[i] = st[i]; st[i] = pow7_b(in[i], bad); }
            if (__builtin_expect(bad != 0, 0)) {
#pragma unroll
                for (int i = 0; i < 12; i++) st[i] = pow7_lazy(in[i]);
                bad = 0;
            }
        }
        mds_layer_mfma(st, m);
    };
#pragma unroll 1
    for (int r = 0; r < 4; r++) full(&POSEIDON_GL_RC[r * 12]);
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        const u64 in = add_lazy_canon(st[0], POSEIDON_GL_PARTIAL_C0[r]);
        if (V == 0) st[0] = pow7_lazy(in);
        else { st[0] = pow7_b(in, bad); if (__builtin_expect(bad != 0, 0)) { st[0] = pow7_lazy(in); bad = 0; } }
        mds_layer_mfma(st, m);
    }
#pragma unroll 1
    for (int r = 26; r < 30; r++) full(r == 26 ? POSEIDON_GL_RC26F : &POSEIDON_GL_RC[r * 12]);
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = canon(st[i]);
}
template <int V>
__global__ void __launch_bounds__(256, 2) k_perm(u64 *out, int iters, u64 seed, int dump) {
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 st[12];
    for (int i = 0; i < 12; i++) st[i] = mix(seed + id * 12 + i);
    MdsMfma m; mds_mfma_init(m);
    for (int i = 0; i < iters; i++) perm_v<V>(st, m);
    if (dump) { for (int i = 0; i < 12; i++) out[id * 12 + i] = st[i]; return; }
    u64 s = 0; for (int i = 0; i < 12; i++) s += st[i] * (i + 1);
    out[id] = s;
}

#define KPROD(NAME, W) __global__ void __launch_bounds__(256, W) NAME(u64 *out, int iters, u64 seed, int dump) { \
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x; u64 st[12]; \
    for (int i = 0; i < 12; i++) st[i] = mix(seed + id * 12 + i); \
    MdsMfma m; mds_mfma_init(m); \
    for (int i = 0; i < iters; i++) poseidon_perm_single(st, m); \
    if (dump) { for (int i = 0; i < 12; i++) out[id * 12 + i] = st[i]; return; } \
    u64 s = 0; for (int i = 0; i < 12; i++) s += st[i] * (i + 1); out[id] = s; }
KPROD(k_prod2, 2)
KPROD(k_prod3, 3)
KPROD(k_prod4, 4)

// the blocked partial rounds (poseidon_blocks.cuh)
#define KBLK(NAME, W) __global__ void __launch_bounds__(256, W) NAME(u64 *out, int iters, u64 seed, int dump) { \
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x; u64 st[12]; \
    for (int i = 0; i < 12; i++) st[i] = mix(seed + id * 12 + i); \
    if (dump == 2) for (int i = 0; i < 12; i++) st[i] = (i & 1) ? ~0ull - id * i : (u64)(id * i) << (i * 5); \
    MdsMfma m; poseidon_init(m); \
    for (int i = 0; i < iters; i++) poseidon_perm(st, m); \
    if (dump) { for (int i = 0; i < 12; i++) out[id * 12 + i] = st[i]; return; } \
    u64 s = 0; for (int i = 0; i < 12; i++) s += st[i] * (i + 1); out[id] = s; }
KBLK(k_blk2, 2)
KBLK(k_blk3, 3)
KBLK(k_blk4, 4)
__global__ void __launch_bounds__(256, 3) k_prod_edge(u64 *out, int iters) {
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x; u64 st[12];
    for (int i = 0; i < 12; i++) st[i] = (i & 1) ? ~0ull - id * i : (u64)(id * i) << (i * 5);
    MdsMfma m; mds_mfma_init(m);
    for (int i = 0; i < iters; i++) poseidon_perm_single(st, m);
    for (int i = 0; i < 12; i++) out[id * 12 + i] = st[i];
}

template <typename F>
float timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}

int main() {
    const int blocks = 256 * 8;
    const size_t n = (size_t)blocks * 256;
    u64 *out; CHECK(hipMalloc((void **)&out, 8 * n * 12));
    u64 *h0 = (u64 *)malloc(8 * n * 12), *h1 = (u64 *)malloc(8 * n * 12);
    // 1. hazard stress
    {
        const int it = 20000;
        hipLaunchKernelGGL(k_hazard<1>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); CHECK(hipMemcpy(h0, out, 8 * n, hipMemcpyDeviceToHost));
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_hazard<0>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); CHECK(hipMemcpy(h1, out, 8 * n, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < n; i++) bad += h0[i] != h1[i];
            printf("carry chain through vcc, unpadded vs padded: %zu of %zu lanes differ (%d chained adds each)\n", bad, n, it);
            hipLaunchKernelGGL(k_hazard<2>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); CHECK(hipMemcpy(h1, out, 8 * n, hipMemcpyDeviceToHost));
            bad = 0; for (size_t i = 0; i < n; i++) bad += h0[i] != h1[i];
            printf("carry chain through an SGPR pair, unpadded vs padded vcc: %zu of %zu lanes differ\n", bad, n);
        }
        hipLaunchKernelGGL(k_hazard<4>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); CHECK(hipMemcpy(h0, out, 8 * n, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_hazard<3>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); CHECK(hipMemcpy(h1, out, 8 * n, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n; i++) bad += h0[i] != h1[i];
        printf("mad carry-out -> addc, unpadded vs padded: %zu of %zu lanes differ\n", bad, n);
        float t0 = timeit([&] { hipLaunchKernelGGL(k_hazard<0>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); });
        float t1 = timeit([&] { hipLaunchKernelGGL(k_hazard<1>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); });
        float t2 = timeit([&] { hipLaunchKernelGGL(k_hazard<2>, dim3(blocks), dim3(256), 0, 0, out, it, 99ull); });
        printf("time: unpadded vcc %.2f ms, padded vcc %.2f ms, unpadded sgpr %.2f ms\n", t0, t1, t2);
    }
    // 2. S-box variants: values (0 = gl::mul_lazy as hipcc compiles it, 3 = carry-out products + fallback per S-box,
    //    4 = carry-out products with the borrow folded back in place)
    for (int dump = 1; dump <= 2; dump++) {
        hipLaunchKernelGGL(k_sbox<0>, dim3(blocks), dim3(256), 0, 0, out, 7, 4242ull, dump); CHECK(hipMemcpy(h0, out, 8 * n * 12, hipMemcpyDeviceToHost));
        for (int v = 3; v <= 4; v++) {
            if (v == 3) hipLaunchKernelGGL(k_sbox<3>, dim3(blocks), dim3(256), 0, 0, out, 7, 4242ull, dump);
            else hipLaunchKernelGGL(k_sbox<4>, dim3(blocks), dim3(256), 0, 0, out, 7, 4242ull, dump);
            CHECK(hipMemcpy(h1, out, 8 * n * 12, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < n * 12; i++) bad += h0[i] != h1[i];
            printf("sbox variant %d vs pow7_lazy (inputs %s): %zu of %zu values differ\n", v, dump == 1 ? "random" : "edge", bad, n * 12);
        }
    }
    // 3. S-box rates
    {
        const int it = 200;
        float t[3];
        t[0] = timeit([&] { hipLaunchKernelGGL(k_sbox<0>, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        t[1] = timeit([&] { hipLaunchKernelGGL(k_sbox<3>, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        t[2] = timeit([&] { hipLaunchKernelGGL(k_sbox<4>, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        const char *nm[3] = { "pow7_lazy (hipcc's plain-C product)", "carry-out products + fallback per S-box", "carry-out products, borrow folded in place" };
        for (int v = 0; v < 3; v++) {
            double ops = (double)n * it * 12;
            printf("%-44s %8.3f ms  %7.2f G sbox/s  (%.1f cyc per wave-sbox per SIMD @2.4GHz)\n", nm[v], t[v], ops / t[v] / 1e6, 2.4e9 * 1024 * 64 / (ops / (t[v] * 1e-3)));
        }
    }
    // 4. permutation: values and rates (0 = pow7_lazy, 4 = pow7_b checked per S-box, 5 = pow7_b checked per layer)
    {
        hipLaunchKernelGGL(k_perm<0>, dim3(blocks), dim3(256), 0, 0, out, 3, 77ull, 1); CHECK(hipMemcpy(h0, out, 8 * n * 12, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_perm<4>, dim3(blocks), dim3(256), 0, 0, out, 3, 77ull, 1); CHECK(hipMemcpy(h1, out, 8 * n * 12, hipMemcpyDeviceToHost));
        { size_t b4 = 0; for (size_t i = 0; i < n * 12; i++) b4 += h0[i] != h1[i]; printf("permutation with pow7_b (check per S-box): %zu words differ\n", b4); }
        hipLaunchKernelGGL(k_perm<5>, dim3(blocks), dim3(256), 0, 0, out, 3, 77ull, 1); CHECK(hipMemcpy(h1, out, 8 * n * 12, hipMemcpyDeviceToHost));
        { size_t b5 = 0; for (size_t i = 0; i < n * 12; i++) b5 += h0[i] != h1[i]; printf("permutation with pow7_b (check per layer): %zu words differ\n", b5); }
        const int it = 20;
        double np = (double)n * it;
        float t0 = timeit([&] { hipLaunchKernelGGL(k_perm<0>, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        float t4 = timeit([&] { hipLaunchKernelGGL(k_perm<4>, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        float t5 = timeit([&] { hipLaunchKernelGGL(k_perm<5>, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        printf("permutation, pow7_lazy        %8.3f ms  %6.3f G perm/s\n", t0, np / t0 / 1e6);
        printf("permutation, pow7_b / S-box   %8.3f ms  %6.3f G perm/s\n", t4, np / t4 / 1e6);
        printf("permutation, pow7_b / layer   %8.3f ms  %6.3f G perm/s\n", t5, np / t5 / 1e6);
    }
    {
        const int it = 20; double np = (double)n * it;
        hipLaunchKernelGGL(k_prod3, dim3(blocks), dim3(256), 0, 0, out, 3, 77ull, 1); CHECK(hipMemcpy(h1, out, 8 * n * 12, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n * 12; i++) bad += h0[i] != h1[i];
        printf("production poseidon_perm vs the pow7_lazy permutation: %zu of %zu words differ\n", bad, n * 12);
        float t2 = timeit([&] { hipLaunchKernelGGL(k_prod2, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        float t3 = timeit([&] { hipLaunchKernelGGL(k_prod3, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        float t4 = timeit([&] { hipLaunchKernelGGL(k_prod4, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        printf("production poseidon_perm, launch bounds 2 / 3 / 4 waves per SIMD: %.3f / %.3f / %.3f G perm/s\n", np / t2 / 1e6, np / t3 / 1e6, np / t4 / 1e6);
    }
    {
        const int it = 100; double np = (double)n * it;
        hipLaunchKernelGGL(k_prod3, dim3(blocks), dim3(256), 0, 0, out, 3, 77ull, 1); CHECK(hipMemcpy(h0, out, 8 * n * 12, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_blk3, dim3(blocks), dim3(256), 0, 0, out, 3, 77ull, 1); CHECK(hipMemcpy(h1, out, 8 * n * 12, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n * 12; i++) bad += h0[i] != h1[i];
        printf("blocked partial rounds vs production poseidon_perm (random states, 3 chained): %zu of %zu words differ\n", bad, n * 12);
        hipLaunchKernelGGL(k_prod_edge, dim3(blocks), dim3(256), 0, 0, out, 2); CHECK(hipMemcpy(h0, out, 8 * n * 12, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_blk3, dim3(blocks), dim3(256), 0, 0, out, 2, 77ull, 2); CHECK(hipMemcpy(h1, out, 8 * n * 12, hipMemcpyDeviceToHost));
        bad = 0; for (size_t i = 0; i < n * 12; i++) bad += h0[i] != h1[i];
        printf("blocked partial rounds vs production poseidon_perm (edge states, 2 chained): %zu of %zu words differ\n", bad, n * 12);
        float t2 = timeit([&] { hipLaunchKernelGGL(k_blk2, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        float t3 = timeit([&] { hipLaunchKernelGGL(k_blk3, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        float t4 = timeit([&] { hipLaunchKernelGGL(k_blk4, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        printf("blocked partial rounds, launch bounds 2 / 3 / 4 waves per SIMD: %.3f / %.3f / %.3f G perm/s\n", np / t2 / 1e6, np / t3 / 1e6, np / t4 / 1e6);
        float p3 = timeit([&] { hipLaunchKernelGGL(k_prod3, dim3(blocks), dim3(256), 0, 0, out, it, 1ull, 0); });
        printf("production again (3 waves): %.3f G perm/s\n", np / p3 / 1e6);
    }
    return 0;
}
