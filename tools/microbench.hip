// Instruction-rate microbenchmarks for the Goldilocks / Poseidon inner loops on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "poseidon_gl.cuh"
using namespace gl;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64 add_c(u64 a, u64 b) { u64 s, t; bool c1 = __builtin_uaddl_overflow(a, b, &s); bool c2 = __builtin_uaddl_overflow(s, EPS, &t); return (c1 | c2) ? t : s; }
__device__ __forceinline__ u64 sub_c(u64 a, u64 b) { u64 d; bool br = __builtin_usubl_overflow(a, b, &d); return br ? d - EPS : d; }
__device__ __forceinline__ u64 add_lazy_c(u64 a, u64 b) { u64 s; bool c = __builtin_uaddl_overflow(a, b, &s); return c ? s + EPS : s; }
__device__ __forceinline__ u64 reduce128_v3(u64 lo, u64 hi) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0; bool br = __builtin_usubl_overflow(lo, (u64)hh, &t0);
    u64 t1 = (u64)hl * EPS;
    u64 t2; bool c = __builtin_uaddl_overflow(t0, t1, &t2);
    u64 adj = (c ? EPS : 0) - (br ? EPS : 0);
    return t2 + adj;
}
__device__ __forceinline__ u64 mul_v3(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 t = (u64)a0 * b0;
    u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v = (u64)a1 * b0 + (u32)u;
    u64 w = (u64)a1 * b1 + (u >> 32) + (v >> 32);
    return reduce128_v3((v << 32) | (u32)t, w);
}

// ---- v4: product as the compiler builds it; reduction around the multiply-add's carry-out and the subtract's borrow
// (with the overflow builtins the compiler re-derives both through 64-bit compares):
//   X + c 2^64 = lo + hl (2^32-1);   Y - b 2^64 = X - hh;   result = Y + (c - b)(2^32-1)    -- no step can wrap twice
__device__ __forceinline__ u64 reduce128_v4(u64 lo, u64 hi) {
    const u32 hl = (u32)hi, hh = (u32)(hi >> 32);
    u64 x, cm;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=&v"(x), "=s"(cm) : "v"(hl), "v"(lo));
    u32 yl = (u32)x, yh = (u32)(x >> 32), e1, nlo, nhi;
    asm("v_sub_co_u32 %0, vcc, %0, %5\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_cndmask_b32 %2, 0, -1, %6\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %3, 0, 1, vcc\n\t"
        "v_cndmask_b32 %4, 0, -1, vcc"
        : "+v"(yl), "+v"(yh), "=&v"(e1), "=&v"(nlo), "=&v"(nhi) : "v"(hh), "s"(cm) : "vcc");
    return (((u64)yh << 32) | yl) + (u64)e1 + (((u64)nhi << 32) | nlo);
}
__device__ __forceinline__ u64 mul_v4(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 t = (u64)a0 * b0;
    u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v = (u64)a1 * b0 + (u32)u;
    u64 w = (u64)a1 * b1 + (u >> 32) + (v >> 32);
    return reduce128_v4((v << 32) | (u32)t, w);
}
__device__ __forceinline__ u64 pow7_v4(u64 x) { u64 x2 = mul_v4(x, x), x3 = mul_v4(x2, x), x4 = mul_v4(x2, x2); return mul_v4(x3, x4); }
// ---- carry-flag variant: product by columns with the multiply-add's carry-out, reduction on borrow/carry chains
__device__ __forceinline__ void acc_mad64(u64 &lo, u32 &hi, u32 x, u32 y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ u64 mul_cf(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    u64 acc = t >> 32; u32 hi = 0;
    acc_mad64(acc, hi, a0, b1);
    acc_mad64(acc, hi, a1, b0);
    const u32 w1 = (u32)acc;
    acc = (acc >> 32) | ((u64)hi << 32);
    acc = (u64)a1 * b1 + acc;                                   // cannot overflow: the full product is < 2^128
    const u32 w2 = (u32)acc, w3 = (u32)(acc >> 32);
    const u64 lo = ((u64)w1 << 32) | (u32)t;
    // lo + w2*EPS - w3 with the wraps folded back: carry -> +EPS, borrow -> -EPS
    u64 r; u32 c, br, rl, rh;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc" : "=&v"(r), "=&v"(c) : "v"(w2), "v"(lo) : "vcc");
    rl = (u32)r; rh = (u32)(r >> 32);
    asm("v_sub_co_u32 %0, vcc, %0, %3\n\tv_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\tv_subb_co_u32 %2, vcc, %4, 0, vcc"
        : "+v"(rl), "+v"(rh), "=&v"(br) : "v"(w3), "v"(c) : "vcc");
    // br = c - 0 - borrow  in {-1, 0, 1} (two's complement): the net number of 2^64 wraps still to add back as EPS
    // r += br * EPS:  br = +1 -> r + EPS = (lo - 1, hi + 1 - borrow);  br = -1 -> r - EPS = (lo - 0xFFFFFFFF, hi - borrow)
    const u32 e = 0u - ((br + 1u) >> 1);                         // 0xFFFFFFFF when br = +1, else 0
    asm("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %1, vcc, %1, %3, vcc" : "+v"(rl), "+v"(rh) : "v"(br), "v"(e) : "vcc");
    return ((u64)rh << 32) | rl;
}
__device__ __forceinline__ u64 pow7_cf(u64 x) { u64 x2 = mul_cf(x, x), x3 = mul_cf(x2, x), x4 = mul_cf(x2, x2); return mul_cf(x3, x4); }

__device__ __forceinline__ u64 pow7_v3(u64 x) { u64 x2 = mul_v3(x, x), x3 = mul_v3(x2, x), x4 = mul_v3(x2, x2); return mul_v3(x3, x4); }

// MDS via v_dot2_u32_u16 on 16-bit limb planes
__device__ __forceinline__ void mds_dot2(u64 st[12]) {
    constexpr u32 MC[12] = { 17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20 };
    u32 Pl[4][6];
#pragma unroll
    for (int jj = 0; jj < 6; jj++) {
        u32 l0 = (u32)st[2 * jj], h0 = (u32)(st[2 * jj] >> 32), l1 = (u32)st[2 * jj + 1], h1 = (u32)(st[2 * jj + 1] >> 32);
        Pl[0][jj] = __builtin_amdgcn_perm(l1, l0, 0x05040100);   // (l0 & 0xFFFF) | (l1 << 16)
        Pl[1][jj] = __builtin_amdgcn_perm(l1, l0, 0x07060302);   // (l0 >> 16) | (l1 & 0xFFFF0000)
        Pl[2][jj] = __builtin_amdgcn_perm(h1, h0, 0x05040100);
        Pl[3][jj] = __builtin_amdgcn_perm(h1, h0, 0x07060302);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u32 s[4];
#pragma unroll
        for (int l = 0; l < 4; l++) {
            u32 acc = 0;
#pragma unroll
            for (int jj = 0; jj < 6; jj++) {
                const u32 m0 = MC[(2 * jj - i + 12) % 12] + ((i == 0 && jj == 0) ? 8u : 0u), m1 = MC[(2 * jj + 1 - i + 12) % 12];
                acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, Pl[l][jj]), __builtin_bit_cast(us2, m0 | (m1 << 16)), acc, false);
            }
            s[l] = acc;
        }
        // value = s0 + s1<<16 + s2<<32 + s3<<48  (< 2^75)
        u64 A = (u64)s[0] + ((u64)s[1] << 16);
        u64 B = (u64)s[2] + ((u64)s[3] << 16);
        u64 lo = A + (B << 32);
        u64 hi = (B >> 32) + (lo < A ? 1 : 0);
        st[i] = reduce128_v3(lo, hi);
    }
}
__device__ __forceinline__ void mds_v3(u64 st[12]) {
    constexpr u32 MC[12] = { 17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20 };
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)st[j]; hi[j] = (u32)(st[j] >> 32); }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u64 al = 0, ah = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) { const u32 m = MC[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u); al += (u64)lo[j] * m; ah += (u64)hi[j] * m; }
        u64 l = al + (ah << 32);
        u64 h = (ah >> 32) + (l < al ? 1 : 0);
        st[i] = reduce128_v3(l, h);
    }
}
template <int V>
__global__ void __launch_bounds__(256) k_mdsv(uint64_t *out, int iters, uint64_t seed) {
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = seed * (i + 1) * 0x9E3779B97F4A7C15ull + threadIdx.x * 0x123456789ull;
    for (int i = 0; i < iters; i++) { if (V == 0) mds_layer(st); if (V == 1) mds_dot2(st); if (V == 2) mds_v3(st); }
    uint64_t s = 0; for (int i = 0; i < 12; i++) s += canon(st[i]) * (i + 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k_ops(uint64_t *out, int iters, uint64_t seed) {
    uint64_t a = seed + threadIdx.x + 1, b = seed * 3 + blockIdx.x + 7, c = seed ^ 0x1234567, d = seed + 99;
    uint32_t a0 = (uint32_t)a, b0 = (uint32_t)b, c0 = (uint32_t)c, d0 = (uint32_t)d;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) { a = mul_lazy(a, b); b = mul_lazy(b, c); c = mul_lazy(c, d); d = mul_lazy(d, a); }
            if (OP == 1) { a = (uint64_t)a0 * b0 + a; b = (uint64_t)b0 * c0 + b; c = (uint64_t)c0 * d0 + c; d = (uint64_t)d0 * a0 + d; a0 = (uint32_t)a; b0 = (uint32_t)b; c0 = (uint32_t)c; d0 = (uint32_t)d; }
            if (OP == 2) { a0 = a0 * b0 + 1; b0 = b0 * c0 + 1; c0 = c0 * d0 + 1; d0 = d0 * a0 + 1; }
            if (OP == 3) { a0 = __umulhi(a0, b0) + 1; b0 = __umulhi(b0, c0) + 1; c0 = __umulhi(c0, d0) + 1; d0 = __umulhi(d0, a0) + 1; }
            if (OP == 4) { a0 = __umul24(a0, b0) + c0; b0 = __umul24(b0, c0) + d0; c0 = __umul24(c0, d0) + a0; d0 = __umul24(d0, a0) + b0; }
            if (OP == 5) { a = add_lazy(a, b); b = add_lazy(b, c); c = add_lazy(c, d); d = add_lazy(d, a); }
            if (OP == 6) { a = a + b; b = b + c; c = c + d; d = d + a; }
            if (OP == 7) { a = pow7_lazy(a); b = pow7_lazy(b); c = pow7_lazy(c); d = pow7_lazy(d); }
            if (OP == 8) { a = add(canon(a), canon(b)); b = sub(canon(b), canon(c)); c = add(canon(c), canon(d)); d = sub(canon(d), canon(a)); }
            if (OP == 10) { a = mul_v3(a, b); b = mul_v3(b, c); c = mul_v3(c, d); d = mul_v3(d, a); }
            if (OP == 11) { a = add_c(a, b); b = sub_c(b, c); c = add_c(c, d); d = sub_c(d, a); }
            if (OP == 12) { a = add_lazy_c(a, b); b = add_lazy_c(b, c); c = add_lazy_c(c, d); d = add_lazy_c(d, a); }
            if (OP == 13) { a0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a0), __builtin_bit_cast(us2, b0), c0, false); b0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, b0), __builtin_bit_cast(us2, c0), d0, false); c0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c0), __builtin_bit_cast(us2, d0), a0, false); d0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, d0), __builtin_bit_cast(us2, a0), b0, false); }
            if (OP == 14) { a0 = __builtin_amdgcn_udot4(a0, b0, c0, false); b0 = __builtin_amdgcn_udot4(b0, c0, d0, false); c0 = __builtin_amdgcn_udot4(c0, d0, a0, false); d0 = __builtin_amdgcn_udot4(d0, a0, b0, false); }
            if (OP == 15) { a0 = __builtin_amdgcn_perm(a0, b0, 0x05040100); b0 = __builtin_amdgcn_perm(b0, c0, 0x07060302); c0 = __builtin_amdgcn_perm(c0, d0, 0x05040100); d0 = __builtin_amdgcn_perm(d0, a0, 0x07060302) + 1; }
            if (OP == 16) { a = pow7_v3(a); b = pow7_v3(b); c = pow7_v3(c); d = pow7_v3(d); }
            if (OP == 17) { a0 = __umul24(a0, b0) + c0; b0 = b0 * 3 + d0; c0 = (c0 << 3) + a0; d0 = d0 ^ b0; }
            if (OP == 18) { a = mul_cf(a, b); b = mul_cf(b, c); c = mul_cf(c, d); d = mul_cf(d, a); }
            if (OP == 19) { a = pow7_cf(a); b = pow7_cf(b); c = pow7_cf(c); d = pow7_cf(d); }
            if (OP == 20) { a = mul_v4(a, b); b = mul_v4(b, c); c = mul_v4(c, d); d = mul_v4(d, a); }
            if (OP == 21) { a = pow7_v4(a); b = pow7_v4(b); c = pow7_v4(c); d = pow7_v4(d); }
            if (OP == 9) { a0 = __mulhi((int)a0, (int)b0) + 1; b0 = (a0 << 3) + c0; c0 = (b0 >> 5) ^ d0; d0 = c0 + a0; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + a0 + b0 + c0 + d0;
}

__global__ void __launch_bounds__(256) k_mds(uint64_t *out, int iters, uint64_t seed) {
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = seed * (i + 1) + threadIdx.x;
    for (int i = 0; i < iters; i++) mds_layer(st);
    uint64_t s = 0; for (int i = 0; i < 12; i++) s += st[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_perm(uint64_t *out, int iters, uint64_t seed) {
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = seed * (i + 1) + threadIdx.x;
    for (int i = 0; i < iters; i++) poseidon_perm(st);
    uint64_t s = 0; for (int i = 0; i < 12; i++) s += st[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}

int main() {
    uint64_t *out; CHECK(hipMalloc((void **)&out, 8ull * 256 * 4096));
    const int blocks = 256 * 8, iters = 2000;
    const char *names[] = { "mul_lazy (GL mul)", "mad_u64_u32", "mul_lo_u32+add", "mul_hi_u32+add", "mul_u24+add", "add_lazy", "add u64", "pow7_lazy", "canon add/sub", "misc32", "mul_v3", "add_c/sub_c", "add_lazy_c", "udot2", "udot4", "v_perm", "pow7_v3", "4 simple", "mul_cf (carry flags)", "pow7_cf", "mul_v4 (carry-out reduce)", "pow7_v4" };
    const double per[] = { 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32 };   // ops per thread per iter
#define RUN(OP) { float ms = timeit([&] { hipLaunchKernelGGL(k_ops<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ull); }); \
    double ops = (double)blocks * 256 * iters * per[OP]; printf("%-20s %8.3f ms  %8.2f Gop/s  (%.2f cyc/wave-op/SIMD @2.4GHz)\n", names[OP], ms, ops / ms / 1e6, 2.4e9 * 1024 * 64 / (ops / (ms * 1e-3))); }
    RUN(20) RUN(21) RUN(0) RUN(7)
    { uint64_t *hv = (uint64_t *)malloc(8 * 256), *hw = (uint64_t *)malloc(8 * 256); int bad = 0;
      for (uint64_t seed : { 0xFFFFFFFF00000000ull, 1ull, 0xFFFFFFFFull, 0x123456789ABCDEFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull }) {
        hipLaunchKernelGGL(k_ops<0>, dim3(1), dim3(256), 0, 0, out, 3, seed); hipMemcpy(hv, out, 2048, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_ops<20>, dim3(1), dim3(256), 0, 0, out, 3, seed); hipMemcpy(hw, out, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 256; i++) bad += (hv[i] % 0xFFFFFFFF00000001ull) != (hw[i] % 0xFFFFFFFF00000001ull); }
      printf("mul_v4 == mul_lazy (mod p): %s\n", bad ? "NO" : "yes"); }
    return 0;
    RUN(18) RUN(19)
    { uint64_t *hv = (uint64_t *)malloc(8 * 256), *hw = (uint64_t *)malloc(8 * 256);
      hipLaunchKernelGGL(k_ops<0>, dim3(1), dim3(256), 0, 0, out, 3, 0xFFFFFFFF00000000ull); hipMemcpy(hv, out, 2048, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(k_ops<18>, dim3(1), dim3(256), 0, 0, out, 3, 0xFFFFFFFF00000000ull); hipMemcpy(hw, out, 2048, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < 256; i++) bad += (hv[i] % 0xFFFFFFFF00000001ull) != (hw[i] % 0xFFFFFFFF00000001ull); printf("mul_cf == mul_lazy (mod p): %s\n", bad ? "NO" : "yes"); }
    RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    { uint64_t h[3][64]; int it = 3;
      hipLaunchKernelGGL(k_mdsv<0>, dim3(1), dim3(64), 0, 0, out, it, 777ull); hipMemcpy(h[0], out, 512, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(k_mdsv<1>, dim3(1), dim3(64), 0, 0, out, it, 777ull); hipMemcpy(h[1], out, 512, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(k_mdsv<2>, dim3(1), dim3(64), 0, 0, out, it, 777ull); hipMemcpy(h[2], out, 512, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < 64; i++) bad += (h[0][i] != h[1][i]) + (h[0][i] != h[2][i]);
      printf("mds variants agree: %s\n", bad ? "NO" : "yes"); }
    { int it = 200; float ms = timeit([&] { hipLaunchKernelGGL(k_mdsv<1>, dim3(blocks), dim3(256), 0, 0, out, it, 777ull); });
      double n = (double)blocks * 256 * it; printf("%-20s %8.3f ms  %8.3f G layers/s (%.0f cyc/wave-layer)\n", "mds_dot2", ms, n / ms / 1e6, 2.4e9 * 1024 * 64 / (n / (ms * 1e-3))); }
    { int it = 200; float ms = timeit([&] { hipLaunchKernelGGL(k_mdsv<2>, dim3(blocks), dim3(256), 0, 0, out, it, 777ull); });
      double n = (double)blocks * 256 * it; printf("%-20s %8.3f ms  %8.3f G layers/s (%.0f cyc/wave-layer)\n", "mds_v3", ms, n / ms / 1e6, 2.4e9 * 1024 * 64 / (n / (ms * 1e-3))); }
    { int it = 200; float ms = timeit([&] { hipLaunchKernelGGL(k_mds, dim3(blocks), dim3(256), 0, 0, out, it, 777ull); });
      double n = (double)blocks * 256 * it; printf("%-20s %8.3f ms  %8.3f G layers/s\n", "mds_layer", ms, n / ms / 1e6); }
    { int it = 20; float ms = timeit([&] { hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, out, it, 777ull); });
      double n = (double)blocks * 256 * it; printf("%-20s %8.3f ms  %8.3f G perm/s\n", "poseidon_perm", ms, n / ms / 1e6); }
    return 0;
}
