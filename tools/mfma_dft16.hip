// The register step of the transform tiles (16-point DFT of one sub-transform per lane) on the matrix cores against the vector form:
// correctness of gl_dft16_mfma.cuh against the definition, and cycles per element and stage of both forms, with and without the seam
// twiddle product, at the occupancy the tile kernels would have (vector form: 4 workgroups of 256 per CU; matrix form: 3, the 16 KB
// operand table takes the fourth's LDS).
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I pil2-stark-js_amd/csrc -I tools tools/mfma_dft16.hip -o tools/mfma_dft16
//        (-DDFT16_ABLATE=2 / 3: the matrix instructions / the vector work alone; -DDFT16_CHAINS=1,2,4; -DDFT16_AREG; -DWAVES=n)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "gl_field.cuh"
#include "gl_fermat.cuh"
#include "gl_dft16_mfma.cuh"
using namespace gl;

#ifndef NTT_MUL
#define NTT_MUL(a, b) mul_lazy_x(a, b)
#endif

__device__ __forceinline__ u32 brev4(u32 i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

// MODE 0: fermat DIF + reduce; 1: mfma; 2: fermat + twiddle; 3: mfma + twiddle
#ifndef WAVES
#define WAVES 3
#endif
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) k_time(u64 *out, int iters, u64 seed, const u32 *gTable, const u64 *gConsts) {
    extern __shared__ u32 lds[];
    dft16::Consts k;
    if (MODE & 1) k = dft16::init(lds, gTable, gConsts);
    u64 x[16];
    unsigned long long c0, r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    u64 tw = canon(seed * 77 + threadIdx.x * 0x9E3779B97F4A7C15ull);
    for (int i = 0; i < 16; i++) x[i] = canon((seed + i) * 0x9E3779B97F4A7C15ull + (threadIdx.x + blockIdx.x * 256) * 0x123456789ull);
    for (int it = 0; it < iters; it++) {
        if (MODE & 1) dft16::run(x, k);
        else {
            fermat::f128 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = fermat::from_gl(x[i]);
            fermat::dft_dif<4, false>(v);
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = fermat::to_gl_lazy(v[i]);
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = NTT_MUL(x[i], tw);
            tw = add(tw, 12345);
        }
    }
    u64 s = 0;
    for (int i = 0; i < 16; i++) s = add(mul(s, 3), canon(x[i]));
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    unsigned long long c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
    if (blockIdx.x == 17 && threadIdx.x == 0) { out[256 * 8192 - 2] = c1 - c0; out[256 * 8192 - 1] = r1 - r0; }
}

// lane (n, h) of a wave holds values 8h..8h+7 of sub-transforms n (group 0) and 32 + n (group 1) of the wave's 64; slot i carries
// X[fq[i]], X[q] = sum_r x[r] w^(q tm[r])
__global__ void __launch_bounds__(256) k_check(const u64 *in, u64 *got, u64 *want, u64 w, int dit, const u32 *gTable, const u64 *gConsts) {
    extern __shared__ u32 lds[];
    dft16::Consts k = dft16::init(lds, gTable, gConsts);
    const u32 lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const size_t wave = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) / 64;
    u64 x[16];
    for (int G = 0; G < 2; G++) {
        const size_t base = (wave * 64 + 32 * G + n) * 16;
        for (int j = 0; j < 8; j++) x[8 * G + j] = in[base + 8 * h + j];
        for (int j = 0; j < 8; j++) {
            const u32 i = 8 * h + j, q = dit ? i : brev4(i);
            u64 acc = 0;
            for (int r = 0; r < 16; r++) acc = add(acc, mul(in[base + r], gl::pow(w, (u64)(q * (dit ? brev4(r) : r)))));
            want[base + i] = acc;
        }
    }
    dft16::run(x, k);
    for (int G = 0; G < 2; G++)
        for (int j = 0; j < 8; j++) got[(wave * 64 + 32 * G + n) * 16 + 8 * h + j] = canon(x[8 * G + j]);
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using gl_dft16_host::hpow; using gl_dft16_host::HP;

template <typename F>
float timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}

int main() {
    const int nT = 256 * 64;
    u64 *din, *dgot, *dwant, *out, *dConsts; u32 *dTable;
    CHECK(hipMalloc((void **)&din, 8ull * nT * 16)); CHECK(hipMalloc((void **)&dgot, 8ull * nT * 16)); CHECK(hipMalloc((void **)&dwant, 8ull * nT * 16));
    CHECK(hipMalloc((void **)&out, 8ull * 256 * 8192)); CHECK(hipMalloc((void **)&dTable, 16384)); CHECK(hipMalloc((void **)&dConsts, 72));
    u64 *h = (u64 *)malloc(8ull * nT * 16);
    u64 s = 88172645463325252ull;
    for (size_t i = 0; i < (size_t)nT * 16; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        u64 v = s;                                                       // any 64-bit word, canonical or not
        if (i % 97 == 0) v = HP - 1; if (i % 101 == 0) v = 0; if (i % 103 == 0) v = 0xFFFFFFFFull; if (i % 107 == 0) v = 0xFFFFFFFFFFFFFFFFull;
        if (i % 109 == 0) v = 0x8080808080808080ull; if (i % 113 == 0) v = 0x7F7F7F7F7F7F7F7Full;
        h[i] = v;
    }
    for (size_t i = 0; i < 16 * 64; i++) h[i] = i < 512 ? 0xFFFFFFFFFFFFFFFFull : 0;          // the largest and the smallest plane sums
    CHECK(hipMemcpy(din, h, 8ull * nT * 16, hipMemcpyHostToDevice));
    int bad = 0;
    const u64 w16 = hpow(7277203076849721926ull, 1ull << 28);          // F.w[4]
    for (int dit = 0; dit < 2; dit++)
        for (int inv = 0; inv < 2; inv++) {
            const u64 w = inv ? hpow(w16, HP - 2) : w16;
            int fq[16], tm[16];
            for (int i = 0; i < 16; i++) { const int b = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); fq[i] = dit ? i : b; tm[i] = dit ? b : i; }
            std::vector<uint32_t> table; uint64_t consts[9];
            gl_dft16_host::build(w, fq, tm, table, consts);
            CHECK(hipMemcpy(dTable, table.data(), 16384, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dConsts, consts, 72, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_check, dim3(nT / 256), dim3(256), 16384, 0, din, dgot, dwant, w, dit, dTable, dConsts);
            CHECK(hipDeviceSynchronize());
            u64 *g = (u64 *)malloc(8ull * nT * 16), *wv = (u64 *)malloc(8ull * nT * 16);
            CHECK(hipMemcpy(g, dgot, 8ull * nT * 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(wv, dwant, 8ull * nT * 16, hipMemcpyDeviceToHost));
            size_t nb = 0; for (size_t i = 0; i < (size_t)nT * 16; i++) nb += g[i] != wv[i];
            printf("dft16 on the matrix cores, %s %s: %s (%zu of %zu differ)\n", inv ? "inverse" : "forward", dit ? "DIT" : "DIF", nb ? "MISMATCH" : "ok", nb, (size_t)nT * 16);
            bad += nb != 0; free(g); free(wv);
        }
    const int iters = 200;
    const char *names[] = { "vector (Z/(2^96+1)) + reduce", "matrix cores", "vector + twiddle product", "matrix cores + twiddle product" };
#define RUN(M, LDS, WG) { CHECK(hipFuncSetAttribute((const void *)k_time<M>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
    const int blocks = 256 * WG * 2; float ms = timeit([&] { hipLaunchKernelGGL(k_time<M>, dim3(blocks), dim3(256), LDS, 0, out, iters, 12345ull, dTable, dConsts); }); \
    double el = (double)blocks * 256 * iters * 16; u64 ck[2]; CHECK(hipMemcpy(ck, out + 256 * 8192 - 2, 16, hipMemcpyDeviceToHost)); \
    printf("%-34s %d workgroups/CU %8.3f ms  %.2f issue cycles per element and stage (2.4 GHz nominal), shader clock in the kernel %.2f GHz\n", names[M], WG, ms, 2.4e9 * 1024 * 64 / (el / (ms * 1e-3)) / 4, ck[1] ? 0.1 * (double)ck[0] / (double)ck[1] : 0.0); }
    RUN(0, 34 * 1024, 4) RUN(1, 50 * 1024, 3) RUN(1, 34 * 1024, 4) RUN(2, 34 * 1024, 4) RUN(3, 50 * 1024, 3) RUN(3, 34 * 1024, 4)
    printf(bad ? "FAILED\n" : "all checks passed\n");
    return bad;
}
