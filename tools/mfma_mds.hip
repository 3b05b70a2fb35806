// Probe + microbenchmark for the matrix-core MDS layer (csrc/poseidon_mds_mfma.cuh) on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc tools/mfma_mds.hip -o tools/mfma_mds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "poseidon_gl.cuh"
#include "poseidon_mds_mfma.cuh"
using namespace gl;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_probe(const v4i *a, const v4i *b, v16i *d) {
    int l = threadIdx.x;
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], c, 0, 0, 0);
    d[l] = c;
}

template <int V>
__global__ void __launch_bounds__(256, 2) k_mds(uint64_t *out, int iters, uint64_t seed, int dump) {
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = (seed + blockIdx.x) * (i + 1) * 0x9E3779B97F4A7C15ull + threadIdx.x * 0x123456789abcdefull;
    if (dump == 2) for (int i = 0; i < 12; i++) st[i] = ~0ull - (uint64_t)i * threadIdx.x;      // all-ones bytes: the largest sums
    MdsMfma m; if (V == 1) mds_mfma_init(m);
    for (int i = 0; i < iters; i++) { if (V == 0) mds_layer(st); else mds_layer_mfma(st, m); }
    if (dump) { for (int i = 0; i < 12; i++) out[(blockIdx.x * blockDim.x + threadIdx.x) * 12 + i] = canon(st[i]); return; }
    uint64_t s = 0; for (int i = 0; i < 12; i++) s += canon(st[i]) * (i + 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
__global__ void __launch_bounds__(256, 2) k_perm(uint64_t *out, int iters, uint64_t seed, int dump) {
    uint64_t st[12];
    for (int i = 0; i < 12; i++) st[i] = (seed + blockIdx.x) * (i + 1) * 0x9E3779B97F4A7C15ull + threadIdx.x * 0x123456789abcdefull;
    MdsMfma m; if (V == 1) mds_mfma_init(m);
    for (int i = 0; i < iters; i++) { if (V == 0) poseidon_perm(st); else poseidon_perm(st, m); }
    if (dump) { for (int i = 0; i < 12; i++) out[(blockIdx.x * blockDim.x + threadIdx.x) * 12 + i] = st[i]; return; }
    uint64_t s = 0; for (int i = 0; i < 12; i++) s += st[i] * (i + 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// issue/throughput of the MFMA itself and of MFMA + independent vector work
template <int NV>
__global__ void __launch_bounds__(256) k_rate(uint64_t *out, int iters, uint64_t seed) {
    v4i a, b; v16i acc[4];
    for (int e = 0; e < 4; e++) { a[e] = (int)(seed * (e + 1) + threadIdx.x); b[e] = (int)(seed * (e + 7) ^ threadIdx.x); }
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) acc[q][r] = 0;
    uint64_t x0 = seed + threadIdx.x, x1 = seed * 3, x2 = seed * 5, x3 = seed * 7;
    uint32_t y0 = (uint32_t)seed | 1, y1 = y0 + 2, y2 = y0 + 4, y3 = y0 + 6;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[q], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) { x0 = (uint64_t)y0 * y1 + x0; x1 = (uint64_t)y1 * y2 + x1; x2 = (uint64_t)y2 * y3 + x2; x3 = (uint64_t)y3 * y0 + x3; y0 = (uint32_t)x0; y1 = (uint32_t)x1; y2 = (uint32_t)x2; y3 = (uint32_t)x3; }
        }
    }
    uint64_t s = x0 + x1 + x2 + x3;
    for (int q = 0; q < 4; q++) for (int r = 0; r < 16; r++) s += (uint32_t)acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
__global__ void __launch_bounds__(256) k_valu(uint64_t *out, int iters, uint64_t seed) {
    uint64_t x0 = seed + threadIdx.x, x1 = seed * 3, x2 = seed * 5, x3 = seed * 7;
    uint32_t y0 = (uint32_t)seed | 1, y1 = y0 + 2, y2 = y0 + 4, y3 = y0 + 6;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int v = 0; v < NV; v++) { x0 = (uint64_t)y0 * y1 + x0; x1 = (uint64_t)y1 * y2 + x1; x2 = (uint64_t)y2 * y3 + x2; x3 = (uint64_t)y3 * y0 + x3; y0 = (uint32_t)x0; y1 = (uint32_t)x1; y2 = (uint32_t)x2; y3 = (uint32_t)x3; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <typename F>
float timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}

int main() {
    // ---- 1. operand / result layout of v_mfma_i32_32x32x32_i8 ----
    {
        int8_t hA[64][16], hB[64][16]; int hD[64][16];
        srand(5);
        for (int l = 0; l < 64; l++) for (int q = 0; q < 16; q++) { hA[l][q] = (int8_t)(rand() % 255 - 127); hB[l][q] = (int8_t)(rand() % 255 - 127); }
        v4i *dA, *dB; v16i *dD;
        CHECK(hipMalloc((void **)&dA, 1024)); CHECK(hipMalloc((void **)&dB, 1024)); CHECK(hipMalloc((void **)&dD, 4096));
        CHECK(hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CHECK(hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost));
        // hypothesis: A lane (i, g), B lane (n, g) pair byte q with byte q; D lane (n, h) VGPR r = row 8*(r/4) + 4h + r%4
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
            int n = l & 31, h = l >> 5, row = 8 * (r / 4) + 4 * h + (r % 4), ref = 0;
            for (int g = 0; g < 2; g++) for (int q = 0; q < 16; q++) ref += (int)hA[row + 32 * g][q] * (int)hB[n + 32 * g][q];
            bad += ref != hD[l][r];
        }
        printf("layout hypothesis (own-K-group pairing, rows 8*(r/4)+4h+r%%4): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
        if (bad) {   // find, for lane 0 / 32, which row each VGPR holds
            for (int l = 0; l < 64; l += 32) for (int r = 0; r < 16; r++) {
                int n = l & 31;
                for (int row = 0; row < 32; row++) { int ref = 0; for (int g = 0; g < 2; g++) for (int q = 0; q < 16; q++) ref += (int)hA[row + 32 * g][q] * (int)hB[n + 32 * g][q]; if (ref == hD[l][r]) printf("  lane %d vgpr %d = row %d\n", l, r, row); }
            }
        }
    }
    uint64_t *out; CHECK(hipMalloc((void **)&out, 8ull * 256 * 4096 * 12));
    // ---- 2. the MFMA layer against the vector-ALU layer ----
    for (int dump = 1; dump <= 2; dump++) for (int it = 1; it <= 5; it += 4) {
        static uint64_t h0[8 * 256 * 12], h1[8 * 256 * 12];
        hipLaunchKernelGGL(k_mds<0>, dim3(8), dim3(256), 0, 0, out, it, 777ull, dump); CHECK(hipMemcpy(h0, out, sizeof h0, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_mds<1>, dim3(8), dim3(256), 0, 0, out, it, 777ull, dump); CHECK(hipMemcpy(h1, out, sizeof h1, hipMemcpyDeviceToHost));
        int bad = 0; for (size_t i = 0; i < 8 * 256 * 12; i++) bad += h0[i] != h1[i];
        printf("mds_layer_mfma == mds_layer (%d layers, inputs %s): %s (%d of %d differ)\n", it, dump == 1 ? "random" : "0xFF..", bad ? "NO" : "yes", bad, 8 * 256 * 12);
        int shown = 0;
        if (bad) for (size_t i = 0; i < 8 * 256 * 12 && shown < 6; i++) if (h0[i] != h1[i]) {
            shown++;
            size_t th = i / 12; int blk = (int)(th / 256), tid = (int)(th % 256);
            printf("   block %d thread %d el %d: valu %016llx mfma %016llx   inputs:", blk, tid, (int)(i % 12), (unsigned long long)h0[i], (unsigned long long)h1[i]);
            for (int j = 0; j < 12; j++) { unsigned long long v = dump == 2 ? ~0ull - (unsigned long long)j * tid : (777ull + blk) * (j + 1) * 0x9E3779B97F4A7C15ull + tid * 0x123456789abcdefull; printf(" %016llx", v); }
            printf("\n");
        }
    }
    {
        static uint64_t h0[8 * 256 * 12], h1[8 * 256 * 12];
        hipLaunchKernelGGL(k_perm<0>, dim3(8), dim3(256), 0, 0, out, 3, 99ull, 1); CHECK(hipMemcpy(h0, out, sizeof h0, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_perm<1>, dim3(8), dim3(256), 0, 0, out, 3, 99ull, 1); CHECK(hipMemcpy(h1, out, sizeof h1, hipMemcpyDeviceToHost));
        int bad = 0; for (size_t i = 0; i < 8 * 256 * 12; i++) bad += h0[i] != h1[i];
        printf("poseidon_perm (mfma) == poseidon_perm (valu), 3 chained permutations: %s (%d differ)\n", bad ? "NO" : "yes", bad);
    }
    // ---- 3. rates ----
    const int blocks = 256 * 8;
    for (int wpb = 64; wpb <= 256; wpb *= 2) {
        int it = 2000;
        float ms = timeit([&] { hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(wpb), 0, 0, out, it, 12345ull); });
        double n = (double)blocks * (wpb / 64) * it * 4;   // wave-MFMAs
        printf("mfma only, %d waves/block: %8.3f ms  %.1f cyc per wave-MFMA per SIMD\n", wpb / 64, ms, 2.4e9 * 1024 / (n / (ms * 1e-3)));
    }
#define RATE(NV) { int it = 1000; \
        float m1 = timeit([&] { hipLaunchKernelGGL(k_rate<NV>, dim3(blocks), dim3(256), 0, 0, out, it, 12345ull); }); \
        float m0 = timeit([&] { hipLaunchKernelGGL(k_valu<NV>, dim3(blocks), dim3(256), 0, 0, out, it, 12345ull); }); \
        double n = (double)blocks * 4 * it * 4; \
        printf("per MFMA + %2d mads: with mfma %.1f cyc, mads alone %.1f cyc (per wave per SIMD)\n", NV * 4, 2.4e9 * 1024 / (n / (m1 * 1e-3)), 2.4e9 * 1024 / (n / (m0 * 1e-3))); }
    RATE(1) RATE(2) RATE(4) RATE(8)
    for (int v = 0; v < 2; v++) {
        int it = 200;
        float ms = v == 0 ? timeit([&] { hipLaunchKernelGGL(k_mds<0>, dim3(blocks), dim3(256), 0, 0, out, it, 777ull, 0); })
                          : timeit([&] { hipLaunchKernelGGL(k_mds<1>, dim3(blocks), dim3(256), 0, 0, out, it, 777ull, 0); });
        double n = (double)blocks * 256 * it;
        printf("%-16s %8.3f ms  %8.3f G layers/s (%.0f cyc/wave-layer/SIMD)\n", v ? "mds_layer_mfma" : "mds_layer", ms, n / ms / 1e6, 2.4e9 * 1024 * 64 / (n / (ms * 1e-3)));
    }
    for (int v = 0; v < 2; v++) {
        int it = 20;
        float ms = v == 0 ? timeit([&] { hipLaunchKernelGGL(k_perm<0>, dim3(blocks), dim3(256), 0, 0, out, it, 777ull, 0); })
                          : timeit([&] { hipLaunchKernelGGL(k_perm<1>, dim3(blocks), dim3(256), 0, 0, out, it, 777ull, 0); });
        double n = (double)blocks * 256 * it;
        printf("%-22s %8.3f ms  %8.3f G perm/s\n", v ? "poseidon_perm (mfma)" : "poseidon_perm (valu)", ms, n / ms / 1e6);
    }
    return 0;
}
