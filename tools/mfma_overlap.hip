// Do matrix instructions overlap with vector-ALU work of OTHER waves on the same SIMD, or only with independent vector
// instructions placed between them in the SAME wave?  Loop body = 8 v_mfma_i32_32x32x32_i8 (two chains) + 48 v_mad_u64_u32
// (eight independent chains), clustered (8 MFMAs, then 48 mads) or interleaved (1 MFMA, 6 mads) x 8, at 1..4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_overlap.hip -o tools/mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef uint64_t u64; typedef uint32_t u32;

template <int MODE>   // 0 clustered, 1 interleaved, 2 mads only, 3 mfma only
__global__ void __launch_bounds__(256) k(u64 *out, int iters) {
    const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
    v4i a = { (int)id, (int)id * 3, 7, 9 }, b = { 1, (int)id, 5, 11 };
    v16i L, H;
    for (int i = 0; i < 16; i++) { L[i] = i; H[i] = 2 * i; }
    u64 x[8]; u32 m = id | 1;
    for (int i = 0; i < 8; i++) x[i] = id * 77 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 8; g++) {
            if (MODE != 2) {
                if (g & 1) H = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, H, 0, 0, 0);
                else L = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, L, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int j = 0; j < 6; j++) { const int c = (g * 6 + j) & 7; x[c] = (u64)(u32)x[c] * m + x[c]; }
            }
            if (MODE == 1) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 6, 0); }
        }
        if (MODE == 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 48; j++) { const int c = j & 7; x[c] = (u64)(u32)x[c] * m + x[c]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    for (int i = 0; i < 16; i++) s += (u32)L[i] + (u32)H[i];
    out[id] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}
int main() {
    u64 *out; hipMalloc((void **)&out, 8ull * 256 * 4 * 256 * 4);
    const int iters = 2000;
    const char *nm[4] = { "clustered (8 MFMA, then 48 mad)", "interleaved (1 MFMA, 6 mad) x 8", "48 mad only", "8 MFMA only" };
    for (int wps = 1; wps <= 4; wps++) {
        // 256 CUs x wps workgroups of 256 threads (4 waves = one per SIMD each)
        const int blocks = 256 * wps;
        for (int mode = 0; mode < 4; mode++) {
            float ms = 0;
            if (mode == 0) ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            if (mode == 1) ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            if (mode == 2) ms = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            if (mode == 3) ms = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            // cycles per loop body per SIMD (all waves of the SIMD together), at a nominal 2.4 GHz
            const double cyc = ms * 1e-3 * 2.4e9 / iters;
            printf("waves/SIMD %d  %-34s %8.3f ms  %7.1f cycles per body (x%d waves) = %6.1f per wave-body\n", wps, nm[mode], ms, cyc, wps, cyc / wps);
        }
    }
    return 0;
}
