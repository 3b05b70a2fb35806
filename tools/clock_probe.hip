// The shader clock under three loads (s_memtime against the 100 MHz s_memrealtime, per workgroup): the production permutation,
// the same with its matrix instructions replaced by two vector operations (-DPBLK_NO_MFMA: wrong results), the plain vector-ALU form.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc -I pil2-stark-js_amd/build [-DPBLK_NO_MFMA] tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#ifdef PBLK_NO_MFMA
typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef int v16i_ __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16i_ fake_mfma(v4i_ a, v4i_ b, v16i_ c) { c[0] += a[0] ^ b[1]; c[5] += a[2] ^ b[3]; c[9] ^= a[1] + b[0]; c[14] ^= a[3] + b[2]; return c; }
#define PBLK_MFMA(a, b, c) fake_mfma(a, b, c)
#endif
#include "poseidon_gl.cuh"
using namespace gl;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V>
__global__ void __launch_bounds__(256, 4) k_perm(u64 *out, int iters, u64 *clk) {
    MdsMfma m;
    poseidon_init(m);
    u64 st[12];
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < 12; j++) st[j] = ((id * 12 + j) * 0x9E3779B97F4A7C15ull) >> 1;
    const u64 c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) { if (V == 0) poseidon_perm<0>(st, m); else poseidon_perm(st); }
    const u64 c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    u64 s = 0;
    for (int j = 0; j < 12; j++) s ^= st[j];
    out[id] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int blocks = 256 * 4 * 8, iters = 40;
    u64 *out, *clk;
    CHECK(hipMalloc(&out, 8ull * blocks * 256)); CHECK(hipMalloc(&clk, 16ull * blocks));
    std::vector<u64> h(2 * blocks);
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 2; v++) {
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            CHECK(hipEventRecord(a));
            if (v == 0) hipLaunchKernelGGL(k_perm<0>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
            else hipLaunchKernelGGL(k_perm<1>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            CHECK(hipMemcpy(h.data(), clk, 16ull * blocks, hipMemcpyDeviceToHost));
            std::vector<double> mhz;
            for (int i = 0; i < blocks; i++) if (h[2 * i + 1]) mhz.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
            std::sort(mhz.begin(), mhz.end());
#ifdef PBLK_NO_MFMA
            const char *name = v == 0 ? "matrix-core form WITHOUT its matrix instructions (wrong results)" : "vector-ALU form";
#else
            const char *name = v == 0 ? "matrix-core form (production)" : "vector-ALU form";
#endif
            printf("%-70s %.2f ms, %.3f G perm/s, shader clock median %.0f MHz (5%% %.0f, 95%% %.0f)\n", name, ms, (double)blocks * 256 * iters / ms / 1e6,
                   mhz[mhz.size() / 2], mhz[mhz.size() / 20], mhz[mhz.size() * 19 / 20]);
        }
    return 0;
}
