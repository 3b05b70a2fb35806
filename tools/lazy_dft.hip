// 16-point register DFTs of the NTT tiles: the Z/(2^96+1) form (gl_fermat.cuh, the current kernels) against plain lazy
// Goldilocks values with carry-flag adds / subs and shift multiplications (gl_lazy.cuh): values and issue rates.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc -I tools tools/lazy_dft.hip -o tools/lazy_dft
// Outcome (profiles/r02_lazy_dft_microbench.txt): 30 % fewer issue cycles in isolation, but inside the NTT tiles (three to four
// waves per SIMD, LDS traffic around it) the passes ran 5-20 % SLOWER than with the Z/(2^96+1) steps -- the carries travel
// through SGPR pairs and each hop stalls the wave (with or without the padding s_nop: same time) -- so ntt.hip keeps the
// Z/(2^96+1) form and this header stays a tools/ experiment.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "gl_field.cuh"
#include "gl_fermat.cuh"
#include "gl_lazy.cuh"
using namespace gl;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; return x; }

// MODE 0: fermat DIT fwd; 1: lazy DIT fwd; 2: fermat DIF inv; 3: lazy DIF inv; 4: fermat DIT inv; 5: lazy DIT inv; 6: fermat DIF fwd; 7: lazy DIF fwd
template <int MODE, bool TW>
__global__ void __launch_bounds__(256) k_dft(u64 *out, int iters, u64 seed, int dump, int edge) {
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x[16];
    for (int i = 0; i < 16; i++) x[i] = mix(seed + id * 16 + i);
    if (edge) for (int i = 0; i < 16; i++) x[i] = (i & 1) ? ~0ull - (id & 3) * i : ((id + i) % 3 ? 0xFFFFFFFF00000000ull + i : (u64)i);
    const u64 tw = canon(mix(seed * 77 + id));
    u64 bad = 0, nbad = 0;
    for (int it = 0; it < iters; it++) {
        if ((MODE & 1) == 0) {
            fermat::f128 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = fermat::from_gl(x[i]);
            if (MODE == 0) fermat::dft_dit<4, false>(v);
            if (MODE == 2) fermat::dft_dif<4, true>(v);
            if (MODE == 4) fermat::dft_dit<4, true>(v);
            if (MODE == 6) fermat::dft_dif<4, false>(v);
#pragma unroll
            for (int i = 0; i < 16; i++) { u64 o = fermat::to_gl_lazy(v[i]); x[i] = TW ? mul_lazy(o, tw) : o; }
        } else {
            if (MODE == 1) lazy::dft16_dit<false>(x, bad);
            if (MODE == 3) lazy::dft16_dif<true>(x, bad);
            if (MODE == 5) lazy::dft16_dit<true>(x, bad);
            if (MODE == 7) lazy::dft16_dif<false>(x, bad);
            if (TW) {
#pragma unroll
                for (int i = 0; i < 16; i++) x[i] = mul_lazy_b(x[i], tw, bad);
            }
        }
        if (dump) break;
    }
    nbad = __popcll(bad);
    if (dump) { for (int i = 0; i < 16; i++) out[id * 17 + i] = canon(x[i]); out[id * 17 + 16] = bad ? 1 : 0; return; }
    u64 s = nbad; for (int i = 0; i < 16; i++) s += canon(x[i]) * (i + 1);
    out[id] = s;
}

template <typename F>
float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms;
}
#define LAUNCH(M, T, ...) hipLaunchKernelGGL((k_dft<M, T>), dim3(blocks), dim3(256), 0, 0, __VA_ARGS__)

int main() {
    const int blocks = 2048; const size_t n = (size_t)blocks * 256;
    u64 *out; CHECK(hipMalloc((void **)&out, 8 * n * 17));
    u64 *h0 = (u64 *)malloc(8 * n * 17), *h1 = (u64 *)malloc(8 * n * 17);
    const char *nm[4] = { "DIT forward", "DIF inverse", "DIT inverse", "DIF forward" };
    for (int edge = 0; edge < 2; edge++) for (int p = 0; p < 4; p++) {
        if (p == 0) { LAUNCH(0, true, out, 1, 5ull, 1, edge); } if (p == 1) { LAUNCH(2, true, out, 1, 5ull, 1, edge); } if (p == 2) { LAUNCH(4, true, out, 1, 5ull, 1, edge); } if (p == 3) { LAUNCH(6, true, out, 1, 5ull, 1, edge); }
        CHECK(hipMemcpy(h0, out, 8 * n * 17, hipMemcpyDeviceToHost));
        if (p == 0) { LAUNCH(1, true, out, 1, 5ull, 1, edge); } if (p == 1) { LAUNCH(3, true, out, 1, 5ull, 1, edge); } if (p == 2) { LAUNCH(5, true, out, 1, 5ull, 1, edge); } if (p == 3) { LAUNCH(7, true, out, 1, 5ull, 1, edge); }
        CHECK(hipMemcpy(h1, out, 8 * n * 17, hipMemcpyDeviceToHost));
        size_t diff = 0, flagged = 0, diff_unflagged = 0;
        for (size_t i = 0; i < n; i++) { bool f = h1[i * 17 + 16] != 0; flagged += f; for (int j = 0; j < 16; j++) { bool d = h0[i * 17 + j] != h1[i * 17 + j]; diff += d; if (!f) diff_unflagged += d; } }
        printf("%s, %s inputs: %zu of %zu values differ (%zu in lanes of unflagged waves); %zu lanes in flagged waves\n", nm[p], edge ? "edge" : "random", diff, n * 16, diff_unflagged, flagged);
    }
    const int it = 100;
    float t;
    double ops = (double)n * it * 16;
#define TIME(M, T, label) t = timeit([&] { LAUNCH(M, T, out, it, 9ull, 0, 0); }); printf("%-58s %8.3f ms  (%.1f cycles per wave-element per SIMD @2.4GHz)\n", label, t, 2.4e9 * 1024 * 64 / (ops / (t * 1e-3)));
    TIME(0, false, "Z/(2^96+1) DIT + reduction")
    TIME(1, false, "lazy DIT")
    TIME(0, true, "Z/(2^96+1) DIT + reduction + twiddle (hipcc product)")
    TIME(1, true, "lazy DIT + twiddle (carry-out product)")
    TIME(2, true, "Z/(2^96+1) inverse DIF + reduction + twiddle")
    TIME(3, true, "lazy inverse DIF + twiddle")
    return 0;
}
