// Issue cost of the BN254 field primitives (csrc/bn_field.cuh) on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc tools/bn_microbench.hip -o tools/bn_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "bn_field.cuh"
using namespace bn;

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(u32 *out, int iters, u32 seed) {
    u32 x[CHAINS][8], y[8], acc[17];
    for (int c = 0; c < CHAINS; c++) for (int l = 0; l < 8; l++) x[c][l] = seed * (c + 3) + threadIdx.x * (l + 1);
    for (int l = 0; l < 8; l++) y[l] = seed + 77 * l + blockIdx.x;
    x[0][7] &= 0x0fffffff; y[7] &= 0x0fffffff;
    for (int l = 0; l < 17; l++) acc[l] = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) fr_mul(x[c], x[c], y);
            if (OP == 1) { mac17(acc, x[c], y); x[c][0] += acc[3]; }
            if (OP == 2) fr_add(x[c], y);
            if (OP == 3) fr_mul_os(x[c], x[c], y);
            if (OP == 4) { mac17_os(acc, x[c], y); x[c][0] += acc[3]; }
        }
        if ((OP == 1 || OP == 4) && (i & 15) == 15) { u32 o[8]; if (OP == 1) redc17(o, acc); else redc17_os(o, acc); for (int l = 0; l < 17; l++) acc[l] = l < 8 ? o[l] : 0; }
    }
    u32 s = 0;
    for (int c = 0; c < CHAINS; c++) for (int l = 0; l < 8; l++) s += x[c][l];
    for (int l = 0; l < 17; l++) s += acc[l];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms;
}
int main() {
    u32 *out; (void)hipMalloc((void **)&out, 4ull * 256 * 8192);
    const int blocks = 256 * 8, iters = 400;
#define RUN(OP, CH, name, steps) { float ms = timeit([&] { hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u); }); \
        double ops = (double)blocks * 256 * iters * CH; double cyc = 2.4e9 * 1024 * 64 / (ops / (ms * 1e-3)); \
        printf("%-28s %8.3f ms  %7.1f cyc per wave-op  (%.1f per mad step)\n", name, ms, cyc, cyc / steps); }
    RUN(0, 1, "fr_mul, 1 chain", 128) RUN(0, 2, "fr_mul, 2 chains", 128) RUN(0, 4, "fr_mul, 4 chains", 128)
    RUN(3, 1, "fr_mul_os, 1 chain", 128) RUN(3, 4, "fr_mul_os, 4 chains", 128)
    { u32 h[2][64]; hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, out, 37, 999u); (void)hipMemcpy(h[0], out, 256, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL((k<3, 1>), dim3(1), dim3(64), 0, 0, out, 37, 999u); (void)hipMemcpy(h[1], out, 256, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < 64; i++) bad += h[0][i] != h[1][i]; printf("fr_mul == fr_mul_os: %s\n", bad ? "NO" : "yes");
      hipLaunchKernelGGL((k<1, 2>), dim3(1), dim3(64), 0, 0, out, 48, 999u); (void)hipMemcpy(h[0], out, 256, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL((k<4, 2>), dim3(1), dim3(64), 0, 0, out, 48, 999u); (void)hipMemcpy(h[1], out, 256, hipMemcpyDeviceToHost);
      bad = 0; for (int i = 0; i < 64; i++) bad += h[0][i] != h[1][i]; printf("mac17/redc17 == operand-scanning forms: %s\n", bad ? "NO" : "yes"); }
    RUN(1, 1, "mac17, 1 chain", 64) RUN(4, 1, "mac17_os, 1 chain", 64)
    RUN(2, 4, "fr_add, 4 chains", 8)
    return 0;
}
