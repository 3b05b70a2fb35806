// The matrix-core loop of the BN254 blocks in isolation: per "column" four operand tiles (1 KB each, global memory) and eight
// v_mfma_i32_32x32x32_i8 into eight accumulators.  Cycles per matrix instruction as one wave sees them, at one and two waves per SIMD,
// with the tiles read from a table of 8 KB (L1), 2.9 MB (L2 / MALL) and with no loads at all.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_tile_stream.hip -o tools/mfma_tile_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef const v4i __attribute__((address_space(1))) *gtile;

template <int MODE>      // 0: no loads; 1: table of `span` tiles, ring of 6 ahead
__global__ void __launch_bounds__(64) k(const v4i *tiles, unsigned span, int cols, unsigned long long *out, int *sink) {
    extern __shared__ int lds[];
    gtile base = (gtile)tiles + threadIdx.x;
    v16i P[8];
    for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) P[i][e] = 1 << 25;
    v4i b0 = { (int)threadIdx.x, 2, 3, 4 }, b1 = { 5, 6, 7, (int)blockIdx.x };
    v4i q[6];
    unsigned pos = blockIdx.x * 37u;
    for (int r = 0; r < 6; r++) q[r] = base[(size_t)((pos + r) % span) * 64];
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int c = 0; c < cols; c++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v4i a;
            if (MODE == 0) a = b1;
            else {
                a = q[0];
#pragma unroll
                for (int r = 0; r < 5; r++) q[r] = q[r + 1];
                q[5] = base[(size_t)((pos + 6) % span) * 64];
                pos++;
            }
            P[2 * i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, P[2 * i], 0, 0, 0);
            P[2 * i + 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, P[2 * i + 1], 0, 0, 0);
        }
        b0[1] += c;
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    int s = 0;
    for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) s += P[i][e];
    if (s == 0x12345) *sink = s;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}
int main() {
    const unsigned big = 2900;
    v4i *tiles; (void)hipMalloc((void **)&tiles, (size_t)(big + 8) * 1024); (void)hipMemset(tiles, 1, (size_t)(big + 8) * 1024);
    unsigned long long *out; int *sink; (void)hipMalloc((void **)&out, 8); (void)hipMalloc((void **)&sink, 4);
    const int cols = 4000;
    for (int wps = 1; wps <= 2; wps++) {
        const size_t lds = wps == 1 ? 38000 : 18000;          // 4 / 8 workgroups of one wave per CU
        for (int mode = 0; mode < 3; mode++) {
            const unsigned span = mode == 1 ? 8 : big;
            const int blocks = 256 * 4 * wps;
            (void)hipMemset(out, 0, 8);
            if (mode == 0) { (void)hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), lds, 0, tiles, span, cols, out, sink); }
            else { (void)hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), lds, 0, tiles, span, cols, out, sink); }
            (void)hipDeviceSynchronize();
            unsigned long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            printf("waves/SIMD %d  %-28s %7.1f cycles per matrix instruction as one wave sees it  (%.1f at the SIMD)\n", wps,
                   mode == 0 ? "no loads" : mode == 1 ? "tiles from an 8 KB window" : "tiles from a 2.9 MB table", (double)h / blocks / cols / 8, (double)h / blocks / cols / 8 / wps);
        }
    }
    return 0;
}
