// The forward NTT passes' memory pattern without the transform: every workgroup reads 256 pieces of 128 bytes `lo` rows apart (a tile of
// 256 rows x 16 columns of an R x C row-major matrix of u64), and writes them back.  lo = 2^8: the near-strided pass; 2^16: the far one.
// Build: hipcc -O3 --offload-arch=gfx950 tools/stride_copy.hip -o tools/stride_copy      Run: tools/stride_copy [log2 rows (24)] [columns, a multiple of 16 (800)] [strides: log2 rows apart, comma separated]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned long u64; typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: read and write back; 1: read only (one word per lane written to `sink`); 2: write only; 3: read pieces `loBits` rows apart and
// write them 2^8 rows apart (an out-of-place pass that gathers far and stores near); 4: the reverse
template <int XCD, int MODE = 0>
__global__ void __launch_bounds__(256) k_tile(u64 *m, u64 C, u32 loBits, u32 nChunks, u32 nRowBits, u64 *sink = nullptr, u32 skewRows = 0) {
    u32 b = blockIdx.x;
    if (XCD) { const u32 per = gridDim.x >> 3; b = (b & 7) * per + (b >> 3); }     // consecutive logical tiles on one XCD, as the library does
    const u32 cc = b % nChunks; b /= nChunks;
    const u32 gt = b & ((1u << loBits) - 1), hi = b >> loBits;                     // row = (hi << (loBits + 8)) + (t << loBits) + gt
    const u32 x = threadIdx.x & 15, y = threadIdx.x >> 4;
    u64 *base = m + ((u64)hi << (loBits + 8)) * C + (u64)gt * C + cc * 16 + x;
    const u64 tStride = C * ((1ull << loBits) + skewRows);      // skewRows: pieces (2^lo + skew) rows apart -- the same span, other low address bits
    // the same tile of the NEAR pattern (pieces 2^8 rows apart), for the mixed modes
    u32 b2 = blockIdx.x; if (XCD) { const u32 per = gridDim.x >> 3; b2 = (b2 & 7) * per + (b2 >> 3); }
    b2 /= nChunks;
    u64 *nbase = m + ((u64)(b2 >> 8) << 16) * C + (u64)(b2 & 255) * C + cc * 16 + x;
    const u64 nStride = C << 8;
    u64 *rb = MODE == 4 ? nbase : base, *wb = MODE == 3 ? nbase : base;
    const u64 rs = MODE == 4 ? nStride : tStride, ws = MODE == 3 ? nStride : tStride;
    u64 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = MODE == 2 ? (u64)i : rb[(u64)(y + 16 * i) * rs];
    if (MODE == 1) { u64 a = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) a ^= v[i];
        if (a == 0x123456789ull) sink[threadIdx.x] = a; return; }
#pragma unroll
    for (int i = 0; i < 16; i++) wb[(u64)(y + 16 * i) * ws] = v[i] + 1;
}

int main(int argc, char **argv) {
    const u32 nRowBits = argc > 1 ? atoi(argv[1]) : 24;
    const u64 R = 1ull << nRowBits, C = argc > 2 ? strtoull(argv[2], 0, 10) : 800;   // 800: the extended config-3 matrix during the passes, N x (100 * 8)
    u64 *m;
    const u32 skewRows = argc > 4 ? (u32)atoi(argv[4]) : 0;
    CHECK(hipMalloc(&m, (R + 256ull * skewRows + 256) * C * 8));
    CHECK(hipMemset(m, 0, (R + 256ull * skewRows + 256) * C * 8));
    const u32 nChunks = (u32)(C / 16);
    const u32 blocks = (u32)((R >> 8) * nChunks);
    std::vector<u32> los = { 0u, 8u, nRowBits - 8 };
    if (argc > 3) { los.clear(); for (char *t = strtok(argv[3], ","); t; t = strtok(nullptr, ",")) los.push_back((u32)atoi(t)); }
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++)
        for (u32 loBits : los)
            for (int xcd = 1; xcd >= 0; xcd--) {
                CHECK(hipEventRecord(a));
                if (xcd) hipLaunchKernelGGL(k_tile<1>, dim3(blocks), dim3(256), 0, 0, m, C, loBits, nChunks, nRowBits, (u64 *)nullptr, skewRows);
                else hipLaunchKernelGGL(k_tile<0>, dim3(blocks), dim3(256), 0, 0, m, C, loBits, nChunks, nRowBits);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                if (xcd && loBits >= 8) {
                    const char *names[] = { "", "read only", "write only", "read far, write 2^8 apart", "read 2^8 apart, write far" };
                    for (int mode = 1; mode <= 4; mode++) {
                        CHECK(hipEventRecord(a));
                        if (mode == 1) hipLaunchKernelGGL((k_tile<1, 1>), dim3(blocks), dim3(256), 0, 0, m, C, loBits, nChunks, nRowBits, m);
                        if (mode == 2) hipLaunchKernelGGL((k_tile<1, 2>), dim3(blocks), dim3(256), 0, 0, m, C, loBits, nChunks, nRowBits, m);
                        if (mode == 3) hipLaunchKernelGGL((k_tile<1, 3>), dim3(blocks), dim3(256), 0, 0, m, C, loBits, nChunks, nRowBits, m);
                        if (mode == 4) hipLaunchKernelGGL((k_tile<1, 4>), dim3(blocks), dim3(256), 0, 0, m, C, loBits, nChunks, nRowBits, m);
                        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                        float ms2; CHECK(hipEventElapsedTime(&ms2, a, b));
                        printf("    pieces 2^%u rows apart, %s: %.2f ms, %.2f TB/s\n", loBits, names[mode], ms2, (mode <= 2 ? 1.0 : 2.0) * R * C * 8 / ms2 / 1e9);
                    }
                }
                printf("rows 2^%u x %lu columns (%.1f GB), pieces 2^%u%s rows apart, %s order: %.2f ms, %.2f TB/s read + written\n", nRowBits, (unsigned long)C, R * C * 8 / 1e9, loBits, skewRows && xcd ? " + skew" : "",
                       xcd ? "XCD-local" : "plain", ms, 2.0 * R * C * 8 / ms / 1e9);
            }
    return 0;
}
