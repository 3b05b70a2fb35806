// Host -> device rate of a row-major witness copied (a) whole, (b) in column chunks with hipMemcpy2DAsync (the pieces a
// column-pipelined upload + LDE would copy: 120-128 B of every 800-byte row), (c) in row blocks.  Pinned host memory.
// Build: hipcc -O2 --offload-arch=gfx950 tools/h2d_2d.hip -o tools/h2d_2d
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 1ull << 22, C = 100, bytes = N * C * 8;
    uint64_t *h, *d;
    CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
    CK(hipMalloc((void **)&d, bytes));
    for (size_t i = 0; i < N * C; i += 512) h[i] = i;
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
        double t1 = now();
        printf("whole matrix (%.2f GB):                 %7.1f ms  %6.1f GB/s\n", bytes / 1e9, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        for (size_t w : { (size_t)15, (size_t)16, (size_t)32, (size_t)50 }) {
            t0 = now();
            for (size_t c0 = 0; c0 < C; c0 += w) {
                size_t ww = c0 + w <= C ? w : C - c0;
                CK(hipMemcpy2DAsync(d + c0, C * 8, h + c0, C * 8, ww * 8, N, hipMemcpyHostToDevice, st));
            }
            CK(hipStreamSynchronize(st));
            t1 = now();
            printf("column chunks of %2zu (2D copies, %3zu B rows): %7.1f ms  %6.1f GB/s\n", w, w * 8, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        }
        t0 = now();
        const size_t blocks = 64, rows = N / blocks;
        for (size_t b = 0; b < blocks; b++) CK(hipMemcpyAsync(d + b * rows * C, h + b * rows * C, rows * C * 8, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t1 = now();
        printf("64 row blocks:                           %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    return 0;
}
