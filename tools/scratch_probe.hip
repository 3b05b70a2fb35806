// Why the first expression evaluator (commit 8435cfa) returned wrong temporaries: a probe of how gfx950 handles the access
// pattern hipcc chose for its per-lane array `u64 tmp[3 * MAXT]` -- 24-byte slots written with
// scratch_store_dwordx4 + scratch_store_dwordx2 and read back through a GENERIC pointer (the operand could also be a global
// address) with flat_load_dwordx2 + flat_load_dwordx4 at +8, i.e. 16-byte private accesses that are only 8-byte aligned.
// Each variant writes a slot, reads it back the way named, and counts lanes whose value differs from what was written.
// Build: hipcc -O3 --offload-arch=gfx950 tools/scratch_probe.hip -o tools/scratch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint64_t u64;
typedef uint32_t u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; return x; }

// SLOT = words per slot (3: the evaluator's layout; 4: padded so that every 16-byte access is 16-byte aligned)
// GENERIC: read through a pointer that is private or global depending on run-time data (flat_load); else directly (scratch_load)
template <int SLOT, bool GENERIC>
__global__ void __launch_bounds__(256) k_probe(const u64 *g, const u32 *prog, int nOps, u64 *bad, u64 seed) {
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    alignas(16) u64 tmp[SLOT * 8];
    u64 nbad = 0;
    for (int i = 0; i < SLOT * 8; i++) tmp[i] = 0;
    for (int k = 0; k < nOps; k++) {
        const u32 w = prog[k];                       // wave-uniform "program": destination slot, source slot, source kind
        const u32 d = w & 7, s = (w >> 3) & 7, kind = (w >> 6) & 1;
        const u64 a = mix(seed + id * 131 + k), b = mix(a), c = mix(b);
        tmp[SLOT * d] = a; tmp[SLOT * d + 1] = b; tmp[SLOT * d + 2] = c;         // the evaluator's destination store
        const u64 *p;
        if (GENERIC) p = kind ? (const u64 *)(tmp + SLOT * s) : g + 3 * s;          // TMP or SCALAR operand: one generic pointer
        else p = tmp + SLOT * s;
        const u64 x = p[0], y = p[1], z = p[2];
        if ((!GENERIC || kind) && s == d) nbad += (x != a) + (y != b) + (z != c);     // read back what was just written
    }
    bad[id] = nbad;
}

int main() {
    const int blocks = 1024, nOps = 4096;
    const size_t n = (size_t)blocks * 256;
    u64 *g, *bad; u32 *prog;
    CHECK(hipMalloc((void **)&g, 8 * 64)); CHECK(hipMemset(g, 0x5a, 8 * 64));
    CHECK(hipMalloc((void **)&bad, 8 * n)); CHECK(hipMalloc((void **)&prog, 4 * nOps));
    u32 *hp = (u32 *)malloc(4 * nOps); u64 *hb = (u64 *)malloc(8 * n);
    u64 r = 12345;
    for (int k = 0; k < nOps; k++) { r = r * 6364136223846793005ull + 1442695040888963407ull; u32 d = (r >> 33) & 7; u32 same = (r >> 40) & 1; u32 s = same ? d : (r >> 45) & 7; hp[k] = d | (s << 3) | (((r >> 50) & 3) ? 64u : 0u); }
    CHECK(hipMemcpy(prog, hp, 4 * nOps, hipMemcpyHostToDevice));
    const char *names[4] = { "24-byte slots, read through a generic pointer (flat)  [the evaluator's pattern]", "24-byte slots, read directly (scratch)",
                             "32-byte slots, read through a generic pointer (flat)", "32-byte slots, read directly (scratch)" };
    for (int v = 0; v < 4; v++) {
        for (int rep = 0; rep < 3; rep++) {
            if (v == 0) hipLaunchKernelGGL((k_probe<3, true>), dim3(blocks), dim3(256), 0, 0, g, prog, nOps, bad, 77ull + rep);
            if (v == 1) hipLaunchKernelGGL((k_probe<3, false>), dim3(blocks), dim3(256), 0, 0, g, prog, nOps, bad, 77ull + rep);
            if (v == 2) hipLaunchKernelGGL((k_probe<4, true>), dim3(blocks), dim3(256), 0, 0, g, prog, nOps, bad, 77ull + rep);
            if (v == 3) hipLaunchKernelGGL((k_probe<4, false>), dim3(blocks), dim3(256), 0, 0, g, prog, nOps, bad, 77ull + rep);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(hb, bad, 8 * n, hipMemcpyDeviceToHost));
            u64 tot = 0, lanes = 0; for (size_t i = 0; i < n; i++) { tot += hb[i]; lanes += hb[i] != 0; }
            printf("%-86s run %d: %llu wrong words in %llu of %zu lanes\n", names[v], rep, (unsigned long long)tot, (unsigned long long)lanes, n);
        }
    }
    return 0;
}
