// What a 12-stage NTT pass would have to move: tiles of T rows x S words where S*8 bytes is LESS than a 128-byte line (S = 4: 32-byte
// pieces, S = 8: 64-byte pieces; S = 16 for reference), rows 2^lo apart in an R x C row-major matrix of u64 -- alone, and with the workgroups
// that own the other pieces of the same lines dealt to the same XCD at the same time (consecutive logical tiles, `xcd` order), reads and
// writes together and apart.  No arithmetic: this prices the memory side of a two-sweep forward LDE (review of round 3, item 1a).
// Build: hipcc -O3 --offload-arch=gfx950 tools/piece_copy.hip -o tools/piece_copy     Run: tools/piece_copy [log2 rows (24)] [columns (800)]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef unsigned long u64; typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ORDER 0: blocks as dispatched (neighbouring column chunks land on different XCDs); 1: consecutive logical tiles on one XCD, column chunk
// fastest (the owners of one line's pieces run on one L2 at about the same time); 2: same, but row group fastest (the owners of a line's
// pieces are far apart in time: what a piece costs when nobody shares it)
// MODE 0: read + write back in place; 1: read only; 2: write only
// SEQ: one workgroup takes the NS = 16/S column chunks of its lines one after the other
template <int S, int EPT, int ORDER, int MODE, bool SEQ>
__global__ void __launch_bounds__(1024) k_piece(u64 *m, u64 C, u32 loBits, u32 tBits, u32 nChunks, u64 *sink) {
    u32 b = blockIdx.x;
    if (ORDER) { const u32 per = gridDim.x >> 3; if (b < (per << 3)) b = (b & 7) * per + (b >> 3); }
    constexpr u32 NS = SEQ ? 16 / S : 1;
    const u32 nCh = nChunks / NS;
    u32 cc, rest;
    const u32 nTilesPerChunk = gridDim.x / nCh;
    if (ORDER == 2) { rest = b % nTilesPerChunk; cc = b / nTilesPerChunk; } else { cc = b % nCh; rest = b / nCh; }
    const u32 gt = rest & ((1u << loBits) - 1), hi = rest >> loBits;          // row = (hi << (loBits + tBits)) + (t << loBits) + gt
    const u32 x = threadIdx.x % S, y = threadIdx.x / S, by = blockDim.x / S;
    u64 *base = m + (((u64)hi << (loBits + tBits)) + gt) * C + (u64)cc * NS * S + x;
    const u64 tStride = C << loBits;
    u64 v[EPT], a = 0;
#pragma unroll 1
    for (u32 s = 0; s < NS; s++) {
#pragma unroll
        for (int i = 0; i < EPT; i++) v[i] = MODE == 2 ? (u64)i + s : base[(u64)(y + by * i) * tStride + s * S];
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < EPT; i++) a ^= v[i];
        } else {
#pragma unroll
            for (int i = 0; i < EPT; i++) base[(u64)(y + by * i) * tStride + s * S] = v[i] + 1;
        }
    }
    if (MODE == 1 && a == 0x123456789ull) sink[threadIdx.x] = a;
}

template <int S, int EPT, int ORDER, int MODE, bool SEQ>
static int run(const char *what, u64 *m, u64 R, u64 C, u32 nRowBits, u32 loBits, u32 tBits, u32 threads, hipEvent_t a, hipEvent_t b) {
    const u32 nChunks = (u32)(C / S);
    const u64 blocks = (R >> tBits) * (nChunks / (SEQ ? 16 / S : 1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k_piece<S, EPT, ORDER, MODE, SEQ>), dim3((unsigned)blocks), dim3(threads), 0, 0, m, C, loBits, tBits, nChunks, m);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (MODE == 0 ? 2.0 : 1.0) * R * C * 8;
    printf("  %3d-byte pieces, %5u rows per tile 2^%-2u rows apart, %4u threads, %-28s %-10s %7.2f ms  %5.2f TB/s\n", S * 8, 1u << tBits, loBits, threads, what,
           MODE == 0 ? "read+write" : MODE == 1 ? "read" : "write", best, bytes / best / 1e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const u32 nRowBits = argc > 1 ? atoi(argv[1]) : 24;
    const u64 R = 1ull << nRowBits, C = argc > 2 ? strtoull(argv[2], 0, 10) : 800;
    u64 *m;
    CHECK(hipMalloc(&m, R * C * 8));
    CHECK(hipMemset(m, 0, R * C * 8));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    printf("piece_copy: %lu x %lu u64 (%.1f GB)\n", (unsigned long)R, (unsigned long)C, R * C * 8 / 1e9);
#define ALLMODES(S_, E_, O_, SEQ_, what, lo, tb, thr)                                  \
    if (run<S_, E_, O_, 0, SEQ_>(what, m, R, C, nRowBits, lo, tb, thr, a, b)) return 1;  \
    if (run<S_, E_, O_, 1, SEQ_>(what, m, R, C, nRowBits, lo, tb, thr, a, b)) return 1;  \
    if (run<S_, E_, O_, 2, SEQ_>(what, m, R, C, nRowBits, lo, tb, thr, a, b)) return 1;
    const u32 los[3] = { 0, 12, nRowBits - 12 };
    // reference: today's tiles (256 rows x 128 bytes, 256 threads x 16) at the strides of today's passes
    printf("reference, 128-byte pieces (today's passes):\n");
    ALLMODES(16, 16, 1, false, "xcd order", 8, 8, 256)
    ALLMODES(16, 16, 1, false, "xcd order", 16, 8, 256)
    for (int li = 0; li < 3; li++) {
        const u32 lo = los[li];
        if (li == 2 && lo == 12) break;
        printf("4096-row tiles, rows 2^%u apart:\n", lo);
        ALLMODES(8, 32, 0, false, "dispatch order", lo, 12, 1024)
        ALLMODES(8, 32, 1, false, "xcd order, line-mates together", lo, 12, 1024)
        ALLMODES(8, 32, 2, false, "xcd order, line-mates apart", lo, 12, 1024)
        ALLMODES(8, 32, 1, true, "one workgroup, 2 pieces in turn", lo, 12, 1024)
        ALLMODES(4, 16, 0, false, "dispatch order", lo, 12, 1024)
        ALLMODES(4, 16, 1, false, "xcd order, line-mates together", lo, 12, 1024)
        ALLMODES(4, 16, 2, false, "xcd order, line-mates apart", lo, 12, 1024)
        ALLMODES(4, 16, 1, true, "one workgroup, 4 pieces in turn", lo, 12, 1024)
    }
    // 10-stage tiles (1024 rows x 128 bytes: the largest tile with whole lines that LDS holds), rows 2^10 and 2^14 apart
    printf("1024-row tiles of 128-byte pieces:\n");
    ALLMODES(16, 16, 1, false, "xcd order", 0, 10, 1024)
    ALLMODES(16, 16, 1, false, "xcd order", 10, 10, 1024)
    ALLMODES(16, 16, 1, false, "xcd order", 14, 10, 1024)
    return 0;
}
