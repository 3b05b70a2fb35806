// Root cause of the first expression evaluator's wrong temporaries (commit 8435cfa; reproduced 100 % by tools/_old/repro.py):
// hipcc ended its op loop with
//     scratch_store_dwordx4 off, v[8:11], s0      ; components 0,1 of the destination slot, address in s0
//     s_add_i32 s0, s0, 16                        ; ... and re-used s0 for the next address at once
//     scratch_store_dwordx2 off, v[14:15], s0     ; component 2
// and components 0,1 never arrived (component 2 did, and the 8 bytes after it were overwritten).  This probe issues exactly
// that sequence and three variants, then reads the slot back: does a scalar write to the address SGPR of a 16-byte scratch
// store that was issued just before it change where that store goes?
// Build: hipcc -O3 --offload-arch=gfx950 tools/sgpr_war_probe.hip -o tools/sgpr_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint64_t u64;
typedef uint32_t u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V>
__global__ void __launch_bounds__(256) k_probe(u64 *out, u32 slot) {
    const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
    volatile u64 pad[16];                               // the kernel's only private object: scratch bytes 0..127
    for (int i = 0; i < 16; i++) pad[i] = 0xEEEEEEEE00000000ull + i;
    const u64 a = 0x1111000000000000ull + id, b = 0x2222000000000000ull + id, c = 0x3333000000000000ull + id;
    v4u ab = { (u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32) };
    v2u cc = { (u32)c, (u32)(c >> 32) };
    u32 off = __builtin_amdgcn_readfirstlane(slot * 24u), off2 = off + 16;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (V == 0)        // the evaluator's sequence
        asm volatile("s_nop 4\n\tscratch_store_dwordx4 off, %1, %0\n\ts_add_i32 %0, %0, 16\n\tscratch_store_dwordx2 off, %2, %0\n\ts_waitcnt vmcnt(0)" : "+s"(off) : "v"(ab), "v"(cc) : "memory");
    if (V == 1)        // one wait state before the address register is rewritten
        asm volatile("s_nop 4\n\tscratch_store_dwordx4 off, %1, %0\n\ts_nop 0\n\ts_add_i32 %0, %0, 16\n\tscratch_store_dwordx2 off, %2, %0\n\ts_waitcnt vmcnt(0)" : "+s"(off) : "v"(ab), "v"(cc) : "memory");
    if (V == 2)        // the second address in a register of its own
        asm volatile("s_nop 4\n\tscratch_store_dwordx4 off, %2, %0\n\tscratch_store_dwordx2 off, %3, %1\n\ts_waitcnt vmcnt(0)" : : "s"(off), "s"(off2), "v"(ab), "v"(cc) : "memory");
    if (V == 3)        // 8-byte store first, then the address rewritten: is the hazard specific to the 16-byte form?
        asm volatile("s_nop 4\n\tscratch_store_dwordx2 off, %2, %0\n\ts_add_i32 %0, %0, 16\n\tscratch_store_dwordx2 off, %2, %0\n\ts_waitcnt vmcnt(0)" : "+s"(off) : "v"(ab), "v"(cc) : "memory");
    if (V == 4)        // immediate offset instead of a second address
        asm volatile("s_nop 4\n\tscratch_store_dwordx4 off, %1, %0\n\tscratch_store_dwordx2 off, %2, %0 offset:16\n\ts_waitcnt vmcnt(0)" : : "s"(off), "v"(ab), "v"(cc) : "memory");
    for (int i = 0; i < 16; i++) out[(u64)id * 16 + i] = pad[i];
}

int main() {
    const int blocks = 64; const size_t n = (size_t)blocks * 256;
    u64 *out; CHECK(hipMalloc((void **)&out, 8 * n * 16));
    u64 *h = (u64 *)malloc(8 * n * 16);
    const char *names[5] = { "x4 store, s_add on its address register, x2 store   [the evaluator's code]", "the same with s_nop 0 after the x4 store", "addresses in two registers",
                             "x2 store, s_add on its address register, x2 store", "x4 store + x2 store with an immediate offset" };
    for (int v = 0; v < 5; v++) {
        for (u32 slot = 0; slot < 3; slot++) {
            if (v == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 0, 0, out, slot);
            if (v == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(256), 0, 0, out, slot);
            if (v == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(256), 0, 0, out, slot);
            if (v == 3) hipLaunchKernelGGL(k_probe<3>, dim3(blocks), dim3(256), 0, 0, out, slot);
            if (v == 4) hipLaunchKernelGGL(k_probe<4>, dim3(blocks), dim3(256), 0, 0, out, slot);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, out, 8 * n * 16, hipMemcpyDeviceToHost));
            size_t okA = 0, okB = 0, okC = 0, aAt2 = 0;
            for (size_t i = 0; i < n; i++) {
                const u64 *p = h + i * 16 + 3 * slot;
                okA += p[0] == 0x1111000000000000ull + i; okB += p[1] == 0x2222000000000000ull + i; okC += p[2] == 0x3333000000000000ull + i;
                aAt2 += p[3] == 0x2222000000000000ull + i;          // component 1 landing 16 bytes late
            }
            if (v == 3) printf("%-80s slot %u: first store's words in place in %zu / %zu lanes, second store's in %zu\n", names[v], slot, okA, n, okC);
            else printf("%-80s slot %u: components in place: c0 %zu  c1 %zu  c2 %zu of %zu lanes; c1 found 16 bytes further in %zu\n", names[v], slot, okA, okB, okC, n, aAt2);
        }
    }
    // slot 0, lane 0 of the evaluator's sequence, as it lies in scratch
    hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(256), 0, 0, out, 0u); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, out, 8 * 16, hipMemcpyDeviceToHost));
    printf("scratch words 0..5 of lane 0 after the evaluator's sequence on slot 0:"); for (int i = 0; i < 6; i++) printf(" %016llx", (unsigned long long)h[i]); printf("\n");
    return 0;
}
