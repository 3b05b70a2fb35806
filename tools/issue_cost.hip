// Issue cost of single vector instructions with the SIMD SATURATED (4 waves per SIMD, 8 independent chains per wave): cycles
// per instruction per SIMD = what an instruction costs in a kernel that is bound by vector issue (the Poseidon and NTT kernels).
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_cost.hip -o tools/issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint64_t u64; typedef uint32_t u32;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void __launch_bounds__(256) k(u64 *out, int iters) {
    const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
    u32 a[8], b[8]; u64 q[8]; u64 cc = 0x5555555555555555ull + (u64)iters;
    for (int i = 0; i < 8; i++) { a[i] = id * 7 + i; b[i] = id ^ (i * 77); q[i] = (u64)id * 1234567 + i; }
    asm volatile("" : "+s"(cc));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define XOR_(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(b[i]));
#define CND(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(cc));
#define ADDCO(i) { u64 s; asm volatile("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(a[i]), "=s"(s) : "v"(b[i])); }
#define ADDC(i) { u64 s; asm volatile("v_addc_co_u32_e64 %0, %1, %0, %2, %3" : "+v"(a[i]), "=s"(s) : "v"(b[i]), "s"(cc)); }
#define ADDCVCC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define MAD(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
#define MADS(i) { u64 s; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(q[i]), "=s"(s) : "v"(a[i]), "v"(b[i])); }
#define LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(q[i]) : "v"(q[(i + 1) & 7]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 12" : "+v"(a[i]) : "v"(b[i]));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define LSHR64(i) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(q[i]));
#define ASHR64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[i]));
#define AND_(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define MAD0(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if (OP == 0) { REP8(MOV) } if (OP == 1) { REP8(ADD) } if (OP == 2) { REP8(XOR_) } if (OP == 3) { REP8(LSHLADD) }
            if (OP == 4) { REP8(CND) } if (OP == 5) { REP8(ADDCO) } if (OP == 6) { REP8(ADDC) } if (OP == 7) { REP8(ADDCVCC) }
            if (OP == 8) { REP8(MAD) } if (OP == 9) { REP8(MADS) } if (OP == 10) { REP8(LSHLADD64) } if (OP == 11) { REP8(MOV64) }
            if (OP == 12) { REP8(PERM) } if (OP == 13) { REP8(ALIGN) } if (OP == 14) { REP8(MULLO) } if (OP == 15) { REP8(ADD3) }
            if (OP == 16) { REP8(LSHR64) } if (OP == 17) { REP8(ASHR64) } if (OP == 18) { REP8(AND_) } if (OP == 19) { REP8(MAD0) }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + q[i];
    out[id] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms;
}
int main() {
    u64 *out; (void)hipMalloc((void **)&out, 8ull * 256 * 4 * 256);
    const int iters = 4000;
    const char *nm[20] = { "v_mov_b32", "v_add_u32", "v_xor_b32", "v_lshl_add_u32", "v_cndmask_b32_e64 (SGPR mask)", "v_add_co_u32_e64 (SGPR out)", "v_addc_co_u32_e64 (SGPR in/out)",
                           "v_addc_co_u32 (vcc)", "v_mad_u64_u32 (vcc out)", "v_mad_u64_u32 (SGPR out)", "v_lshl_add_u64", "v_mov_b64", "v_perm_b32", "v_alignbit_b32", "v_mul_lo_u32", "v_add3_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_and_b32", "v_mad_u64_u32 (addend 0)" };
    for (int wps = 4; wps >= 1; wps = wps == 4 ? 2 : wps - 1) {
        const int blocks = 256 * wps;
        for (int op = 0; op < 20; op++) {
            float ms = 0;
#define RUN(N) if (op == N) ms = timeit([&] { hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19)
            const double insts = (double)iters * 32 * wps;            // per SIMD (one wave of each block per SIMD)
            printf("waves/SIMD %d  %-34s %7.3f ms  %5.2f cycles per instruction per SIMD (2.4 GHz nominal)\n", wps, nm[op], ms, ms * 1e-3 * 2.4e9 / insts);
        }
    }
    return 0;
}
