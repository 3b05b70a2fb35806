// Unit checks of the pieces of csrc/bn_mfma.cuh on the device: the hand-scheduled carry / finish against a plain C restatement,
// the reduction-free product against the canonical one, on random and extreme operands.
// Build: hipcc -O3 --offload-arch=gfx950 -I pil2-stark-js_amd/csrc tools/bn_mfma_unit.hip -o tools/bn_mfma_unit
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "bn_mfma.cuh"
#include "bn_field29.cuh"
using namespace bn;

__device__ u32 rnd(u32 &s) { s = s * 1664525u + 1013904223u; return s ^ (s >> 15); }

// plain restatement: positions -> value -> (V + K 2^32 + m r) / 2^32, canonical
__device__ void finish_ref(const int *p0, const int *p1, const u32 *kc, u32 out[8]) {
    // this lane holds 16 positions of tile 0 (p0) and of tile 1 (p1); partner lane^32 holds the other halves
    u32 w[10];
    for (int h = 0; h < 2; h++) {
        const int *p = h ? p1 : p0;
        u64 acc = 0;
        u32 limb[5] = { 0, 0, 0, 0, 0 };
        // 16 positions -> 5 limbs by schoolbook
        unsigned __int128 lo = 0; u64 hi = 0;
        for (int k = 15; k >= 0; k--) { hi = (hi << 8) | (u64)(lo >> 120); lo = (lo << 8) + (u32)p[k]; }
        (void)acc;
        limb[0] = (u32)lo; limb[1] = (u32)(lo >> 32); limb[2] = (u32)(lo >> 64); limb[3] = (u32)(lo >> 96); limb[4] = (u32)hi;
        for (int l = 0; l < 5; l++) w[5 * h + l] = limb[l];
    }
    u32 lo5[5], hi5[5];
    const bool upper = (threadIdx.x & 63) >= 32;
    for (int l = 0; l < 5; l++) {
        const u32 mine = upper ? w[l] : w[5 + l];            // what the partner needs: upper gives its tile-0 words, lower its tile-1 words
        const u32 got = __shfl_xor(mine, 32);
        lo5[l] = upper ? got : w[l];
        hi5[l] = upper ? w[5 + l] : got;
    }
    u32 t[9];
    u64 c = 0;
    for (int l = 0; l < 9; l++) { c += (l < 5 ? lo5[l] : 0u); if (l >= 4) c += hi5[l - 4]; t[l] = (u32)c; c >>= 32; }
    c = 0;
    for (int l = 1; l < 9; l++) { c += (u64)t[l] + kc[l - 1]; t[l] = (u32)c; c >>= 32; }
    const u32 m = t[0] * N0INV;
    u64 v = (u64)m * r_limb(0) + t[0];
    u32 o[9];
    for (int l = 1; l < 8; l++) { v = (u64)m * r_limb(l) + t[l] + (v >> 32); o[l - 1] = (u32)v; }
    o[7] = t[8] + (u32)(v >> 32); o[8] = 0;
    cond_sub_r(o); cond_sub_r(o);
    for (int l = 0; l < 8; l++) out[l] = o[l];
}

__global__ void k_finish(int *bad, u32 seed, int mode) {
    u32 s = seed * 977u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    bnm::v16i a0, a1;
    for (int k = 0; k < 16; k++) {
        a0[k] = mode == 0 ? (int)(rnd(s) & 0x3ffffff) : mode == 1 ? 0x3ffffff : (int)(rnd(s) & 1 ? 0x3ffffff : 1);
        a1[k] = mode == 0 ? (int)(rnd(s) & 0x3ffffff) : mode == 1 ? 0x3ffffff : (int)(rnd(s) & 1 ? 0x3ffffff : 1);
    }
    // make the values pass through a vector instruction like an MFMA result would (compiler-visible producer)
    u32 kc[8];
    u32 s2 = seed * 31u + blockIdx.x;                     // wave-uniform constant below r
    for (int l = 0; l < 8; l++) kc[l] = rnd(s2);
    kc[7] &= 0x0fffffff;
    int p0[16], p1[16];
    for (int k = 0; k < 16; k++) { p0[k] = a0[k]; p1[k] = a1[k]; }
    u32 want[8], got[8];
    finish_ref(p0, p1, kc, want);
    const bnm::Sh sh = bnm::sh_init();
    bnm::finish_row(a0, a1, kc, got, sh);
    bnm::canon(got);
    int b = 0;
    for (int l = 0; l < 8; l++) b |= want[l] != got[l];
    if (b) atomicAdd(bad, 1);
}

__global__ void k_mul(int *bad, u32 seed, int mode) {
    u32 s = seed * 131u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    u32 a[8], b[8];
    for (int l = 0; l < 8; l++) { a[l] = mode == 1 ? 0xffffffffu : rnd(s); b[l] = mode == 1 ? 0xffffffffu : rnd(s); }
    a[7] &= 0x7fffffffu; b[7] &= 0x7fffffffu;             // lazy representatives below 2^255
    u32 ac[8], bc[8];
    for (int l = 0; l < 8; l++) { ac[l] = a[l]; bc[l] = b[l]; }
    bnm::canon(ac); bnm::canon(bc);
    u32 want[8], got[8];
    fr_mul(want, ac, bc);
    fr_mul_nr(got, a, b);
    int bd = got[7] >> 31;                                // must stay below 2^255
    bnm::canon(got);
    for (int l = 0; l < 8; l++) bd |= want[l] != got[l];
    // x^5 through three reduction-free products
    u32 x2[8], x4[8], x5[8], y2[8], y4[8], y5[8];
    fr_mul_nr(x2, a, a); fr_mul_nr(x4, x2, x2); fr_mul_nr(x5, x4, a);
    fr_mul(y2, ac, ac); fr_mul(y4, y2, y2); fr_mul(y5, y4, ac);
    bd |= x5[7] >> 31;
    bnm::canon(x5);
    for (int l = 0; l < 8; l++) bd |= y5[l] != x5[l];
    if (bd) atomicAdd(bad, 1);
}

// the radix-2^29 S-box: x^5 / 2^1044 of a lazy x, i.e. the Montgomery (2^256) form of x^5 times 2^-20
__global__ void k_pow29(int *bad, u32 seed, int mode) {
    u32 s = seed * 257u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    u32 a[8];
    for (int l = 0; l < 8; l++) a[l] = mode == 1 ? 0xffffffffu : rnd(s);
    if (mode == 1) a[7] = 0xb0000000u; else if (mode == 2) a[7] = a[7] % 0xb0000000u; else a[7] &= 0x7fffffffu;      // below 0.69 * 2^256
    u32 ac[8], y[8], want[8], x2[8], x4[8];
    for (int l = 0; l < 8; l++) { ac[l] = a[l]; y[l] = a[l]; }
    bnm::canon(ac); bnm::canon(ac);                       // up to 0.69 * 2^256 < 4r: a second pair of subtractions
    fr_mul(x2, ac, ac); fr_mul(x4, x2, x2); fr_mul(want, x4, ac);
    bn29::pow5(y);
    int bd = y[7] >> 31;
    u32 t20[8] = { 1u << 20, 0, 0, 0, 0, 0, 0, 0 }, r2[8], m20[8], got[8];
    for (int l = 0; l < 8; l++) r2[l] = r2_limb(l);
    fr_mul(m20, t20, r2);                                 // 2^20 in Montgomery form
    bnm::canon(y);
    fr_mul(got, y, m20);                                  // y * 2^20
    for (int l = 0; l < 8; l++) bd |= want[l] != got[l];
    if (bd) atomicAdd(bad, 1);
}

int main() {
    int *bad; (void)hipMalloc((void **)&bad, 4);
    for (int mode = 0; mode < 3; mode++) {
        (void)hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_finish, dim3(4096), dim3(64), 0, 0, bad, 12345u + mode, mode);
        int h = -1; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("finish_row vs plain restatement, mode %d: %d of %d lanes differ\n", mode, h, 4096 * 64);
    }
    for (int mode = 0; mode < 2; mode++) {
        (void)hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_mul, dim3(4096), dim3(64), 0, 0, bad, 777u + mode, mode);
        int h = -1; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("fr_mul_nr / x^5 on lazy operands vs canonical, mode %d: %d of %d lanes differ\n", mode, h, 4096 * 64);
    }
    for (int mode = 0; mode < 3; mode++) {
        (void)hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k_pow29, dim3(4096), dim3(64), 0, 0, bad, 4242u + mode, mode);
        int h = -1; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("radix-2^29 x^5 (times 2^20) vs the 32-bit-limb S-box, mode %d: %d of %d lanes differ\n", mode, h, 4096 * 64);
    }
    return 0;
}
