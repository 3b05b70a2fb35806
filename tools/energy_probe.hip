// Energy per vector / matrix instruction on the whole chip: the Poseidon kernels run at the board's power cap (tools/power_probe.py), so what an
// instruction costs in JOULES decides the time, not what it costs in issue cycles.  Every class runs alone on all 256 CUs (two waves per SIMD, eight
// independent chains per wave) for ~1.5 s while a host thread samples the card's hwmon files; reported: board power, shader clock, issue cycles per
// instruction per SIMD at that clock, and (power - resident-idle power) x time / wave-instructions = nJ per wave-instruction.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/energy_probe.hip -o /tmp/energy_probe -lpthread && /tmp/energy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <glob.h>
#include <atomic>
#include <thread>
#include <vector>
#include <string>
#include <chrono>
typedef uint64_t u64; typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { OP_SLEEP, OP_ADD, OP_XOR, OP_ADDC, OP_MULLO, OP_MULHI, OP_MAD64, OP_MAD64_ZERO, OP_MAD64_SMALL, OP_MAD24, OP_MFMA, OP_MFMA_ZERO, OP_MIX_MAD_MFMA, OP_LSHR64, OP_PERM, OP_MFMA_SPARSE_A, OP_MFMA444, OP_MFMA444_ZERO, OP_MFMA16X16X32, OP_DOT4, OP_MIX_GL, OP_MIX_GL_NOMFMA, OP_MIX_GL_SPARSE, OP_LAYER18, OP_LAYER8P, OP_LAYER8S, N_OPS };
static const char *NAMES[N_OPS] = { "s_sleep (waves resident, nothing issued)", "v_add_u32", "v_xor_b32", "v_addc_co_u32 (vcc)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32 (random operands)",
                                    "v_mad_u64_u32 (zero operands)", "v_mad_u64_u32 (16-bit operands)", "v_mad_u32_u24", "v_mfma_i32_32x32x32_i8 (random bytes)", "v_mfma_i32_32x32x32_i8 (zero operands)",
                                    "40 v_mad_u64_u32 : 8 v_mfma (both pipes saturated)", "v_lshrrev_b64", "v_perm_b32", "v_mfma_i32_32x32x32_i8 (A = the GL MDS tile: 1/8 dense 6-bit, B random)", "v_mfma_i32_4x4x4_16b_i8 (random bytes)",
                                    "v_mfma_i32_4x4x4_16b_i8 (zero operands)", "v_mfma_i32_16x16x32_i8 (random bytes)", "v_dot4_u32_u8",
                                    "GL hash mix: 40 v_mad_u64_u32 + 104 v_add/v_xor : 8 v_mfma (random A)", "the same vector stream without the matrix instructions", "GL hash mix with A = the MDS tile",
                                    "one MDS layer as built: 90 v_mad_u64_u32 + 234 plain : 18 v_mfma", "the layer with byte-transposed operands: + 48 v_perm_b32 : 8 v_mfma", "the same with 72 SDWA byte moves instead of the 48 v_perm_b32" };
// instructions counted per inner repetition (8 chains): the mix counts its multiply-adds (the matrix instructions ride along)
template <int OP>
__global__ void __launch_bounds__(256) k(u64 *out, int iters, u32 seed) {
    const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
    u32 a[8], b[8]; u64 q[8];
    for (int i = 0; i < 8; i++) {
        u32 x = (id * 2654435761u) ^ (seed + i * 0x9E3779B9u); x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
        u32 y = x * 0xC2B2AE35u; y ^= y >> 16;
        if (OP == OP_MAD64_ZERO || OP == OP_MFMA_ZERO || OP == OP_MFMA444_ZERO) { x = 0; y = 0; }
        if (OP == OP_MAD64_SMALL) { x &= 0xFFFF; y &= 0xFFFF; }
        a[i] = x; b[i] = y; q[i] = (OP == OP_MAD64_ZERO) ? 0 : ((u64)x << 32) | y;
    }
    v4i ta = { (int)a[0], (int)a[1], (int)a[2], (int)a[3] }, tb = { (int)b[0], (int)b[1], (int)b[2], (int)b[3] };
    v16i c0 = { 0 }, c1 = { 0 };
    v4i d0 = { 0, 0, 0, 0 }, d1 = { 0, 0, 0, 0 }, d2 = { 0, 0, 0, 0 }, d3 = { 0, 0, 0, 0 };
    long t2a = ((long)a[0] << 32) | a[1], t2b = ((long)b[0] << 32) | b[1];
    if (OP == OP_MFMA_SPARSE_A || OP == OP_MIX_GL_SPARSE || OP == OP_LAYER18) {                   // poseidon_mds_mfma.cuh's operand: rows owned by the lane's half, one 6-bit byte per K-dword
        const u32 lane = threadIdx.x & 63, i = lane & 31, g = lane >> 5;
        const bool owned = ((i >> 2) & 1) == g;
        for (int e = 0; e < 4; e++) ta[e] = owned ? (int)(((a[e] & 63u) | 1u) << (8 * (i & 3))) : 0;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define XOR_(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define ADDCVCC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define MAD(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define LSHR64(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(q[i]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            if (OP == OP_SLEEP) { asm volatile("s_sleep 8"); }
            if (OP == OP_ADD) { REP8(ADD) } if (OP == OP_XOR) { REP8(XOR_) } if (OP == OP_ADDC) { REP8(ADDCVCC) }
            if (OP == OP_MULLO) { REP8(MULLO) } if (OP == OP_MULHI) { REP8(MULHI) }
            if (OP == OP_MAD64 || OP == OP_MAD64_ZERO || OP == OP_MAD64_SMALL) { REP8(MAD) }
            if (OP == OP_MAD24) { REP8(MAD24) } if (OP == OP_LSHR64) { REP8(LSHR64) } if (OP == OP_PERM) { REP8(PERM) }
            if (OP == OP_MFMA444 || OP == OP_MFMA444_ZERO) {
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %2, %0" : "+v"(d0) : "v"(a[0]), "v"(b[0]));
                    asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %2, %0" : "+v"(d1) : "v"(a[1]), "v"(b[1]));
                    asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %2, %0" : "+v"(d2) : "v"(a[2]), "v"(b[2]));
                    asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %2, %0" : "+v"(d3) : "v"(a[3]), "v"(b[3]));
                }
            }
            if (OP == OP_MFMA16X16X32) {
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(d0) : "v"(t2a), "v"(t2b));
                    asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(d1) : "v"(t2b), "v"(t2a));
                    asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(d2) : "v"(t2a), "v"(t2b));
                    asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(d3) : "v"(t2b), "v"(t2a));
                }
            }
#define DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            if (OP == OP_DOT4) { REP8(DOT4) }
            if (OP == OP_MFMA || OP == OP_MFMA_ZERO || OP == OP_MFMA_SPARSE_A) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c0) : "v"(ta), "v"(tb));
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c1) : "v"(tb), "v"(ta));
                }
            }
            if (OP == OP_MIX_GL || OP == OP_MIX_GL_NOMFMA || OP == OP_MIX_GL_SPARSE) {      // per matrix instruction: 5 multiply-adds, 13 plain instructions
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    if (OP != OP_MIX_GL_NOMFMA) {
                        if (m & 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c1) : "v"(ta), "v"(tb));
                        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c0) : "v"(ta), "v"(tb));
                    }
                    MAD(0) MAD(1) MAD(2) MAD(3) MAD(4)
                    REP8(ADD) XOR_(0) XOR_(1) XOR_(2) XOR_(3) XOR_(4)
                }
            }
            if (OP == OP_LAYER18 || OP == OP_LAYER8P || OP == OP_LAYER8S) {      // r counts quarter layers: the body below is half a layer, run for r = 0, 2
                if ((r & 1) == 0) {
                    constexpr int NM = OP == OP_LAYER18 ? 9 : 4;
#pragma unroll
                    for (int m = 0; m < 9; m++) {
                        if (m < NM) {
                            if (m & 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c1) : "v"(ta), "v"(tb));
                            else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c0) : "v"(ta), "v"(tb));
                        }
                        MAD(0) MAD(1) MAD(2) MAD(3) MAD(4)
                        REP8(ADD) XOR_(0) XOR_(1) XOR_(2) XOR_(3) XOR_(4)
                        if (OP == OP_LAYER8P && m < 3) { REP8(PERM) }
#define SDWA(i) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a[i]) : "v"(b[i]));
                        if (OP == OP_LAYER8S && m < 4) { REP8(SDWA) if (m == 0) { SDWA(0) SDWA(1) SDWA(2) SDWA(3) } }
                    }
                }
            }
            if (OP == OP_MIX_MAD_MFMA) {          // 40 multiply-adds : 8 matrix instructions
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c0) : "v"(ta), "v"(tb));
                    REP8(MAD)
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c1) : "v"(tb), "v"(ta));
                    if (m < 1) { REP8(MAD) }
                }
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + q[i];
    for (int i = 0; i < 16; i++) s += (u32)c0[i] + (u32)c1[i];
    for (int i = 0; i < 4; i++) s += (u32)d0[i] + (u32)d1[i] + (u32)d2[i] + (u32)d3[i];
    out[id] = s;
}

struct Sampler {
    std::vector<std::string> pw, fq;
    std::atomic<bool> run{ false }, stop{ false };
    std::vector<std::vector<double>> p, f;
    std::thread th;
    static double rd(const std::string &path) { FILE *fp = fopen(path.c_str(), "r"); if (!fp) return -1; double v = -1; if (fscanf(fp, "%lf", &v) != 1) v = -1; fclose(fp); return v; }
    void open() {
        glob_t g;
        if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g) == 0) {
            for (size_t i = 0; i < g.gl_pathc; i++) { std::string s = g.gl_pathv[i]; pw.push_back(s); fq.push_back(s.substr(0, s.rfind('/')) + "/freq1_input"); }
            globfree(&g);
        }
        if (pw.empty() && glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", 0, nullptr, &g) == 0) {
            for (size_t i = 0; i < g.gl_pathc; i++) { std::string s = g.gl_pathv[i]; pw.push_back(s); fq.push_back(s.substr(0, s.rfind('/')) + "/freq1_input"); }
            globfree(&g);
        }
        p.resize(pw.size()); f.resize(pw.size());
        th = std::thread([this] {
            while (!stop) {
                if (run) for (size_t i = 0; i < pw.size(); i++) { p[i].push_back(rd(pw[i]) / 1e6); f[i].push_back(rd(fq[i]) / 1e6); }
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
        });
    }
    void begin() { for (auto &v : p) v.clear(); for (auto &v : f) v.clear(); run = true; }
    // the card under load = the sensor with the largest mean; the first quarter of the samples (ramp) is dropped
    void end(double &power, double &clock, int &which) {
        run = false; std::this_thread::sleep_for(std::chrono::milliseconds(12));
        power = clock = 0; which = -1;
        for (size_t i = 0; i < p.size(); i++) {
            double s = 0, c = 0; size_t n = 0;
            for (size_t k = p[i].size() / 4; k < p[i].size(); k++) { s += p[i][k]; c += f[i][k]; n++; }
            if (n && s / n > power) { power = s / n; clock = c / n; which = (int)i; }
        }
    }
    void close() { stop = true; th.join(); }
};

template <int OP> static void run_one(u64 *out, Sampler &S, double idle_resident, double *idle_out) {
    const int blocks = 256 * 2;                                          // two workgroups of four waves per CU: two waves per SIMD
    int iters = OP == OP_SLEEP ? 20000 : (OP == OP_MFMA || OP == OP_MFMA_ZERO || OP == OP_MFMA_SPARSE_A) ? 20000 : 40000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 200, 1u); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms1; (void)hipEventElapsedTime(&ms1, e0, e1);
    int reps = (int)(1500.0 / ms1) + 1;
    S.begin();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u + r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    double power, clock; int which; S.end(power, clock, which);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_rep8 = (OP == OP_MFMA || OP == OP_MFMA_ZERO || OP == OP_MFMA_SPARSE_A) ? 8 : (OP == OP_MIX_MAD_MFMA || OP == OP_MIX_GL || OP == OP_MIX_GL_NOMFMA || OP == OP_MIX_GL_SPARSE) ? 40 : (OP == OP_LAYER18 || OP == OP_LAYER8P || OP == OP_LAYER8S) ? 22.5 : OP == OP_SLEEP ? 1 : 8;
    const double wave_insts = (double)blocks * 4 * reps * iters * 4 * per_rep8;      // per wave: iters x 4 repetitions x instructions
    const double simd_cycles = ms * 1e-3 * clock * 1e6;                               // cycles every SIMD ran
    const double insts_per_simd = wave_insts / (256.0 * 4);
    if (idle_out) *idle_out = power;
    printf("%-82s %7.1f W  %5.0f MHz  %8.2f ms/launch  %6.2f cycles/inst/SIMD  %7.3f nJ/wave-inst (above resident idle %.0f W)\n", NAMES[OP], power, clock, ms / reps,
           simd_cycles / insts_per_simd, idle_resident > 0 ? (power - idle_resident) * ms * 1e-3 / wave_insts * 1e9 : 0.0, idle_resident);
    fflush(stdout);
}

int main(int argc, char **argv) {
    u64 *out; (void)hipMalloc((void **)&out, 8ull * 256 * 2 * 256);
    Sampler S; S.open();
    printf("sensors: %zu\n", S.pw.size());
    double idle = 0;
    run_one<OP_SLEEP>(out, S, 0, &idle);
    if (argc < 2) {
    run_one<OP_ADD>(out, S, idle, nullptr);
    run_one<OP_XOR>(out, S, idle, nullptr);
    run_one<OP_ADDC>(out, S, idle, nullptr);
    run_one<OP_PERM>(out, S, idle, nullptr);
    run_one<OP_LSHR64>(out, S, idle, nullptr);
    run_one<OP_MULLO>(out, S, idle, nullptr);
    run_one<OP_MULHI>(out, S, idle, nullptr);
    run_one<OP_MAD24>(out, S, idle, nullptr);
    run_one<OP_MAD64>(out, S, idle, nullptr);
    run_one<OP_MAD64_SMALL>(out, S, idle, nullptr);
    run_one<OP_MAD64_ZERO>(out, S, idle, nullptr);
    run_one<OP_MFMA>(out, S, idle, nullptr);
    run_one<OP_MFMA_ZERO>(out, S, idle, nullptr);
    run_one<OP_MIX_MAD_MFMA>(out, S, idle, nullptr);
    run_one<OP_MFMA_SPARSE_A>(out, S, idle, nullptr);
    run_one<OP_MFMA444>(out, S, idle, nullptr);
    run_one<OP_MFMA444_ZERO>(out, S, idle, nullptr);
    run_one<OP_MFMA16X16X32>(out, S, idle, nullptr);
    run_one<OP_DOT4>(out, S, idle, nullptr);
    run_one<OP_MIX_GL>(out, S, idle, nullptr);
    run_one<OP_MIX_GL_SPARSE>(out, S, idle, nullptr);
    run_one<OP_MIX_GL_NOMFMA>(out, S, idle, nullptr);
    }
    run_one<OP_LAYER18>(out, S, idle, nullptr);
    run_one<OP_LAYER8P>(out, S, idle, nullptr);
    run_one<OP_LAYER8S>(out, S, idle, nullptr);
    run_one<OP_LAYER18>(out, S, idle, nullptr);
    run_one<OP_LAYER8P>(out, S, idle, nullptr);
    run_one<OP_LAYER8S>(out, S, idle, nullptr);
    S.close();
    return 0;
}
