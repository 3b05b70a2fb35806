// libpil2gl runtime: device selection, error state, tables, scratch and raw device buffers.
#include "common.h"
#include "gl_field.cuh"
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace pil2gl {

static thread_local char g_err[512] = "";
static bool g_ready = false;
static int g_device = -1;
static Tables g_tables = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
static u64 *g_tables_mem = nullptr;
static const u32 N_SCRATCH = 14;      // 12, 13: device copies of the host-pointer NTT entry points (ntt.hip host_wrap)
static u64 *g_scratch[N_SCRATCH] = { nullptr };
static u64 g_scratch_words[N_SCRATCH] = { 0 };

int fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    (void)hipGetLastError();
    return (e == hipErrorOutOfMemory) ? PIL2GL_ENOMEM : (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? PIL2GL_ENODEV : PIL2GL_EHIP;
}

// ---- host Goldilocks (tables and scalar parameters only) ----
static const u64 HP = 0xFFFFFFFF00000001ull;
u64 h_mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % HP); }
u64 h_add(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % HP); }
u64 h_sub(u64 a, u64 b) { return a >= b ? a - b : HP - b + a; }
u64 h_pow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = h_mul(r, a); a = h_mul(a, a); e >>= 1; } return r; }
u64 h_inv(u64 a) { return h_pow(a, HP - 2); }
u64 h_root(u32 bits) {                      // f3g.js:40, fft/fft.js:39-50
    u64 w = 7277203076849721926ull;
    for (u32 i = 32; i > bits; i--) w = h_mul(w, w);
    return w;
}
void h_e3_mul(const u64 a[3], const u64 b[3], u64 r[3]) {   // f3g.js:94-102
    u64 A = h_mul(h_add(a[0], a[1]), h_add(b[0], b[1]));
    u64 B = h_mul(h_add(a[0], a[2]), h_add(b[0], b[2]));
    u64 C = h_mul(h_add(a[1], a[2]), h_add(b[1], b[2]));
    u64 D = h_mul(a[0], b[0]), E = h_mul(a[1], b[1]), F = h_mul(a[2], b[2]);
    u64 G = h_sub(D, E);
    u64 r0 = h_sub(h_add(C, G), F);
    u64 r1 = h_sub(h_sub(h_sub(h_add(A, C), E), E), D);
    u64 r2 = h_sub(B, G);
    r[0] = r0; r[1] = r1; r[2] = r2;
}

static void fill_pow256(u64 *T, u64 g) {
    for (int t = 0; t < 4; t++) {
        u64 acc = 1;
        for (int i = 0; i < 256; i++) { T[t * 256 + i] = acc; acc = h_mul(acc, g); }
        g = acc;                            // g^(256) = next level's base
    }
}

const Tables &tables() { return g_tables; }
hipStream_t as_stream(void *s) { return (hipStream_t)s; }

int ensure_init() {
    if (g_ready) return PIL2GL_OK;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    return pil2gl_init(dev);
}

// The reference issues its operator calls one after the other from one JS thread and the ABI is specified for
// that: ONE caller at a time, any stream.  The lock below only keeps the process-global state (device tables, scratch slots,
// compiled-kernel cache) consistent if two host threads do enter at once; it does not make one scratch slot serve two
// concurrent calls.
std::recursive_mutex &runtime_lock() { static std::recursive_mutex m; return m; }

int scratch(u32 slot, u64 nWords, u64 **out) {
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    if (slot >= N_SCRATCH) return fail(PIL2GL_EINVAL, "bad scratch slot");
    if (g_scratch_words[slot] < nWords) {
        if (g_scratch[slot]) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g_scratch[slot])); g_scratch[slot] = nullptr; g_scratch_words[slot] = 0; }
        HIP_TRY(hipMalloc((void **)&g_scratch[slot], nWords * 8));
        g_scratch_words[slot] = nWords;
    }
    *out = g_scratch[slot];
    return PIL2GL_OK;
}

int stage_acquire(u64 nWords, u64 **out, bool *owned) {
    if (nWords <= (2ull << 20)) { *owned = false; return scratch(11, nWords + 1, out); }
    *owned = true;
    HIP_TRY(hipMalloc((void **)out, (nWords + 1) * 8));
    return PIL2GL_OK;
}
void stage_release(u64 *p, bool owned) { if (owned && p) (void)hipFree(p); }

}  // namespace pil2gl

using namespace pil2gl;

extern "C" {

int pil2gl_version(void) { return 1; }
const char *pil2gl_last_error(void) { return g_err; }

int pil2gl_init(int device) {
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) { (void)hipGetLastError(); return fail(PIL2GL_ENODEV, "no HIP device available (libpil2gl has no CPU fallback)"); }
    if (device < 0 || device >= n) return fail(PIL2GL_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    if (g_ready && g_device == device) return PIL2GL_OK;
    if (g_ready) pil2gl_shutdown();
    HIP_TRY(hipSetDevice(device));
    std::vector<u64> host(6 * 1024);
    u64 w32 = h_root(32);
    fill_pow256(&host[0], w32);
    fill_pow256(&host[1024], h_inv(w32));
    fill_pow256(&host[2048], 7);
    fill_pow256(&host[3072], h_inv(7));
    {
        const u64 w10 = h_root(10), w10i = h_inv(w10);
        u64 a = 1, b = 1;
        for (int j = 0; j < 1024; j++) { host[4096 + j] = a; host[5120 + j] = b; a = h_mul(a, w10); b = h_mul(b, w10i); }
    }
    HIP_TRY(hipMalloc((void **)&g_tables_mem, host.size() * 8));
    HIP_TRY(hipMemcpy(g_tables_mem, host.data(), host.size() * 8, hipMemcpyHostToDevice));
    g_tables.powW = g_tables_mem;
    g_tables.powWi = g_tables_mem + 1024;
    g_tables.pow7 = g_tables_mem + 2048;
    g_tables.pow7i = g_tables_mem + 3072;
    g_tables.tw1024 = g_tables_mem + 4096;
    g_tables.tw1024i = g_tables_mem + 5120;
    g_device = device;
    g_ready = true;
    return PIL2GL_OK;
}

void pil2gl_shutdown(void) {
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    if (!g_ready) return;
    (void)hipDeviceSynchronize();
    jit_clear();                                          // compiled expression kernels belong to the device being left
    for (u32 i = 0; i < N_SCRATCH; i++) { if (g_scratch[i]) (void)hipFree(g_scratch[i]); g_scratch[i] = nullptr; g_scratch_words[i] = 0; }
    if (g_tables_mem) (void)hipFree(g_tables_mem);
    g_tables_mem = nullptr;
    g_ready = false; g_device = -1;
}

int pil2gl_device_info(char *name, uint32_t nameLen, uint32_t *numCUs, uint64_t *totalMem) {
    P2_TRY(ensure_init());
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, g_device));
    if (name && nameLen) { strncpy(name, p.gcnArchName, nameLen - 1); name[nameLen - 1] = 0; }
    if (numCUs) *numCUs = (uint32_t)p.multiProcessorCount;
    if (totalMem) *totalMem = (uint64_t)p.totalGlobalMem;
    return PIL2GL_OK;
}

int pil2gl_dev_alloc(uint64_t nWords, uint64_t **out) {
    P2_TRY(ensure_init());
    if (!out) return fail(PIL2GL_EINVAL, "null out pointer");
    HIP_TRY(hipMalloc((void **)out, (nWords ? nWords : 1) * 8));
    return PIL2GL_OK;
}
int pil2gl_dev_free(uint64_t *p) { if (p) HIP_TRY(hipFree(p)); return PIL2GL_OK; }
int pil2gl_dev_zero(uint64_t *p, uint64_t nWords, void *stream) { if (!nWords) return PIL2GL_OK; if (!p) return fail(PIL2GL_EINVAL, "null buffer"); HIP_TRY(hipMemsetAsync(p, 0, nWords * 8, as_stream(stream))); return PIL2GL_OK; }
int pil2gl_dev_upload(uint64_t *dst, const uint64_t *hostSrc, uint64_t nWords) { if (!nWords) return PIL2GL_OK; if (!dst || !hostSrc) return fail(PIL2GL_EINVAL, "null buffer"); HIP_TRY(hipMemcpy(dst, hostSrc, nWords * 8, hipMemcpyHostToDevice)); return PIL2GL_OK; }
int pil2gl_dev_download(uint64_t *hostDst, const uint64_t *src, uint64_t nWords) { if (!nWords) return PIL2GL_OK; if (!hostDst || !src) return fail(PIL2GL_EINVAL, "null buffer"); HIP_TRY(hipMemcpy(hostDst, src, nWords * 8, hipMemcpyDeviceToHost)); return PIL2GL_OK; }
// The scalar exports of the reference's WASM module (glwasm.js:1269-1275: add, mul, square on i64 words): host arithmetic on canonical
// operands, the same the library's own planners use for twiddles and shifts.  Not a compute path: one element per call, no device involved.
uint64_t pil2gl_add(uint64_t a, uint64_t b) { return h_add(a % 0xFFFFFFFF00000001ull, b % 0xFFFFFFFF00000001ull); }
uint64_t pil2gl_mul(uint64_t a, uint64_t b) { return h_mul(a % 0xFFFFFFFF00000001ull, b % 0xFFFFFFFF00000001ull); }
uint64_t pil2gl_square(uint64_t a) { return h_mul(a % 0xFFFFFFFF00000001ull, a % 0xFFFFFFFF00000001ull); }
int pil2gl_sync(void *stream) { HIP_TRY(hipStreamSynchronize(as_stream(stream))); return PIL2GL_OK; }

}  // extern "C"

// ---- selftests (device arithmetic on arbitrary operands, used by tests/) ----
__global__ void selftest_field_kernel(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *m, uint64_t *s, uint64_t *d) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    m[i] = gl::mul(a[i], b[i]);
    s[i] = gl::add(a[i], b[i]);
    d[i] = gl::sub(a[i], b[i]);
}
__global__ void selftest_ext_kernel(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *m, uint64_t *iv) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl::E3 x = { { a[3 * i], a[3 * i + 1], a[3 * i + 2] } }, y = { { b[3 * i], b[3 * i + 1], b[3 * i + 2] } };
    gl::E3 r = gl::e3_mul(x, y), q = gl::e3_inv(x);
    for (int c = 0; c < 3; c++) { m[3 * i + c] = r.v[c]; iv[3 * i + c] = q.v[c]; }
}

extern "C" int pil2gl_selftest_field(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *mul, uint64_t *add, uint64_t *sub) {
    P2_TRY(ensure_init());
    if (!n) return PIL2GL_OK;
    u64 *d;
    HIP_TRY(hipMalloc((void **)&d, 5 * n * 8));
    HIP_TRY(hipMemcpy(d, a, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d + n, b, n * 8, hipMemcpyHostToDevice));
    selftest_field_kernel<<<(unsigned)((n + 255) / 256), 256>>>(d, d + n, n, d + 2 * n, d + 3 * n, d + 4 * n);
    KERNEL_CHECK();
    HIP_TRY(hipMemcpy(mul, d + 2 * n, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(add, d + 3 * n, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sub, d + 4 * n, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return PIL2GL_OK;
}
// the hand-written products on arbitrary operands (any u64 representatives): x = mul_lazy_x (exact), b = mul_lazy_b with its
// "recompute me" flag (flag[i] != 0: the lane's last subtraction borrowed, b[i] is then not to be used)
__global__ void selftest_products_kernel(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *x, uint64_t *pb, uint64_t *flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    if (!live) i = n - 1;
    const uint64_t va = a[i], vb = b[i];
    uint64_t bad = 0;
    const uint64_t rb = gl::mul_lazy_b(va, vb, bad);          // bad: one bit per lane of the wave
    const uint64_t rx = gl::mul_lazy_x(va, vb);
    if (!live) return;
    x[i] = gl::canon(rx); pb[i] = gl::canon(rb);
    flag[i] = (bad >> (threadIdx.x & 63)) & 1;
}
extern "C" int pil2gl_selftest_products(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *x, uint64_t *pb, uint64_t *flag) {
    P2_TRY(ensure_init());
    if (!n) return PIL2GL_OK;
    u64 *d;
    HIP_TRY(hipMalloc((void **)&d, 5 * n * 8));
    HIP_TRY(hipMemcpy(d, a, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d + n, b, n * 8, hipMemcpyHostToDevice));
    selftest_products_kernel<<<(unsigned)((n + 255) / 256), 256>>>(d, d + n, n, d + 2 * n, d + 3 * n, d + 4 * n);
    KERNEL_CHECK();
    HIP_TRY(hipMemcpy(x, d + 2 * n, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pb, d + 3 * n, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(flag, d + 4 * n, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return PIL2GL_OK;
}
extern "C" int pil2gl_selftest_ext(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *mul, uint64_t *inv) {
    P2_TRY(ensure_init());
    if (!n) return PIL2GL_OK;
    u64 *d;
    HIP_TRY(hipMalloc((void **)&d, 12 * n * 8));
    HIP_TRY(hipMemcpy(d, a, 3 * n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d + 3 * n, b, 3 * n * 8, hipMemcpyHostToDevice));
    selftest_ext_kernel<<<(unsigned)((n + 63) / 64), 64>>>(d, d + 3 * n, n, d + 6 * n, d + 9 * n);
    KERNEL_CHECK();
    HIP_TRY(hipMemcpy(mul, d + 6 * n, 3 * n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(inv, d + 9 * n, 3 * n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return PIL2GL_OK;
}
