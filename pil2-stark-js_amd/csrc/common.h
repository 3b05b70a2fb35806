// Host-side shared declarations of libpil2gl (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include "../../include/pil2gl.h"

namespace pil2gl {

typedef uint64_t u64;
typedef uint32_t u32;

// largest transform (rows = 2^this) the NTT / LDE / FRI-fold entry points accept on one device
#define PIL2GL_MAX_NTT_BITS 30

// error plumbing ------------------------------------------------------------
int  fail(int code, const char *fmt, ...);          // records the message, returns code
int  hip_fail(hipError_t e, const char *what);      // -> PIL2GL_EHIP
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return ::pil2gl::hip_fail(e_, #expr); } while (0)
#define P2_TRY(expr)  do { int rc_ = (expr); if (rc_ != PIL2GL_OK) return rc_; } while (0)
#define KERNEL_CHECK() HIP_TRY(hipGetLastError())

int ensure_init();                                   // pil2gl_init(current device) on first use
std::recursive_mutex &runtime_lock();                // guards the process-global runtime state (tables, scratch, kernel cache)
void jit_clear();                                    // unloads the run-time compiled expression kernels (expr.hip)

// host-side Goldilocks (table construction and scalar parameters only) ---------
u64 h_mul(u64 a, u64 b);
u64 h_add(u64 a, u64 b);
u64 h_sub(u64 a, u64 b);
u64 h_pow(u64 a, u64 e);
u64 h_inv(u64 a);
u64 h_root(u32 bits);                                // F.w[bits]
void h_e3_mul(const u64 a[3], const u64 b[3], u64 r[3]);

// device tables -------------------------------------------------------------------
// pow256 layout: T[t*256 + i] = g^(i << (8t)), t = 0..3  (any 32-bit exponent in 3 multiplications)
struct Tables {
    const u64 *powW;      // g = w[32]   (forward roots of unity)
    const u64 *powWi;     // g = w[32]^-1
    const u64 *pow7;      // g = 7       (coset shift)
    const u64 *pow7i;     // g = 7^-1
    const u64 *tw1024;    // tw1024[j] = w[10]^j, j < 1024 (tile twiddles of the NTT passes)
    const u64 *tw1024i;   // w[10]^-j
};
const Tables &tables();

// scratch (grown on demand, kept across calls) ------------------------------------------
int scratch(u32 slot, u64 nWords, u64 **out);

// device staging for the host-pointer entry points: small requests (<= 16 MB) come from a persistent slot -- a hipMalloc /
// hipFree pair per call costs a device synchronisation each, which dominated the transcript's single permutations --
// larger ones are allocated and released per call.  stage_release() frees only what stage_acquire() allocated.
int stage_acquire(u64 nWords, u64 **out, bool *owned);
void stage_release(u64 *p, bool owned);

hipStream_t as_stream(void *s);

// internal launchers shared between translation units --------------------------------------
int ntt_launch(const u64 *src, u64 nPols, u32 nBits, u64 *dst, bool inverse, hipStream_t st);
int lde_launch(const u64 *src, u64 nPols, u32 nBits, u64 *dst, u32 nBitsExt, hipStream_t st, u32 cosetBegin, u32 cosetCount, u64 *work, bool unitShift, bool coefIn = false);

}  // namespace pil2gl
