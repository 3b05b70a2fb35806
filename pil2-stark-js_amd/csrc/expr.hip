// Expression (constraint / FRI / intermediate polynomial) evaluator, gfx950.
//
// Replaces src/prover/prover_helpers.js:23-259 (callCalculateExps -> calculateExps: the op-list is
// compiled to JavaScript and called once per row, :31-45,:83-107) and its worker variant
// (:360-546, src/prover/stark_prover_worker.js:6-44).  Op-list encoding: include/pil2gl_expr.h.
//
// One lane evaluates one row.  The program is wave-uniform (scalar loads, no divergence); the
// temporaries are renumbered on the host by live range so that they fit a small per-lane array.
#include "common.h"
#include "gl_field.cuh"
#include <vector>
#include <string.h>
#include <algorithm>

using namespace gl;

namespace {

// Device form of one op: everything the interpreter branches on is a 32-bit word (decoded once on the host).
struct DevRef { u32 kind_dim;    // kind | dim << 8
                u32 section;
                int32_t rowOff;  // prime << primeShift, already scaled
                u32 index; };
struct DevOp { u32 op; u32 pad_; DevRef dest, src[2]; };     // 56 bytes

#define GLX_MAX_SECTIONS 24
struct DevCtx {
    const DevOp *__restrict__ ops; u32 nOps;
    u32 nBits;
    const u64 *__restrict__ scalars;
    u64 *secPtr[GLX_MAX_SECTIONS];
    u32 secWidth[GLX_MAX_SECTIONS];
};

// Temporaries live in a [slot*3+component][lane] array (LDS when it fits, global scratch otherwise), so
// that slot numbers -- which are data of the program, not compile-time constants -- index memory and the
// access of a wave stays one contiguous row.
__device__ __forceinline__ void load_ref(const DevRef r, const DevCtx &c, u64 row, const u64 *T, u32 stride, u64 &v0, u64 &v1, u64 &v2) {
    const u32 kind = r.kind_dim & 0xff, dim = r.kind_dim >> 8;
    if (kind == GLX_TMP) {
        const u64 *t = T + (size_t)(3 * r.index) * stride;
        v0 = t[0];
        v1 = dim == 3 ? t[stride] : 0;
        v2 = dim == 3 ? t[2 * (size_t)stride] : 0;
    } else if (kind == GLX_SCALAR) {
        const u64 *p = c.scalars + r.index;
        v0 = p[0];
        v1 = dim == 3 ? p[1] : 0;
        v2 = dim == 3 ? p[2] : 0;
    } else {
        const u64 mask = (1ull << c.nBits) - 1;
        const u64 rr = (row + (u64)(int64_t)r.rowOff) & mask;                        // prover_helpers.js:220-233
        const u64 *p = c.secPtr[r.section] + rr * c.secWidth[r.section] + r.index;
        v0 = p[0];
        v1 = dim == 3 ? p[1] : 0;
        v2 = dim == 3 ? p[2] : 0;
    }
}

template <bool LDS_TMP>
__global__ void __launch_bounds__(256) eval_kernel(DevCtx c, u64 *gtmp) {
    extern __shared__ u64 lds_tmp[];
    u64 *T; u32 stride;
    if (LDS_TMP) { T = lds_tmp + threadIdx.x; stride = blockDim.x; }
    else { T = gtmp + (size_t)blockIdx.x * blockDim.x + threadIdx.x; stride = gridDim.x * blockDim.x; }
    const u64 nRows = 1ull << c.nBits;
    for (u64 row = (u64)blockIdx.x * blockDim.x + threadIdx.x; row < nRows; row += (u64)gridDim.x * blockDim.x) {
        for (u32 k = 0; k < c.nOps; k++) {
            const DevOp op = c.ops[k];
            u64 a0, a1, a2, b0 = 0, b1 = 0, b2 = 0, r0, r1, r2;
            const u32 da = op.src[0].kind_dim >> 8;
            u32 db = 1;
            load_ref(op.src[0], c, row, T, stride, a0, a1, a2);
            if (op.op != GLX_OP_COPY) { load_ref(op.src[1], c, row, T, stride, b0, b1, b2); db = op.src[1].kind_dim >> 8; }
            if (op.op == GLX_OP_ADD) {                                  // f3g.js:47-58
                r0 = add(a0, b0);
                if (da == 3 && db == 3) { r1 = add(a1, b1); r2 = add(a2, b2); }
                else if (da == 3) { r1 = a1; r2 = a2; } else { r1 = b1; r2 = b2; }
            } else if (op.op == GLX_OP_SUB) {                           // f3g.js:60-71
                r0 = sub(a0, b0);
                if (da == 3 && db == 3) { r1 = sub(a1, b1); r2 = sub(a2, b2); }
                else if (da == 3) { r1 = a1; r2 = a2; } else { r1 = neg(b1); r2 = neg(b2); }
            } else if (op.op == GLX_OP_MUL) {                           // f3g.js:82-103
                if (da == 3 && db == 3) { E3 x = { { a0, a1, a2 } }, y = { { b0, b1, b2 } }; E3 z = e3_mul(x, y); r0 = z.v[0]; r1 = z.v[1]; r2 = z.v[2]; }
                else if (da == 3) { r0 = mul(a0, b0); r1 = mul(a1, b0); r2 = mul(a2, b0); }
                else { r0 = mul(a0, b0); r1 = mul(a0, b1); r2 = mul(a0, b2); }
            } else { r0 = a0; r1 = a1; r2 = a2; }                       // copy
            const DevRef d = op.dest;
            if ((d.kind_dim & 0xff) == GLX_TMP) {
                u64 *t = T + (size_t)(3 * d.index) * stride;
                t[0] = r0; t[stride] = r1; t[2 * (size_t)stride] = r2;
            } else {
                const u64 mask = (1ull << c.nBits) - 1;
                const u64 rr = (row + (u64)(int64_t)d.rowOff) & mask;
                u64 *q = c.secPtr[d.section] + rr * c.secWidth[d.section] + d.index;
                q[0] = r0;
                if ((d.kind_dim >> 8) == 3) { q[1] = r1; q[2] = r2; }
            }
        }
    }
}

}  // namespace

using namespace pil2gl;

// Validates the op-list and renumbers its temporaries by live range (linear scan): code.tmpUsed counts one
// slot per op of the largest expression (codegen.js:83), far more than are ever live at once.
static int compact_program(const glx_program *prog, const glx_ctx *ctx, std::vector<glx_op> &ops, u32 &nSlots) {
    ops.assign(prog->ops, prog->ops + prog->nOps);
    const u32 NONE = 0xFFFFFFFFu;
    std::vector<u32> lastUse(prog->nTmp, NONE);
    for (u32 k = 0; k < prog->nOps; k++) {
        const glx_op &o = ops[k];
        if (o.op > GLX_OP_COPY) return fail(PIL2GL_EINVAL, "Invalid op: %u", o.op);      // prover_helpers.js:96
        const int ns = o.op == GLX_OP_COPY ? 1 : 2;
        for (int s = 0; s < ns + 1; s++) {
            const glx_ref &r = s < ns ? o.src[s] : o.dest;
            if (r.dim != 1 && r.dim != 3) return fail(PIL2GL_EINVAL, "invalid dim %u in op %u", r.dim, k);
            if (r.kind == GLX_TMP) { if (r.index >= prog->nTmp) return fail(PIL2GL_EINVAL, "tmp %u out of range in op %u", r.index, k); if (s < ns) lastUse[r.index] = k; }
            else if (r.kind == GLX_SEC) {
                if (!ctx) continue;
                if (r.section >= ctx->nSections) return fail(PIL2GL_EINVAL, "section %u out of range in op %u", r.section, k);
                if ((u64)r.index + r.dim > ctx->sections[r.section].width) return fail(PIL2GL_EINVAL, "column %u out of range in op %u", r.index, k);
            } else if (r.kind == GLX_SCALAR) {
                if (s == ns) return fail(PIL2GL_EINVAL, "Invalid reference type set");     // prover_helpers.js:148
                if (ctx && (u64)r.index + r.dim > ctx->nScalars) return fail(PIL2GL_EINVAL, "scalar %u out of range in op %u", r.index, k);
            } else return fail(PIL2GL_EINVAL, "Invalid reference type get");
        }
    }
    std::vector<u32> slotOf(prog->nTmp, NONE), freeSlots;
    nSlots = 0;
    for (u32 k = 0; k < prog->nOps; k++) {
        glx_op &o = ops[k];
        const int ns = o.op == GLX_OP_COPY ? 1 : 2;
        u32 dying[2]; int nd = 0;
        for (int s = 0; s < ns; s++) {
            glx_ref &r = o.src[s];
            if (r.kind != GLX_TMP) continue;
            const u32 id = r.index;
            if (slotOf[id] == NONE) return fail(PIL2GL_EINVAL, "tmp %u read before written in op %u", id, k);
            r.index = slotOf[id];
            if (lastUse[id] == k && !(nd == 1 && dying[0] == id)) dying[nd++] = id;
        }
        // a lane reads both sources before it writes the destination, so a dying source's slot may be reused at once
        for (int f = 0; f < nd; f++) { freeSlots.push_back(slotOf[dying[f]]); slotOf[dying[f]] = NONE; }
        if (o.dest.kind == GLX_TMP) {
            const u32 id = o.dest.index;
            if (slotOf[id] == NONE) {
                if (!freeSlots.empty()) { slotOf[id] = freeSlots.back(); freeSlots.pop_back(); }
                else slotOf[id] = nSlots++;
            }
            o.dest.index = slotOf[id];
            if (lastUse[id] == NONE || lastUse[id] <= k) { freeSlots.push_back(slotOf[id]); slotOf[id] = NONE; }   // never read again
        }
    }
    return PIL2GL_OK;
}

// test hook (host only, no GPU needed): the compacted op-list that the kernel would run
extern "C" int pil2gl_debug_compact_program(const glx_program *prog, glx_op *outOps, uint32_t *nSlots) {
    if (!prog || !outOps || !nSlots) return fail(PIL2GL_EINVAL, "null argument");
    std::vector<glx_op> ops; u32 n = 0;
    P2_TRY(compact_program(prog, nullptr, ops, n));
    memcpy(outOps, ops.data(), ops.size() * sizeof(glx_op));
    *nSlots = n;
    return PIL2GL_OK;
}

extern "C" int pil2gl_eval_program_dev(const glx_program *prog, const glx_ctx *ctx, void *stream) {
    P2_TRY(ensure_init());
    if (!prog || !ctx || (prog->nOps && !prog->ops)) return fail(PIL2GL_EINVAL, "null program");
    if (ctx->nBits > 31) return fail(PIL2GL_EINVAL, "domain too large");
    if (prog->nOps == 0) return PIL2GL_OK;
    hipStream_t st = as_stream(stream);
    std::vector<glx_op> ops; u32 nSlots = 0;
    P2_TRY(compact_program(prog, ctx, ops, nSlots));

    if (ctx->nSections > GLX_MAX_SECTIONS) return fail(PIL2GL_EINVAL, "too many sections (%u > %d)", ctx->nSections, GLX_MAX_SECTIONS);
    // device form of the program (scratch slot 4): ops, then the scalar pool
    std::vector<DevOp> dops(ops.size());
    for (size_t k = 0; k < ops.size(); k++) {
        auto cv = [&](const glx_ref &r) { DevRef d; d.kind_dim = (u32)r.kind | ((u32)r.dim << 8); d.section = r.section; d.rowOff = (int32_t)((int64_t)r.prime * ((int64_t)1 << ctx->primeShift)); d.index = r.index; return d; };
        dops[k].op = ops[k].op; dops[k].pad_ = 0; dops[k].dest = cv(ops[k].dest); dops[k].src[0] = cv(ops[k].src[0]); dops[k].src[1] = cv(ops[k].src[1]);
        for (int s = 0; s < 3; s++) { const glx_ref &r = s < 2 ? ops[k].src[s] : ops[k].dest; if ((int64_t)r.prime * ((int64_t)1 << ctx->primeShift) != (int64_t)(int32_t)((int64_t)r.prime * ((int64_t)1 << ctx->primeShift))) return fail(PIL2GL_EINVAL, "row offset overflow in op %zu", k); }
    }
    const u64 opsWords = ((u64)dops.size() * sizeof(DevOp) + 7) / 8;
    u64 *d;
    P2_TRY(scratch(4, opsWords + ctx->nScalars + 1, &d));
    HIP_TRY(hipMemcpyAsync(d, dops.data(), dops.size() * sizeof(DevOp), hipMemcpyHostToDevice, st));
    if (ctx->nScalars) HIP_TRY(hipMemcpyAsync(d + opsWords, ctx->scalars, (u64)ctx->nScalars * 8, hipMemcpyHostToDevice, st));
    DevCtx c;
    c.ops = (const DevOp *)d; c.nOps = (u32)dops.size();
    c.scalars = d + opsWords;
    c.nBits = ctx->nBits;
    for (u32 i = 0; i < GLX_MAX_SECTIONS; i++) { c.secPtr[i] = i < ctx->nSections ? ctx->sections[i].ptr : nullptr; c.secWidth[i] = i < ctx->nSections ? (u32)ctx->sections[i].width : 0; }
    for (u32 i = 0; i < ctx->nSections; i++) if (ctx->sections[i].width >> 32) return fail(PIL2GL_EINVAL, "section %u too wide", i);
    const u64 nRows = 1ull << ctx->nBits;
    const u32 slots = nSlots ? nSlots : 1;
    if ((size_t)slots * 3 * 64 * 8 <= 60 * 1024) {               // temporaries fit LDS at some block size
        u32 threads = 256;
        while ((size_t)slots * 3 * threads * 8 > 60 * 1024) threads /= 2;
        const size_t lds = (size_t)slots * 3 * threads * 8;
        HIP_TRY(hipFuncSetAttribute((const void *)eval_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        const u64 blocks = std::min<u64>((nRows + threads - 1) / threads, 256ull * 64);
        eval_kernel<true><<<(unsigned)blocks, threads, lds, st>>>(c, nullptr);
    } else {                                                     // spill to a [slot][lane] global array, persistent lanes
        const u32 threads = 256;
        const u64 blocks = std::min<u64>((nRows + threads - 1) / threads, 256ull * 8);
        u64 *gtmp;
        P2_TRY(scratch(5, (u64)slots * 3 * blocks * threads, &gtmp));
        eval_kernel<false><<<(unsigned)blocks, threads, 0, st>>>(c, gtmp);
    }
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(st));          // the staging copies above come from stack/heap buffers
    return PIL2GL_OK;
}
