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

using namespace gl;

namespace {

struct DevCtx {
    const glx_op *ops; u32 nOps;
    const glx_section *sections;
    const u64 *scalars;
    u32 nBits, primeShift;
};

__device__ __forceinline__ void load_ref(const glx_ref &r, const DevCtx &c, u64 row, const u64 *tmp, u64 v[3]) {
    const u64 *p;
    if (r.kind == GLX_TMP) p = tmp + 3 * r.index;
    else if (r.kind == GLX_SCALAR) p = c.scalars + r.index;
    else {
        const glx_section s = c.sections[r.section];
        const u64 mask = (1ull << c.nBits) - 1;
        const u64 rr = (row + (u64)((int64_t)r.prime << c.primeShift)) & mask;      // prover_helpers.js:220-233
        p = s.ptr + rr * s.width + r.index;
    }
    v[0] = p[0];
    if (r.dim == 3) { v[1] = p[1]; v[2] = p[2]; } else { v[1] = 0; v[2] = 0; }
}

template <int MAXT>
__global__ void __launch_bounds__(256) eval_kernel(DevCtx c) {
    const u64 row = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (1ull << c.nBits)) return;
    u64 tmp[3 * MAXT];
    for (u32 k = 0; k < c.nOps; k++) {
        const glx_op op = c.ops[k];
        u64 a[3], b[3] = { 0, 0, 0 }, r[3];
        const u32 da = op.src[0].dim;
        u32 db = 1;
        load_ref(op.src[0], c, row, tmp, a);
        if (op.op != GLX_OP_COPY) { load_ref(op.src[1], c, row, tmp, b); db = op.src[1].dim; }
        switch (op.op) {
        case GLX_OP_ADD:                                        // f3g.js:47-58
            r[0] = add(a[0], b[0]);
            if (da == 3 && db == 3) { r[1] = add(a[1], b[1]); r[2] = add(a[2], b[2]); }
            else if (da == 3) { r[1] = a[1]; r[2] = a[2]; } else { r[1] = b[1]; r[2] = b[2]; }
            break;
        case GLX_OP_SUB:                                        // f3g.js:60-71
            r[0] = sub(a[0], b[0]);
            if (da == 3 && db == 3) { r[1] = sub(a[1], b[1]); r[2] = sub(a[2], b[2]); }
            else if (da == 3) { r[1] = a[1]; r[2] = a[2]; } else { r[1] = neg(b[1]); r[2] = neg(b[2]); }
            break;
        case GLX_OP_MUL:                                        // f3g.js:82-103
            if (da == 3 && db == 3) { E3 x = { { a[0], a[1], a[2] } }, y = { { b[0], b[1], b[2] } }; E3 z = e3_mul(x, y); r[0] = z.v[0]; r[1] = z.v[1]; r[2] = z.v[2]; }
            else if (da == 3) { r[0] = mul(a[0], b[0]); r[1] = mul(a[1], b[0]); r[2] = mul(a[2], b[0]); }
            else { r[0] = mul(a[0], b[0]); r[1] = mul(a[0], b[1]); r[2] = mul(a[0], b[2]); }
            break;
        default: r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; break;  // copy
        }
        const glx_ref d = op.dest;
        if (d.kind == GLX_TMP) { tmp[3 * d.index] = r[0]; tmp[3 * d.index + 1] = r[1]; tmp[3 * d.index + 2] = r[2]; }
        else {
            const glx_section s = c.sections[d.section];
            const u64 mask = (1ull << c.nBits) - 1;
            const u64 rr = (row + (u64)((int64_t)d.prime << c.primeShift)) & mask;
            u64 *q = s.ptr + rr * s.width + d.index;
            q[0] = r[0];
            if (d.dim == 3) { q[1] = r[1]; q[2] = r[2]; }
        }
    }
}

}  // namespace

using namespace pil2gl;

extern "C" int pil2gl_eval_program_dev(const glx_program *prog, const glx_ctx *ctx, void *stream) {
    P2_TRY(ensure_init());
    if (!prog || !ctx || (prog->nOps && !prog->ops)) return fail(PIL2GL_EINVAL, "null program");
    if (ctx->nBits > 31) return fail(PIL2GL_EINVAL, "domain too large");
    if (prog->nOps == 0) return PIL2GL_OK;
    hipStream_t st = as_stream(stream);

    // validate and renumber temporaries by live range (linear scan): code.tmpUsed counts one slot per
    // op of the largest expression (codegen.js:83), far more than are ever live at once
    std::vector<glx_op> ops(prog->ops, prog->ops + prog->nOps);
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
                if (r.section >= ctx->nSections) return fail(PIL2GL_EINVAL, "section %u out of range in op %u", r.section, k);
                if ((u64)r.index + r.dim > ctx->sections[r.section].width) return fail(PIL2GL_EINVAL, "column %u out of range in op %u", r.index, k);
            } else if (r.kind == GLX_SCALAR) {
                if (s == ns) return fail(PIL2GL_EINVAL, "Invalid reference type set");     // prover_helpers.js:148
                if ((u64)r.index + r.dim > ctx->nScalars) return fail(PIL2GL_EINVAL, "scalar %u out of range in op %u", r.index, k);
            } else return fail(PIL2GL_EINVAL, "Invalid reference type get");
        }
    }
    std::vector<u32> slotOf(prog->nTmp, NONE), freeSlots;
    u32 nSlots = 0;
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

    // device copies of ops / sections / scalars (scratch slot 4, laid out back to back)
    const u64 opsWords = ((u64)ops.size() * sizeof(glx_op) + 7) / 8;
    const u64 secWords = ((u64)ctx->nSections * sizeof(glx_section) + 7) / 8;
    u64 *d;
    P2_TRY(scratch(4, opsWords + secWords + ctx->nScalars + 1, &d));
    HIP_TRY(hipMemcpyAsync(d, ops.data(), ops.size() * sizeof(glx_op), hipMemcpyHostToDevice, st));
    if (ctx->nSections) HIP_TRY(hipMemcpyAsync(d + opsWords, ctx->sections, ctx->nSections * sizeof(glx_section), hipMemcpyHostToDevice, st));
    if (ctx->nScalars) HIP_TRY(hipMemcpyAsync(d + opsWords + secWords, ctx->scalars, (u64)ctx->nScalars * 8, hipMemcpyHostToDevice, st));
    DevCtx c;
    c.ops = (const glx_op *)d; c.nOps = (u32)ops.size();
    c.sections = (const glx_section *)(d + opsWords);
    c.scalars = d + opsWords + secWords;
    c.nBits = ctx->nBits; c.primeShift = ctx->primeShift;
    const unsigned blocks = (unsigned)(((1ull << ctx->nBits) + 255) / 256);
    if (nSlots <= 8) eval_kernel<8><<<blocks, 256, 0, st>>>(c);
    else if (nSlots <= 32) eval_kernel<32><<<blocks, 256, 0, st>>>(c);
    else if (nSlots <= 128) eval_kernel<128><<<blocks, 256, 0, st>>>(c);
    else if (nSlots <= 1024) eval_kernel<1024><<<blocks, 256, 0, st>>>(c);
    else return fail(PIL2GL_EINVAL, "program needs %u live temporaries (max 1024)", nSlots);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(st));          // the staging copies above come from stack/heap buffers
    return PIL2GL_OK;
}
