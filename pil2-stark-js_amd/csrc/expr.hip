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
#include <array>
#include <map>
#include <stdlib.h>

using namespace gl;

namespace {

// Device form of one op: everything the interpreter branches on is a 32-bit word (decoded once on the host).
struct DevRef { u32 kind_dim;    // kind | dim << 8
                u32 section;
                int32_t rowOff;  // prime << primeShift, already scaled
                u32 index; };
struct DevOp { u32 op; u32 aux; DevRef dest, src[2]; };      // 56 bytes; aux: limb-pool offset of a GLX_LZ_MAD weight

// internal ops produced by the host-side optimiser (never part of the ABI):
// a Horner chain  t = X*t + c_i  over an extension constant X is evaluated as sum_i c_i * X^(n-i) with the lazy
// 22-bit-limb accumulation of csrc/dot.hip: LZ_BEGIN zeroes the 3x6 partial sums, LZ_MAD adds value * weight
// (the weight's limbs come from the limb pool), LZ_END folds them into the destination.
enum { GLX_LZ_BEGIN = 4, GLX_LZ_MAD = 5, GLX_LZ_END = 6 };

#define GLX_MAX_SECTIONS 24
struct DevCtx {
    const DevOp *__restrict__ ops; u32 nOps;
    u32 nBits;
    const u64 *__restrict__ scalars;
    const u32 *__restrict__ limbs;
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

__device__ __forceinline__ u64 fold6(const u64 S[6]) {       // see csrc/dot.hip
    u64 r = canon(S[0]);
    r = add(r, mul(S[1], 1ull << 22));
    r = add(r, mul(S[2], 1ull << 44));
    r = add(r, mul(S[3], 1ull << 32));
    r = add(r, mul(S[4], 1ull << 54));
    r = add(r, mul(S[5], ((1ull << 32) - 1) << 12));
    return r;
}
__device__ __forceinline__ void lz_mad(u64 S[3][6], u64 v, const u32 *__restrict__ L) {
    const u32 p0 = (u32)v, p1 = (u32)(v >> 32);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const u32 w0 = L[3 * k], w1 = L[3 * k + 1], w2 = L[3 * k + 2];
        S[k][0] += (u64)p0 * w0; S[k][1] += (u64)p0 * w1; S[k][2] += (u64)p0 * w2;
        S[k][3] += (u64)p1 * w0; S[k][4] += (u64)p1 * w1; S[k][5] += (u64)p1 * w2;
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
        u64 LZ[3][6];
        for (u32 k = 0; k < c.nOps; k++) {
            const DevOp op = c.ops[k];
            u64 a0, a1, a2, b0 = 0, b1 = 0, b2 = 0, r0, r1, r2;
            if (op.op == GLX_LZ_BEGIN) {
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int i = 0; i < 6; i++) LZ[q][i] = 0;
                continue;
            }
            if (op.op == GLX_LZ_MAD) {
                load_ref(op.src[0], c, row, T, stride, a0, a1, a2);
                const u32 *L = c.limbs + op.aux;
                lz_mad(LZ, a0, L);
                if ((op.src[0].kind_dim >> 8) == 3) { lz_mad(LZ, a1, L + 9); lz_mad(LZ, a2, L + 18); }
                continue;
            }
            const u32 da = op.src[0].kind_dim >> 8;
            u32 db = 1;
            if (op.op != GLX_LZ_END) load_ref(op.src[0], c, row, T, stride, a0, a1, a2);
            if (op.op < GLX_OP_COPY) { load_ref(op.src[1], c, row, T, stride, b0, b1, b2); db = op.src[1].kind_dim >> 8; }
            if (op.op == GLX_LZ_END) { r0 = fold6(LZ[0]); r1 = fold6(LZ[1]); r2 = fold6(LZ[2]); }
            else if (op.op == GLX_OP_ADD) {                             // f3g.js:47-58
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

// ------------------------------------------------------------------------------------------------ host-side passes
struct IOp { u32 op; u32 aux; glx_ref dest; glx_ref src[2]; };
static inline int n_src(u32 op) { return op < GLX_OP_COPY ? 2 : (op == GLX_OP_COPY || op == GLX_LZ_MAD) ? 1 : 0; }
static inline bool has_dest(u32 op) { return op <= GLX_OP_COPY || op == GLX_LZ_END; }

static int validate_program(const glx_program *prog, const glx_ctx *ctx) {
    for (u32 k = 0; k < prog->nOps; k++) {
        const glx_op &o = prog->ops[k];
        if (o.op > GLX_OP_COPY) return fail(PIL2GL_EINVAL, "Invalid op: %u", o.op);      // prover_helpers.js:96
        const int ns = o.op == GLX_OP_COPY ? 1 : 2;
        for (int s = 0; s < ns + 1; s++) {
            const glx_ref &r = s < ns ? o.src[s] : o.dest;
            if (r.dim != 1 && r.dim != 3) return fail(PIL2GL_EINVAL, "invalid dim %u in op %u", r.dim, k);
            if (r.kind == GLX_TMP) { if (r.index >= prog->nTmp) return fail(PIL2GL_EINVAL, "tmp %u out of range in op %u", r.index, k); }
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
    return PIL2GL_OK;
}

// Pass 1: value numbering.  Every distinct (section, column, row offset) operand is loaded once into a temporary
// (the reference's op-lists re-read the same cell for every use, and every use is a strided, uncoalesced access
// for a lane-per-row evaluator), identical sub-expressions are computed once, copies between temporaries vanish.
// The output is in single-assignment form: temporary ids are value numbers.
static int value_number(const glx_program *prog, const glx_ctx *ctx, std::vector<IOp> &out, u32 &nVal) {
    std::map<std::array<u64, 4>, u32> scalarIndex;  // (dim, words) -> first pool offset holding that constant
    struct Key { u32 op; u64 a, b; bool operator<(const Key &o) const { return op != o.op ? op < o.op : a != o.a ? a < o.a : b < o.b; } };
    std::map<Key, u32> table;                       // expression -> value number
    std::map<std::pair<u64, u64>, u32> loads;       // (section|dim|prime, column) -> value number
    std::vector<u32> valDim;                        // dim of each value
    const u32 NONE = 0xFFFFFFFFu;
    std::vector<u32> cur(prog->nTmp, NONE);         // value currently held by each original tmp id
    auto tmp_ref = [&](u32 v) { glx_ref r; memset(&r, 0, sizeof r); r.kind = GLX_TMP; r.dim = (uint8_t)valDim[v]; r.index = v; return r; };
    auto operand_key = [](const glx_ref &r) { return ((u64)r.kind << 60) | ((u64)r.dim << 56) | (u64)r.index; };
    out.clear();
    for (u32 k = 0; k < prog->nOps; k++) {
        const glx_op &o = prog->ops[k];
        const int ns = o.op == GLX_OP_COPY ? 1 : 2;
        glx_ref src[2]; memset(src, 0, sizeof src);
        for (int s = 0; s < ns; s++) {
            const glx_ref &r = o.src[s];
            if (r.kind == GLX_TMP) {
                if (cur[r.index] == NONE) return fail(PIL2GL_EINVAL, "tmp %u read before written in op %u", r.index, k);
                src[s] = tmp_ref(cur[r.index]);
                src[s].dim = r.dim < src[s].dim ? r.dim : src[s].dim;       // a dim-1 read of a triple sees component 0 only
            } else if (r.kind == GLX_SEC) {
                const std::pair<u64, u64> lk(((u64)r.section << 40) | ((u64)r.dim << 32) | (u32)r.prime, r.index);
                auto it = loads.find(lk);
                if (it == loads.end()) {
                    const u32 v = (u32)valDim.size(); valDim.push_back(r.dim);
                    IOp ld; memset(&ld, 0, sizeof ld); ld.op = GLX_OP_COPY; ld.dest = tmp_ref(v); ld.src[0] = r;
                    out.push_back(ld);
                    it = loads.emplace(lk, v).first;
                }
                src[s] = tmp_ref(it->second);
            } else {                                                         // scalar pool: wave-uniform, read in place
                src[s] = r;
                if (ctx && ctx->scalars) {                                   // encoders append one pool entry per use: merge equal constants
                    const u64 *w = ctx->scalars + r.index;
                    const std::array<u64, 4> key = { r.dim, w[0], r.dim == 3 ? w[1] : 0, r.dim == 3 ? w[2] : 0 };
                    src[s].index = scalarIndex.emplace(key, r.index).first->second;
                }
            }
        }
        if (o.dest.kind == GLX_SEC) {                                        // stores are never merged; they invalidate loads
            IOp st; memset(&st, 0, sizeof st); st.op = o.op; st.dest = o.dest; st.src[0] = src[0]; st.src[1] = src[1];
            out.push_back(st);
            for (auto it = loads.begin(); it != loads.end();) { if ((it->first.first >> 40) == o.dest.section) it = loads.erase(it); else ++it; }
            continue;
        }
        if (o.op == GLX_OP_COPY && src[0].kind == GLX_TMP && src[0].dim == o.dest.dim) { cur[o.dest.index] = src[0].index; continue; }
        Key key = { o.op | ((u32)o.dest.dim << 8), operand_key(src[0]), ns == 2 ? operand_key(src[1]) : 0 };
        if ((o.op == GLX_OP_ADD || o.op == GLX_OP_MUL) && key.b < key.a) std::swap(key.a, key.b);
        auto it = table.find(key);
        if (it == table.end()) {
            const u32 v = (u32)valDim.size(); valDim.push_back(o.dest.dim);
            IOp n; memset(&n, 0, sizeof n); n.op = o.op; n.dest = tmp_ref(v); n.src[0] = src[0]; n.src[1] = src[1];
            out.push_back(n);
            it = table.emplace(key, v).first;
        }
        cur[o.dest.index] = it->second;
    }
    nVal = (u32)valDim.size();
    return PIL2GL_OK;
}

static void push_limbs(std::vector<u32> &pool, u64 v) {
    pool.push_back((u32)v & 0x3FFFFF); pool.push_back((u32)(v >> 22) & 0x3FFFFF); pool.push_back((u32)(v >> 44));
}

// Pass 2: Horner chains  t_i = X * t_(i-1) + c_i  over a scalar-pool extension constant X  ->  lazy dot product
// sum_i c_i X^(n-i)  (same field element, see the kernel comment).  Chains must not interleave (one accumulator set).
static void fuse_horner(std::vector<IOp> &ops, u32 nVal, const glx_ctx *ctx, std::vector<u32> &limbPool) {
    const u32 NONE = 0xFFFFFFFFu;
    std::vector<u32> def(nVal, NONE), uses(nVal, 0);
    for (u32 k = 0; k < ops.size(); k++) {
        for (int s = 0; s < n_src(ops[k].op); s++) if (ops[k].src[s].kind == GLX_TMP) uses[ops[k].src[s].index]++;
        if (has_dest(ops[k].op) && ops[k].dest.kind == GLX_TMP) def[ops[k].dest.index] = k;
    }
    struct Link { u32 mulIdx, addIdx; glx_ref X, A, C; };
    // link ending at ADD k: one operand is a single-use MUL(scalar dim3, value)
    auto link_at = [&](u32 k, Link &L) {
        const IOp &a = ops[k];
        if (a.op != GLX_OP_ADD || a.dest.dim != 3) return false;
        for (int s = 0; s < 2; s++) {
            const glx_ref &m = a.src[s];
            if (m.kind != GLX_TMP || uses[m.index] != 1 || def[m.index] == NONE) continue;
            const IOp &mu = ops[def[m.index]];
            if (mu.op != GLX_OP_MUL) continue;
            for (int t = 0; t < 2; t++) {
                if (mu.src[t].kind == GLX_SCALAR && mu.src[t].dim == 3 && mu.src[1 - t].kind != GLX_SEC) {
                    L.mulIdx = def[m.index]; L.addIdx = k; L.X = mu.src[t]; L.A = mu.src[1 - t]; L.C = a.src[1 - s];
                    return L.C.kind != GLX_SEC;
                }
            }
        }
        return false;
    };
    std::vector<bool> removed(ops.size(), false);
    std::vector<std::vector<IOp>> insertAt(ops.size());      // replacement ops for position k
    u32 busyUntil = 0;                                       // chains may not overlap in program order
    std::vector<bool> consumed(ops.size(), false);
    for (u32 k = (u32)ops.size(); k-- > 0;) {               // find chain ends from the back
        Link L;
        if (consumed[k] || !link_at(k, L)) continue;
        // is this ADD's result the A of a later link? then it is not an end (it will be reached from that end)
        std::vector<Link> chain; chain.push_back(L);
        for (;;) {                                           // walk towards the start
            const glx_ref &A = chain.back().A;
            if (A.kind != GLX_TMP || uses[A.index] != 1 || def[A.index] == NONE) break;
            Link P2;
            if (!link_at(def[A.index], P2) || P2.X.index != L.X.index) break;
            chain.push_back(P2);
        }
        const u32 n = (u32)chain.size();
        if (n < 4 || n > 300) continue;                     // <= 1024 terms of < 2^54 per partial sum (3 per triple)
        const u32 first = chain.back().mulIdx, last = chain.front().addIdx;
        if (busyUntil && last >= busyUntil) continue;        // overlaps a chain already taken (we scan backwards)
        bool ordered = true;
        for (u32 i = 0; i + 1 < n; i++) ordered &= chain[i + 1].addIdx < chain[i].mulIdx;
        if (!ordered) continue;
        busyUntil = first;
        // weights X^0 .. X^n
        const u64 *xp = ctx->scalars + L.X.index;
        std::vector<std::array<u64, 3>> pw(n + 1);
        pw[0] = { 1, 0, 0 };
        for (u32 i = 1; i <= n; i++) { u64 t[3]; h_e3_mul(pw[i - 1].data(), xp, t); pw[i] = { t[0], t[1], t[2] }; }
        auto weight = [&](const std::array<u64, 3> &W, u32 dim) {
            const u32 off = (u32)limbPool.size();
            std::array<u64, 3> w = W;
            for (u32 t = 0; t < dim; t++) {
                for (int q = 0; q < 3; q++) push_limbs(limbPool, w[q]);
                w = { w[2], h_add(w[0], w[2]), w[1] };      // times x  (x^3 = x + 1)
            }
            return off;
        };
        auto mad = [&](const glx_ref &v, const std::array<u64, 3> &W) {
            IOp m; memset(&m, 0, sizeof m); m.op = GLX_LZ_MAD; m.src[0] = v; m.aux = weight(W, v.dim); return m;
        };
        // chain[n-1] is the innermost link: its A is t_0 (weight X^n), its C is c_1 (weight X^(n-1)), ...
        for (u32 i = 0; i < n; i++) { removed[chain[i].mulIdx] = true; removed[chain[i].addIdx] = true; consumed[chain[i].addIdx] = true; }
        IOp bg; memset(&bg, 0, sizeof bg); bg.op = GLX_LZ_BEGIN;
        insertAt[first].push_back(bg);
        insertAt[first].push_back(mad(chain[n - 1].A, pw[n]));
        for (u32 i = 0; i < n; i++) {
            const Link &lk = chain[n - 1 - i];               // link i+1 in program order
            insertAt[lk.addIdx].push_back(mad(lk.C, pw[n - 1 - i]));
        }
        IOp en; memset(&en, 0, sizeof en); en.op = GLX_LZ_END; en.dest = ops[last].dest;
        insertAt[last].push_back(en);
    }
    std::vector<IOp> res;
    for (u32 k = 0; k < ops.size(); k++) {
        for (const IOp &i : insertAt[k]) res.push_back(i);
        if (!removed[k]) res.push_back(ops[k]);
    }
    ops.swap(res);
}

// Pass 3: renumber temporaries by live range (linear scan); the input is single-assignment.
static int allocate_slots(std::vector<IOp> &ops, u32 nVal, u32 &nSlots) {
    const u32 NONE = 0xFFFFFFFFu;
    std::vector<u32> lastUse(nVal, NONE);
    for (u32 k = 0; k < ops.size(); k++)
        for (int s = 0; s < n_src(ops[k].op); s++) if (ops[k].src[s].kind == GLX_TMP) lastUse[ops[k].src[s].index] = k;
    std::vector<u32> slotOf(nVal, NONE), freeSlots;
    nSlots = 0;
    for (u32 k = 0; k < ops.size(); k++) {
        IOp &o = ops[k];
        u32 dying[2]; int nd = 0;
        for (int s = 0; s < n_src(o.op); s++) {
            glx_ref &r = o.src[s];
            if (r.kind != GLX_TMP) continue;
            const u32 id = r.index;
            if (slotOf[id] == NONE) return fail(PIL2GL_EINVAL, "internal: value %u used before definition", id);
            r.index = slotOf[id];
            if (lastUse[id] == k && !(nd == 1 && dying[0] == id)) dying[nd++] = id;
        }
        // a lane reads both sources before it writes the destination, so a dying source's slot may be reused at once
        for (int f = 0; f < nd; f++) { freeSlots.push_back(slotOf[dying[f]]); slotOf[dying[f]] = NONE; }
        if (has_dest(o.op) && o.dest.kind == GLX_TMP) {
            const u32 id = o.dest.index;
            if (!freeSlots.empty()) { slotOf[id] = freeSlots.back(); freeSlots.pop_back(); }
            else slotOf[id] = nSlots++;
            o.dest.index = slotOf[id];
            if (lastUse[id] == NONE) { freeSlots.push_back(slotOf[id]); slotOf[id] = NONE; }   // never read
        }
    }
    return PIL2GL_OK;
}

static int compile_program(const glx_program *prog, const glx_ctx *ctx, std::vector<IOp> &ops, u32 &nSlots, std::vector<u32> &limbPool, bool fuse) {
    P2_TRY(validate_program(prog, ctx));
    u32 nVal = 0;
    P2_TRY(value_number(prog, ctx, ops, nVal));
    if (fuse && ctx && ctx->scalars) fuse_horner(ops, nVal, ctx, limbPool);
    return allocate_slots(ops, nVal, nSlots);
}

// test hook (host only, no GPU needed): the program after value numbering and slot allocation (no Horner fusion),
// in the public op encoding, as the kernel would run it
extern "C" int pil2gl_debug_compact_program(const glx_program *prog, glx_op *outOps, uint32_t *nSlots /* [2]: slots, ops */) {
    if (!prog || !outOps || !nSlots) return fail(PIL2GL_EINVAL, "null argument");
    std::vector<IOp> ops; std::vector<u32> pool; u32 n = 0;
    P2_TRY(compile_program(prog, nullptr, ops, n, pool, false));
    if (ops.size() > 2ull * prog->nOps + 16) return fail(PIL2GL_EINVAL, "internal: program grew unexpectedly");
    for (size_t k = 0; k < ops.size(); k++) { memset(&outOps[k], 0, sizeof(glx_op)); outOps[k].op = ops[k].op; outOps[k].dest = ops[k].dest; outOps[k].src[0] = ops[k].src[0]; outOps[k].src[1] = ops[k].src[1]; }
    nSlots[0] = n; nSlots[1] = (uint32_t)ops.size();
    return PIL2GL_OK;
}

// ------------------------------------------------------------------------------------------------ run-time compilation
// The reference turns the op-list into JavaScript source and calls `new Function` on it (prover_helpers.js:31-45,
// :83-107).  The same is done here with HIP source and hiprtc: the optimised op-list becomes one straight-line
// kernel (temporaries in registers, dims and column offsets resolved statically), compiled once per distinct
// program and cached.  Long programs on large domains take this path; short ones use the interpreter above.
#include <hip/hiprtc.h>
#include <sstream>
#include <set>
#include <string>

static const char *kFieldSrc =
#include "gl_field_src.inc"
    ;

struct JitArgs {            // kernel argument block (by value)
    const u64 *scalars; const u32 *limbs; u64 *sec[GLX_MAX_SECTIONS]; u32 nBits; u32 pad_;
};
struct JitEntry { hipModule_t mod; hipFunction_t fn; };
static std::map<std::string, JitEntry> g_jit_cache;       // modules of the device the library is initialised on (cleared by pil2gl_shutdown)
namespace pil2gl {
void jit_clear() {
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    for (auto &kv : g_jit_cache) (void)hipModuleUnload(kv.second.mod);
    g_jit_cache.clear();
}
}

static std::string jit_source(const std::vector<IOp> &ops, u32 nSlots, const glx_ctx *ctx) {
    std::ostringstream o;
    // no #include: hiprtc pre-includes its built-in device runtime header; only the fixed-width typedefs are needed
    o << "typedef unsigned long uint64_t; typedef unsigned int uint32_t; typedef long int64_t; typedef int int32_t;\n" << kFieldSrc << "\nusing namespace gl;\n";
    o << "struct JitArgs { const u64 *scalars; const u32 *limbs; u64 *sec[" << GLX_MAX_SECTIONS << "]; u32 nBits; u32 pad_; };\n";
    o << "__device__ __forceinline__ u64 fold6(const u64 S[6]) { u64 r = canon(S[0]); r = add(r, mul(S[1], 1ull << 22)); r = add(r, mul(S[2], 1ull << 44));"
         " r = add(r, mul(S[3], 1ull << 32)); r = add(r, mul(S[4], 1ull << 54)); r = add(r, mul(S[5], ((1ull << 32) - 1) << 12)); return r; }\n";
    o << "__device__ __forceinline__ void lz_mad(u64 S[3][6], u64 v, const u32 *__restrict__ L) { const u32 p0 = (u32)v, p1 = (u32)(v >> 32);\n"
         " _Pragma(\"unroll\") for (int k = 0; k < 3; k++) { const u32 w0 = L[3*k], w1 = L[3*k+1], w2 = L[3*k+2];\n"
         "  S[k][0] += (u64)p0 * w0; S[k][1] += (u64)p0 * w1; S[k][2] += (u64)p0 * w2; S[k][3] += (u64)p1 * w0; S[k][4] += (u64)p1 * w1; S[k][5] += (u64)p1 * w2; } }\n";
    o << "__device__ __noinline__ E3 e3_mul_call(E3 a, E3 b) { return e3_mul(a, b); }\n";
    // Long programs call their modular multiplications instead of inlining them: the straight-line kernel of the 666-op
    // constraint program of the bench AIR is 65 KB of code with inlined products, more than the 64 KB instruction cache two CUs
    // share, and every wave walks through all of it once; with calls it is 34 KB and the kernel 14 % faster (48.8 -> 42.1 ms
    // at config 3).  Short programs keep the inlined form.  PIL2GL_EXPR_MULCALL=0|1 overrides.
    const char *mcEnv = getenv("PIL2GL_EXPR_MULCALL");
    const bool mulCall = mcEnv ? mcEnv[0] == '1' : ops.size() >= 200;
    // The products are the carry-out form of gl_field.cuh (mul_lazy_x: 15 instructions, any representatives in, a lazy one out).
    // A product whose every reader is another product or a lazy multiply-accumulate stays lazy; one that is added, subtracted,
    // copied or stored is made canonical (four more instructions).  PIL2GL_EXPR_LAZYMUL=0: every product canonical, hipcc's form.
    const char *lzEnv = getenv("PIL2GL_EXPR_LAZYMUL");
    const bool lazyMul = !(lzEnv && lzEnv[0] == '0');
    // (only in the called form: inlined, the asm products are slower than hipcc's own -- 59 against 39 ms for the config-3 constraint
    // program, 60 against 49 for the permutation AIR's -- the scheduler can no longer interleave them with the loads around them)
    const std::string MULC = mulCall ? "mul_call(" : "mul(", MULC_END = ")";
    const std::string MULL = (mulCall && lazyMul) ? "mull_call(" : MULC, MULL_END = ")";
    if (mulCall) {
        o << "__device__ __noinline__ u64 mul_call(u64 a, u64 b) { return " << (lazyMul ? "canon(mul_lazy_x(a, b))" : "mul(a, b)") << "; }\n";
        if (lazyMul) o << "__device__ __noinline__ u64 mull_call(u64 a, u64 b) { return mul_lazy_x(a, b); }\n";
    }
    // lazyOK[k]: op k writes a temporary that only products and multiply-accumulates read before the slot is written again
    std::vector<bool> lazyOK(ops.size(), false);
    for (size_t k = 0; k < ops.size(); k++) {
        const IOp &p = ops[k];
        if (p.op != GLX_OP_MUL || p.dest.kind != GLX_TMP || (p.src[0].dim == 3 && p.src[1].dim == 3)) continue;
        bool ok = true, live = true;
        for (size_t j = k + 1; j < ops.size() && ok && live; j++) {
            const IOp &q = ops[j];
            for (int t = 0; t < n_src(q.op); t++) {
                if (q.src[t].kind != GLX_TMP || q.src[t].index != p.dest.index) continue;
                const bool tolerant = q.op == GLX_LZ_MAD || (q.op == GLX_OP_MUL && !(q.src[0].dim == 3 && q.src[1].dim == 3));
                ok &= tolerant;
            }
            if (has_dest(q.op) && q.dest.kind == GLX_TMP && q.dest.index == p.dest.index) live = false;
        }
        lazyOK[k] = ok;
    }
    // (Reads through LDS tiles -- synchronous 16-column tiles, direct-to-LDS 8-column tiles filled ahead -- and loads grouped by line
    // between scheduling fences were built, were bit-exact, and lost (48.5 / 47.0 / 39.5-47.9 ms against 39.2-39.6): LAB_NOTES.md 9.5, 10.)
    auto row_off = [&](const glx_ref &r) { return (int64_t)r.prime * ((int64_t)1 << ctx->primeShift); };
    o << "extern \"C\" __global__ void __launch_bounds__(256) jit_eval(JitArgs A) {\n";
    o << " const u64 row = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (row >= (1ull << A.nBits)) return;\n";
    o << " const u64 mask = (1ull << A.nBits) - 1; const u64 *__restrict__ SC = A.scalars; const u32 *__restrict__ LM = A.limbs;\n";
    o << " u64 LZ[3][6];\n";
    for (u32 s = 0; s < nSlots; s++) o << " u64 t" << s << "_0 = 0, t" << s << "_1 = 0, t" << s << "_2 = 0;\n";
    auto addr = [&](const glx_ref &r) {
        std::ostringstream a;
        a << "A.sec[" << r.section << "] + ((row + (u64)(" << row_off(r) << "ll)) & mask) * " << ctx->sections[r.section].width << "ull + " << r.index;
        return a.str();
    };
    auto rd = [&](const glx_ref &r, int c) {        // component c of an operand (0 beyond its dim)
        std::ostringstream a;
        if (c >= (int)r.dim) { a << "0ull"; return a.str(); }
        if (r.kind == GLX_TMP) a << "t" << r.index << "_" << c;
        else if (r.kind == GLX_SCALAR) a << "SC[" << (r.index + c) << "]";
        else a << "(" << addr(r) << ")[" << c << "]";
        return a.str();
    };
    for (size_t k = 0; k < ops.size(); k++) {
        const IOp &p = ops[k];
        if (p.op == GLX_LZ_BEGIN) { o << " for (int q = 0; q < 3; q++) for (int i = 0; i < 6; i++) LZ[q][i] = 0;\n"; continue; }
        if (p.op == GLX_LZ_MAD) {
            o << " lz_mad(LZ, " << rd(p.src[0], 0) << ", LM + " << p.aux << ");";
            if (p.src[0].dim == 3) o << " lz_mad(LZ, " << rd(p.src[0], 1) << ", LM + " << p.aux + 9 << "); lz_mad(LZ, " << rd(p.src[0], 2) << ", LM + " << p.aux + 18 << ");";
            o << "\n"; continue;
        }
        const glx_ref &a = p.src[0], &b = p.src[1];
        std::string r[3];
        const u32 da = a.dim, db = b.dim;
        switch (p.op) {
        case GLX_LZ_END: r[0] = "fold6(LZ[0])"; r[1] = "fold6(LZ[1])"; r[2] = "fold6(LZ[2])"; break;
        case GLX_OP_ADD:
            r[0] = "add(" + rd(a, 0) + ", " + rd(b, 0) + ")";
            for (int c = 1; c < 3; c++) r[c] = (da == 3 && db == 3) ? "add(" + rd(a, c) + ", " + rd(b, c) + ")" : da == 3 ? rd(a, c) : rd(b, c);
            break;
        case GLX_OP_SUB:
            r[0] = "sub(" + rd(a, 0) + ", " + rd(b, 0) + ")";
            for (int c = 1; c < 3; c++) r[c] = (da == 3 && db == 3) ? "sub(" + rd(a, c) + ", " + rd(b, c) + ")" : da == 3 ? rd(a, c) : "neg(" + rd(b, c) + ")";
            break;
        case GLX_OP_MUL:
            if (da == 3 && db == 3) {
                o << " { E3 x_ = { { " << rd(a, 0) << ", " << rd(a, 1) << ", " << rd(a, 2) << " } }, y_ = { { " << rd(b, 0) << ", " << rd(b, 1) << ", " << rd(b, 2) << " } }; E3 z_ = e3_mul_call(x_, y_);";
                r[0] = "z_.v[0]"; r[1] = "z_.v[1]"; r[2] = "z_.v[2]";
            } else {
                const std::string &M = lazyOK[k] ? MULL : MULC, &ME = lazyOK[k] ? MULL_END : MULC_END;
                if (da == 3) { for (int c = 0; c < 3; c++) r[c] = M + rd(a, c) + ", " + rd(b, 0) + ME; }
                else { for (int c = 0; c < 3; c++) r[c] = c < (int)db || c == 0 ? M + rd(a, 0) + ", " + rd(b, c) + ME : "0ull"; }
            }
            break;
        default: for (int c = 0; c < 3; c++) r[c] = rd(a, c); break;     // copy
        }
        const glx_ref &d = p.dest;
        const bool braced = (p.op == GLX_OP_MUL && da == 3 && db == 3);
        if (!braced) o << " {";
        if (d.kind == GLX_TMP) {
            o << " const u64 n0_ = " << r[0] << ", n1_ = " << (d.dim == 3 ? r[1] : "0ull") << ", n2_ = " << (d.dim == 3 ? r[2] : "0ull") << ";";
            o << " t" << d.index << "_0 = n0_; t" << d.index << "_1 = n1_; t" << d.index << "_2 = n2_; }\n";
        } else {
            o << " u64 *q_ = " << addr(d) << "; q_[0] = " << r[0] << ";";
            if (d.dim == 3) o << " q_[1] = " << r[1] << "; q_[2] = " << r[2] << ";";
            o << " }\n";
        }
    }
    o << "}\n";
    return o.str();
}

// target of the run-time compiled kernels: the device the library is initialised on (gfx950 when no device is present:
// the host-only compile hook of the CPU tests)
static std::string jit_arch() {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.gcnArchName[0]) {
        std::string a(pr.gcnArchName);
        return a.substr(0, a.find(':'));
    }
    (void)hipGetLastError();
    return "gfx950";
}

static int jit_build(const std::string &src, std::vector<char> &code) {
    if (const char *dump = getenv("PIL2GL_EXPR_DUMP")) {        // debugging aid: the generated kernel's source, last program compiled
        if (FILE *f = fopen(dump, "w")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "pil2gl_expr.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return fail(PIL2GL_EHIP, "hiprtcCreateProgram failed");
    const std::string arch = "--offload-arch=" + jit_arch();
    const char *opts[] = { arch.c_str(), "-O3", "-ffp-contract=off" };
    hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, 0); if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(PIL2GL_EHIP, "hiprtc compile failed: %.300s", log.c_str());
    }
    size_t sz = 0; hiprtcGetCodeSize(prog, &sz);
    code.resize(sz); hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return PIL2GL_OK;
}
static int jit_get(const std::string &src, hipFunction_t *fn) {
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    auto it = g_jit_cache.find(src);
    if (it != g_jit_cache.end()) { *fn = it->second.fn; return PIL2GL_OK; }
    std::vector<char> code;
    P2_TRY(jit_build(src, code));
    JitEntry e;
    HIP_TRY(hipModuleLoadData(&e.mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&e.fn, e.mod, "jit_eval"));
    g_jit_cache.emplace(src, e);
    *fn = e.fn;
    return PIL2GL_OK;
}

// test hook (host only): optimise the program, generate its kernel source and compile it with hiprtc (no GPU needed)
extern "C" int pil2gl_debug_jit_compile(const glx_program *prog, const glx_ctx *ctx, uint64_t *codeBytes, uint32_t *fusedOps) {
    if (!prog || !ctx || !codeBytes) return fail(PIL2GL_EINVAL, "null argument");
    std::vector<IOp> ops; std::vector<u32> pool; u32 n = 0;
    P2_TRY(compile_program(prog, ctx, ops, n, pool, true));
    std::vector<char> code;
    P2_TRY(jit_build(jit_source(ops, n, ctx), code));
    *codeBytes = code.size();
    if (fusedOps) { u32 f = 0; for (const IOp &o : ops) f += o.op == GLX_LZ_MAD; *fusedOps = f; }
    return PIL2GL_OK;
}

extern "C" int pil2gl_eval_program_dev(const glx_program *prog, const glx_ctx *ctx, void *stream) {
    P2_TRY(ensure_init());
    if (!prog || !ctx || (prog->nOps && !prog->ops)) return fail(PIL2GL_EINVAL, "null program");
    if (ctx->nBits > 31) return fail(PIL2GL_EINVAL, "domain too large");
    if (prog->nOps == 0) return PIL2GL_OK;
    hipStream_t st = as_stream(stream);
    std::vector<IOp> ops; std::vector<u32> limbPool; u32 nSlots = 0;
    P2_TRY(compile_program(prog, ctx, ops, nSlots, limbPool, getenv("PIL2GL_EXPR_NOFUSE") == nullptr));

    if (ctx->nSections > GLX_MAX_SECTIONS) return fail(PIL2GL_EINVAL, "too many sections (%u > %d)", ctx->nSections, GLX_MAX_SECTIONS);
    // device form of the program (scratch slot 4): ops, then the scalar pool
    std::vector<DevOp> dops(ops.size());
    for (size_t k = 0; k < ops.size(); k++) {
        auto cv = [&](const glx_ref &r) { DevRef d; d.kind_dim = (u32)r.kind | ((u32)r.dim << 8); d.section = r.section; d.rowOff = (int32_t)((int64_t)r.prime * ((int64_t)1 << ctx->primeShift)); d.index = r.index; return d; };
        dops[k].op = ops[k].op; dops[k].aux = ops[k].aux; dops[k].dest = cv(ops[k].dest); dops[k].src[0] = cv(ops[k].src[0]); dops[k].src[1] = cv(ops[k].src[1]);
        for (int s = 0; s < 3; s++) { const glx_ref &r = s < 2 ? ops[k].src[s] : ops[k].dest; if ((int64_t)r.prime * ((int64_t)1 << ctx->primeShift) != (int64_t)(int32_t)((int64_t)r.prime * ((int64_t)1 << ctx->primeShift))) return fail(PIL2GL_EINVAL, "row offset overflow in op %zu", k); }
    }
    const u64 opsWords = ((u64)dops.size() * sizeof(DevOp) + 7) / 8, limbWords = (limbPool.size() * 4 + 7) / 8;
    u64 *d;
    P2_TRY(scratch(4, opsWords + ctx->nScalars + limbWords + 2, &d));
    HIP_TRY(hipMemcpyAsync(d, dops.data(), dops.size() * sizeof(DevOp), hipMemcpyHostToDevice, st));
    if (ctx->nScalars) HIP_TRY(hipMemcpyAsync(d + opsWords, ctx->scalars, (u64)ctx->nScalars * 8, hipMemcpyHostToDevice, st));
    if (!limbPool.empty()) HIP_TRY(hipMemcpyAsync(d + opsWords + ctx->nScalars + 1, limbPool.data(), limbPool.size() * 4, hipMemcpyHostToDevice, st));
    DevCtx c;
    c.ops = (const DevOp *)d; c.nOps = (u32)dops.size();
    c.scalars = d + opsWords;
    c.limbs = (const u32 *)(d + opsWords + ctx->nScalars + 1);
    c.nBits = ctx->nBits;
    for (u32 i = 0; i < GLX_MAX_SECTIONS; i++) { c.secPtr[i] = i < ctx->nSections ? ctx->sections[i].ptr : nullptr; c.secWidth[i] = i < ctx->nSections ? (u32)ctx->sections[i].width : 0; }
    for (u32 i = 0; i < ctx->nSections; i++) if (ctx->sections[i].width >> 32) return fail(PIL2GL_EINVAL, "section %u too wide", i);
    const u64 nRows = 1ull << ctx->nBits;
    {   // run-time compiled path for long programs on large domains (PIL2GL_EXPR_JIT=1 forces it, =0 disables it)
        const char *e = getenv("PIL2GL_EXPR_JIT");
        const bool want = e ? atoi(e) != 0 : (ops.size() >= 64 && ctx->nBits >= 16);
        if (want && nSlots <= 200) {
            hipFunction_t fn;
            if (jit_get(jit_source(ops, nSlots, ctx), &fn) != PIL2GL_OK) {
                static bool warned = false;
                if (!warned) { fprintf(stderr, "pil2gl: run-time compilation unavailable (%s); using the interpreter kernel\n", pil2gl_last_error()); warned = true; }
                goto interpreter;
            }
            if (getenv("PIL2GL_JIT_INFO") && getenv("PIL2GL_JIT_INFO")[0] == '1') {
                int regs = 0, lds = 0, loc = 0, maxt = 0;
                (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn); (void)hipFuncGetAttribute(&lds, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fn);
                (void)hipFuncGetAttribute(&loc, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn); (void)hipFuncGetAttribute(&maxt, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, fn);
                fprintf(stderr, "pil2gl jit_eval: %zu ops, regs %d, lds %d, scratch %d, max threads %d\n", ops.size(), regs, lds, loc, maxt);
            }
            JitArgs A; memset(&A, 0, sizeof A);
            A.scalars = c.scalars; A.limbs = c.limbs; A.nBits = ctx->nBits;
            for (u32 i = 0; i < ctx->nSections; i++) A.sec[i] = ctx->sections[i].ptr;
            size_t asz = sizeof A;
            void *cfg[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END };
            // (held to fewer resident waves by padding with unused LDS, the config-3 constraint kernel does not get faster: 39.8 ms at
            // five and four waves per SIMD, 48.6 at two, 84.5 at one -- its 2x re-read of the trace is not an occupancy effect)
            HIP_TRY(hipModuleLaunchKernel(fn, (unsigned)((nRows + 255) / 256), 1, 1, 256, 1, 1, 0, st, nullptr, cfg));
            HIP_TRY(hipStreamSynchronize(st));
            return PIL2GL_OK;
        }
    }
interpreter:
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

// ---- constraint checking (calculateExps with debug = true, prover_helpers.js:46-70) --------------------------------------------
// The reference evaluates a constraint on the rows of its boundary, one after the other, and stops at the first row where the value
// is not zero.  Here the program has run on the whole domain (pil2gl_eval_program_dev with the last destination sent to a column);
// what is left is the smallest row of [first, last) whose value is non-zero: one atomic minimum per block that saw one.
__global__ void first_nonzero_kernel(const u64 *__restrict__ col, u32 dim, u64 first, u64 last, unsigned long long *__restrict__ best) {
    const u64 r = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    bool nz = false;
    if (r < last) for (u32 k = 0; k < dim; k++) nz |= col[r * dim + k] != 0;
    const u64 mask = __ballot(nz);
    if (mask && (threadIdx.x & 63) == 0) atomicMin(best, (unsigned long long)(r + __builtin_ctzll(mask)));
}
extern "C" int pil2gl_first_nonzero_row_dev(const uint64_t *col, uint32_t dim, uint64_t first, uint64_t last, uint64_t *hostRow, uint64_t *hostVal, void *stream) {
    P2_TRY(ensure_init());
    if (!hostRow || !hostVal) return fail(PIL2GL_EINVAL, "null buffer");
    if (dim != 1 && dim != 3) return fail(PIL2GL_EINVAL, "dim must be 1 or 3 (got %u)", dim);
    if (last < first) return fail(PIL2GL_EINVAL, "empty range [%llu, %llu)", (unsigned long long)first, (unsigned long long)last);
    *hostRow = ~0ull;
    for (u32 k = 0; k < dim; k++) hostVal[k] = 0;
    if (last == first) return PIL2GL_OK;
    if (!col) return fail(PIL2GL_EINVAL, "null buffer");
    const u64 blocks = (last - first + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");
    hipStream_t st = as_stream(stream);
    u64 *d;
    P2_TRY(scratch(4, 1, &d));
    HIP_TRY(hipMemsetAsync(d, 0xff, 8, st));
    first_nonzero_kernel<<<(unsigned)blocks, 256, 0, st>>>(col, dim, first, last, (unsigned long long *)d);
    KERNEL_CHECK();
    HIP_TRY(hipMemcpyAsync(hostRow, d, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (*hostRow != ~0ull) HIP_TRY(hipMemcpy(hostVal, col + *hostRow * dim, (size_t)dim * 8, hipMemcpyDeviceToHost));
    return PIL2GL_OK;
}
