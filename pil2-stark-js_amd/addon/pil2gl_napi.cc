// N-API binding of the libpil2gl C ABI (include/pil2gl.h) for Node.js >= 12.17 (N-API 6: BigUint64Array).
//
// Thin by design: every export is one C-ABI call on raw pointers.  Host buffers are BigUint64Array
// (what the reference hands to its workers: fft_p.js:246, merklehash_p.js:70); device buffers are
// addresses carried as BigInt.  The CommonJS modules in ../js give these the reference's own
// function names and signatures.  Calls are synchronous (the reference awaits each one anyway).
#include <node_api.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include "pil2gl.h"

#define NAPI_CALL(env, call)                                                         \
    do { if ((call) != napi_ok) { napi_throw_error(env, nullptr, "N-API call failed: " #call); return nullptr; } } while (0)

static napi_value throw_rc(napi_env env, int rc) {
    std::string m = "pil2gl error " + std::to_string(rc) + ": " + pil2gl_last_error();
    napi_throw_error(env, nullptr, m.c_str());
    return nullptr;
}
#define P2(env, call) do { int rc_ = (call); if (rc_ != PIL2GL_OK) return throw_rc(env, rc_); } while (0)

struct Args {
    napi_env env; size_t argc; napi_value argv[12]; bool ok;
    Args(napi_env e, napi_callback_info info) : env(e), argc(12), ok(true) {
        if (napi_get_cb_info(e, info, &argc, argv, nullptr, nullptr) != napi_ok) ok = false;
    }
    bool fail(const char *m) { if (ok) napi_throw_type_error(env, nullptr, m); ok = false; return false; }
    uint64_t u64(size_t i) {                       // Number or BigInt
        if (!ok || i >= argc) { fail("missing argument"); return 0; }
        napi_valuetype t; napi_typeof(env, argv[i], &t);
        if (t == napi_bigint) { uint64_t v = 0; bool lossless; napi_get_value_bigint_uint64(env, argv[i], &v, &lossless); return v; }
        if (t == napi_number) { double d; napi_get_value_double(env, argv[i], &d); if (d < 0) { fail("negative size"); return 0; } return (uint64_t)d; }
        if (t == napi_boolean) { bool b; napi_get_value_bool(env, argv[i], &b); return b; }
        fail("expected a number or BigInt"); return 0;
    }
    bool is_nullish(size_t i) {
        if (i >= argc) return true;
        napi_valuetype t; napi_typeof(env, argv[i], &t);
        return t == napi_undefined || t == napi_null;
    }
    uint64_t *arr(size_t i, uint64_t minWords, uint64_t *len = nullptr) {   // BigUint64Array
        if (!ok || i >= argc) { fail("missing buffer argument"); return nullptr; }
        bool is; napi_is_typedarray(env, argv[i], &is);
        if (!is) { fail("expected a BigUint64Array"); return nullptr; }
        napi_typedarray_type ty; size_t n; void *data; napi_value ab; size_t off;
        napi_get_typedarray_info(env, argv[i], &ty, &n, &data, &ab, &off);
        if (ty != napi_biguint64_array && ty != napi_bigint64_array) { fail("expected a BigUint64Array"); return nullptr; }
        if (n < minWords) { fail("buffer too small"); return nullptr; }
        if (len) *len = n;
        return (uint64_t *)data;
    }
    void *stream(size_t i) { return is_nullish(i) ? nullptr : (void *)(uintptr_t)u64(i); }
};

static napi_value mk_undefined(napi_env env) { napi_value v; napi_get_undefined(env, &v); return v; }
static napi_value mk_bigint(napi_env env, uint64_t x) { napi_value v; napi_create_bigint_uint64(env, x, &v); return v; }

#define FN(name) static napi_value name(napi_env env, napi_callback_info info)

FN(Init) { Args a(env, info); int dev = a.is_nullish(0) ? 0 : (int)a.u64(0); if (!a.ok) return nullptr; P2(env, pil2gl_init(dev)); return mk_undefined(env); }
FN(Shutdown) { pil2gl_shutdown(); return mk_undefined(env); }
FN(DeviceInfo) {
    char name[64]; uint32_t cus; uint64_t mem;
    P2(env, pil2gl_device_info(name, sizeof name, &cus, &mem));
    napi_value o, v; napi_create_object(env, &o);
    napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "name", v);
    napi_create_uint32(env, cus, &v); napi_set_named_property(env, o, "computeUnits", v);
    napi_set_named_property(env, o, "totalMem", mk_bigint(env, mem));
    return o;
}

// ---- device buffers ----
FN(DevAlloc) { Args a(env, info); uint64_t n = a.u64(0); if (!a.ok) return nullptr; uint64_t *p; P2(env, pil2gl_dev_alloc(n, &p)); return mk_bigint(env, (uint64_t)(uintptr_t)p); }
FN(DevFree) { Args a(env, info); uint64_t p = a.u64(0); if (!a.ok) return nullptr; P2(env, pil2gl_dev_free((uint64_t *)(uintptr_t)p)); return mk_undefined(env); }
FN(DevZero) { Args a(env, info); uint64_t p = a.u64(0), off = a.u64(1), n = a.u64(2); if (!a.ok) return nullptr; P2(env, pil2gl_dev_zero((uint64_t *)(uintptr_t)p + off, n, nullptr)); return mk_undefined(env); }
FN(DevUpload) {      // (devPtr, offsetWords, BigUint64Array)
    Args a(env, info); uint64_t p = a.u64(0), off = a.u64(1), n = 0; uint64_t *h = a.arr(2, 0, &n); if (!a.ok) return nullptr;
    P2(env, pil2gl_dev_upload((uint64_t *)(uintptr_t)p + off, h, n)); return mk_undefined(env);
}
FN(DevDownload) {    // (BigUint64Array, devPtr, offsetWords)
    Args a(env, info); uint64_t n = 0; uint64_t *h = a.arr(0, 0, &n); uint64_t p = a.u64(1), off = a.u64(2); if (!a.ok) return nullptr;
    P2(env, pil2gl_dev_download(h, (uint64_t *)(uintptr_t)p + off, n)); return mk_undefined(env);
}
FN(Sync) { P2(env, pil2gl_sync(nullptr)); return mk_undefined(env); }

// ---- NTT ----
FN(Interpolate) {    // (src, nPols, nBits, dst, nBitsExt)  fft_p.js:187
    Args a(env, info); uint64_t nPols = a.u64(1); uint32_t nBits = (uint32_t)a.u64(2), nBitsExt = (uint32_t)a.u64(4);
    if (nBits > 40 || nBitsExt > 40) a.fail("bad size");
    uint64_t *s = a.arr(0, a.ok ? nPols << nBits : 0), *d = a.arr(3, a.ok ? nPols << nBitsExt : 0); if (!a.ok) return nullptr;
    P2(env, pil2gl_interpolate(s, nPols, nBits, d, nBitsExt)); return mk_undefined(env);
}
static napi_value fft_common(napi_env env, napi_callback_info info, bool inverse) {
    Args a(env, info); uint64_t nPols = a.u64(1); uint32_t nBits = (uint32_t)a.u64(2);
    if (nBits > 40) a.fail("bad size");
    uint64_t *s = a.arr(0, a.ok ? nPols << nBits : 0), *d = a.arr(3, a.ok ? nPols << nBits : 0); if (!a.ok) return nullptr;
    P2(env, inverse ? pil2gl_ifft(s, nPols, nBits, d) : pil2gl_fft(s, nPols, nBits, d)); return mk_undefined(env);
}
FN(Fft) { return fft_common(env, info, false); }
FN(Ifft) { return fft_common(env, info, true); }
FN(InterpolateDev) {
    Args a(env, info); uint64_t s = a.u64(0), nPols = a.u64(1), nBits = a.u64(2), d = a.u64(3), nBitsExt = a.u64(4); if (!a.ok) return nullptr;
    P2(env, pil2gl_interpolate_dev((const uint64_t *)(uintptr_t)s, nPols, (uint32_t)nBits, (uint64_t *)(uintptr_t)d, (uint32_t)nBitsExt, a.stream(5))); return mk_undefined(env);
}
FN(FftDev) {
    Args a(env, info); uint64_t s = a.u64(0), nPols = a.u64(1), nBits = a.u64(2), d = a.u64(3); if (!a.ok) return nullptr;
    P2(env, pil2gl_fft_dev((const uint64_t *)(uintptr_t)s, nPols, (uint32_t)nBits, (uint64_t *)(uintptr_t)d, a.stream(4))); return mk_undefined(env);
}
FN(IfftDev) {
    Args a(env, info); uint64_t s = a.u64(0), nPols = a.u64(1), nBits = a.u64(2), d = a.u64(3); if (!a.ok) return nullptr;
    P2(env, pil2gl_ifft_dev((const uint64_t *)(uintptr_t)s, nPols, (uint32_t)nBits, (uint64_t *)(uintptr_t)d, a.stream(4))); return mk_undefined(env);
}

// ---- hashing ----
FN(Poseidon) {       // (in BigUint64Array(8*count), cap BigUint64Array(4*count)|null, count, nOut, out)
    Args a(env, info); uint64_t count = a.u64(2); uint32_t nOut = (uint32_t)a.u64(3);
    uint64_t *in = a.arr(0, 8 * count), *cap = a.is_nullish(1) ? nullptr : a.arr(1, 4 * count), *out = a.arr(4, (uint64_t)nOut * count); if (!a.ok) return nullptr;
    P2(env, pil2gl_poseidon(in, cap, count, nOut, out)); return mk_undefined(env);
}
FN(LinearHashRows) { // (in, width, height, split, out)  merklehash_worker.js:37
    Args a(env, info); uint64_t w = a.u64(1), h = a.u64(2); int split = (int)a.u64(3);
    uint64_t *in = a.arr(0, w * h), *out = a.arr(4, 4 * h); if (!a.ok) return nullptr;
    P2(env, pil2gl_linear_hash_rows(in, w, h, split, out)); return mk_undefined(env);
}
FN(MerkelizeLevel) { // (in, nOps, out)  merklehash_worker.js:86
    Args a(env, info); uint64_t n = a.u64(1); uint64_t *in = a.arr(0, 8 * n), *out = a.arr(2, 4 * n); if (!a.ok) return nullptr;
    P2(env, pil2gl_merkelize_level(in, n, out)); return mk_undefined(env);
}
FN(MerkleNumNodes) { Args a(env, info); uint64_t h = a.u64(0); if (!a.ok) return nullptr; napi_value v; napi_create_double(env, (double)pil2gl_merkle_num_nodes(h), &v); return v; }
FN(Merkelize) {      // (elems, width, height, split, nodes)  merklehash_p.js:44
    Args a(env, info); uint64_t w = a.u64(1), h = a.u64(2); int split = (int)a.u64(3);
    uint64_t *el = a.arr(0, w * h), *nodes = a.arr(4, pil2gl_merkle_num_nodes(h)); if (!a.ok) return nullptr;
    P2(env, pil2gl_merkelize(el, w, h, split, nodes)); return mk_undefined(env);
}
FN(MerkelizeDev) {   // (devElems, width, height, split, devNodes)
    Args a(env, info); uint64_t el = a.u64(0), w = a.u64(1), h = a.u64(2); int split = (int)a.u64(3); uint64_t nodes = a.u64(4); if (!a.ok) return nullptr;
    P2(env, pil2gl_merkelize_dev((const uint64_t *)(uintptr_t)el, w, h, split, (uint64_t *)(uintptr_t)nodes, a.stream(5))); return mk_undefined(env);
}
FN(GroupProofDev) {  // (devElems, devNodes, width, height, idx, vals BigUint64Array(width), siblings BigUint64Array(4*64)) -> nLevels
    Args a(env, info); uint64_t el = a.u64(0), nodes = a.u64(1), w = a.u64(2), h = a.u64(3), idx = a.u64(4);
    uint64_t *vals = a.arr(5, w), *sib = a.arr(6, 4 * 64); if (!a.ok) return nullptr;
    uint32_t nl = 0;
    P2(env, pil2gl_group_proof_dev((const uint64_t *)(uintptr_t)el, (const uint64_t *)(uintptr_t)nodes, w, h, idx, vals, sib, &nl));
    napi_value v; napi_create_uint32(env, nl, &v); return v;
}

FN(GroupProofsDev) { // (devElems, devNodes, width, height, idxs BigUint64Array(n), out BigUint64Array(n*(width+4*levels))) -> nLevels: every query of a tree in one gather
    Args a(env, info); uint64_t el = a.u64(0), nodes = a.u64(1), w = a.u64(2), h = a.u64(3);
    uint64_t n = 0; uint64_t *idxs = a.arr(4, 1, &n); uint64_t *out = a.arr(5, n * w); if (!a.ok) return nullptr;
    uint32_t nl = 0;
    for (uint64_t m = h * 4; m > 4; m = ((m - 1) / 8 + 1) * 4) nl++;
    uint64_t outLen = 0; a.arr(5, n * (w + 4ull * nl), &outLen); if (!a.ok) return nullptr;
    P2(env, pil2gl_group_proofs_dev((const uint64_t *)(uintptr_t)el, (const uint64_t *)(uintptr_t)nodes, w, h, idxs, (uint32_t)n, out, &nl));
    napi_value v; napi_create_uint32(env, nl, &v); return v;
}

FN(SpongeAbsorb) {  // (blocks BigUint64Array(8*n), n, cap BigUint64Array(4), out BigUint64Array(12))  transcript.js:49-66 for a list
    Args a(env, info); uint64_t n = a.u64(1);
    uint64_t *blocks = a.arr(0, 8 * n), *cap = a.arr(2, 4), *out = a.arr(3, 12); if (!a.ok) return nullptr;
    P2(env, pil2gl_sponge_absorb(blocks, n, cap, out)); return mk_undefined(env);
}
FN(RootsFromGroupProofs) {  // (packed BigUint64Array(n*(width+4*levels)), width, levels, idxs BigUint64Array(n), n, split, roots BigUint64Array(4n))
    Args a(env, info); uint64_t w = a.u64(1), lv = a.u64(2), n = a.u64(4); int split = (int)a.u64(5);
    uint64_t *packed = a.arr(0, n * (w + 4 * lv)), *idx = a.arr(3, n), *roots = a.arr(6, 4 * n); if (!a.ok) return nullptr;
    P2(env, pil2gl_roots_from_group_proofs(packed, w, (uint32_t)lv, idx, (uint32_t)n, split, roots)); return mk_undefined(env);
}

// ---- STARK step helpers and stage-2 hints (device pointers; js/stark_gen_helpers.js and js/polutils.js stage host buffers) ----
#define DP(i) ((uint64_t *)(uintptr_t)a.u64(i))
FN(BuildXDev) {        // (nBits, shift, dX)  stark_gen_helpers.js:111-116,139-144
    Args a(env, info); uint32_t nBits = (uint32_t)a.u64(0); uint64_t shift = a.u64(1); uint64_t *x = DP(2); if (!a.ok) return nullptr;
    P2(env, pil2gl_build_x_dev(nBits, shift, x, a.stream(3))); return mk_undefined(env);
}
FN(BuildZhInvDev) {    // (nBits, nBitsExt, dOut)  polutils.js:39-55
    Args a(env, info); uint32_t nb = (uint32_t)a.u64(0), nbe = (uint32_t)a.u64(1); uint64_t *o = DP(2); if (!a.ok) return nullptr;
    P2(env, pil2gl_build_zhinv_dev(nb, nbe, o, a.stream(3))); return mk_undefined(env);
}
FN(BuildOneRowZerofierInvDev) {   // (nBits, nBitsExt, rowIndex, dOut)  polutils.js:57-71
    Args a(env, info); uint32_t nb = (uint32_t)a.u64(0), nbe = (uint32_t)a.u64(1); uint64_t row = a.u64(2); uint64_t *o = DP(3); if (!a.ok) return nullptr;
    P2(env, pil2gl_build_one_row_zerofier_inv_dev(nb, nbe, row, o, a.stream(4))); return mk_undefined(env);
}
FN(BuildFrameZerofierDev) {       // (nBits, nBitsExt, offsetMin, offsetMax, dOut)  polutils.js:74-102
    Args a(env, info); uint32_t nb = (uint32_t)a.u64(0), nbe = (uint32_t)a.u64(1); uint64_t mn = a.u64(2), mx = a.u64(3); uint64_t *o = DP(4); if (!a.ok) return nullptr;
    P2(env, pil2gl_build_frame_zerofier_dev(nb, nbe, mn, mx, o, a.stream(5))); return mk_undefined(env);
}
FN(ComputeQSplitDev) { // (dQq1, nBits, nBitsExt, qDim, qDeg, dQq2)  stark_gen_helpers.js:179-190
    Args a(env, info); uint64_t *q1 = DP(0); uint32_t nb = (uint32_t)a.u64(1), nbe = (uint32_t)a.u64(2), qDim = (uint32_t)a.u64(3), qDeg = (uint32_t)a.u64(4); uint64_t *q2 = DP(5); if (!a.ok) return nullptr;
    P2(env, pil2gl_compute_q_split_dev(q1, nb, nbe, qDim, qDeg, q2, a.stream(6))); return mk_undefined(env);
}
FN(ComputeQSplitBrevDev) { // (dQq1, nBits, nBitsExt, qDim, qDeg, dCoefBrev): the pieces as N coefficient rows, row bitrev(i) = coefficient i
    Args a(env, info); uint64_t *q1 = DP(0); uint32_t nb = (uint32_t)a.u64(1), nbe = (uint32_t)a.u64(2), qDim = (uint32_t)a.u64(3), qDeg = (uint32_t)a.u64(4); uint64_t *c = DP(5); if (!a.ok) return nullptr;
    P2(env, pil2gl_compute_q_split_brev_dev(q1, nb, nbe, qDim, qDeg, c, a.stream(6))); return mk_undefined(env);
}
FN(ExtendCoefsBrevDev) { // (dCoefBrev, nPols, nBits, dDst, nBitsExt): fft(nBitsExt) of the zero-padded coefficients (stark_gen_helpers.js:192)
    Args a(env, info); uint64_t s = a.u64(0), nPols = a.u64(1), nBits = a.u64(2), d = a.u64(3), nBitsExt = a.u64(4); if (!a.ok) return nullptr;
    P2(env, pil2gl_extend_coefs_brev_dev((const uint64_t *)(uintptr_t)s, nPols, (uint32_t)nBits, (uint64_t *)(uintptr_t)d, (uint32_t)nBitsExt, a.stream(5))); return mk_undefined(env);
}
FN(XDivXSubXiDev) {    // (nBitsExt, xi BigUint64Array(3), nOpen, iOpen, dOut)  stark_gen_helpers.js:293-322
    Args a(env, info); uint32_t nbe = (uint32_t)a.u64(0); uint64_t *xi = a.arr(1, 3); uint64_t nOpen = a.u64(2), iOpen = a.u64(3); uint64_t *o = DP(4); if (!a.ok) return nullptr;
    P2(env, pil2gl_x_div_x_sub_xi_dev(nbe, xi, nOpen, iOpen, o, a.stream(5))); return mk_undefined(env);
}
FN(BuildLevDev) {      // (nBits, xi BigUint64Array(3), dLev)  stark_gen_helpers.js:216-231
    Args a(env, info); uint32_t nb = (uint32_t)a.u64(0); uint64_t *xi = a.arr(1, 3); uint64_t *lev = DP(2); if (!a.ok) return nullptr;
    P2(env, pil2gl_build_lev_dev(nb, xi, lev, a.stream(3))); return mk_undefined(env);
}
FN(ComputeEvalsDev) {  // (descs BigUint64Array(5*nEvals) = [dBuf,width,offset,dim,levIndex]*, nEvals, nBits, extendBits, levs BigUint64Array(nLev) device ptrs, out BigUint64Array(3*nEvals))
    Args a(env, info); uint64_t nEv = a.u64(1); uint64_t *d = a.arr(0, 5 * nEv); uint32_t nb = (uint32_t)a.u64(2), eb = (uint32_t)a.u64(3);
    uint64_t nLev = 0; uint64_t *levs = a.arr(4, 1, &nLev); uint64_t *out = a.arr(5, 3 * nEv); if (!a.ok) return nullptr;
    std::vector<pil2gl_eval_desc> descs(nEv);
    for (uint64_t i = 0; i < nEv; i++) { descs[i].buf = (const uint64_t *)(uintptr_t)d[5 * i]; descs[i].width = d[5 * i + 1]; descs[i].offset = d[5 * i + 2]; descs[i].dim = (uint32_t)d[5 * i + 3]; descs[i].levIndex = (uint32_t)d[5 * i + 4]; }
    std::vector<const uint64_t *> lp(nLev);
    for (uint64_t i = 0; i < nLev; i++) lp[i] = (const uint64_t *)(uintptr_t)levs[i];
    P2(env, pil2gl_compute_evals_dev(descs.data(), (uint32_t)nEv, nb, eb, lp.data(), (uint32_t)nLev, out, nullptr)); return mk_undefined(env);
}
// the two stages that are matrix-vector products (csrc/dot.hip): FRI polynomial as row sums, evaluations as column sums
FN(RowsDotExtDev) {    // (dBuf, width, nRows, coef BigUint64Array(nOut*width*3), nOut, dAcc, accumulate)
    Args a(env, info); uint64_t *buf = DP(0); uint64_t width = a.u64(1), nRows = a.u64(2); uint32_t nOut = (uint32_t)a.u64(4);
    uint64_t *coef = a.arr(3, (uint64_t)nOut * width * 3); uint64_t *acc = DP(5); int accumulate = (int)a.u64(6); if (!a.ok) return nullptr;
    P2(env, pil2gl_rows_dot_ext_dev(buf, width, nRows, coef, nOut, acc, accumulate, nullptr)); return mk_undefined(env);
}
FN(RowsDotExtMultiDev) {   // (dBufs BigUint64Array(n) device ptrs, widths BigUint64Array(n), nRows, coefs Array(n) of BigUint64Array(nOut*width*3), nOut, dAcc, accumulate)
    Args a(env, info); uint64_t n = 0; uint64_t *ptrs = a.arr(0, 1, &n); uint64_t *widths = a.arr(1, n); uint64_t nRows = a.u64(2);
    uint32_t nOut = (uint32_t)a.u64(4); uint64_t *acc = DP(5); int accumulate = (int)a.u64(6); if (!a.ok || n == 0 || n > 16) return nullptr;
    std::vector<const uint64_t *> bp(n), cp(n);
    for (uint64_t k = 0; k < n; k++) {
        bp[k] = (const uint64_t *)(uintptr_t)ptrs[k];
        napi_value el; if (napi_get_element(env, a.argv[3], (uint32_t)k, &el) != napi_ok) { napi_throw_error(env, nullptr, "coefs must be an array of BigUint64Array"); return nullptr; }
        napi_typedarray_type ty; size_t len; void *data; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, el, &ty, &len, &data, &ab, &off) != napi_ok || ty != napi_biguint64_array || len < (size_t)nOut * widths[k] * 3) {
            napi_throw_error(env, nullptr, "coefs[k] must be a BigUint64Array of nOut*width*3 words"); return nullptr; }
        cp[k] = (const uint64_t *)data;
    }
    P2(env, pil2gl_rows_dot_ext_multi_dev(bp.data(), widths, (uint32_t)n, nRows, cp.data(), nOut, acc, accumulate, nullptr)); return mk_undefined(env);
}
FN(FriCombineDev) {    // (dAcc, K BigUint64Array(nOpen*3), vf1 BigUint64Array(3), dXDivXSubXi, nOpen, nRows, dF[, order BigUint64Array(nOpen)])
    Args a(env, info); uint64_t *acc = DP(0); uint32_t nOpen = (uint32_t)a.u64(4); uint64_t *K = a.arr(1, 3ull * nOpen), *vf1 = a.arr(2, 3);
    uint64_t *x = DP(3); uint64_t nRows = a.u64(5); uint64_t *f = DP(6);
    uint32_t order[4] = { 0, 1, 2, 3 };
    if (!a.is_nullish(7)) { uint64_t *o = a.arr(7, nOpen); if (a.ok && nOpen <= 4) for (uint32_t k = 0; k < nOpen; k++) order[k] = (uint32_t)o[k]; }
    if (!a.ok) return nullptr;
    P2(env, pil2gl_fri_combine_order_dev(acc, K, vf1, x, nOpen, order, nRows, f, nullptr)); return mk_undefined(env);
}
FN(ColsDotExtDev) {    // (dBuf, width, nRows, rowStep, levs BigUint64Array(nLev) device ptrs, out BigUint64Array(nLev*width*3))
    Args a(env, info); uint64_t *buf = DP(0); uint64_t width = a.u64(1), nRows = a.u64(2), rowStep = a.u64(3);
    uint64_t nLev = 0; uint64_t *levs = a.arr(4, 1, &nLev); uint64_t *out = a.arr(5, nLev * width * 3); if (!a.ok) return nullptr;
    std::vector<const uint64_t *> lp(nLev);
    for (uint64_t i = 0; i < nLev; i++) lp[i] = (const uint64_t *)(uintptr_t)levs[i];
    P2(env, pil2gl_cols_dot_ext_dev(buf, width, nRows, rowStep, lp.data(), (uint32_t)nLev, out, nullptr)); return mk_undefined(env);
}
FN(ColsDotExtMultiDev) {   // (dBufs BigUint64Array(n) device ptrs, widths BigUint64Array(n), nRows, rowStep, levs BigUint64Array(nLev) device ptrs, outs Array(n) of BigUint64Array(nLev*width*3))
    Args a(env, info); uint64_t n = 0; uint64_t *ptrs = a.arr(0, 1, &n); uint64_t *widths = a.arr(1, n); uint64_t nRows = a.u64(2), rowStep = a.u64(3);
    uint64_t nLev = 0; uint64_t *levs = a.arr(4, 1, &nLev); if (!a.ok || n == 0 || n > 8) return nullptr;
    std::vector<const uint64_t *> bp(n), lp(nLev); std::vector<uint64_t *> op(n);
    for (uint64_t i = 0; i < nLev; i++) lp[i] = (const uint64_t *)(uintptr_t)levs[i];
    for (uint64_t k = 0; k < n; k++) {
        bp[k] = (const uint64_t *)(uintptr_t)ptrs[k];
        napi_value el; napi_typedarray_type ty; size_t len; void *data; napi_value ab; size_t off;
        if (napi_get_element(env, a.argv[5], (uint32_t)k, &el) != napi_ok || napi_get_typedarray_info(env, el, &ty, &len, &data, &ab, &off) != napi_ok ||
            ty != napi_biguint64_array || len < (size_t)nLev * widths[k] * 3) { napi_throw_error(env, nullptr, "outs[k] must be a BigUint64Array of nLev*width*3 words"); return nullptr; }
        op[k] = (uint64_t *)data;
    }
    P2(env, pil2gl_cols_dot_ext_multi_dev(bp.data(), widths, (uint32_t)n, nRows, rowStep, lp.data(), (uint32_t)nLev, op.data(), nullptr)); return mk_undefined(env);
}
FN(SynthFibonacciDev) { // (nBits, nPairs, init BigUint64Array(2*nPairs), dCm): synthetic witness for benchmarks (pil2gl.h)
    Args a(env, info); uint32_t nb = (uint32_t)a.u64(0), np = (uint32_t)a.u64(1); uint64_t *init = a.arr(2, 2ull * np); uint64_t *cm = DP(3); if (!a.ok) return nullptr;
    P2(env, pil2gl_synth_fibonacci_dev(nb, np, init, cm, nullptr)); return mk_undefined(env);
}
FN(GprodDev) {         // (dNum, dimNum, dDen, dimDen, n, dOut)  polutils.js:128-143
    Args a(env, info); uint64_t *num = DP(0); uint32_t dn = (uint32_t)a.u64(1); uint64_t *den = DP(2); uint32_t dd = (uint32_t)a.u64(3); uint64_t n = a.u64(4); uint64_t *o = DP(5); if (!a.ok) return nullptr;
    P2(env, pil2gl_gprod_dev(num, dn, den, dd, n, o, a.stream(6))); return mk_undefined(env);
}
FN(GsumDev) {          // (dNum (one element), dimNum, dDen, dimDen, n, dOut)  polutils.js:145-164
    Args a(env, info); uint64_t *num = DP(0); uint32_t dn = (uint32_t)a.u64(1); uint64_t *den = DP(2); uint32_t dd = (uint32_t)a.u64(3); uint64_t n = a.u64(4); uint64_t *o = DP(5); if (!a.ok) return nullptr;
    P2(env, pil2gl_gsum_dev(num, dn, den, dd, n, o, a.stream(6))); return mk_undefined(env);
}
FN(H1H2Dev) {          // (dF, dT, n, dim, dH1, dH2)  polutils.js:105-126
    Args a(env, info); uint64_t *f = DP(0), *t = DP(1); uint64_t n = a.u64(2); uint32_t dim = (uint32_t)a.u64(3); uint64_t *h1 = DP(4), *h2 = DP(5); if (!a.ok) return nullptr;
    P2(env, pil2gl_h1h2_dev(f, t, n, dim, h1, h2, a.stream(6))); return mk_undefined(env);
}

// ---- BN128 Merkle commitment (merklehash_bn128_p.js / merklehash_bn128_worker.js) ----
FN(Bn128Poseidon) {  // (in BigUint64Array(4*nIn*count) normal form, init BigUint64Array(4*count)|null, count, nIn, nOut, out(4*nOut*count))
    Args a(env, info); uint64_t count = a.u64(2), nIn = a.u64(3), nOut = a.u64(4);
    uint64_t *in = a.arr(0, 4 * nIn * count), *init = a.is_nullish(1) ? nullptr : a.arr(1, 4 * count), *out = a.arr(5, 4 * nOut * count); if (!a.ok) return nullptr;
    P2(env, pil2gl_bn128_poseidon(in, init, count, (uint32_t)nIn, (uint32_t)nOut, out)); return mk_undefined(env);
}
FN(Bn128SpongeAbsorb) {  // (blocks BigUint64Array(4*nIn*n) normal form, n, nIn, init BigUint64Array(4), out BigUint64Array(4*(nIn+1)))  transcript.bn128.js:56-83 for a list
    Args a(env, info); uint64_t n = a.u64(1), nIn = a.u64(2);
    uint64_t *blocks = a.arr(0, 4 * nIn * n), *init = a.arr(3, 4), *out = a.arr(4, 4 * (nIn + 1)); if (!a.ok) return nullptr;
    P2(env, pil2gl_bn128_sponge_absorb(blocks, n, (uint32_t)nIn, init, out)); return mk_undefined(env);
}
FN(Bn128LinearHashRows) { // (in, width, height, arity, custom, out(4*height) Montgomery)  merklehash_bn128_worker.js:13
    Args a(env, info); uint64_t w = a.u64(1), h = a.u64(2), arity = a.u64(3); int custom = (int)a.u64(4);
    uint64_t *in = a.arr(0, w * h), *out = a.arr(5, 4 * h); if (!a.ok) return nullptr;
    P2(env, pil2gl_bn128_linear_hash_rows(in, w, h, (uint32_t)arity, custom, out)); return mk_undefined(env);
}
FN(Bn128MerkleNumNodes) { Args a(env, info); uint64_t h = a.u64(0), arity = a.u64(1); if (!a.ok) return nullptr; napi_value v; napi_create_double(env, (double)pil2gl_bn128_merkle_num_nodes(h, (uint32_t)arity), &v); return v; }
FN(Bn128Merkelize) { // (elems, width, height, arity, custom, nodes(4*nNodes))  merklehash_bn128_p.js:47
    Args a(env, info); uint64_t w = a.u64(1), h = a.u64(2), arity = a.u64(3); int custom = (int)a.u64(4);
    uint64_t *el = a.arr(0, w * h), *nodes = a.arr(5, 4 * pil2gl_bn128_merkle_num_nodes(h, (uint32_t)arity)); if (!a.ok) return nullptr;
    P2(env, pil2gl_bn128_merkelize(el, w, h, (uint32_t)arity, custom, nodes)); return mk_undefined(env);
}
FN(Bn128MerkelizeDev) { // (devElems, width, height, arity, custom, devNodes)
    Args a(env, info); uint64_t el = a.u64(0), w = a.u64(1), h = a.u64(2), arity = a.u64(3); int custom = (int)a.u64(4); uint64_t nodes = a.u64(5); if (!a.ok) return nullptr;
    P2(env, pil2gl_bn128_merkelize_dev((const uint64_t *)(uintptr_t)el, w, h, (uint32_t)arity, custom, (uint64_t *)(uintptr_t)nodes, a.stream(6))); return mk_undefined(env);
}
FN(Bn128Convert) {   // (in BigUint64Array(4n), n, toMontgomery, out)
    Args a(env, info); uint64_t n = a.u64(1); int toM = (int)a.u64(2);
    uint64_t *in = a.arr(0, 4 * n), *out = a.arr(3, 4 * n); if (!a.ok) return nullptr;
    P2(env, pil2gl_bn128_convert(in, n, toM, out)); return mk_undefined(env);
}

// ---- FRI ----
FN(FriFold) {        // (pol, polBits, outBits, shiftInv, challenge[3], out)  fri.js:22
    Args a(env, info); uint32_t pb = (uint32_t)a.u64(1), ob = (uint32_t)a.u64(2); uint64_t sinv = a.u64(3);
    if (pb > 40 || ob > pb) a.fail("Invalid polynomial size");
    uint64_t *pol = a.arr(0, a.ok ? 3ull << pb : 0), *ch = a.arr(4, 3), *out = a.arr(5, a.ok ? 3ull << ob : 0); if (!a.ok) return nullptr;
    P2(env, pil2gl_fri_fold(pol, pb, ob, sinv, ch, out)); return mk_undefined(env);
}
FN(FriVerifyFold) {  // (groups BigUint64Array(3*2^foldBits*nQ), foldBits, nQ, sinv BigUint64Array(nQ), challenge[3], out BigUint64Array(3*nQ))  fri.js:121-127
    Args a(env, info); uint32_t fb = (uint32_t)a.u64(1), nq = (uint32_t)a.u64(2);
    if (fb > 20 || nq == 0) a.fail("Invalid group size or query count");
    uint64_t *g = a.arr(0, a.ok ? (3ull << fb) * nq : 0), *sinv = a.arr(3, nq), *ch = a.arr(4, 3), *out = a.arr(5, 3ull * nq); if (!a.ok) return nullptr;
    P2(env, pil2gl_fri_verify_fold(g, fb, nq, sinv, ch, out)); return mk_undefined(env);
}
FN(FriTranspose) {   // (pol, polBits, transposeBits, out)  fri.js:187
    Args a(env, info); uint32_t pb = (uint32_t)a.u64(1), tb = (uint32_t)a.u64(2);
    if (pb > 40) a.fail("Invalid polynomial size");
    uint64_t *pol = a.arr(0, a.ok ? 3ull << pb : 0), *out = a.arr(3, a.ok ? 3ull << pb : 0); if (!a.ok) return nullptr;
    P2(env, pil2gl_fri_transpose(pol, pb, tb, out)); return mk_undefined(env);
}

FN(FriFoldDev) {     // (dPol, polBits, outBits, shiftInv, challenge BigUint64Array(3), dOut)
    Args a(env, info); uint64_t pol = a.u64(0); uint32_t pb = (uint32_t)a.u64(1), ob = (uint32_t)a.u64(2); uint64_t sinv = a.u64(3);
    uint64_t *ch = a.arr(4, 3); uint64_t out = a.u64(5); if (!a.ok) return nullptr;
    P2(env, pil2gl_fri_fold_dev((const uint64_t *)(uintptr_t)pol, pb, ob, sinv, ch, (uint64_t *)(uintptr_t)out, a.stream(6))); return mk_undefined(env);
}
FN(FriTransposeDev) { // (dPol, polBits, transposeBits, dOut)
    Args a(env, info); uint64_t pol = a.u64(0); uint32_t pb = (uint32_t)a.u64(1), tb = (uint32_t)a.u64(2); uint64_t out = a.u64(3); if (!a.ok) return nullptr;
    P2(env, pil2gl_fri_transpose_dev((const uint64_t *)(uintptr_t)pol, pb, tb, (uint64_t *)(uintptr_t)out, a.stream(4))); return mk_undefined(env);
}

// ---- expression evaluator ----
// (opsBuf BigUint64Array = packed glx_op[], nOps, nTmp, nBits, primeShift, sectionPtrs BigUint64Array, sectionWidths BigUint64Array, scalars BigUint64Array)
FN(EvalProgramDev) {
    Args a(env, info); uint64_t nOps = a.u64(1), nTmp = a.u64(2), nBits = a.u64(3), ps = a.u64(4);
    uint64_t nSec = 0, nSec2 = 0, nSc = 0;
    uint64_t *ops = a.arr(0, nOps * (sizeof(glx_op) / 8)), *ptrs = a.arr(5, 0, &nSec), *widths = a.arr(6, 0, &nSec2), *scal = a.arr(7, 0, &nSc);
    if (a.ok && nSec != nSec2) a.fail("section arrays differ in length");
    if (!a.ok) return nullptr;
    std::vector<glx_section> secs(nSec);
    for (uint64_t i = 0; i < nSec; i++) { secs[i].ptr = (uint64_t *)(uintptr_t)ptrs[i]; secs[i].width = widths[i]; }
    glx_program prog = { (uint32_t)nOps, (uint32_t)nTmp, (const glx_op *)ops };
    glx_ctx ctx = { (uint32_t)nBits, (uint32_t)ps, (uint32_t)nSec, (uint32_t)nSc, secs.data(), scal };
    P2(env, pil2gl_eval_program_dev(&prog, &ctx, nullptr)); return mk_undefined(env);
}

// (devCol, dim, first, last) -> [row, v0, v1, v2] as BigInt, row = 2^64-1 when the column is zero on [first, last)
FN(FirstNonzeroRowDev) {
    Args a(env, info); uint64_t col = a.u64(0), dim = a.u64(1), first = a.u64(2), last = a.u64(3);
    if (!a.ok) return nullptr;
    uint64_t out[4] = { 0, 0, 0, 0 };
    P2(env, pil2gl_first_nonzero_row_dev((const uint64_t *)(uintptr_t)col, (uint32_t)dim, first, last, &out[0], &out[1], nullptr));
    napi_value arr; NAPI_CALL(env, napi_create_array_with_length(env, 4, &arr));
    for (uint32_t i = 0; i < 4; i++) { napi_value v; NAPI_CALL(env, napi_create_bigint_uint64(env, out[i], &v)); NAPI_CALL(env, napi_set_element(env, arr, i, v)); }
    return arr;
}

// worker-level operators (fft_worker.js:6-67) on a device block, in place
FN(FftBlockDev) {
    Args a(env, info); uint64_t buf = a.u64(0), startPos = a.u64(1), nPols = a.u64(2), nBits = a.u64(3), s = a.u64(4), blockBits = a.u64(5), layers = a.u64(6);
    if (!a.ok) return nullptr;
    P2(env, pil2gl_fft_block_dev((uint64_t *)(uintptr_t)buf, startPos, nPols, (uint32_t)nBits, (uint32_t)s, (uint32_t)blockBits, (uint32_t)layers, a.stream(7)));
    return mk_undefined(env);
}
FN(InterpolatePrepareBlockDev) {
    Args a(env, info); uint64_t buf = a.u64(0), width = a.u64(1), height = a.u64(2), start = a.u64(3), inc = a.u64(4);
    if (!a.ok) return nullptr;
    P2(env, pil2gl_interpolate_prepare_block_dev((uint64_t *)(uintptr_t)buf, width, height, start, inc, a.stream(5)));
    return mk_undefined(env);
}

static napi_value ModuleInit(napi_env env, napi_value exports) {
    struct { const char *name; napi_callback fn; } fns[] = {
        { "init", Init }, { "shutdown", Shutdown }, { "deviceInfo", DeviceInfo },
        { "devAlloc", DevAlloc }, { "devFree", DevFree }, { "devZero", DevZero }, { "devUpload", DevUpload }, { "devDownload", DevDownload }, { "sync", Sync },
        { "interpolate", Interpolate }, { "fft", Fft }, { "ifft", Ifft },
        { "interpolateDev", InterpolateDev }, { "fftDev", FftDev }, { "ifftDev", IfftDev },
        { "poseidon", Poseidon }, { "linearHashRows", LinearHashRows }, { "merkelizeLevel", MerkelizeLevel },
        { "merkleNumNodes", MerkleNumNodes }, { "merkelize", Merkelize }, { "merkelizeDev", MerkelizeDev }, { "groupProofDev", GroupProofDev }, { "groupProofsDev", GroupProofsDev },
        { "rootsFromGroupProofs", RootsFromGroupProofs }, { "spongeAbsorb", SpongeAbsorb },
        { "bn128Poseidon", Bn128Poseidon }, { "bn128SpongeAbsorb", Bn128SpongeAbsorb }, { "bn128LinearHashRows", Bn128LinearHashRows }, { "bn128MerkleNumNodes", Bn128MerkleNumNodes },
        { "bn128Merkelize", Bn128Merkelize }, { "bn128MerkelizeDev", Bn128MerkelizeDev }, { "bn128Convert", Bn128Convert },
        { "buildXDev", BuildXDev }, { "buildZhInvDev", BuildZhInvDev }, { "buildOneRowZerofierInvDev", BuildOneRowZerofierInvDev },
        { "buildFrameZerofierDev", BuildFrameZerofierDev }, { "computeQSplitDev", ComputeQSplitDev }, { "computeQSplitBrevDev", ComputeQSplitBrevDev }, { "extendCoefsBrevDev", ExtendCoefsBrevDev }, { "xDivXSubXiDev", XDivXSubXiDev },
        { "buildLevDev", BuildLevDev }, { "computeEvalsDev", ComputeEvalsDev }, { "gprodDev", GprodDev }, { "gsumDev", GsumDev }, { "h1h2Dev", H1H2Dev },
        { "rowsDotExtDev", RowsDotExtDev }, { "rowsDotExtMultiDev", RowsDotExtMultiDev }, { "friCombineDev", FriCombineDev }, { "colsDotExtDev", ColsDotExtDev }, { "colsDotExtMultiDev", ColsDotExtMultiDev }, { "synthFibonacciDev", SynthFibonacciDev },
        { "friFoldDev", FriFoldDev }, { "friTransposeDev", FriTransposeDev },
        { "friFold", FriFold }, { "friVerifyFold", FriVerifyFold }, { "friTranspose", FriTranspose }, { "evalProgramDev", EvalProgramDev }, { "firstNonzeroRowDev", FirstNonzeroRowDev },
        { "fftBlockDev", FftBlockDev }, { "interpolatePrepareBlockDev", InterpolatePrepareBlockDev },
    };
    for (auto &f : fns) {
        napi_value v;
        NAPI_CALL(env, napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v));
        NAPI_CALL(env, napi_set_named_property(env, exports, f.name, v));
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, ModuleInit)
