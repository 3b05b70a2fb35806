// Drop-in for the evaluator entry point of src/prover/prover_helpers.js:
//   callCalculateExps(stage, code, dom, ctx, parallelExec, useThreads, debug, global)   (prover_helpers.js:23)
// The op-list `code.code` is encoded into the binary form of include/pil2gl_expr.h and run on the GPU for every
// row of the domain; operand resolution follows getRef/setRef/evalMap (prover_helpers.js:109-259).
// ctx buffers (const_n, cm{s}_n/_ext, x_n/x_ext, Zi_ext, xDivXSubXi_ext, q_ext, f_ext) may be BigUint64Array or
// BigBuffer-like; referenced sections are staged to the device, destinations copied back.
"use strict";
const { addon, isDev, upload, download } = require("./native.js");
const P = 0xFFFFFFFF00000001n;
const OP = { add: 0, sub: 1, mul: 2, copy: 3 };
const TMP = 0, SEC = 1, SCALAR = 2;
const e = (a) => { let v = BigInt(a) % P; if (v < 0n) v += P; return v; };

function encode(code, dom, ctx, global) {
    const info = ctx.pilInfo;
    const sections = [];            // {name, buf, width, written}
    const secIndex = new Map();
    const scalars = [];
    function section(name, width, zi) {
        const key = zi === undefined ? name : name + "#" + zi;
        if (!secIndex.has(key)) { secIndex.set(key, sections.length); sections.push({ name, width, zi, written: false }); }
        return secIndex.get(key);
    }
    function scalar(v) {            // base element or [3] extension element -> word offset
        const off = scalars.length;
        if (Array.isArray(v)) { for (const c of v) scalars.push(e(c)); return [off, 3]; }
        scalars.push(e(v)); return [off, 1];
    }
    function cmRef(r) {             // evalMap, prover_helpers.js:220-259
        const p = info.cmPolsMap[r.id];
        const st = "cm" + p.stage;
        return { kind: SEC, dim: p.dim, section: section(st + "_" + dom, info.mapSectionsN[st]), prime: r.prime || 0, index: p.stagePos };
    }
    function ref(r, isDest) {
        switch (r.type) {
            case "tmp": return { kind: TMP, dim: r.dim, section: 0, prime: 0, index: r.id };
            case "$ret": return { kind: SEC, dim: r.dim, section: section("$ret", r.dim), prime: 0, index: 0 };      // calculateExps(..., ret = true)
            case "$arg": return { kind: SEC, dim: r.dim, section: section("$arg", r.dim), prime: 0, index: 0 };      // setPol: the caller's column, staged
            case "const": return { kind: SEC, dim: 1, section: section("const_" + dom, info.nConstants), prime: r.prime || 0, index: r.id };
            case "cm": return cmRef(r);
            case "q": if (dom !== "ext") throw new Error("Accessing q in domain n");
                return { kind: SEC, dim: r.dim, section: section("q_ext", info.qDim), prime: 0, index: 0 };
            case "f": if (dom !== "ext") throw new Error("Accessing q in domain n");
                return { kind: SEC, dim: 3, section: section("f_ext", 3), prime: 0, index: 0 };
            case "x": return { kind: SEC, dim: r.dim || 1, section: section("x_" + dom, r.dim || 1), prime: 0, index: 0 };
            case "Zi": {
                const boundary = info.boundaries[r.boundaryId];
                const ziIndex = boundary.name === "everyFrame"
                    ? info.boundaries.findIndex((b) => b.name === "everyFrame" && b.offsetMin === boundary.offsetMin && b.offsetMax === boundary.offsetMax)
                    : info.boundaries.findIndex((b) => b.name === boundary.name);
                if (ziIndex === -1) throw new Error("Something went wrong");
                return { kind: SEC, dim: 1, section: section("Zi_ext", 1, ziIndex), prime: 0, index: 0 };
            }
            case "xDivXSubXi": return { kind: SEC, dim: 3, section: section("xDivXSubXi_ext", 3 * info.openingPoints.length), prime: 0, index: 3 * r.id };
            case "number": { const [o, d] = scalar(BigInt(r.value)); return { kind: SCALAR, dim: d, section: 0, prime: 0, index: o }; }
            case "public": { const [o, d] = scalar(ctx.publics[r.id]); return { kind: SCALAR, dim: d, section: 0, prime: 0, index: o }; }
            case "challenge": { const [o] = scalar(ctx.challenges[r.stage - 1][r.stageId]); return { kind: SCALAR, dim: 3, section: 0, prime: 0, index: o }; }
            case "subproofValue": { const v = global ? ctx.subproofValues[r.subproofId][r.id] : ctx.subproofValues[r.id]; const [o, d] = scalar(v); return { kind: SCALAR, dim: d, section: 0, prime: 0, index: o }; }
            case "eval": { const [o] = scalar(ctx.evals[r.id]); return { kind: SCALAR, dim: 3, section: 0, prime: 0, index: o }; }
            default:
                if (!isDest && /^tree[0-9]+$/.test(r.type)) {       // verifierInfo.queryVerifier: a witness column by its place in the stage-N opening (stark_verify.js:245-246)
                    const st = "cm" + r.type.slice(4);
                    return { kind: SEC, dim: r.dim, section: section(st + "_" + dom, info.mapSectionsN[st]), prime: 0, index: r.treePos };
                }
                throw new Error((isDest ? "Invalid reference type set: " : "Invalid reference type get: ") + r.type);
        }
    }
    // glx_op: u32 op, u32 pad, 3 x glx_ref{u8 kind,u8 dim,u16 section,i32 prime,u32 index,u32 pad} = 56 bytes
    let nTmp = 0;
    for (const c of code) for (const r of [c.dest, ...c.src]) if (r.type === "tmp") nTmp = Math.max(nTmp, r.id + 1);
    // `muladd` (verifier code only: codegen.js:137-165, stark_verify.js:234) = F.add(F.mul(a, b), c): two ops through one more temporary
    const list = [];
    let hasMulAdd = false;
    for (const c of code) {
        if (c.op === "muladd") {
            hasMulAdd = true;
            const prod = { type: "tmp", id: nTmp, dim: Math.max(c.src[0].dim || 1, c.src[1].dim || 1) };
            list.push({ op: "mul", dest: prod, src: [c.src[0], c.src[1]] }, { op: "add", dest: c.dest, src: [prod, c.src[2]] });
        } else list.push(c);
    }
    if (hasMulAdd) nTmp++;
    const buf = new ArrayBuffer(list.length * 56), dv = new DataView(buf);
    const put = (o, r) => { dv.setUint8(o, r.kind); dv.setUint8(o + 1, r.dim); dv.setUint16(o + 2, r.section, true); dv.setInt32(o + 4, r.prime, true); dv.setUint32(o + 8, r.index, true); };
    for (let j = 0; j < list.length; j++) {
        const c = list[j], o = j * 56;
        if (!(c.op in OP)) throw new Error("Invalid op:" + c.op);
        dv.setUint32(o, OP[c.op], true);
        const d = ref(c.dest, true);
        if (d.kind === SCALAR) throw new Error("Invalid reference type set: " + c.dest.type);
        if (d.kind === SEC) sections[d.section].written = true;
        put(o + 8, d);
        put(o + 24, ref(c.src[0], false));
        if (c.op !== "copy") put(o + 40, ref(c.src[1], false));
    }
    return { ops: new BigUint64Array(buf), nOps: list.length, nTmp, sections, scalars: BigUint64Array.from(scalars.length ? scalars : [0n]) };
}

// calculateExps(ctx, code, dom, debug, ret, global)   (prover_helpers.js:31-72).  ret: the value the LAST op produced, for every row of
// the domain -- an array of BigInt (dim 1) or of [a, b, c] (dim 3) -- which is how the hints read a numerator / denominator / lookup
// column that is an expression rather than a committed polynomial (getHintField op "tmp", hints_helpers.js:32).  The last op's
// destination is redirected to a scratch column on the device; nothing else of the program changes.
// ret === "dev": the column stays in HBM and `onDev(devPtr, dim, rows)` reads what it needs of it (constraint checking).
function run(code, dom, ctx, global, ret, onDev) {
    let ops = code.code, retDim = 0;
    if (ret) {
        if (!ops.length) throw new Error("calculateExps: an empty program returns nothing");
        const last = ops[ops.length - 1];
        retDim = last.dest.dim || 1;
        // a temporary is simply redirected; any other destination is written as the program says AND copied out (compileCode returns getRef(dest))
        if (last.dest.type === "tmp") ops = ops.slice(0, -1).concat([{ op: last.op, dest: { type: "$ret", dim: retDim }, src: last.src }]);
        else ops = ops.concat([{ op: "copy", dest: { type: "$ret", dim: retDim }, src: [last.dest] }]);
    }
    const enc = encode(ops, dom, ctx, global);
    const nBits = dom === "n" ? ctx.nBits : ctx.nBitsExt;
    const rows = 2 ** nBits;
    const ptrs = new BigUint64Array(enc.sections.length), widths = new BigUint64Array(enc.sections.length), devs = [];
    let out;
    try {
        enc.sections.forEach((s, i) => {
            if (s.name === "$ret") { const d = addon.devAlloc(rows * s.width); devs.push(d); ptrs[i] = d; widths[i] = BigInt(s.width); return; }
            if (s.name === "$arg") { const d = addon.devAlloc(rows * s.width); devs.push(d); addon.devUpload(d, 0, ctx.$arg); ptrs[i] = d; widths[i] = BigInt(s.width); return; }
            const host = ctx[s.name];
            if (!host) throw new Error("ctx." + s.name + " is not allocated");
            const n = rows * s.width, base = s.zi === undefined ? 0 : s.zi * rows;
            if (isDev(host)) { ptrs[i] = host.addr(base); widths[i] = BigInt(s.width); devs.push(null); return; }     // resident section
            const d = addon.devAlloc(n);
            devs.push(d);
            if (base === 0 && n === host.length) upload(d, host, n);
            else addon.devUpload(d, 0, host.slice(base, base + n));
            ptrs[i] = d; widths[i] = BigInt(s.width);
        });
        addon.evalProgramDev(enc.ops, enc.nOps, enc.nTmp, nBits, dom === "n" ? 0 : ctx.extendBits, ptrs, widths, enc.scalars);
        enc.sections.forEach((s, i) => {
            if (s.name === "$ret" && ret === "dev") out = onDev(devs[i], retDim, rows);
            else if (s.name === "$ret") {
                const flat = new BigUint64Array(rows * retDim);
                addon.devDownload(flat, devs[i], 0);
                out = new Array(rows);
                if (retDim === 1) for (let r = 0; r < rows; r++) out[r] = flat[r];
                else for (let r = 0; r < rows; r++) out[r] = [flat[3 * r], flat[3 * r + 1], flat[3 * r + 2]];
            } else if (s.written && devs[i] !== null) {
                const base = s.zi === undefined ? 0 : s.zi * rows;
                if (base === 0) download(ctx[s.name], devs[i], rows * s.width);
                else { const t = new BigUint64Array(rows * s.width); addon.devDownload(t, devs[i], 0); ctx[s.name].set(t, base); }
            }
        });
    } finally {
        for (const d of devs) if (d !== null) addon.devFree(d);
    }
    return out;
}
// debug = true (prover_helpers.js:46-70): `code` is one constraint (expressionsInfo.constraints[i]: an op-list whose last op holds the
// constraint's value, with its boundary and source line).  The reference walks the rows of the boundary and records the FIRST row whose
// value is not zero; here the op-list runs on the whole domain on the device and pil2gl_first_nonzero_row_dev finds that row: the same
// message lands in ctx.errors, and only the row index and its value cross PCIe.
function checkConstraint(ctx, code, dom, global) {
    const N = dom === "n" ? 2 ** ctx.nBits : 2 ** ctx.nBitsExt;
    let first, last;
    if (code.boundary === "everyRow") { first = 0; last = N; }
    else if (code.boundary === "firstRow" || code.boundary === "finalProof") { first = 0; last = 1; }
    else if (code.boundary === "lastRow") { first = N - 1; last = N; }
    else if (code.boundary === "everyFrame") { first = code.offsetMin; last = N - code.offsetMax; }
    else throw new Error("Invalid boundary: " + code.boundary);
    if (!ctx.errors) ctx.errors = [];
    if (last <= first) return;
    const hit = run(code, dom, ctx, global, "dev", (dev, dim) => {
        const [row, v0, v1, v2] = addon.firstNonzeroRowDev(dev, dim, first, last);
        return row === 0xFFFFFFFFFFFFFFFFn ? null : { row, val: dim === 1 ? v0 : [v0, v1, v2] };
    });
    if (hit) ctx.errors.push(`${code.line}: identity does not match w=${hit.row} val=${dim1or3(hit.val)} `);
}
const dim1or3 = (v) => (Array.isArray(v) ? v.map((c) => c.toString(10)) : v.toString(10));     // F3g.toString, f3g.js:313-320
module.exports.calculateExps = function calculateExps(ctx, code, dom, debug, ret, global) {
    if (debug) return checkConstraint(ctx, code, dom, !!global);
    return run(code, dom, ctx, !!global, !!ret);
};
// calculateExpAtPoint(ctx, code, i) (prover_helpers.js:74-80): the value of the op-list's last op at ONE row of the trace domain.  The
// program runs on the whole domain (rows are independent on the device; one row is not cheaper) and one value is read back.
module.exports.calculateExpAtPoint = function calculateExpAtPoint(ctx, code, i) {
    return run(code, "n", ctx, false, "dev", (dev, dim) => {
        const t = new BigUint64Array(dim);
        addon.devDownload(t, dev, Number(i) * dim);
        return dim === 1 ? t[0] : [t[0], t[1], t[2]];
    });
};
module.exports.calculateExpressionAtRow = function calculateExpressionAtRow(ctx, expId, row) {       // prover_helpers.js:18-21
    const expressionCode = ctx.expressionsInfo.expressionsCode.find((e) => e && e.expId === expId);
    return module.exports.calculateExpAtPoint(ctx, expressionCode.code, row);
};
// calculateExpression(ctx, expId)   (prover_helpers.js:10-16): the expression's column on the trace domain
module.exports.calculateExpression = function calculateExpression(ctx, expId, debug = false) {
    const expressionCode = ctx.expressionsInfo.expressionsCode.find((e) => e && e.expId === expId);
    if (!expressionCode) throw new Error("expression " + expId + " not found");
    return module.exports.calculateExps(ctx, expressionCode.code, "n", debug, true);
};

// getPolRef / getPol / setPol / getFixedPol (prover_helpers.js:261-358): one column of a stage buffer as an array of BigInt (dim 1) or
// [a, b, c] (dim 3).  The reference walks the column with getElement / setElement, one element at a time; on a buffer that lives in
// HBM (native.js DevBuffer) that would be one transfer per row, so there the column is gathered / scattered by a one-op program of
// the evaluator and crosses PCIe once.  Host buffers are walked directly.
module.exports.getPolRef = function getPolRef(ctx, idPol, dom, isFixed = false) {
    if (!["n", "ext"].includes(dom)) throw new Error("invalid stage");
    const deg = dom === "ext" ? 2 ** ctx.nBitsExt : 2 ** ctx.nBits;
    const p = isFixed ? (ctx.pilInfo.constPolsMap || [])[idPol] || { dim: 1 } : ctx.pilInfo.cmPolsMap[idPol];
    const st = isFixed ? "const" : "cm" + p.stage;
    const stage = st + "_" + dom;
    return { stage, buffer: ctx[stage], deg, offset: isFixed ? idPol : p.stagePos, size: isFixed ? ctx.pilInfo.nConstants : ctx.pilInfo.mapSectionsN[st], dim: p.dim };
};
module.exports.getPol = function getPol(ctx, idPol, dom, isFixed = false) {
    const p = module.exports.getPolRef(ctx, idPol, dom, isFixed);
    if (p.dim !== 1 && p.dim !== 3) throw new Error("invalid dim" + p.dim);
    if (isDev(p.buffer))
        return run({ code: [{ op: "copy", dest: { type: "tmp", id: 0, dim: p.dim }, src: [{ type: isFixed ? "const" : "cm", id: idPol, prime: 0, dim: p.dim }] }] }, dom, ctx, false, true);
    const res = new Array(p.deg), b = p.buffer, flat = b instanceof BigUint64Array;
    const at = flat ? (k) => b[k] : (k) => b.getElement(k);
    if (p.dim === 1) for (let i = 0; i < p.deg; i++) res[i] = at(p.offset + i * p.size);
    else for (let i = 0; i < p.deg; i++) { const k = p.offset + i * p.size; res[i] = [at(k), at(k + 1), at(k + 2)]; }
    return res;
};
module.exports.getFixedPol = function getFixedPol(ctx, idPol) { return module.exports.getPol(ctx, idPol, "n", true); };
module.exports.setPol = function setPol(ctx, idPol, pol, dom, options) {
    const p = module.exports.getPolRef(ctx, idPol, dom);
    if (p.dim !== 1 && p.dim !== 3) throw new Error("invalid dim" + p.dim);
    const word = (v) => { let x = BigInt(v) % P; if (x < 0n) x += P; return x; };
    if (isDev(p.buffer)) {
        const col = new BigUint64Array(p.deg * p.dim);
        if (p.dim === 1) for (let i = 0; i < p.deg; i++) col[i] = word(pol[i]);
        else for (let i = 0; i < p.deg; i++) { const v = Array.isArray(pol[i]) ? pol[i] : [pol[i], 0n, 0n]; col[3 * i] = word(v[0]); col[3 * i + 1] = word(v[1]); col[3 * i + 2] = word(v[2]); }
        ctx.$arg = col;
        try { run({ code: [{ op: "copy", dest: { type: "cm", id: idPol, dim: p.dim }, src: [{ type: "$arg", dim: p.dim }] }] }, dom, ctx, false, false); }
        finally { delete ctx.$arg; }
        return;
    }
    const b = p.buffer, flat = b instanceof BigUint64Array;
    const put = flat ? (k, v) => { b[k] = word(v); } : (k, v) => b.setElement(k, word(v));
    if (p.dim === 1) for (let i = 0; i < p.deg; i++) put(p.offset + i * p.size, pol[i]);
    else for (let i = 0; i < p.deg; i++) {
        const v = Array.isArray(pol[i]) ? pol[i] : [pol[i], 0n, 0n], k = p.offset + i * p.size;
        put(k, v[0]); put(k + 1, v[1]); put(k + 2, v[2]);
    }
};

module.exports.callCalculateExps = async function callCalculateExps(stage, code, dom, ctx, parallelExec, useThreads, debug, global = false) {
    module.exports.calculateExps(ctx, code, dom, debug, false, global);                            // prover_helpers.js:23-29 (no worker pool here)
};
module.exports.encode = encode;
