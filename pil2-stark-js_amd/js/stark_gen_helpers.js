// Drop-ins for the data-parallel stages of src/stark/stark_gen_helpers.js, same names, `ctx` object and return values:
//   buildXTables(ctx)            the x_n / x_ext loops of initProverStark             stark_gen_helpers.js:111-116,139-144
//   computeQStark(ctx, options)                                                       stark_gen_helpers.js:168-208
//   computeEvalsStark(ctx, options)                                                   stark_gen_helpers.js:210-273
//   computeFRIStark(ctx, options)                                                     stark_gen_helpers.js:275-335
//   calculateHashStark(ctx, inputs)   (starkStruct.hashCommits)                       stark_gen_helpers.js:442-461
// ctx is the reference's prover context (ctx.q_ext, ctx.cm<k>_ext, ctx.const_ext, ctx.x_ext, ctx.xDivXSubXi_ext, ctx.f_ext
// are BigBuffers / BigUint64Arrays; ctx.MH, ctx.trees, ctx.challenges, ctx.pilInfo, ctx.expressionsInfo as the reference
// builds them).  Each stage uploads what it reads, runs on the device and stores what the reference stores.
"use strict";
const { addon, isDev, upload, download } = require("./native.js");
const { callCalculateExps } = require("./prover_helpers.js");

const P = 0xFFFFFFFF00000001n;
const SHIFT = 7n;
const W32 = 7277203076849721926n;                  // F.w[32] (f3g.js:40)
const mulm = (a, b) => (a * b) % P;
function powm(a, e) { let r = 1n; a %= P; while (e > 0n) { if (e & 1n) r = mulm(r, a); a = mulm(a, a); e >>= 1n; } return r; }
const invm = (a) => powm(a, P - 2n);
const rootOfUnity = (bits) => powm(W32, 1n << BigInt(32 - bits));
function e3mul(a, b) {      // f3g.js:94-102
    const A = mulm(a[0] + a[1], b[0] + b[1]), B = mulm(a[0] + a[2], b[0] + b[2]), C = mulm(a[1] + a[2], b[1] + b[2]);
    const D = mulm(a[0], b[0]), E = mulm(a[1], b[1]), F = mulm(a[2], b[2]), G = (D - E + P) % P;
    return [(C + G - F + P) % P, (((A + C - E - E - D) % P) + 2n * P) % P, (B - G + P) % P];
}
const e3scale = (a, s) => [mulm(a[0], s), mulm(a[1], s), mulm(a[2], s)];
const asE3 = (v) => (Array.isArray(v) ? v.map(BigInt) : [BigInt(v), 0n, 0n]);

function devTmp(n) { return addon.devAlloc(n); }

module.exports.buildXTables = function buildXTables(ctx) {
    for (const [buf, bits, shift] of [[ctx.x_n, ctx.nBits, 1n], [ctx.x_ext, ctx.nBitsExt, SHIFT]]) {
        if (!buf) continue;
        const n = 1 << bits;
        if (isDev(buf)) { addon.buildXDev(bits, shift, buf.ptr); continue; }
        const d = devTmp(n);
        try { addon.buildXDev(bits, shift, d); download(buf, d, n); } finally { addon.devFree(d); }
    }
};

module.exports.computeQStark = async function computeQStark(ctx, options) {
    const qStage = ctx.pilInfo.nStages + 1, qDim = ctx.pilInfo.qDim, qDeg = ctx.pilInfo.qDeg, extN = ctx.extN;
    const qIn = ctx.q_ext, qOut = ctx["cm" + qStage + "_ext"];
    const dQ = isDev(qIn) ? null : devTmp(qDim * extN), dQ1 = devTmp(qDim * extN), dQ2 = isDev(qOut) ? null : devTmp(qDim * qDeg * extN);
    const dC = devTmp(qDim * qDeg * ctx.N);
    try {
        if (dQ !== null) upload(dQ, qIn, qDim * extN);
        const pQ = dQ !== null ? dQ : qIn.ptr, pQ2 = dQ2 !== null ? dQ2 : qOut.ptr;
        addon.ifftDev(pQ, qDim, ctx.nBitsExt, dQ1);                                         // :177
        // :179-192, same values: the pieces as their N coefficient rows (bit-reversed order), extended from there -- the zero-padded
        // 2^nBitsExt-row matrix of the reference is never built and its first extendBits stages are not run
        addon.computeQSplitBrevDev(dQ1, ctx.nBits, ctx.nBitsExt, qDim, qDeg, dC);
        addon.extendCoefsBrevDev(dC, qDim * qDeg, ctx.nBits, pQ2, ctx.nBitsExt);
        if (dQ2 !== null) download(qOut, dQ2, qDim * qDeg * extN);
    } finally { if (dQ !== null) addon.devFree(dQ); addon.devFree(dQ1); addon.devFree(dC); if (dQ2 !== null) addon.devFree(dQ2); }
    const nPolsQ = ctx.pilInfo.mapSectionsN["cm" + qStage] || 0;
    ctx.trees[qStage] = await ctx.MH.merkelize(ctx["cm" + qStage + "_ext"], nPolsQ, extN);   // :197
    return [ctx.MH.root(ctx.trees[qStage])];
};

function openingXi(ctx, opening, divideByShift) {
    let w = 1n;
    const wN = rootOfUnity(ctx.nBits);
    for (let j = 0; j < Math.abs(Number(opening)); ++j) w = mulm(w, wN);
    if (Number(opening) < 0) w = invm(w);
    const xiChallenge = asE3(ctx.challenges[ctx.pilInfo.nStages + 1][0]);
    let xi = e3scale(xiChallenge, w);
    if (divideByShift) xi = e3scale(xi, invm(SHIFT));
    return xi;
}

module.exports.computeEvalsStark = async function computeEvalsStark(ctx, options) {
    const N = ctx.N, nOpen = ctx.pilInfo.openingPoints.length;
    const ptrs = [];
    try {
        const levs = new BigUint64Array(nOpen);
        for (let i = 0; i < nOpen; i++) {                                                    // :216-231
            const xi = openingXi(ctx, ctx.pilInfo.openingPoints[i], true);
            const d = devTmp(3 * N); ptrs.push(d);
            addon.buildLevDev(ctx.nBits, BigUint64Array.from(xi), d);
            levs[i] = d;
        }
        const uploaded = new Map();                                                          // each section once
        const devOf = (name, buf, words) => {
            if (isDev(buf)) return buf.ptr;                                                  // resident section
            if (!uploaded.has(name)) { const d = devTmp(words); ptrs.push(d); upload(d, buf, words); uploaded.set(name, d); }
            return uploaded.get(name);
        };
        const nEv = ctx.pilInfo.evMap.length;
        const place = (ev) => {                                                              // :233-247
            if (ev.type == "const") return { name: "const_ext", size: ctx.pilInfo.nConstants, offset: ev.id, dim: 1 };
            if (ev.type == "cm") { const p = ctx.pilInfo.cmPolsMap[ev.id]; return { name: "cm" + p.stage + "_ext", size: ctx.pilInfo.mapSectionsN["cm" + p.stage], offset: p.stagePos, dim: p.dim }; }
            throw new Error("Invalid ev type: " + ev.type);
        };
        const openIdx = (ev) => ctx.pilInfo.openingPoints.findIndex((p) => p === ev.prime);
        ctx.evals = [];
        if (nOpen <= 64) {                                     // (a sweep of the library weighs four opening points; more of them take more sweeps inside the call)
            // eval_e = sum_k v_e[k << b] LEv[k] (:250-264) for EVERY column of a section and every opening in one sweep of
            // the section (pil2gl_cols_dot_ext_dev); a dim-3 polynomial q0 + q1 x + q2 x^2 is assembled from its base columns
            const sums = new Map(), secs = [];
            for (const ev of ctx.pilInfo.evMap) {
                const pl = place(ev);
                if (sums.has(pl.name)) continue;
                sums.set(pl.name, new BigUint64Array(nOpen * pl.size * 3)); secs.push(pl);
            }
            for (let g0 = 0; g0 < secs.length; g0 += 8) {              // the matrices of a group in one sweep of the weights (pil2gl.h)
                const grp = secs.slice(g0, g0 + 8);
                addon.colsDotExtMultiDev(BigUint64Array.from(grp.map((pl) => BigInt(devOf(pl.name, ctx[pl.name], pl.size * ctx.extN)))), BigUint64Array.from(grp.map((pl) => BigInt(pl.size))),
                    N, 1 << ctx.extendBits, levs, grp.map((pl) => sums.get(pl.name)));
            }
            const mulX = (a) => [a[2], (a[0] + a[2]) % P, a[1]];                             // times x in F[x]/(x^3 - x - 1), f3g.js:94-102
            for (let i = 0; i < nEv; i++) {
                const ev = ctx.pilInfo.evMap[i], pl = place(ev), out = sums.get(pl.name), li = openIdx(ev);
                const col = (c) => { const o = 3 * (li * pl.size + c); return [out[o], out[o + 1], out[o + 2]]; };
                let acc = col(pl.offset);
                for (let t = 1; t < pl.dim; t++) { let term = col(pl.offset + t); for (let k = 0; k < t; k++) term = mulX(term); acc = acc.map((v, j) => (v + term[j]) % P); }
                ctx.evals[i] = acc;
            }
        } else {
            const descs = new BigUint64Array(5 * nEv);
            for (let i = 0; i < nEv; i++) {
                const ev = ctx.pilInfo.evMap[i], pl = place(ev);
                descs.set([devOf(pl.name, ctx[pl.name], pl.size * ctx.extN), BigInt(pl.size), BigInt(pl.offset), BigInt(pl.dim), BigInt(openIdx(ev))], 5 * i);
            }
            const out = new BigUint64Array(3 * nEv);
            addon.computeEvalsDev(descs, nEv, ctx.nBits, ctx.extendBits, levs, out);         // :248-264
            for (let i = 0; i < nEv; i++) ctx.evals[i] = [out[3 * i], out[3 * i + 1], out[3 * i + 2]];
        }
    } finally { for (const d of ptrs) addon.devFree(d); }
    if (ctx.pilInfo.starkStruct.hashCommits) return [await module.exports.calculateHashStark(ctx, ctx.evals)];      // :267-272
    return ctx.evals;
};

// stark_gen_helpers.js:442-461: the hash of a list of values = the state of a fresh transcript that absorbed them
// (used for the publics, the evaluations and the last FRI polynomial when starkStruct.hashCommits is set)
module.exports.calculateHashStark = async function calculateHashStark(ctx, inputs) {
    const ss = ctx.pilInfo.starkStruct;
    let transcript;
    if (ss.verificationHashType === "GL") {
        transcript = new (require("./transcript.js"))(require("./poseidon.js")());
    } else if (ss.verificationHashType === "BN128") {
        transcript = new (require("./transcript_bn128.js"))(ss.merkleTreeCustom ? ss.merkleTreeArity : 16);
    } else throw new Error("Invalid Hash Type: " + ss.verificationHashType);
    for (let i = 0; i < inputs.length; i++) transcript.put(inputs[i]);
    return transcript.getState();
};

// The FRI polynomial the friExp op-list computes (friPolinomial.js:26-50) is, per opening o and evMap order
// j = 1..n_o,  F_o = sum_j (p_j - ev_j) vf2^(n_o-j)  and  f = Horner in vf1 over the openings of F_o xDivXSubXi_o: per section
// one extension weight per base column (pil2gl_rows_dot_ext_dev), then one combine kernel.  Same field elements as the
// op-list, about a tenth of its multiplications.  The Horner order is the reference's: friPolinomial.js:42 walks
// Object.keys(friExps) -- an object's integer-like keys come first, ascending, then the others ("-1") as they were
// inserted, i.e. as the opening first appears in evMap -- which with a previous-row opening ([-1, 0, 1]) is 0, 1, -1
// and NOT the order of openingPoints; everything here is indexed by the position in openingPoints, `order` lists the
// terms.  Only for device-resident sections and the layout the formula covers (the openings of evMap are exactly
// openingPoints, at most four); otherwise the op-list runs.
function friOpeningOrder(info) {
    const keys = {};
    for (const ev of info.evMap) if (!(ev.prime in keys)) keys[ev.prime] = true;       // the same object, so the same key order
    return Object.keys(keys).map(Number);
}
module.exports.friOpeningOrder = friOpeningOrder;
function friPolynomialAsRowSums(ctx) {
    const info = ctx.pilInfo, nOpen = info.openingPoints.length, extN = ctx.extN;
    const openings = info.openingPoints, hornerOrder = friOpeningOrder(info);
    if (nOpen > 4 || hornerOrder.length !== nOpen || hornerOrder.some((o) => openings.indexOf(o) < 0)) return false;
    if (!isDev(ctx.f_ext) || !isDev(ctx.xDivXSubXi_ext)) return false;
    const vf1 = asE3(ctx.challenges[info.nStages + 2][0]), vf2 = asE3(ctx.challenges[info.nStages + 2][1]);
    const mulX = (a) => [a[2], (a[0] + a[2]) % P, a[1]];
    const coefs = new Map(), K = new BigUint64Array(3 * nOpen);
    for (let oi = 0; oi < nOpen; oi++) {
        const terms = []; info.evMap.forEach((ev, i) => { if (ev.prime === openings[oi]) terms.push(i); });
        const ws = []; let w = [1n, 0n, 0n];
        for (let k = 0; k < terms.length; k++) { ws.push(w); w = e3mul(w, vf2); }
        let kacc = [0n, 0n, 0n];
        terms.forEach((i, j) => {
            const ev = info.evMap[i], W = ws[terms.length - 1 - j];
            const t = e3mul(asE3(ctx.evals[i]), W); kacc = kacc.map((v, c) => (v + t[c]) % P);
            let name, width, col, dim;
            if (ev.type == "const") { name = "const_ext"; width = info.nConstants; col = ev.id; dim = 1; }
            else { const p = info.cmPolsMap[ev.id]; name = "cm" + p.stage + "_ext"; width = info.mapSectionsN["cm" + p.stage]; col = p.stagePos; dim = p.dim; }
            if (!coefs.has(name)) coefs.set(name, { width, c: new BigUint64Array(nOpen * width * 3) });
            const c = coefs.get(name).c;
            let Wt = W;
            for (let t2 = 0; t2 < dim; t2++) { const o = 3 * (oi * width + col + t2); for (let q = 0; q < 3; q++) c[o + q] = (c[o + q] + Wt[q]) % P; Wt = mulX(Wt); }
        });
        K.set(kacc, 3 * oi);
    }
    for (const name of coefs.keys()) if (!isDev(ctx[name])) return false;
    const acc = devTmp(3 * nOpen * extN);
    try {
        // all the matrices the polynomial reads in ONE pass when they fit the matrix-core kernel side by side (pil2gl.h)
        const names = [...coefs.keys()];
        addon.rowsDotExtMultiDev(BigUint64Array.from(names.map((nm) => BigInt(ctx[nm].ptr))), BigUint64Array.from(names.map((nm) => BigInt(coefs.get(nm).width))),
            extN, names.map((nm) => coefs.get(nm).c), nOpen, acc, 0);
        addon.friCombineDev(acc, K, BigUint64Array.from(vf1), ctx.xDivXSubXi_ext.ptr, nOpen, extN, ctx.f_ext.ptr,
            BigUint64Array.from(hornerOrder.map((o) => BigInt(openings.indexOf(o)))));
    } finally { addon.devFree(acc); }
    return true;
}

module.exports.computeFRIStark = async function computeFRIStark(ctx, options) {
    const stage = ctx.pilInfo.nStages + 2, nOpen = ctx.pilInfo.openingPoints.length, extN = ctx.extN;
    ctx.friPol = []; ctx.friProof = []; ctx.friTrees = [];
    const s0_trees = [];
    for (let i = 0; i < ctx.pilInfo.nStages + 1; ++i) s0_trees.push(ctx.trees[i + 1]);
    s0_trees.push(ctx.constTree);
    ctx.friTrees[0] = s0_trees;
    ctx.friProof[0] = {};
    const xRes = isDev(ctx.xDivXSubXi_ext);
    const dX = xRes ? ctx.xDivXSubXi_ext.ptr : devTmp(3 * extN * nOpen);
    try {
        for (let i = 0; i < nOpen; i++)                                                      // :293-322
            addon.xDivXSubXiDev(ctx.nBitsExt, BigUint64Array.from(openingXi(ctx, ctx.pilInfo.openingPoints[i], false)), nOpen, i, dX);
        if (!xRes) download(ctx.xDivXSubXi_ext, dX, 3 * extN * nOpen);
    } finally { if (!xRes) addon.devFree(dX); }
    if (!friPolynomialAsRowSums(ctx))
        await callCalculateExps(stage, ctx.expressionsInfo.expressionsCode.find((e) => e.expId === ctx.pilInfo.friExpId).code, "ext", ctx,
                                options.parallelExec, options.useThreads, false);           // :324
    if (isDev(ctx.f_ext)) { ctx.friPol[0] = ctx.f_ext; return; }     // resident: FRI.fold takes the device polynomial as it is
    ctx.friPol[0] = new Array(extN);
    for (let i = 0; i < extN; i++) {
        const g = (k) => (ctx.f_ext instanceof BigUint64Array ? ctx.f_ext[k] : ctx.f_ext.getElement(k));
        ctx.friPol[0][i] = [g(i * 3), g(i * 3 + 1), g(i * 3 + 2)];
    }
};
