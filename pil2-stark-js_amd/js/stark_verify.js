// Drop-in for src/stark/stark_verify.js and src/stark/calculateTranscriptVerify.js, same exports and argument lists:
//   starkVerify(proof, publics, constRoot, challenges, starkInfo, verifierInfo, options) -> Promise<boolean>     stark_verify.js:8-218
//   starkVerify.executeCode(F, ctx, code, global)                                                                 stark_verify.js:222-298
//   calculateTranscript(F, starkInfo, proof, publics, constRoot, options), calculateFRIQueries(starkInfo, challenge, options)
// What is data-parallel runs on the device in batches (SURVEY.md 8 row f4): the openings of one tree for ALL queries in one call
// (MH.verifyGroupProofs instead of the per-query loop of :165-178), the query program verifierInfo.queryVerifier for all queries
// as one evaluator launch on an nQueries-row domain whose section rows are the opened values (:180-215), and the FRI layers
// through the drop-in FRI.verify (fri.js:107-174).  The transcript, the zerofier values and the evaluation identity are a few
// hundred host-side BigInt operations, as in the reference.  `F` is accepted and ignored (the field is Goldilocks).
"use strict";
const FRI = require("./fri.js");
const buildMerkleHashGL = require("./merklehash_p.js");
const buildMerkleHashBN128 = require("./merklehash_bn128_p.js");
const getPoseidon = require("./poseidon.js");
const Transcript = require("./transcript.js");
const TranscriptBN128 = require("./transcript_bn128.js");
const { calculateHashStark } = require("./stark_gen_helpers.js");
const { calculateExps } = require("./prover_helpers.js");

const P = 0xFFFFFFFF00000001n, SHIFT = 7n, W32 = 7277203076849721926n;
const m = (a) => { a %= P; return a < 0n ? a + P : a; };
const powm = (a, e) => { let r = 1n; a = m(a); while (e > 0n) { if (e & 1n) r = r * a % P; a = a * a % P; e >>= 1n; } return r; };
const invm = (a) => powm(a, P - 2n);
const rootOfUnity = (bits) => powm(W32, 1n << BigInt(32 - bits));
const is3 = (a) => Array.isArray(a);
// cubic extension x^3 = x + 1 (f3g.js:47-172), operands a base element (BigInt) or a triple
function add(a, b) { if (is3(a)) return is3(b) ? [m(a[0] + b[0]), m(a[1] + b[1]), m(a[2] + b[2])] : [m(a[0] + b), a[1], a[2]]; return is3(b) ? [m(a + b[0]), b[1], b[2]] : m(a + b); }
function sub(a, b) { if (is3(a)) return is3(b) ? [m(a[0] - b[0]), m(a[1] - b[1]), m(a[2] - b[2])] : [m(a[0] - b), a[1], a[2]]; return is3(b) ? [m(a - b[0]), m(-b[1]), m(-b[2])] : m(a - b); }
function mul(a, b) {
    if (!is3(a)) return is3(b) ? [m(a * b[0]), m(a * b[1]), m(a * b[2])] : m(a * b);
    if (!is3(b)) return [m(a[0] * b), m(a[1] * b), m(a[2] * b)];
    const A = m((a[0] + a[1]) * (b[0] + b[1])), B = m((a[0] + a[2]) * (b[0] + b[2])), C = m((a[1] + a[2]) * (b[1] + b[2]));
    const D = m(a[0] * b[0]), E = m(a[1] * b[1]), F = m(a[2] * b[2]), G = m(D - E);
    return [m(C + G - F), m(A + C - E - E - D), m(B - G)];
}
function inv(a) {
    if (!is3(a)) return invm(a);
    const aa = m(a[0] * a[0]), ac = m(a[0] * a[2]), ba = m(a[1] * a[0]), bb = m(a[1] * a[1]), bc = m(a[1] * a[2]), cc = m(a[2] * a[2]);
    const t = m(-aa * a[0] - 2n * aa * a[2] + 3n * ba * a[2] + ba * a[1] - ac * a[2] - bb * a[1] + bc * a[2] - cc * a[2]);
    const ti = invm(t);
    return [m((-aa - 2n * ac + bc + bb - cc) * ti), m((ba - cc) * ti), m((-bb + ac + cc) * ti)];      // adjugate over the norm, f3g.js:136-172
}
function exp3(a, e) { let r = [1n, 0n, 0n]; let b = a; while (e > 0n) { if (e & 1n) r = mul(r, b); b = mul(b, b); e >>= 1n; } return r; }
const big = (v) => (Array.isArray(v) ? v.map(big) : BigInt(v));
const eq = (a, b) => { const x = is3(a) ? a : [a, 0n, 0n], y = is3(b) ? b : [b, 0n, 0n]; return m(x[0] - y[0]) === 0n && m(x[1] - y[1]) === 0n && m(x[2] - y[2]) === 0n; };

function newTranscript(ss) {
    if (ss.verificationHashType === "GL") return new Transcript(getPoseidon());
    if (ss.verificationHashType === "BN128") return new TranscriptBN128(ss.merkleTreeCustom ? ss.merkleTreeArity : 16);
    throw new Error("Invalid Hash Type: " + ss.verificationHashType);
}

// calculateTranscriptVerify.js:7-103
async function calculateTranscript(F, starkInfo, proof, publics, constRoot, options) {
    const ss = starkInfo.starkStruct, transcript = newTranscript(ss), ctx = { pilInfo: starkInfo };
    const challenges = [];
    const absorb = async (list) => { if (!ss.hashCommits) transcript.put(list); else transcript.put(await calculateHashStark(ctx, list)); };
    transcript.put(constRoot);
    await absorb(publics);
    for (let i = 0; i < starkInfo.nStages; i++) {
        const stage = i + 1, n = starkInfo.challengesMap.filter((c) => c.stage === stage).length;
        challenges[stage - 1] = [];
        for (let j = 0; j < n; j++) challenges[stage - 1][j] = transcript.getField();
        transcript.put(proof["root" + stage]);
    }
    const qStep = starkInfo.nStages;
    challenges[qStep] = [transcript.getField()];
    transcript.put(proof["root" + (qStep + 1)]);
    challenges[qStep + 1] = [transcript.getField()];
    await absorb(proof.evals);
    challenges[qStep + 2] = [transcript.getField(), transcript.getField()];
    const challengesFRISteps = [];
    for (let step = 0; step < ss.steps.length; step++) {
        challengesFRISteps[step] = transcript.getField();
        if (step < ss.steps.length - 1) transcript.put(proof.fri[step + 1].root);
        else await absorb(proof.fri[proof.fri.length - 1]);
    }
    challengesFRISteps[ss.steps.length] = transcript.getField();
    return { challenges, challengesFRISteps };
}
// calculateTranscriptVerify.js:106-125
async function calculateFRIQueries(starkInfo, challenge, options) {
    const ss = starkInfo.starkStruct, t = newTranscript(ss);
    t.put(challenge);
    return t.getPermutations(ss.nQueries, ss.steps[0].nBits);
}

// stark_verify.js:222-298 on host integers
function executeCode(F, ctx, code, global) {
    const tmp = [];
    const get = (r) => {
        if (r.type.startsWith("tree")) { const a = ctx[r.type]; return r.dim === 1 ? BigInt(a[r.treePos]) : a.slice(r.treePos, r.treePos + 3).map(BigInt); }
        switch (r.type) {
            case "tmp": return tmp[r.id];
            case "const": return BigInt(ctx.consts[r.id]);
            case "eval": return big(ctx.evals[r.id]);
            case "number": return m(BigInt(r.value));
            case "public": return BigInt(ctx.publics[r.id]);
            case "challenge": return big(ctx.challenges[r.stage - 1][r.stageId]);
            case "subproofValue": return big(global ? ctx.subproofValues[r.subproofId][r.id] : ctx.subproofValues[r.id]);
            case "xDivXSubXi": return ctx.xDivXSubXi[r.id];
            case "x": return big(ctx.challenges[ctx.starkInfo.nStages + 1][0]);
            case "Zi": {
                const b = ctx.starkInfo.boundaries[r.boundaryId];
                if (b.name === "everyRow") return ctx.Z;
                if (b.name === "firstRow") return ctx.Z_fr;
                if (b.name === "lastRow") return ctx.Z_lr;
                if (b.name === "everyFrame") return ctx["Z_frame" + ctx.starkInfo.boundaries.filter((x) => x.name === "everyFrame").findIndex((x) => x.offsetMin === b.offsetMin && x.offsetMax === b.offsetMax)];
                throw new Error("Invalid boundary: " + b.name);
            }
            default: throw new Error("Invalid reference type get: " + r.type);
        }
    };
    for (const c of code) {
        const s = c.src.map(get);
        let res;
        switch (c.op) {
            case "add": res = add(s[0], s[1]); break;
            case "sub": res = sub(s[0], s[1]); break;
            case "mul": res = mul(s[0], s[1]); break;
            case "muladd": res = add(mul(s[0], s[1]), s[2]); break;
            case "copy": res = s[0]; break;
            default: throw new Error("Invalid op:" + c.op);
        }
        if (c.dest.type !== "tmp") throw new Error("Invalid reference type set: " + c.dest.type);
        tmp[c.dest.id] = res;
    }
    return get(code[code.length - 1].dest);
}

async function starkVerify(proof, publics, constRoot, challenges, starkInfo, verifierInfo, options = {}) {
    const logger = options.logger;
    const ss = starkInfo.starkStruct;
    let MH;
    if (ss.verificationHashType === "GL") MH = await buildMerkleHashGL(ss.splitLinearHash);
    else if (ss.verificationHashType === "BN128") MH = await buildMerkleHashBN128(ss.merkleTreeArity, ss.merkleTreeCustom);
    else throw new Error("Invalid Hash Type: " + ss.verificationHashType);
    const nBits = ss.nBits, N = 1n << BigInt(nBits), extendBits = ss.nBitsExt - ss.nBits;
    if (nBits + extendBits !== ss.steps[0].nBits) throw new Error("First step must be just one");
    const nStages = starkInfo.nStages, qStage = nStages + 1, evalsStage = nStages + 1;
    const ctx = { evals: proof.evals, subproofValues: proof.subproofValues, publics, starkInfo, proof };
    const tr = challenges || await calculateTranscript(null, starkInfo, proof, publics, constRoot, options);
    ctx.challenges = tr.challenges; ctx.challengesFRISteps = tr.challengesFRISteps;
    // The query positions ALWAYS come out of the last FRI challenge (stark_verify.js:93) -- positions handed in by the caller would not be
    // bound to the transcript.  One exception, behind an explicit option: proofs of the older pil-stark transcript layout (the reference's
    // test/final proof) draw them from the main transcript; options.legacyTranscriptQueries = true takes challenges.friQueries for those.
    if (options.legacyTranscriptQueries && tr.friQueries) ctx.friQueries = tr.friQueries.slice();
    else {
        const seed = ctx.challengesFRISteps[ss.steps.length];
        if (seed === undefined) { if (logger) logger.warn("No challenge to draw the query positions from"); return false; }
        const derived = await calculateFRIQueries(starkInfo, seed, options);
        if (tr.friQueries && (tr.friQueries.length !== derived.length || tr.friQueries.some((q, i) => Number(q) !== Number(derived[i])))) {
            if (logger) logger.warn("Query positions do not follow from the FRI challenge");
            return false;
        }
        ctx.friQueries = derived;
    }

    // evaluations (:95-152)
    const xi = big(ctx.challenges[evalsStage][0]);
    const xN = exp3(xi, N), zh = sub(xN, 1n), wN = rootOfUnity(nBits);
    ctx.Z = inv(zh);
    const names = starkInfo.boundaries.map((b) => b.name);
    if (names.includes("firstRow")) ctx.Z_fr = mul(zh, inv(sub(xi, 1n)));
    if (names.includes("lastRow")) ctx.Z_lr = mul(zh, inv(sub(xi, powm(wN, N - 1n))));
    starkInfo.boundaries.filter((b) => b.name === "everyFrame").forEach((frame, i) => {
        let z = [1n, 0n, 0n];
        for (let j = 0; j < frame.offsetMin; j++) z = mul(z, sub(xi, powm(wN, BigInt(j))));
        for (let j = 0; j < frame.offsetMax; j++) z = mul(z, sub(xi, powm(wN, N - BigInt(j) - 1n)));
        ctx["Z_frame" + i] = z;
    });
    const res = executeCode(null, ctx, verifierInfo.qVerifier.code);
    let xAcc = 1n, q = 0n;
    const qIndex = starkInfo.cmPolsMap.findIndex((p) => p.stage === qStage && p.stageId === 0);
    for (let i = 0; i < starkInfo.qDeg; i++) {
        const evId = starkInfo.evMap.findIndex((e) => e.type === "cm" && e.id === qIndex + i);
        q = add(q, mul(xAcc, big(ctx.evals[evId])));
        xAcc = mul(xAcc, xN);
    }
    if (!eq(res, q)) { if (logger) logger.warn("Invalid evaluations"); return false; }

    // openings of every tree, all queries of a tree in one call (:165-178)
    const nQ = ss.nQueries, pq0 = proof.fri[0].polQueries, queries = ctx.friQueries;
    // a proof whose openings are short (fewer queries than the struct asks for, a tree missing, fewer values than the section is wide) is
    // an invalid proof, not an exception: the reference's paths all end in `return false`
    const widthOf = (j) => (j < qStage ? (starkInfo.mapSectionsN["cm" + (j + 1)] || 0) : starkInfo.nConstants);
    const wellFormed = Array.isArray(pq0) && pq0.length >= nQ && queries.length >= nQ && pq0.slice(0, nQ).every((q) => Array.isArray(q) && q.length >= qStage + 1 &&
        q.slice(0, qStage + 1).every((o, j) => Array.isArray(o) && Array.isArray(o[0]) && Array.isArray(o[1]) && (o[0].length >= widthOf(j) || (o[0].length === 0 && o[1].length === 0))));
    if (!wellFormed) { if (logger) logger.warn("Malformed openings"); return false; }
    const roots = [];
    for (let st = 1; st <= qStage; st++) roots.push(proof["root" + st]);
    roots.push(constRoot);
    for (let j = 0; j < roots.length; j++) {
        if (j < nStages && !(starkInfo.mapSectionsN["cm" + (j + 1)]) && !pq0.some((q) => q[j][1].length)) continue;   // a stage that commits nothing: a zkin file carries no openings for it (proof2zkin.js:37-41)
        const openings = [];
        for (let i = 0; i < nQ; i++) openings.push([pq0[i][j][0], pq0[i][j][1]]);
        if (!MH.verifyGroupProofs(roots[j], openings, queries.slice(0, nQ))) { if (logger) logger.warn(j < qStage ? "Invalid root" + (j + 1) : "Invalid constRoot"); return false; }
    }

    // the FRI polynomial at the query points: the query program on an nQueries-row domain (:180-215)
    let qb = 1; while ((1 << qb) < nQ) qb++;
    const rows = 1 << qb, nOpen = starkInfo.openingPoints.length;
    const wE = rootOfUnity(nBits + extendBits);
    const fake = { nBits: qb, nBitsExt: qb, extendBits: 0, publics: publics.map(BigInt), challenges: ctx.challenges, evals: ctx.evals,
        subproofValues: proof.subproofValues || [], pilInfo: starkInfo };
    for (let j = 0; j <= qStage; j++) {
        const name = j < qStage ? "cm" + (j + 1) + "_ext" : "const_ext";
        const width = j < qStage ? (starkInfo.mapSectionsN["cm" + (j + 1)] || 0) : starkInfo.nConstants;
        const a = new BigUint64Array(Math.max(1, rows * width));
        for (let i = 0; i < nQ; i++) { const v = pq0[i][j][0]; if (v.length < width) continue; for (let c = 0; c < width; c++) a[i * width + c] = m(BigInt(v[c])); }
        fake[name] = a;
    }
    const xdiv = new BigUint64Array(rows * 3 * nOpen);
    for (let i = 0; i < nQ; i++) {
        const x = SHIFT * powm(wE, BigInt(queries[i])) % P;
        for (let k = 0; k < nOpen; k++) {
            const opening = Number(starkInfo.openingPoints[k]);
            let w = powm(wN, BigInt(Math.abs(opening)));
            if (opening < 0) w = invm(w);
            const v = mul(inv(sub(x, mul(xi, w))), x);                                         // F.div(x, F.sub(x, xi w))  (:199-213)
            xdiv.set(v, (i * nOpen + k) * 3);
        }
    }
    fake.xDivXSubXi_ext = xdiv;
    const vals = calculateExps(fake, { code: verifierInfo.queryVerifier.code }, "ext", false, true);
    const byQuery = new Map();
    for (let i = 0; i < nQ; i++) byQuery.set(pq0[i], [is3(vals[i]) ? vals[i] : [vals[i], 0n, 0n]]);

    const fri = new FRI(ss, MH);
    return fri.verify(ctx.challengesFRISteps, queries, proof.fri, (query) => byQuery.get(query) || false);
}

module.exports = starkVerify;
module.exports.executeCode = executeCode;
module.exports.calculateTranscript = calculateTranscript;
module.exports.calculateFRIQueries = calculateFRIQueries;
