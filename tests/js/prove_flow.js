// A whole proof driven from Node in the order of src/prover/prover.js:7-127 (nStages = 1) with the reference's own
// function boundaries, every data-parallel piece coming from the drop-in modules of pil2-stark-js_amd/js:
//   extendAndMerkelize (fft_p.interpolate + MH.merkelize), callCalculateExps, computeQStark, computeEvalsStark,
//   computeFRIStark, FRI.fold / proofQueries, Transcript over the device Poseidon.
// The proof must equal, field by field, the proof the CPU checker wrote for the same AIR and witness
// (tests/golden/fib_flow.json, oracle/gen_fib_flow_golden.py).
"use strict";
const fs = require("fs");
const path = require("path");
const assert = require("assert");
const root = path.join(__dirname, "..", "..");
const J = (p) => path.join(root, "pil2-stark-js_amd/js", p);
const { interpolate } = require(J("fft_p.js"));
const buildMH = require(J("merklehash_p.js"));
const getPoseidon = require(J("poseidon.js"));
const Transcript = require(J("transcript.js"));
const FRI = require(J("fri.js"));
const PH = require(J("prover_helpers.js"));
const PU = require(J("polutils.js"));
const { callCalculateExps } = PH;
const SGH = require(J("stark_gen_helpers.js"));
const { buildZhInv, buildOneRowZerofierInv, buildFrameZerofierInv } = require(J("polutils.js"));
const { DevBuffer } = require(J("native.js"));

const big = (v) => (Array.isArray(v) ? v.map(big) : (typeof v === "string" && /^[0-9]+$/.test(v) ? BigInt(v) : v));
function bigProof(p) {      // golden proof: decimal strings -> BigInt, everything else untouched
    if (Array.isArray(p)) return p.map(bigProof);
    if (p && typeof p === "object") { const o = {}; for (const k of Object.keys(p)) o[k] = bigProof(p[k]); return o; }
    return typeof p === "string" ? BigInt(p) : p;
}

// computeStage for a witness stage (prover.js:193-214): the stage's hints (hints_helpers.js:81-123 restated on the drop-ins: calculateExpression
// for fields that are expressions, calculateZ / calculateS, setPol), then the intermediate polynomials of the last witness stage
// (:212-214: op-lists with destinations of type cm, trace domain)
async function witnessStage(st, ctx) {
    const pilInfo = ctx.pilInfo;
    if (st > 1) {
        for (const hint of ctx.expressionsInfo.hintsInfo || []) {
            const fld = (name) => hint.fields.find((f) => f.name === name);
            const ref = fld("reference");
            if (!ref || pilInfo.cmPolsMap[ref.id].stage !== st) continue;
            const col = (f) => (f.op === "cm" ? PH.getPol(ctx, f.id, "n") : f.op === "const" ? PH.getFixedPol(ctx, f.id) : f.op === "tmp" ? PH.calculateExpression(ctx, f.id) : BigInt(f.value));
            if (hint.name === "gprod") PH.setPol(ctx, ref.id, await PU.calculateZ(null, col(fld("numerator")), col(fld("denominator"))), "n");
            else if (hint.name === "gsum") PH.setPol(ctx, ref.id, await PU.calculateS(null, col(fld("numerator")), col(fld("denominator"))), "n");
            else throw new Error("hint " + hint.name + " is not part of this flow");
        }
    }
    const im = (ctx.expressionsInfo.imPolsCode || [])[st - 1];
    if (st === pilInfo.nStages && im && im.code.length) await callCalculateExps(st, im, "n", ctx, false, false);
}

// The constraint check every reference integration test starts with (test/stark/helpers.js:23-33: starkGen with options.debug): the
// witness stages are computed as in a proof but nothing is committed -- each stage's challenges are arbitrary (prover.js:65) -- and
// after each stage its constraints are evaluated on the rows of their boundaries (computeStage, prover.js:223-230, through
// callCalculateExps with debug = true).  -> ctx.errors, empty for a valid witness (proofGen returns true, prover.js:73-85).
async function debugCheck(g, resident) {
    const fromHost = (a) => (resident ? DevBuffer.from(a) : a);
    const pilInfo = g.pilInfo, ss = pilInfo.starkStruct, N = 1 << ss.nBits;
    const asBuf = (v) => fromHost(BigUint64Array.from(v, BigInt));
    const ctx = { prover: "stark", pilInfo, expressionsInfo: g.expressionsInfo, nBits: ss.nBits, nBitsExt: ss.nBitsExt, extendBits: ss.nBitsExt - ss.nBits, N,
        publics: g.publics.map(BigInt), challenges: [], evals: [], subproofValues: [], errors: [] };
    for (let i = 0; i < pilInfo.nStages + 3; i++) ctx.challenges.push([]);
    ctx.const_n = asBuf(g.consts);
    ctx.cm1_n = asBuf(g.cm1);
    for (let st = 2; st <= pilInfo.nStages; st++) { const n = pilInfo.mapSectionsN["cm" + st] * N; ctx["cm" + st + "_n"] = resident ? new DevBuffer(n) : new BigUint64Array(n); if (resident) ctx["cm" + st + "_n"].zero(); }
    let seed = 0x9E3779B97F4A7C15n;
    const rnd = () => { seed = (seed * 6364136223846793005n + 1442695040888963407n) & 0xFFFFFFFFFFFFFFFFn; return seed % 0xFFFFFFFF00000001n; };
    for (let st = 1; st <= pilInfo.nStages; st++) {
        ctx.challenges[st - 1] = pilInfo.challengesMap.filter((c) => c.stage === st).map(() => [rnd(), rnd(), rnd()]);
        await witnessStage(st, ctx);
        for (const constraint of (ctx.expressionsInfo.constraints || []).filter((c) => c.stage === st))
            await callCalculateExps(st, constraint, "n", ctx, false, false, true);
    }
    if (resident) freeCtx(ctx);
    return ctx.errors;
}

async function prove(g, resident) {
    // resident: every large buffer is a DevBuffer (HBM); the modules then work in place and only roots, evaluations,
    // the last FRI polynomial and the opened rows ever reach the JS heap
    const alloc = (n) => (resident ? new DevBuffer(n) : new BigUint64Array(n));
    const fromHost = (a) => (resident ? DevBuffer.from(a) : a);
    const pilInfo = g.pilInfo, ss = pilInfo.starkStruct;
    const poseidon = getPoseidon(), MH = await buildMH(false);
    const nBits = ss.nBits, nBitsExt = ss.nBitsExt, N = 1 << nBits, extN = 1 << nBitsExt;
    const ctx = { prover: "stark", pilInfo, expressionsInfo: g.expressionsInfo, nBits, nBitsExt, extendBits: nBitsExt - nBits, N, extN, MH,
        publics: g.publics.map(BigInt), challenges: [], evals: [], subproofValues: [], trees: [] };
    const nStages = pilInfo.nStages, qStage = nStages + 1;
    for (let i = 0; i < nStages + 3; i++) ctx.challenges.push([]);
    // setup (stark_buildConstTree.js:6-43) and initProverStark (stark_gen_helpers.js:104-160)
    const asBuf = (v) => (v instanceof DevBuffer ? v : fromHost(v instanceof BigUint64Array ? v : BigUint64Array.from(v, BigInt)));
    ctx.const_n = asBuf(g.consts);
    ctx.const_ext = alloc(pilInfo.nConstants * extN);
    await interpolate(ctx.const_n, pilInfo.nConstants, nBits, ctx.const_ext, nBitsExt);
    ctx.constTree = await MH.merkelize(ctx.const_ext, pilInfo.nConstants, extN);
    assert.deepStrictEqual(MH.root(ctx.constTree), g.constRoot.map(BigInt), "constant tree root");
    ctx.cm1_n = asBuf(g.cm1);
    for (let st = 1; st <= qStage; st++) {
        if (st > 1 && st <= nStages) { ctx["cm" + st + "_n"] = alloc(pilInfo.mapSectionsN["cm" + st] * N); if (resident) ctx["cm" + st + "_n"].zero(); }
        ctx["cm" + st + "_ext"] = alloc(pilInfo.mapSectionsN["cm" + st] * extN);
    }
    ctx.q_ext = alloc(pilInfo.qDim * extN);
    ctx.f_ext = alloc(3 * extN);
    ctx.x_n = alloc(N); ctx.x_ext = alloc(extN);
    ctx.Zi_ext = alloc(pilInfo.boundaries.length * extN);
    ctx.xDivXSubXi_ext = alloc(3 * extN * pilInfo.openingPoints.length);
    SGH.buildXTables(ctx);
    for (let i = 0; i < pilInfo.boundaries.length; i++) {                                        // stark_gen_helpers.js:146-160
        const bd = pilInfo.boundaries[i];
        if (bd.name === "everyRow") buildZhInv(ctx.Zi_ext, i * extN, null, nBits, nBitsExt, true);
        else if (bd.name === "firstRow") buildOneRowZerofierInv(ctx.Zi_ext, i * extN, null, null, nBits, nBitsExt, 0, true);
        else if (bd.name === "lastRow") buildOneRowZerofierInv(ctx.Zi_ext, i * extN, null, null, nBits, nBitsExt, N - 1, true);
        else if (bd.name === "everyFrame") buildFrameZerofierInv(ctx.Zi_ext, i * extN, null, null, nBits, nBitsExt, bd, true);
    }
    ctx.fri = new FRI(ss, MH);
    require(J("native.js")).addon.sync();
    const tStart = process.hrtime.bigint();
    const transcript = new Transcript(poseidon);
    transcript.put(MH.root(ctx.constTree));                                                     // prover.js:148-189
    transcript.put(ss.hashCommits ? await SGH.calculateHashStark(ctx, ctx.publics) : ctx.publics);
    // witness stages (prover.js:41-66, computeStage :193-228): challenges of the stage, its hints (hints_helpers.js:81-123 restated on the
    // drop-ins: calculateExpression for fields that are expressions, calculateZ / calculateS, setPol), the intermediate polynomials of the
    // last witness stage (:212-214: op-lists with destinations of type cm, trace domain), then extendAndMerkelize (stark_gen_helpers.js:388-412)
    const roots = {};
    for (let st = 1; st <= nStages; st++) {
        if (st > 1) {
            const nCh = pilInfo.challengesMap.filter((c) => c.stage === st).length;
            ctx.challenges[st - 1] = [];
            for (let k = 0; k < nCh; k++) ctx.challenges[st - 1].push(transcript.getField());
        }
        await witnessStage(st, ctx);
        const w = pilInfo.mapSectionsN["cm" + st];
        await interpolate(ctx["cm" + st + "_n"], w, nBits, ctx["cm" + st + "_ext"], nBitsExt);
        ctx.trees[st] = await MH.merkelize(ctx["cm" + st + "_ext"], w, extN);
        roots[st] = MH.root(ctx.trees[st]); transcript.put(roots[st]);
    }
    // quotient stage (challenges are stored at [stage - 1], setChallengesStark :414-431)
    ctx.challenges[qStage - 1] = [transcript.getField()];
    await callCalculateExps(qStage, ctx.expressionsInfo.expressionsCode.find((e) => e.expId === pilInfo.cExpId).code, "ext", ctx, false, false, false);
    [roots[qStage]] = await SGH.computeQStark(ctx, {}); transcript.put(roots[qStage]);
    // evaluations
    ctx.challenges[qStage] = [transcript.getField()];
    const evals = await SGH.computeEvalsStark(ctx, {});
    transcript.put(evals);          // prover.js absorbs them one by one (addTranscriptStark); as one list the drop-in transcript chains the permutations in one device call -- same state
    ctx.challenges[qStage + 1] = [transcript.getField(), transcript.getField()];
    await SGH.computeFRIStark(ctx, { parallelExec: false, useThreads: false });
    // FRI folding (computeFRIFolding :337-356) and queries (:474-493, fri.js:83-105)
    for (let step = 0; step < ss.steps.length; step++) {
        const challenge = transcript.getField();
        const sp = await ctx.fri.fold(step, ctx.friPol[step], challenge);
        ctx.friPol[step + 1] = sp.pol; ctx.friProof[step + 1] = sp.proof;
        if (step < ss.steps.length - 1) { ctx.friTrees[step + 1] = sp.tree; transcript.put(sp.proof.root); }
        else if (ss.hashCommits) transcript.put(await SGH.calculateHashStark(ctx, sp.proof));   // stark_gen_helpers.js:349-351
        else transcript.put(sp.proof);
    }
    const tq = new Transcript(poseidon); tq.put(transcript.getField());
    const friQueries = tq.getPermutations(ss.nQueries, ss.steps[0].nBits);
    assert.deepStrictEqual(friQueries, g.queries, "query positions");
    ctx.fri.proofQueries(ctx.friProof, ctx.friTrees, friQueries.slice());
    const proof = {};                                                                            // genProofStark :362-386: roots, evaluations, FRI
    for (let st = 1; st <= qStage; st++) proof["root" + st] = roots[st];
    proof.evals = ctx.evals; proof.fri = ctx.friProof;
    require(J("native.js")).addon.sync();
    const seconds = Number(process.hrtime.bigint() - tStart) / 1e9;
    if (!g.proof) return { proof, ctx, seconds };
    const want = bigProof(g.proof);
    assert.deepStrictEqual(ctx.challenges, bigProof(g.challenges), "challenges");
    for (let st = 1; st <= qStage; st++) assert.deepStrictEqual(proof["root" + st], want["root" + st], "root" + st);
    assert.deepStrictEqual(proof.evals, want.evals, "evals");
    assert.deepStrictEqual(proof.fri.length, want.fri.length);
    for (let s = 0; s < want.fri.length; s++) assert.deepStrictEqual(proof.fri[s], want.fri[s], "fri[" + s + "]");
    if (resident) assert(ctx.trees[1].nodes instanceof DevBuffer && (!ctx.friTrees[1] || ctx.friTrees[1].nodes instanceof DevBuffer) && (ss.steps.length < 2 || ctx.friPol[1] instanceof DevBuffer));   // (the last step's polynomial is the proof's: host values)
    return { proof, ctx, seconds };
}
// DevBuffers have no finalizer: release what a resident prove() allocated (everything reachable from ctx except the
// caller's own inputs), so that a config-3 proof (107 GB extension) can be repeated in one process
function freeCtx(ctx, keep = []) {
    const seen = new Set(keep);
    (function walk(v, depth) {
        if (!v || typeof v !== "object" || depth > 4) return;
        if (v instanceof DevBuffer) { if (!seen.has(v)) { seen.add(v); v.free(); } return; }
        if (ArrayBuffer.isView(v)) return;
        for (const x of (Array.isArray(v) ? v : Object.values(v))) walk(x, depth + 1);
    })(ctx, 0);
}
module.exports = { prove, freeCtx, debugCheck };

if (require.main === module) (async () => {
    for (const name of ["fib_flow.json", "fib_flow_hashcommits.json", "fib_flow_prevrow.json", "fib_flow_impols.json", "fib_flow_boundaries.json", "fib_flow_boundaries_only.json", "perm_flow_hints.json"]) {
        const g = JSON.parse(fs.readFileSync(path.join(root, "tests/golden", name)));
        for (const resident of [false, true]) {
            assert.deepStrictEqual(await debugCheck(g, resident), [], name + ": the witness does not satisfy its constraints");    // options.debug pre-run, test/stark/helpers.js:23-33
            await prove(g, resident);
        }
    }
    console.log("prove flow OK");
})().catch((e) => { console.error(e); process.exit(1); });
