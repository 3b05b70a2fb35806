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
const { callCalculateExps } = require(J("prover_helpers.js"));
const SGH = require(J("stark_gen_helpers.js"));
const { buildZhInv, buildOneRowZerofierInv, buildFrameZerofierInv } = require(J("polutils.js"));
const { DevBuffer } = require(J("native.js"));

const big = (v) => (Array.isArray(v) ? v.map(big) : (typeof v === "string" && /^[0-9]+$/.test(v) ? BigInt(v) : v));
function bigProof(p) {      // golden proof: decimal strings -> BigInt, everything else untouched
    if (Array.isArray(p)) return p.map(bigProof);
    if (p && typeof p === "object") { const o = {}; for (const k of Object.keys(p)) o[k] = bigProof(p[k]); return o; }
    return typeof p === "string" ? BigInt(p) : p;
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
        publics: g.publics.map(BigInt), challenges: [[], [], [], []], evals: [], subproofValues: [], trees: [] };
    // setup (stark_buildConstTree.js:6-43) and initProverStark (stark_gen_helpers.js:104-160)
    const asBuf = (v) => (v instanceof DevBuffer ? v : fromHost(v instanceof BigUint64Array ? v : BigUint64Array.from(v, BigInt)));
    ctx.const_n = asBuf(g.consts);
    ctx.const_ext = alloc(pilInfo.nConstants * extN);
    await interpolate(ctx.const_n, pilInfo.nConstants, nBits, ctx.const_ext, nBitsExt);
    ctx.constTree = await MH.merkelize(ctx.const_ext, pilInfo.nConstants, extN);
    assert.deepStrictEqual(MH.root(ctx.constTree), g.constRoot.map(BigInt), "constant tree root");
    ctx.cm1_n = asBuf(g.cm1);
    ctx.cm1_ext = alloc(pilInfo.mapSectionsN.cm1 * extN);
    ctx.cm2_ext = alloc(pilInfo.mapSectionsN.cm2 * extN);
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
    // intermediate polynomials of the last witness stage (prover.js:212-214): op-lists with destinations of type cm, trace domain
    const im = (ctx.expressionsInfo.imPolsCode || [])[0];
    if (im && im.code.length) await callCalculateExps(1, im, "n", ctx, false, false);
    // stage 1: extendAndMerkelize (stark_gen_helpers.js:388-412)
    await interpolate(ctx.cm1_n, pilInfo.mapSectionsN.cm1, nBits, ctx.cm1_ext, nBitsExt);
    ctx.trees[1] = await MH.merkelize(ctx.cm1_ext, pilInfo.mapSectionsN.cm1, extN);
    const root1 = MH.root(ctx.trees[1]); transcript.put(root1);
    // stage 2: quotient (challenges are stored at [stage - 1], setChallengesStark :414-431)
    ctx.challenges[1] = [transcript.getField()];
    await callCalculateExps(2, ctx.expressionsInfo.expressionsCode.find((e) => e.expId === pilInfo.cExpId).code, "ext", ctx, false, false, false);
    const [root2] = await SGH.computeQStark(ctx, {}); transcript.put(root2);
    // evaluations
    ctx.challenges[2] = [transcript.getField()];
    const evals = await SGH.computeEvalsStark(ctx, {});
    transcript.put(evals);          // prover.js absorbs them one by one (addTranscriptStark); as one list the drop-in transcript chains the permutations in one device call -- same state
    ctx.challenges[3] = [transcript.getField(), transcript.getField()];
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
    const proof = { root1, root2, evals: ctx.evals, fri: ctx.friProof };                         // genProofStark :362-386
    require(J("native.js")).addon.sync();
    const seconds = Number(process.hrtime.bigint() - tStart) / 1e9;
    if (!g.proof) return { proof, ctx, seconds };
    const want = bigProof(g.proof);
    assert.deepStrictEqual(ctx.challenges, bigProof(g.challenges), "challenges");
    assert.deepStrictEqual(proof.root1, want.root1, "root1");
    assert.deepStrictEqual(proof.root2, want.root2, "root2");
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
module.exports = { prove, freeCtx };

if (require.main === module) (async () => {
    for (const name of ["fib_flow.json", "fib_flow_hashcommits.json", "fib_flow_prevrow.json", "fib_flow_impols.json", "fib_flow_boundaries.json"]) {
        const g = JSON.parse(fs.readFileSync(path.join(root, "tests/golden", name)));
        await prove(g, false);
        await prove(g, true);
    }
    console.log("prove flow OK");
})().catch((e) => { console.error(e); process.exit(1); });
