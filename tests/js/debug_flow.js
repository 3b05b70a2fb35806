// Constraint checking (calculateExps with debug = true, prover_helpers.js:46-70) through the JS drop-ins on the device: for every
// witness of the job (a golden AIR with one cell of its stage-1 trace altered, or none) the errors the check records -- from host
// buffers and from HBM-resident ones -- are printed as one JSON line; tests/test_node_boundary.py compares them with the messages
// the reference's row-by-row loop would write (first failing row of each constraint's boundary, its value).
//   node tests/js/debug_flow.js job.json        job = { golden, variants: [ { flips: [[row, col, delta], ...] } ] }
"use strict";
const fs = require("fs");
const path = require("path");
const root = path.join(__dirname, "..", "..");
const { debugCheck } = require("./prove_flow.js");
const PH = require(path.join(root, "pil2-stark-js_amd/js/prover_helpers.js"));
const P = 0xFFFFFFFF00000001n;

(async () => {
    const job = JSON.parse(fs.readFileSync(process.argv[2]));
    const g0 = JSON.parse(fs.readFileSync(path.join(root, "tests/golden", job.golden)));
    const width = g0.pilInfo.mapSectionsN.cm1;
    const out = [];
    for (const v of job.variants) {
        const g = Object.assign({}, g0, { cm1: g0.cm1.slice() });
        for (const [row, col, delta] of v.flips || []) g.cm1[row * width + col] = ((BigInt(g.cm1[row * width + col]) + BigInt(delta)) % P).toString();
        const host = await debugCheck(g, false), dev = await debugCheck(g, true);
        out.push({ host, dev });
    }
    // calculateExpAtPoint / calculateExpressionAtRow (prover_helpers.js:18-21,74-80): the first constraint's value at a few rows
    const ss = g0.pilInfo.starkStruct;
    const ctx = { pilInfo: g0.pilInfo, expressionsInfo: g0.expressionsInfo, nBits: ss.nBits, nBitsExt: ss.nBitsExt, publics: g0.publics.map(BigInt), challenges: [[], [], [], [], []], evals: [], subproofValues: [],
        const_n: BigUint64Array.from(g0.consts, BigInt), cm1_n: BigUint64Array.from(g0.cm1, BigInt) };
    const c0 = g0.expressionsInfo.constraints.find((c) => c.stage === 1);
    const at = c0 ? (job.points || []).map((i) => { const v = PH.calculateExpAtPoint(ctx, c0, i); return Array.isArray(v) ? v.map(String) : String(v); }) : [];
    console.log(JSON.stringify({ results: out, at }));
    console.log("debug flow OK");
})().catch((e) => { console.error(e); process.exit(1); });
