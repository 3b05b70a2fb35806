// A whole proof at BASELINE config 3's size (2^24 rows x 100 columns, blow-up 8, FRI 27/22/17/12/7, 64 queries) DRIVEN FROM
// NODE: prover.js's stage order (tests/js/prove_flow.js) over the JS drop-in modules, every large buffer a DevBuffer in HBM.
// The job (pilInfo, expressionsInfo, start values of the witness, constant root, publics, expected query rows) comes from
// the file named on the command line -- tests/test_node_boundary.py writes it from the Python-driven proof of the same
// witness -- and the digest of the proof's canonical text is printed for the caller to compare with that proof's.
//   node tests/js/prove_c3.js job.json [repeats]
"use strict";
const fs = require("fs");
const path = require("path");
const crypto = require("crypto");
const root = path.join(__dirname, "..", "..");
const { prove, freeCtx } = require("./prove_flow.js");
const { addon, DevBuffer } = require(path.join(root, "pil2-stark-js_amd/js/native.js"));

function canon(v) {
    if (Array.isArray(v)) return "[" + v.map(canon).join(",") + "]";
    if (v && typeof v === "object") return "{" + Object.keys(v).map((k) => '"' + k + '":' + canon(v[k])).join(",") + "}";
    return '"' + BigInt(v).toString() + '"';
}

(async () => {
    const g = JSON.parse(fs.readFileSync(process.argv[2]));
    const repeats = Number(process.argv[3] || 2);
    const nBits = g.pilInfo.starkStruct.nBits, N = 2 ** nBits, K = g.start.length / 2;
    // witness of K Fibonacci machines (sm_fibonacci.js:12-23) generated in HBM, constants L1 / LLAST uploaded
    const cm1 = new DevBuffer(N * 2 * K);
    addon.synthFibonacciDev(nBits, K, BigUint64Array.from(g.start, BigInt), cm1.ptr);
    const consts = new BigUint64Array(N * 2); consts[0] = 1n; consts[(N - 1) * 2 + 1] = 1n;
    const job = { pilInfo: g.pilInfo, expressionsInfo: g.expressionsInfo, cm1, consts: DevBuffer.from(consts), publics: g.publics, constRoot: g.constRoot, queries: g.queries };
    let best = Infinity, res;
    for (let it = 0; it < repeats; it++) { res = await prove(job, true); best = Math.min(best, res.seconds); freeCtx(res.ctx, [job.cm1, job.consts]); res.ctx = null; }
    const digest = crypto.createHash("sha256").update(canon(res.proof)).digest("hex");
    console.log(JSON.stringify({ config: "2^" + nBits + " x " + 2 * K + ", blow-up 8, Node-driven, device-resident", proof_seconds: best, cells_per_s: N * 2 * K / best, proofSha256: digest }));
    console.log("prove c3 OK");
})().catch((e) => { console.error(e); process.exit(1); });
