// BASELINE config 2's shape (2^20 rows x 8 columns, blow-up 8, FRI 23/18/13/8, 16 queries) proved from Node with every large
// buffer resident in HBM (DevBuffer): the JS stage loop of prove_flow.js drives, the data never enters the JS heap.
// The proof's canonical text must hash to the digest of the CPU checker's proof (tests/golden/fib_c2.json).
"use strict";
const fs = require("fs");
const path = require("path");
const assert = require("assert");
const crypto = require("crypto");
const root = path.join(__dirname, "..", "..");
const { prove } = require("./prove_flow.js");

const P = 0xFFFFFFFF00000001n;
function canon(v) {
    if (Array.isArray(v)) return "[" + v.map(canon).join(",") + "]";
    if (v && typeof v === "object") return "{" + Object.keys(v).map((k) => '"' + k + '":' + canon(v[k])).join(",") + "}";
    return '"' + BigInt(v).toString() + '"';
}

(async () => {
    const g = JSON.parse(fs.readFileSync(path.join(root, "tests/golden/fib_c2.json")));
    const nBits = g.pilInfo.starkStruct.nBits, N = 1 << nBits, K = g.start.length;
    // witness of K Fibonacci machines (sm_fibonacci.js:12-23) and the constants L1 / LLAST
    const cm1 = new BigUint64Array(N * 2 * K), consts = new BigUint64Array(N * 2);
    for (let k = 0; k < K; k++) {
        let a = BigInt(g.start[k][0]), b = BigInt(g.start[k][1]);
        for (let i = 0; i < N; i++) { cm1[i * 2 * K + 2 * k] = a; cm1[i * 2 * K + 2 * k + 1] = b; const t = (a * a + b * b) % P; b = a; a = t; }
    }
    consts[0] = 1n; consts[(N - 1) * 2 + 1] = 1n;
    const job = { pilInfo: g.pilInfo, expressionsInfo: g.expressionsInfo, cm1, consts, publics: g.publics, constRoot: g.constRoot, queries: g.queries };
    let best = Infinity, res;
    for (let it = 0; it < 3; it++) { res = await prove(job, true); best = Math.min(best, res.seconds); }
    assert.deepStrictEqual(res.proof.root1.map(String), g.root1, "root1");
    assert.deepStrictEqual(res.proof.root2.map(String), g.root2, "root2");
    const digest = crypto.createHash("sha256").update(canon(res.proof)).digest("hex");
    assert.strictEqual(digest, g.proofSha256, "proof digest");
    console.log(JSON.stringify({ config: "2^20 x 8, blow-up 8, Node-driven, device-resident", proof_seconds: best, cells_per_s: N * 2 * K / best }));
    console.log("prove c2 OK");
})().catch((e) => { console.error(e); process.exit(1); });
