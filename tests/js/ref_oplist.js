// The reference's 3 257-op sample program (tests/golden/ref_verify_evals_code.json.gz, see tests/test_ref_oplist.py) through
// the JS drop-in callCalculateExps: node ref_oplist.js <inputs.json> prints f_ext as hex words; the Python test compares
// it with the CPU checker's result on the same inputs.
"use strict";
const fs = require("fs");
const path = require("path");
const zlib = require("zlib");
const ROOT = path.join(__dirname, "..", "..");
const { callCalculateExps } = require(path.join(ROOT, "pil2-stark-js_amd", "js", "prover_helpers.js"));

(async () => {
    const inp = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
    const code = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(ROOT, "tests", "golden", "ref_verify_evals_code.json.gz"))).toString()).code;
    const last = code[code.length - 1].dest;
    code.push({ op: "copy", dest: { type: "f", dim: 3 }, src: [{ type: "tmp", id: last.id, dim: last.dim }] });
    // the sample predates the staged challenge schema ({id} only): one stage per flat id
    for (const c of code) for (const r of c.src) if (r.type === "challenge" && r.stage === undefined) { r.stage = r.id + 1; r.stageId = 0; }
    const big = (a) => a.map((v) => BigInt(v));
    const rows = 2 ** inp.nBits;
    const ctx = {
        nBits: inp.nBits, nBitsExt: inp.nBits, extendBits: 0, pilInfo: {},
        evals: inp.evals.map(big), challenges: inp.challenges.map((c) => [big(c)]), publics: big(inp.publics),
        x_ext: BigUint64Array.from(big(inp.x)), f_ext: new BigUint64Array(3 * rows),
    };
    await callCalculateExps("verifier", { code }, "ext", ctx, false, false, false);
    console.log(JSON.stringify(Array.from(ctx.f_ext, (v) => v.toString(16))));
})().catch((e) => { console.error(e); process.exit(1); });
