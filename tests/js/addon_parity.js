// GPU parity of the Node.js boundary: the drop-in modules in pil2-stark-js_amd/js (same names and signatures as the
// reference's fft_p.js / merklehash_p.js / poseidon.js / fri.js) against the committed golden vectors, which were
// produced by the reference's own modules (oracle/gen_golden.js).  Shapes follow test/fft_p.test.js and
// test/merklehash_p.test.js.  Usage: node tests/js/addon_parity.js
"use strict";
const fs = require("fs");
const path = require("path");
const assert = require("assert");
const root = path.join(__dirname, "..", "..");
const { fft, ifft, interpolate } = require(path.join(root, "pil2-stark-js_amd/js/fft_p.js"));
const buildMH = require(path.join(root, "pil2-stark-js_amd/js/merklehash_p.js"));
const getPoseidon = require(path.join(root, "pil2-stark-js_amd/js/poseidon.js"));
const FRI = require(path.join(root, "pil2-stark-js_amd/js/fri.js"));
const G = (n) => JSON.parse(fs.readFileSync(path.join(root, "tests/golden", n)));
const H = (v) => Array.isArray(v) ? v.map(H) : (typeof v === "string" ? BigInt("0x" + v) : v);

// minimal chunked container with pilcom.BigBuffer's surface, to drive the staged (non-flat) path
class ChunkedBuffer {
    constructor(n, chunk = 1000) { this.length = n; this.chunk = chunk; this.bufs = []; for (let o = 0; o < n; o += chunk) this.bufs.push(new BigUint64Array(Math.min(chunk, n - o))); }
    getElement(i) { return this.bufs[Math.floor(i / this.chunk)][i % this.chunk]; }
    setElement(i, v) { this.bufs[Math.floor(i / this.chunk)][i % this.chunk] = v; }
    slice(a, b) { const r = new BigUint64Array(b - a); for (let i = a; i < b; i++) r[i - a] = this.getElement(i); return r; }
    set(arr, off) { for (let i = 0; i < arr.length; i++) this.setElement(off + i, arr[i]); }
}

(async () => {
    // --- fft_p: "Check fft" / "Check interpolate" shapes of test/fft_p.test.js (value = row index) vs per-column goldens
    const ntt = G("ntt.json");
    for (const c of ntt.cases) {
        if (c.nBits < 1) continue;
        const n = 1 << c.nBits, nPols = 2;
        const p = H(c.p), buff = new BigUint64Array(n * nPols), out = new BigUint64Array(n * nPols);
        for (let j = 0; j < n; j++) for (let i = 0; i < nPols; i++) buff[j * nPols + i] = p[j];
        await fft(buff, nPols, c.nBits, out);
        const f = H(c.fft); for (let j = 0; j < n; j++) for (let i = 0; i < nPols; i++) assert.strictEqual(out[j * nPols + i], f[j], "fft " + c.name);
        await ifft(buff, nPols, c.nBits, out);
        const fi = H(c.ifft); for (let j = 0; j < n; j++) assert.strictEqual(out[j * nPols + 1], fi[j], "ifft " + c.name);
        for (const eb of Object.keys(c.ext)) {
            const e = H(c.ext[eb]), ext = new BigUint64Array(e.length * nPols);
            await interpolate(buff, nPols, c.nBits, ext, c.nBits + Number(eb));
            for (let j = 0; j < e.length; j++) for (let i = 0; i < nPols; i++) assert.strictEqual(ext[j * nPols + i], e[j], "interpolate " + c.name);
            // same through a chunked (BigBuffer-like) container
            const cb = new ChunkedBuffer(n * nPols, 7), co = new ChunkedBuffer(e.length * nPols, 13);
            cb.set(buff, 0);
            await interpolate(cb, nPols, c.nBits, co, c.nBits + Number(eb));
            for (let j = 0; j < e.length; j++) assert.strictEqual(co.getElement(j * nPols), e[j], "interpolate(chunked) " + c.name);
        }
    }
    // --- poseidon KATs (test/poseidon.test.js)
    const poseidon = getPoseidon();
    for (const [inp, cap, out] of H(G("poseidon.json"))) assert.deepStrictEqual(poseidon(inp, cap, 12), out);
    assert.deepStrictEqual(poseidon([0, 0, 0, 0, 0, 0, 0, 0]).map((x) => x.toString(16)), ["3c18a9786cb0b359", "c4055e3364a246c3", "7953db0ab48808f4", "c71603f33a1144ca"]);
    assert.deepStrictEqual(poseidon([-1, -1, -1, -1, -1, -1, -1, -1], [-1, -1, -1, -1])[0].toString(16), "be0085cfc57a8357");
    assert.throws(() => poseidon([1, 2, 3]), /Invalid Input size/);
    // --- merklehash_p: test/merklehash_p.test.js shapes, roots from the golden file
    for (const [N, w, split, rootG] of H(G("merkle.json"))) {
        const MH = await buildMH(!!split);
        const pols = new BigUint64Array(N * w);
        for (let i = 0; i < N; i++) for (let j = 0; j < w; j++) pols[i * w + j] = BigInt(i + j * 1000);
        const tree = await MH.merkelize(pols, w, N);
        assert.deepStrictEqual(MH.root(tree), rootG, `root ${N}x${w} split=${split}`);
        const idx = Math.min(3, N - 1);
        const [groupElements, mp] = MH.getGroupProof(tree, idx);
        assert(MH.verifyGroupProof(MH.root(tree), mp, idx, groupElements));
        groupElements[0] ^= 1n;
        assert(!MH.verifyGroupProof(MH.root(tree), mp, idx, groupElements));
        assert.throws(() => MH.getGroupProof(tree, N), /Out of range/);
        if (N === 256 && w === 9) {     // chunked container + file round trip
            const cb = new ChunkedBuffer(N * w, 100); cb.set(pols, 0);
            const t2 = await MH.merkelize(cb, w, N);
            assert.deepStrictEqual(MH.root(t2), rootG);
            const fn = path.join(require("os").tmpdir(), `pil2gl_${process.pid}.consttree`);
            await MH.writeToFile(tree, fn);
            const t3 = await MH.readFromFile(fn);
            fs.unlinkSync(fn);
            assert.strictEqual(t3.width, w); assert.strictEqual(t3.height, N);
            assert.deepStrictEqual(Array.from(t3.nodes), Array.from(tree.nodes));
            assert.deepStrictEqual(Array.from(t3.elements), Array.from(pols));
        }
    }
    // --- FRI.fold vs golden folds (fri.js:22-61), steps chosen so that step 1 folds polBits -> outBits
    const MHf = await buildMH(false);
    for (const [polBits, outBits, bits0, , ch, pol, res] of H(G("fri_fold.json"))) {
        if (bits0 !== polBits) continue;        // FRI class derives shiftInv from steps[0] - steps[step-1]
        const fri = new FRI({ nBits: polBits - 1, nBitsExt: polBits, nQueries: 4, steps: [{ nBits: polBits }, { nBits: outBits }] }, MHf);
        const r = await fri.fold(1, pol, ch);
        assert.deepStrictEqual(r.pol, res, `fold ${polBits}->${outBits}`);
        assert.deepStrictEqual(r.proof, res);
        const r0 = await fri.fold(0, pol, ch);  // step 0: identity + tree over transposed groups
        assert.strictEqual(r0.pol, pol);
        const nGroups = 1 << outBits, gp = MHf.getGroupProof(r0.tree, nGroups - 1);
        assert(MHf.verifyGroupProof(r0.proof.root, gp[1], nGroups - 1, gp[0]));
        for (let j = 0; j < pol.length / nGroups; j++) assert.strictEqual(gp[0][3 * j], pol[j * nGroups + nGroups - 1][0]);
    }
    // --- callCalculateExps: a small op-list over a fake ctx (prover_helpers.js:23-259 operand kinds), checked by BigInt math
    {
        const { callCalculateExps } = require(path.join(root, "pil2-stark-js_amd/js/prover_helpers.js"));
        const P = 0xFFFFFFFF00000001n, nBits = 4, nBitsExt = 6, extN = 64, ext = 4;
        const ctx = {
            nBits, nBitsExt, extendBits: 2, publics: [5n], challenges: [[], [[3n, 1n, 4n]]], evals: [], subproofValues: [],
            pilInfo: { nConstants: 2, qDim: 3, openingPoints: [0, 1], boundaries: [{ name: "everyRow" }], mapSectionsN: { cm1: 3 },
                cmPolsMap: [{ stage: 1, dim: 1, stagePos: 0 }, { stage: 1, dim: 1, stagePos: 2 }] },
            const_ext: new BigUint64Array(extN * 2), cm1_ext: new BigUint64Array(extN * 3), x_ext: new BigUint64Array(extN),
            Zi_ext: new BigUint64Array(extN), q_ext: new BigUint64Array(extN * 3),
        };
        for (let i = 0; i < extN; i++) { ctx.const_ext[2 * i + 1] = BigInt(7 * i + 1); ctx.cm1_ext[3 * i] = BigInt(i * i + 3); ctx.cm1_ext[3 * i + 2] = P - BigInt(i + 1); ctx.x_ext[i] = BigInt(1000 + i); ctx.Zi_ext[i] = BigInt(2 * i + 9); }
        // q = ((cm0' - cm1) * const1 + public0 + x) * Zi * challenge        (cm0' = next row: prime 1)
        const code = { tmpUsed: 5, code: [
            { op: "sub", dest: { type: "tmp", id: 0, dim: 1 }, src: [{ type: "cm", id: 0, prime: 1, dim: 1 }, { type: "cm", id: 1, prime: 0, dim: 1 }] },
            { op: "mul", dest: { type: "tmp", id: 1, dim: 1 }, src: [{ type: "tmp", id: 0, dim: 1 }, { type: "const", id: 1, prime: 0, dim: 1 }] },
            { op: "add", dest: { type: "tmp", id: 2, dim: 1 }, src: [{ type: "tmp", id: 1, dim: 1 }, { type: "public", id: 0, dim: 1 }] },
            { op: "add", dest: { type: "tmp", id: 3, dim: 1 }, src: [{ type: "tmp", id: 2, dim: 1 }, { type: "x", dim: 1 }] },
            { op: "mul", dest: { type: "tmp", id: 4, dim: 1 }, src: [{ type: "tmp", id: 3, dim: 1 }, { type: "Zi", boundaryId: 0, dim: 1 }] },
            { op: "mul", dest: { type: "q", dim: 3 }, src: [{ type: "tmp", id: 4, dim: 1 }, { type: "challenge", stage: 2, stageId: 0, id: 0, dim: 3 }] },
        ] };
        await callCalculateExps(2, code, "ext", ctx, false, false, false);
        const mod = (a) => ((a % P) + P) % P;
        for (let i = 0; i < extN; i++) {
            const nx = (i + ext) % extN;
            const t = mod(mod(mod(ctx.cm1_ext[3 * nx] - ctx.cm1_ext[3 * i + 2]) * ctx.const_ext[2 * i + 1] + 5n + ctx.x_ext[i]) * ctx.Zi_ext[i]);
            assert.deepStrictEqual([ctx.q_ext[3 * i], ctx.q_ext[3 * i + 1], ctx.q_ext[3 * i + 2]], [mod(t * 3n), mod(t * 1n), mod(t * 4n)], "callCalculateExps row " + i);
        }
    }
    // --- BN128 Merkle commitment (merklehash_bn128_p.js, linearhash.bn128.js, transcript.bn128.js)
    {
        const buildMHBN = require(path.join(root, "pil2-stark-js_amd/js/merklehash_bn128_p.js"));
        const TranscriptBN = require(path.join(root, "pil2-stark-js_amd/js/transcript_bn128.js"));
        const g = JSON.parse(fs.readFileSync(path.join(root, "tests/golden/bn128_merkle.json")));
        assert.strictEqual(buildMHBN.poseidon([1n, 2n], 0n).toString(), g.poseidon.out_t3);
        assert.deepStrictEqual(buildMHBN.poseidon(g.poseidon.in.map(BigInt), BigInt(g.poseidon.init), 17).map(String), g.poseidon.out17);
        for (const t of g.trees) {          // test/merklehash_bn128_p.test.js shapes
            const MH = await buildMHBN(t.arity, t.custom);
            const pols = new BigUint64Array(t.N * t.nPols);
            for (let i = 0; i < t.N; i++) for (let j = 0; j < t.nPols; j++) pols[i * t.nPols + j] = BigInt(i + j * 1000);
            const cbn = new ChunkedBuffer(pols.length, 1000); cbn.set(pols, 0);
            for (const buf of [pols, cbn]) {
                const tree = await MH.merkelize(buf, t.nPols, t.N);
                assert.strictEqual(MH.root(tree).toString(), t.root, `bn128 root arity ${t.arity}`);
                const [v, mp] = MH.getGroupProof(tree, t.idx);
                assert.deepStrictEqual(mp.map((l) => l.map(String)), t.proof);
                assert(MH.verifyGroupProof(MH.root(tree), mp, t.idx, v));
                v[0] = v[0] + 1n;
                assert(!MH.verifyGroupProof(MH.root(tree), mp, t.idx, v));
                if (buf === pols) {
                    const f = path.join(require("os").tmpdir(), `pil2gl_bn_${process.pid}.bin`);
                    await MH.writeToFile(tree, f);
                    const t2 = await MH.readFromFile(f);
                    fs.unlinkSync(f);
                    assert.deepStrictEqual(t2.nodes, tree.nodes);
                    assert.deepStrictEqual(t2.elements, pols);
                }
            }
        }
        // a proof the reference prover wrote (test/final/verifier.proof.zkin.json): transcript -> query positions -> openings
        const p = JSON.parse(fs.readFileSync(path.join(root, "tests/golden/ref_final_verifier.proof.zkin.json")));
        const T = new TranscriptBN(16);
        T.put(p.publics.map(BigInt)); T.put(BigInt(p.root1)); T.getField(); T.getField();
        T.put(BigInt(p.root2)); T.getField(); T.getField();
        T.put(BigInt(p.root3)); T.getField();
        T.put(BigInt(p.rootQ)); T.getField();
        T.put(p.evals.map((e) => e.map(BigInt))); T.getField(); T.getField(); T.getField();
        for (let s = 1; s <= 4; s++) { T.put(BigInt(p[`s${s}_root`])); T.getField(); }
        T.put(p.finalPol.map((e) => e.map(BigInt)));
        const ys = T.getPermutations(32, 17);
        const MH4 = await buildMHBN(4, false);
        for (const q of [0, 13, 31]) {
            assert(MH4.verifyGroupProof(BigInt(p.root1), p.s0_siblings1[q], ys[q], p.s0_vals1[q].map(BigInt)), "final proof root1 q" + q);
            assert(MH4.verifyGroupProof(BigInt(p.rootQ), p.s0_siblingsQ[q], ys[q], p.s0_valsQ[q].map(BigInt)), "final proof rootQ q" + q);
            assert(MH4.verifyGroupProof(BigInt(p.s2_root), p.s2_siblings[q], ys[q] % (1 << 11), p.s2_vals[q].map(BigInt)), "final proof s2 q" + q);
        }
    }
    console.log("addon parity OK");
})().catch((e) => { console.error(e); process.exit(1); });
