// Generates tests/golden/*.json by RUNNING the reference's own dependency-free
// modules under node (f3g.js, fft/fft.js, polutils.js, hash/poseidon/poseidon.js,
// hash/linearhash/linearhash{,_gpu}.js, transcript/transcript.js).  The reference
// is only present in the build container, so the outputs are committed as data.
//
//   node oracle/gen_golden.js          (reads /root/reference or $PIL2_REFERENCE)
//
// Modules that need absent third-party packages (fft_p.js, merklehash_p.js,
// fri.js: pilcom/workerpool/chai) are NOT loaded and NOT shimmed.  Where a
// vector below needs their behaviour it is composed here from the loaded
// reference primitives exactly as the cited lines do, and says so ("composed").
"use strict";
const fs = require("fs");
const path = require("path");
const ref = process.env.PIL2_REFERENCE || "/root/reference";
const F3g = require(path.join(ref, "src/helpers/f3g.js"));
const { extendPol, polMulAxi, evalPol, buildZhInv, buildOneRowZerofierInv, buildFrameZerofierInv } =
    require(path.join(ref, "src/helpers/polutils.js"));
const getPoseidon = require(path.join(ref, "src/helpers/hash/poseidon/poseidon.js"));
const LinearHash = require(path.join(ref, "src/helpers/hash/linearhash/linearhash.js"));
const LinearHashGPU = require(path.join(ref, "src/helpers/hash/linearhash/linearhash_gpu.js"));
const Transcript = require(path.join(ref, "src/helpers/transcript/transcript.js"));

const F = new F3g();
const poseidon = getPoseidon();
const outDir = path.join(__dirname, "..", "tests", "golden");
fs.mkdirSync(outDir, { recursive: true });

// deterministic PRNG (splitmix64), canonical field elements
let sm = 0x5EED0000n;
const M64 = (1n << 64n) - 1n;
function rnd64() {
    sm = (sm + 0x9E3779B97F4A7C15n) & M64;
    let z = sm;
    z = ((z ^ (z >> 30n)) * 0xBF58476D1CE4E5B9n) & M64;
    z = ((z ^ (z >> 27n)) * 0x94D049BB133111EBn) & M64;
    return z ^ (z >> 31n);
}
const rndF = () => rnd64() % F.p;
const rnd3 = () => [rndF(), rndF(), rndF()];
const hx = (v) => {
    if (Array.isArray(v)) return v.map(hx);
    if (typeof v === "bigint") return v.toString(16);
    if (v !== null && typeof v === "object") { const o = {}; for (const k of Object.keys(v)) o[k] = hx(v[k]); return o; }
    return v;       // plain JS numbers (sizes, bit counts) stay numbers
};
const write = (name, obj) => {
    fs.writeFileSync(path.join(outDir, name), JSON.stringify(obj, null, 0).replace(/\],/g, "],\n") + "\n");
    console.log("wrote", name);
};
const EDGE = [0n, 1n, 2n, F.p - 1n, F.p - 2n, 0xFFFFFFFFn, 0x100000000n, 0xFFFFFFFF00000000n, 0x8000000000000000n % F.p, 0xFFFFFFFEFFFFFFFFn];

// ---------------------------------------------------------------- field
{
    const mul = [], ext = [], inv = [];
    for (const a of EDGE) for (const b of EDGE) mul.push([a, b, F.mul(a, b), F.add(a, b), F.sub(a, b)]);
    for (let i = 0; i < 200; i++) { const a = rndF(), b = rndF(); mul.push([a, b, F.mul(a, b), F.add(a, b), F.sub(a, b)]); }
    for (let i = 0; i < 40; i++) { const a = rndF() || 1n; inv.push([a, F.inv(a)]); }
    ext.push({ a: [1n, 2n, 3n], b: [4n, 5n, F.p - 1n], mul: F.mul([1n, 2n, 3n], [4n, 5n, F.p - 1n]), inv: F.inv([1n, 2n, 3n]) });   // test/f3g.test.js:33-38
    for (let i = 0; i < 60; i++) { const a = rnd3(), b = rnd3(); ext.push({ a, b, mul: F.mul(a, b), inv: F.inv(a) }); }
    const bi = []; for (let i = 0; i < 17; i++) bi.push(rndF() || 1n);
    const bi3 = []; for (let i = 0; i < 9; i++) bi3.push(rnd3());
    write("field.json", hx({
        p: F.p, w: F.w, wi: F.wi, shiftInv: F.shiftInv, mul, inv, ext: ext.map((e) => [e.a, e.b, e.mul, e.inv]),
        batchInverse: [bi, F.batchInverse(bi)], batchInverse3: [bi3, F.batchInverse(bi3)],
    }));
}

// ---------------------------------------------------------------- scalar NTT / extendPol  (fft.js:118-174, polutils.js:18-30)
{
    const cases = [];
    const mk = (name, nBits, gen) => { const p = []; for (let i = 0; i < (1 << nBits); i++) p.push(gen(i)); return { name, nBits, p }; };
    const ins = [mk("index3", 3, (i) => BigInt(i)), mk("index5", 5, (i) => BigInt(i)), mk("one0", 0, () => 5n),
        mk("rand1", 1, rndF), mk("rand4", 4, rndF), mk("rand7", 7, rndF), mk("rand10", 10, rndF),
        mk("edge4", 4, (i) => EDGE[i % EDGE.length])];
    for (const c of ins) {
        const o = { name: c.name, nBits: c.nBits, p: c.p, fft: F.fft(c.p.slice()), ifft: F.ifft(c.p.slice()), ext: {} };
        for (const eb of [1, 2, 3]) if (c.nBits >= 1 && c.nBits + eb <= 11) o.ext[eb] = extendPol(F, c.p.slice(), eb);
        cases.push(o);
    }
    // extension-field ifft (used by FRI fold, fri.js:55) on triples
    const e3 = []; for (let i = 0; i < 16; i++) e3.push(rnd3());
    write("ntt.json", hx({ cases, ext3: { p: e3, fft: F.fft(e3.slice()), ifft: F.ifft(e3.slice()) } }));
}

// ---------------------------------------------------------------- Poseidon (test/poseidon.test.js:14,26,38 + random)
{
    const v = [];
    const m1 = F.p - 1n;
    v.push({ in: [0n, 0n, 0n, 0n, 0n, 0n, 0n, 0n], cap: [0n, 0n, 0n, 0n] });
    v.push({ in: [0n, 1n, 2n, 3n, 4n, 5n, 6n, 7n], cap: [8n, 9n, 10n, 11n] });
    v.push({ in: [m1, m1, m1, m1, m1, m1, m1, m1], cap: [m1, m1, m1, m1] });
    for (let i = 0; i < 64; i++) { const a = [], c = []; for (let k = 0; k < 8; k++) a.push(rndF()); for (let k = 0; k < 4; k++) c.push(rndF()); v.push({ in: a, cap: c }); }
    for (let i = 0; i < 8; i++) { const a = [], c = []; for (let k = 0; k < 8; k++) a.push(EDGE[(i + k) % EDGE.length]); for (let k = 0; k < 4; k++) c.push(EDGE[(i + 3 * k) % EDGE.length]); v.push({ in: a, cap: c }); }
    write("poseidon.json", hx(v.map((x) => [x.in, x.cap, poseidon(x.in, x.cap, 12)])));
}

// ---------------------------------------------------------------- linear hash, both variants (test/glwasm.test.js:17-45,198-230)
{
    const lh = new LinearHash(poseidon), lhg = new LinearHashGPU(poseidon);
    const idx = [], rand = [];
    for (let w = 0; w <= 50; w++) { const a = []; for (let i = 0; i < w; i++) a.push(BigInt(i)); idx.push([w, lh.hash(a), lhg.hash(a)]); }
    for (const w of [1, 3, 4, 5, 8, 9, 12, 16, 17, 31, 32, 33, 40, 64, 100, 128, 129, 200]) {
        const a = []; for (let i = 0; i < w; i++) a.push(rndF());
        rand.push([a, lh.hash(a), lhg.hash(a)]);
    }
    write("linearhash.json", hx({ index: idx, random: rand }));
}

// ---------------------------------------------------------------- Merkle roots ("composed": leaves = lh.hash(row), parent =
// poseidon([...l, ...r]) with zero-digest padding of odd levels, as merklehash_p.js:44-133 / merklehash.js:66-90 do)
{
    function root(N, w, split) {
        const lh = split ? new LinearHashGPU(poseidon) : new LinearHash(poseidon);
        let lvl = [];
        for (let i = 0; i < N; i++) { const r = []; for (let j = 0; j < w; j++) r.push(BigInt(i + 1000 * j)); lvl.push(lh.hash(r)); }   // test/merklehash_p.test.js:27-31
        const levels = [lvl];
        while (lvl.length > 1) {
            if (lvl.length & 1) lvl.push([0n, 0n, 0n, 0n]);
            const nx = []; for (let i = 0; i < lvl.length; i += 2) nx.push(poseidon([...lvl[i], ...lvl[i + 1]]));
            lvl = nx; levels.push(lvl);
        }
        return { N, w, split, root: lvl[0], leaf0: levels[0][0], leafLast: levels[0][N - 1] };
    }
    const out = [];
    for (const [N, w] of [[256, 3], [256, 9], [33, 6], [256, 10], [2, 9], [3, 5], [64, 100], [1024, 8]]) for (const s of [false, true]) out.push(root(N, w, s));
    write("merkle.json", hx(out.map((o) => [o.N, o.w, o.split ? 1 : 0, o.root, o.leaf0, o.leafLast])));
}

// ---------------------------------------------------------------- transcript (transcript.js:2-85)
{
    const out = [];
    { const t = new Transcript(poseidon); t.put([1n, 2n, 3n, 4n]); out.push({ put: [[1n, 2n, 3n, 4n]], perms: [8, 11, t.getPermutations(8, 11)] }); }
    { const t = new Transcript(poseidon); const a = []; for (let i = 0; i < 13; i++) a.push(rndF()); t.put(a); const f1 = t.getField(); const b = rnd3(); t.put(b); const f2 = t.getField(); const f3 = t.getField();
        out.push({ put: [a, b], fields: [f1, f2, f3], state: t.getState() }); }
    { const t = new Transcript(poseidon); const a = []; for (let i = 0; i < 8; i++) a.push(rndF()); t.put(a); const f1 = t.getField(); out.push({ put: [a], fields: [f1], perms: [32, 17, t.getPermutations(32, 17)] }); }
    { const t = new Transcript(poseidon); const a = []; for (let i = 0; i < 20; i++) a.push(rndF()); t.put(a); out.push({ put: [a], state: t.getState() }); }
    write("transcript.json", hx(out));
}

// ---------------------------------------------------------------- FRI fold ("composed" from F.ifft/polMulAxi/evalPol exactly as fri.js:31-60)
{
    function fold(pol, polBits, outBits, bits0, bitsPrev, challenge) {
        let shiftInv = F.shiftInv;
        for (let j = 0; j < bits0 - bitsPrev; j++) shiftInv = F.mul(shiftInv, shiftInv);
        const pol2N = 1 << outBits, nX = pol.length / pol2N;
        const res = new Array(pol2N);
        let sinv = shiftInv;
        const wi = F.inv(F.w[polBits]);
        for (let g = 0; g < pol2N; g++) {
            const ppar = new Array(nX);
            for (let i = 0; i < nX; i++) ppar[i] = pol[i * pol2N + g];
            const ppar_c = F.ifft(ppar);
            polMulAxi(F, ppar_c, F.one, sinv);
            res[g] = evalPol(F, ppar_c, challenge);
            sinv = F.mul(sinv, wi);
        }
        return res;
    }
    const out = [];
    for (const [polBits, outBits, bits0] of [[7, 3, 11], [11, 7, 11], [6, 5, 9], [8, 4, 8], [5, 0, 9], [10, 5, 12]]) {
        const pol = []; for (let i = 0; i < (1 << polBits); i++) pol.push(rnd3());
        const ch = rnd3();
        out.push([polBits, outBits, bits0, polBits /* = bitsPrev */, ch, pol, fold(pol, polBits, outBits, bits0, polBits, ch)]);
    }
    write("fri_fold.json", hx(out));
}

// ---------------------------------------------------------------- zerofiers (polutils.js:39-102), run as-is with an array-backed buffer
{
    const mkbuf = (n) => { const a = new Array(n).fill(0n); return { a, getElement: (i) => a[i], setElement: (i, v) => { a[i] = v; } }; };
    const out = [];
    for (const [nBits, nBitsExt] of [[3, 4], [4, 6], [5, 8]]) {
        const extN = 1 << nBitsExt;
        const zh = mkbuf(extN); buildZhInv(zh, 0, F, nBits, nBitsExt, true);
        const first = mkbuf(extN); buildOneRowZerofierInv(first, 0, F, zh, nBits, nBitsExt, 0, true);
        const last = mkbuf(extN); buildOneRowZerofierInv(last, 0, F, zh, nBits, nBitsExt, (1 << nBits) - 1, true);
        const frame = mkbuf(extN); buildFrameZerofierInv(frame, 0, F, zh, nBits, nBitsExt, { offsetMin: 2, offsetMax: 1 }, true);
        out.push([nBits, nBitsExt, zh.a, first.a, last.a, frame.a]);
    }
    write("zerofiers.json", hx(out));
}

// ---------------------------------------------------------------- stage-2 hints: calculateZ / calculateS / calculateH1H2 (polutils.js:105-164), run as they are
async function hints() {
    const { calculateZ, calculateS, calculateH1H2 } = require(path.join(ref, "src/helpers/polutils.js"));
    const col = (n, dim) => { const a = []; for (let i = 0; i < n; i++) a.push(dim === 1 ? (rndF() || 1n) : rnd3()); return a; };
    const z = [], s = [], h = [];
    for (const [n, dn, dd] of [[1, 1, 1], [2, 1, 1], [16, 1, 1], [37, 3, 1], [64, 1, 3], [50, 3, 3], [300, 3, 3]]) {
        const num = col(n, dn), den = col(n, dd);
        z.push({ n, dimNum: dn, dimDen: dd, num, den, gprod: await calculateZ(F, num, den) });
        const one = dn === 1 ? rndF() : rnd3();            // calculateS takes ONE numerator (polutils.js:153)
        s.push({ n, dimNum: dn, dimDen: dd, num: one, den, gsum: await calculateS(F, one, den) });
    }
    // a grand product that closes: numerators = the denominators rotated by one row (a permutation argument)
    { const den = col(33, 3), num = den.slice(1).concat([den[0]]); z.push({ n: 33, dimNum: 3, dimDen: 3, num, den, gprod: await calculateZ(F, num, den) }); }
    for (const [n, dim, distinct] of [[1, 1, 1], [8, 1, 3], [64, 1, 64], [200, 3, 17], [128, 3, 128], [100, 1, 1]]) {
        const vals = col(distinct, dim);
        const t = []; for (let i = 0; i < n; i++) t.push(distinct < n ? vals[Number(rnd64() % BigInt(distinct))] : vals[i]);
        const f = []; for (let i = 0; i < n; i++) f.push(t[Number(rnd64() % BigInt(n))]);
        if (n >= 8) for (let i = 0; i < n / 4; i++) f[i] = t[0];                       // one heavily used entry
        const [h1, h2] = calculateH1H2(F, f, t);
        h.push({ n, dim, f, t, h1, h2 });
    }
    write("hints.json", hx({ gprod: z, gsum: s, h1h2: h }));
}

// ---------------------------------------------------------------- proof2zkin (src/proof2zkin.js:1-75) on synthetic proof objects of every shape it distinguishes
function zkin() {
    const { proof2zkin } = require(path.join(ref, "src/proof2zkin.js"));
    const dig = () => [rndF(), rndF(), rndF(), rndF()];
    const list = (n, g) => { const a = []; for (let i = 0; i < n; i++) a.push(g()); return a; };
    const out = [];
    for (const [nStages, widths, nSub, steps, nQueries] of [[1, { cm1: 2, cm2: 3 }, 0, [5, 3, 1], 2], [2, { cm1: 3, cm2: 4, cm3: 6 }, 0, [6, 2], 3],
        [3, { cm1: 2, cm2: 0, cm3: 5, cm4: 3 }, 2, [7, 4, 2], 2], [2, { cm1: 1, cm2: 2, cm3: 3 }, 1, [4], 1]]) {
        const qStage = nStages + 1;
        const starkInfo = { starkStruct: { nQueries, steps: steps.map((b) => ({ nBits: b })) }, nStages, nSubproofValues: nSub, mapSectionsN: widths };
        const p = { evals: list(5, rnd3), subproofValues: list(nSub, rnd3), fri: [] };
        for (let s = 1; s <= qStage; s++) p["root" + s] = dig();
        const opening = (w, levels) => [list(w, rndF), list(levels, dig)];
        p.fri.push({ polQueries: list(nQueries, () => { const q = []; for (let s = 1; s <= qStage; s++) q.push(opening(widths["cm" + s], steps[0])); q.push(opening(2, steps[0])); return q; }) });
        for (let i = 1; i < steps.length; i++) p.fri.push({ root: dig(), polQueries: list(nQueries, () => opening(3 << (steps[i - 1] - steps[i]), steps[i])) });
        p.fri.push(list(1 << steps[steps.length - 1], rnd3));
        out.push({ starkInfo, proof: p, zkin: proof2zkin(p, starkInfo) });
    }
    write("proof2zkin.json", hx(out));
}

hints().then(zkin);
