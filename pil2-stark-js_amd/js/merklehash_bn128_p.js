// Drop-in for src/helpers/hash/merklehash/merklehash_bn128_p.js: `await buildMerkleHash(arity, custom)` -> MH with
// merkelize / getElement / getGroupProof / calculateRootFromGroupProof / verifyGroupProof / eqRoot / root /
// writeToFile / readFromFile (merklehash_bn128_p.js:10-285).  tree.nodes is a BigUint64Array of Montgomery-form field
// elements (4 words each) laid out as the reference's (:31-45, :89-101); roots and siblings are BigInt in normal form.
// The reference obtains its permutation from circomlibjs / wasmcurves; here it is libpil2gl's (csrc/bn128.hip).
"use strict";
const fs = require("fs");
const { addon, isFlat, upload } = require("./native.js");

const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const M64 = 0xFFFFFFFFFFFFFFFFn;

function toWords(vals) {
    const a = new BigUint64Array(4 * vals.length);
    for (let i = 0; i < vals.length; i++) {
        let v = BigInt(vals[i]) % R; if (v < 0n) v += R;
        for (let k = 0; k < 4; k++) a[4 * i + k] = (v >> BigInt(64 * k)) & M64;
    }
    return a;
}
function fromWords(a, i) { return a[4 * i] | (a[4 * i + 1] << 64n) | (a[4 * i + 2] << 128n) | (a[4 * i + 3] << 192n); }

// circomlibjs poseidon(inputs, initState, nOut) with BigInt in / out
function poseidon(inputs, initState, nOut) {
    nOut = nOut || 1;
    const out = new BigUint64Array(4 * nOut);
    addon.bn128Poseidon(toWords(inputs), toWords([initState || 0n]), 1, inputs.length, nOut, out);
    const res = [];
    for (let i = 0; i < nOut; i++) res.push(fromWords(out, i));
    return nOut === 1 ? res[0] : res;
}
// transcript.bn128.js:56-66 for a list: the full blocks of nIn elements in `flat` absorbed one after the other (each
// permutation's output 0 is the next one's state element 0) in one device call -> the nIn+1 outputs of the last one
poseidon.absorbChain = function (flat, initState, nIn) {
    const out = new BigUint64Array(4 * (nIn + 1));
    addon.bn128SpongeAbsorb(toWords(flat), flat.length / nIn, nIn, toWords([initState || 0n]), out);
    const res = [];
    for (let i = 0; i <= nIn; i++) res.push(fromWords(out, i));
    return res;
};
function fromMontgomery(words) {
    const n = words.length / 4, out = new BigUint64Array(words.length);
    addon.bn128Convert(words, n, 0, out);
    const res = [];
    for (let i = 0; i < n; i++) res.push(fromWords(out, i));
    return res;
}

class LinearHashBN {    // linearhash.bn128.js:4-62
    constructor(arity, custom) { this.arity = arity; this.custom = custom; }
    hash(vals) {
        const flat = [];
        for (let i = 0; i < vals.length; i++) {
            if (Array.isArray(vals[i])) for (let k = 0; k < vals[i].length; k++) flat.push(BigInt(vals[i][k])); else flat.push(BigInt(vals[i]));
        }
        const vals3 = [];
        for (let i = 0; i < flat.length; i += 3) {
            let acc = 0n;
            for (let k = 0; k < 3 && i + k < flat.length; k++) acc += flat[i + k] << BigInt(64 * k);
            vals3.push(acc % R);
        }
        if (vals3.length == 0) return 0n;
        if (vals3.length == 1) return vals3[0];
        let st = 0n, inHash = [];
        for (let i = 0; i < vals3.length; i++) {
            inHash.push(vals3[i]);
            if (inHash.length == this.arity) { st = poseidon(inHash, st); inHash = []; }
        }
        if (inHash.length > 0) {
            while (inHash.length % this.arity !== 0 && this.custom) inHash.push(0n);
            st = poseidon(inHash, st);
        }
        return st;
    }
}

class MerkleHash {
    constructor(arity, custom) {
        this.arity = arity; this.custom = custom;
        this.lh = new LinearHashBN(arity, custom);
        this.poseidon = poseidon;
    }

    _getNNodes(n) { return addon.bn128MerkleNumNodes(n, this.arity); }      // merklehash_bn128_p.js:31-45

    async merkelize(buff, width, height) {
        const tree = { elements: buff, nodes: new BigUint64Array(this._getNNodes(height) * 4), width, height };
        if (isFlat(buff)) {
            addon.bn128Merkelize(buff, width, height, this.arity, this.custom ? 1 : 0, tree.nodes);
        } else {
            const dEl = addon.devAlloc(width * height);
            let dNodes;
            try {
                dNodes = addon.devAlloc(tree.nodes.length);
                upload(dEl, buff, width * height);
                addon.bn128MerkelizeDev(dEl, width, height, this.arity, this.custom ? 1 : 0, dNodes);
                addon.devDownload(tree.nodes, dNodes, 0);
            } finally {
                addon.devFree(dEl);
                if (dNodes !== undefined) addon.devFree(dNodes);
            }
        }
        return tree;
    }

    getElement(tree, idx, subIdx) {
        const e = tree.elements;
        return isFlat(e) ? e[tree.width * idx + subIdx] : e.getElement(tree.width * idx + subIdx);
    }

    getGroupProof(tree, idx) {          // merklehash_bn128_p.js:142-182
        if ((idx < 0) || (idx >= tree.height)) throw new Error("Out of range");
        const v = new Array(tree.width);
        for (let i = 0; i < tree.width; i++) v[i] = this.getElement(tree, idx, i);
        const nBitsArity = Math.ceil(Math.log2(this.arity));
        const mp = [];
        let offset = 0, n = tree.height;
        while (n > 1) {
            const si = idx ^ (idx & (this.arity - 1));
            const grp = fromMontgomery(tree.nodes.slice((offset + si) * 4, (offset + si + this.arity) * 4));
            mp.push(grp.map((g, i) => (i < n ? g : 0n)));
            const nextN = Math.floor((n - 1) / this.arity) + 1;
            offset += nextN * this.arity; n = nextN; idx = idx >> nBitsArity;
        }
        return [v, mp];
    }

    calculateRootFromGroupProof(mp, idx, vals) {    // merklehash_bn128_p.js:184-232
        let value = this.lh.hash(vals);
        const nBitsArity = Math.ceil(Math.log2(this.arity));
        for (let o = 0; o < mp.length; o++) {
            const curIdx = idx & (this.arity - 1);
            idx = idx >> nBitsArity;
            const group = mp[o].map((x) => BigInt(x));
            group[curIdx] = value;
            value = poseidon(group, 0n);
        }
        return value;
    }

    // batch form for the verifier's loops over queries: every sponge chunk and every level is one batched permutation call
    // over all the openings (one BN254 permutation alone has the latency of a wave of them)
    calculateRootsFromGroupProofs(proofs, idxs) {
        const n = proofs.length;
        if (n === 0) return [];
        const batch = (rows, init) => {         // rows: n x nIn BigInt -> n outputs
            const nIn = rows[0].length, out = new BigUint64Array(4 * n);
            addon.bn128Poseidon(toWords([].concat(...rows)), init ? toWords(init) : null, n, nIn, 1, out);
            const res = [];
            for (let q = 0; q < n; q++) res.push(fromWords(out, q));
            return res;
        };
        const els = proofs.map(([vals]) => {
            const flat = [];
            for (const v of vals) { if (Array.isArray(v)) for (const x of v) flat.push(BigInt(x)); else flat.push(BigInt(v)); }
            const e = [];
            for (let i = 0; i < flat.length; i += 3) { let acc = 0n; for (let k = 0; k < 3 && i + k < flat.length; k++) acc += flat[i + k] << BigInt(64 * k); e.push(acc % R); }
            return e;
        });
        const nEl = els[0].length, nl = proofs[0][1].length;
        if (els.some((e) => e.length !== nEl) || proofs.some((p) => p[1].length !== nl)) throw new Error("openings of different shapes in one batch");
        let value;
        if (nEl === 0) value = els.map(() => 0n);
        else if (nEl === 1) value = els.map((e) => e[0]);
        else {
            value = els.map(() => 0n);
            for (let i = 0; i < nEl; i += this.arity) {
                let chunks = els.map((e) => e.slice(i, i + this.arity));
                if (chunks[0].length < this.arity && this.custom) chunks = chunks.map((c) => c.concat(new Array(this.arity - c.length).fill(0n)));
                value = batch(chunks, value);
            }
        }
        const nBitsArity = Math.ceil(Math.log2(this.arity)), pos = idxs.map((i) => Number(i));
        for (let o = 0; o < nl; o++) {
            const groups = proofs.map(([, mp], q) => { const g = mp[o].map((x) => BigInt(x)); g[pos[q] & (this.arity - 1)] = value[q]; pos[q] = pos[q] >> nBitsArity; return g; });
            value = batch(groups, null);
        }
        return value;
    }
    verifyGroupProofs(root, proofs, idxs) { return this.calculateRootsFromGroupProofs(proofs, idxs).every((r) => this.eqRoot(r, root)); }

    eqRoot(r1, r2) { return BigInt(r1) === BigInt(r2); }

    verifyGroupProof(root, mp, idx, groupElements) {
        return this.eqRoot(this.calculateRootFromGroupProof(mp, idx, groupElements), root);
    }

    root(tree) { return fromMontgomery(tree.nodes.slice(tree.nodes.length - 4))[0]; }

    async writeToFile(tree, fileName) {     // merklehash_bn128_p.js:243-263
        const fd = await fs.promises.open(fileName, "w+");
        await fd.write(new Uint8Array(BigUint64Array.from([BigInt(tree.width), BigInt(tree.height)]).buffer));
        const el = tree.elements;
        const n = tree.width * tree.height;
        const chunk = 1 << 22;
        for (let i = 0; i < n; i += chunk) {
            const sb = isFlat(el) ? el.subarray(i, Math.min(n, i + chunk)) : el.slice(i, Math.min(n, i + chunk));
            await fd.write(new Uint8Array(sb.buffer, sb.byteOffset, sb.byteLength));
        }
        await fd.write(new Uint8Array(tree.nodes.buffer, tree.nodes.byteOffset, tree.nodes.byteLength));
        await fd.close();
    }

    async readFromFile(fileName) {
        const fd = await fs.promises.open(fileName, "r");
        const header = new BigUint64Array(2);
        await fd.read(new Uint8Array(header.buffer), 0, 16, 0);
        const tree = { width: Number(header[0]), height: Number(header[1]) };
        tree.elements = new BigUint64Array(tree.width * tree.height);
        tree.nodes = new BigUint64Array(this._getNNodes(tree.height) * 4);
        await fd.read(new Uint8Array(tree.elements.buffer), 0, tree.elements.byteLength, 16);
        await fd.read(new Uint8Array(tree.nodes.buffer), 0, tree.nodes.byteLength, 16 + tree.elements.byteLength);
        await fd.close();
        return tree;
    }
}

module.exports = async function buildMerkleHash(arity, custom) { return new MerkleHash(arity, !!custom); };
module.exports.poseidon = poseidon;
module.exports.LinearHashBN = LinearHashBN;
