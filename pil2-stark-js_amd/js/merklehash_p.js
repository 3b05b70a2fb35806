// Drop-in for src/helpers/hash/merklehash/merklehash_p.js: `await buildMerkleHash(splitLinearHash)` -> MH with
// merkelize / getElement / getGroupProof / calculateRootFromGroupProof / verifyGroupProof / eqRoot / root /
// writeToFile / readFromFile (merklehash_p.js:12-279).  tree = {elements, nodes, width, height}; elements aliases
// the caller's buffer (:47), nodes is a new BigUint64Array laid out exactly as the reference's (:28-42, :87-103).
"use strict";
const fs = require("fs");
const { addon, isFlat, isDev, DevBuffer, upload } = require("./native.js");
const getPoseidon = require("./poseidon.js");

module.exports = async function buildMerkleHash(splitLinearHash = false) {
    return new MerkleHash(getPoseidon(), splitLinearHash);
};

class LinearHash {      // linearhash.js / linearhash_gpu.js: hash(vals) -> [4 BigInt]
    constructor(split) { this.split = split; }
    hash(vals) {
        const flat = [];
        for (let i = 0; i < vals.length; i++) {
            if (Array.isArray(vals[i])) for (let j = 0; j < vals[i].length; j++) flat.push(vals[i][j]); else flat.push(vals[i]);
        }
        if (flat.length === 0) return [0n, 0n, 0n, 0n];
        const out = new BigUint64Array(4);
        addon.linearHashRows(BigUint64Array.from(flat, BigInt), flat.length, 1, this.split ? 1 : 0, out);
        return Array.from(out);
    }
}

class MerkleHash {
    constructor(poseidon, splitLinearHash = false) {
        this.poseidon = poseidon;
        this.splitLinearHash = splitLinearHash;
        this.lh = new LinearHash(splitLinearHash);
    }

    _getNNodes(n) { return addon.merkleNumNodes(n / 4); }      // merklehash_p.js:28-42, n = 4*height

    async merkelize(buff, width, height) {
        if (isDev(buff)) {          // resident: the tree is built next to its leaves and stays there
            const nodes = new DevBuffer(this._getNNodes(height * 4));
            addon.merkelizeDev(buff.ptr, width, height, this.splitLinearHash ? 1 : 0, nodes.ptr);
            return { elements: buff, nodes, width, height };
        }
        const tree = { elements: buff, nodes: new BigUint64Array(this._getNNodes(height * 4)), width, height };
        if (isFlat(buff)) {
            addon.merkelize(buff, width, height, this.splitLinearHash ? 1 : 0, tree.nodes);
        } else {
            const dEl = addon.devAlloc(width * height);
            let dNodes;
            try {
                dNodes = addon.devAlloc(tree.nodes.length);
                upload(dEl, buff, width * height);
                addon.merkelizeDev(dEl, width, height, this.splitLinearHash ? 1 : 0, dNodes);
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
        return isFlat(e) ? e[tree.width * idx + subIdx] : e.getElement(tree.width * idx + subIdx);     // DevBuffer answers getElement too
    }

    // getGroupProof for every query of a tree (fri.js:83-105 opens each tree at all query rows): for a device-resident
    // tree one gather kernel and one copy back instead of a copy per row and per level
    getGroupProofs(tree, idxs) {
        if (!(isDev(tree.elements) && isDev(tree.nodes)) || idxs.length === 0) return idxs.map((i) => this.getGroupProof(tree, i));
        for (const idx of idxs) if ((idx < 0) || (idx >= tree.height)) throw new Error("Out of range");
        let nl = 0; for (let n = tree.height * 4; n > 4; n = (Math.floor((n - 1) / 8) + 1) * 4) nl++;
        const w = tree.width, stride = w + 4 * nl, out = new BigUint64Array(idxs.length * stride);
        addon.groupProofsDev(tree.elements.ptr, tree.nodes.ptr, w, tree.height, BigUint64Array.from(idxs, BigInt), out);
        return idxs.map((_, q) => {
            const o = q * stride, mp = [];
            for (let l = 0; l < nl; l++) mp.push([out[o + w + 4 * l], out[o + w + 4 * l + 1], out[o + w + 4 * l + 2], out[o + w + 4 * l + 3]]);
            return [Array.from(out.subarray(o, o + w)), mp];
        });
    }

    getGroupProof(tree, idx) {          // merklehash_p.js:142-168
        if ((idx < 0) || (idx >= tree.height)) throw new Error("Out of range");
        if (isDev(tree.elements) && isDev(tree.nodes)) {        // only the opened row and its siblings cross PCIe
            const vals = new BigUint64Array(Math.max(1, tree.width)), sib = new BigUint64Array(4 * 64);
            const nl = addon.groupProofDev(tree.elements.ptr, tree.nodes.ptr, tree.width, tree.height, idx, vals, sib);
            const mpd = [];
            for (let l = 0; l < nl; l++) mpd.push([sib[4 * l], sib[4 * l + 1], sib[4 * l + 2], sib[4 * l + 3]]);
            return [Array.from(vals.subarray(0, tree.width)), mpd];
        }
        const v = new Array(tree.width);
        for (let i = 0; i < tree.width; i++) v[i] = this.getElement(tree, idx, i);
        const mp = [];
        let offset = 0, n = tree.height * 4;
        while (n > 4) {
            const si = (idx ^ 1) * 4;
            mp.push(isFlat(tree.nodes) ? [tree.nodes[offset + si], tree.nodes[offset + si + 1], tree.nodes[offset + si + 2], tree.nodes[offset + si + 3]] : Array.from(tree.nodes.slice(offset + si, offset + si + 4)));
            const nextN = (Math.floor((n - 1) / 8) + 1) * 4;
            offset += nextN * 2; n = nextN; idx = idx >> 1;
        }
        return [v, mp];
    }

    calculateRootFromGroupProof(mp, idx, vals) {    // merklehash_p.js:170-210
        let value = this.lh.hash(vals);
        for (let o = 0; o < mp.length; o++) {
            value = (idx & 1) === 0 ? this.poseidon([...value, ...mp[o]]) : this.poseidon([...mp[o], ...value]);
            idx = Math.floor(idx / 2);
        }
        return value;
    }

    // batch form for the verifier's loops over queries (stark_verify.js:165-178, fri.js:140): proofs = [[vals, siblings], ...]
    // of one tree -> their roots, leaf hashes and path walks on the device in one call
    calculateRootsFromGroupProofs(proofs, idxs) {
        if (proofs.length === 0) return [];
        const width = proofs[0][0].length, nl = proofs[0][1].length, stride = width + 4 * nl, n = proofs.length;
        const packed = new BigUint64Array(n * stride), ii = new BigUint64Array(n), roots = new BigUint64Array(4 * n);
        for (let q = 0; q < n; q++) {
            const [vals, sib] = proofs[q];
            if (vals.length !== width || sib.length !== nl) throw new Error("openings of different shapes in one batch");
            for (let i = 0; i < width; i++) packed[q * stride + i] = BigInt(vals[i]);
            for (let l = 0; l < nl; l++) for (let k = 0; k < 4; k++) packed[q * stride + width + 4 * l + k] = BigInt(sib[l][k]);
            ii[q] = BigInt(idxs[q]);
        }
        addon.rootsFromGroupProofs(packed, width, nl, ii, n, this.splitLinearHash ? 1 : 0, roots);
        const out = [];
        for (let q = 0; q < n; q++) out.push([roots[4 * q], roots[4 * q + 1], roots[4 * q + 2], roots[4 * q + 3]]);
        return out;
    }
    verifyGroupProofs(root, proofs, idxs) { return this.calculateRootsFromGroupProofs(proofs, idxs).every((r) => this.eqRoot(r, root)); }

    eqRoot(r1, r2) { for (let k = 0; k < 4; k++) if (BigInt(r1[k]) !== BigInt(r2[k])) return false; return true; }
    verifyGroupProof(root, mp, idx, groupElements) { return this.eqRoot(this.calculateRootFromGroupProof(mp, idx, groupElements), root); }
    root(tree) { return [...tree.nodes.slice(-4)]; }

    async writeToFile(tree, fileName) {             // merklehash_p.js:228-246
        const fd = await fs.promises.open(fileName, "w+");
        const header = new BigUint64Array(2);
        header[0] = BigInt(tree.width); header[1] = BigInt(tree.height);
        await fd.write(new Uint8Array(header.buffer));
        const MaxBuffSize = 1024 * 1024 * 32;
        for (const buff of [tree.elements, tree.nodes]) {
            for (let i = 0; i < buff.length; i += MaxBuffSize) {
                const sb = buff.slice(i, Math.min(buff.length, i + MaxBuffSize));
                await fd.write(new Uint8Array(sb.buffer, sb.byteOffset, sb.byteLength));
            }
        }
        await fd.close();
    }

    async readFromFile(fileName) {                  // merklehash_p.js:248-278 (elements returned as one BigUint64Array)
        const fd = await fs.promises.open(fileName, "r");
        const header = new BigUint64Array(2);
        await fd.read(new Uint8Array(header.buffer), 0, 16, 0);
        const tree = { width: Number(header[0]), height: Number(header[1]) };
        tree.elements = new BigUint64Array(tree.width * tree.height);
        tree.nodes = new BigUint64Array(this._getNNodes(tree.height * 4));
        let pos = 16;
        for (const buff of [tree.elements, tree.nodes]) {
            const b8 = new Uint8Array(buff.buffer, buff.byteOffset, buff.byteLength);
            const CH = 1 << 28;
            for (let o = 0; o < b8.length; o += CH) await fd.read(b8, o, Math.min(CH, b8.length - o), pos + o);
            pos += b8.length;
        }
        await fd.close();
        return tree;
    }
}
module.exports.MerkleHash = MerkleHash;
// The reference's own operator granularity (merklehash_worker.js:37-117, the functions its worker pool runs on slices of the rows):
//   linearHash(buffIn, width, st_i, st_n, splitLinearHash) -> BigUint64Array(height * 4)      one digest per row of buffIn
//   merkelizeLevel(buffIn, st_i, st_n)                     -> BigUint64Array(nOps * 4)        one parent per 8 input words
// st_i / st_n (the slice's position, used there for logging only) are accepted and ignored.
module.exports.linearHash = async function linearHash(buffIn, width, st_i, st_n, splitLinearHash) {
    const height = width ? buffIn.length / width : 0;
    const out = new BigUint64Array(height * 4);
    if (height) addon.linearHashRows(buffIn, width, height, splitLinearHash ? 1 : 0, out);
    return out;
};
module.exports.merkelizeLevel = async function merkelizeLevel(buffIn, st_i, st_n) {
    const nOps = buffIn.length / 8;
    const out = new BigUint64Array(nOps * 4);
    if (nOps) addon.merkelizeLevel(buffIn, nOps, out);
    return out;
};
