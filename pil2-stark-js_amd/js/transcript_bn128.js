// Drop-in for src/helpers/transcript/transcript.bn128.js (constructor (poseidon, nInputs); put, getField, getFields1,
// getFields253, getState, getPermutations).  `poseidon` is the BigInt form exported by ./merklehash_bn128_p.js (no F.e /
// F.toObject needed).  Rate nInputs, nInputs+1 outputs per permutation, output 0 is the new state; a challenge limb is
// one of the three low 64-bit words of an output, and 253 bits of an output feed the query indices.
"use strict";
const { Duplex, indicesFromFields, absorbAll } = require("./transcript_core.js");
const { poseidon: devicePoseidon } = require("./merklehash_bn128_p.js");
const W64 = 0xFFFFFFFFFFFFFFFFn;

module.exports = class Transcript {
    constructor(poseidon, nInputs) {
        if (typeof poseidon === "number") { nInputs = poseidon; poseidon = undefined; }
        this.poseidon = poseidon || devicePoseidon;
        this.nInputs = nInputs || 16;
        this.limbs = [];                        // 64-bit words of the output currently being cut up (the reference's out3)
        // a permutation drops limbs still waiting (transcript.bn128.js:62); absorbing alone does not (:78-83)
        this.core = new Duplex((block, st) => this.poseidon(block, st, this.nInputs + 1), this.nInputs, 0n, (out) => out[0], () => { this.limbs = []; });
        if (this.poseidon.absorbChain) this.core.chain = (blocks, st) => this.poseidon.absorbChain(blocks, st, this.nInputs);   // lists: one device call
    }
    get state() { return this.core.state; }
    put(a) { absorbAll(this.core, a, BigInt); }
    getFields253() { return this.core.next(); }
    getFields1() {
        if (this.limbs.length === 0) {
            const v = this.core.next();         // may permute, which clears limbs: take the output first, cut it after
            this.limbs = [v & W64, (v >> 64n) & W64, (v >> 128n) & W64];
        }
        return this.limbs.shift();
    }
    getField() { return [this.getFields1(), this.getFields1(), this.getFields1()]; }
    getState() { return this.core.settle(); }
    getPermutations(n, nBits) { return indicesFromFields(() => this.core.next(), n, nBits, 253); }
};
