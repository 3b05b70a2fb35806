// Drop-in for src/helpers/transcript/transcript.js (same constructor and methods: put, getField, getFields1, getState,
// getPermutations), over the device Poseidon of ./poseidon.js.  Rate 8, twelve outputs per permutation, the first four of
// which are the new state; 63 bits of an output are used per query-index word (transcript.js:62).
"use strict";
const { Duplex, indicesFromFields, absorbAll } = require("./transcript_core.js");

module.exports = class Transcript {
    constructor(poseidon) {
        this.poseidon = poseidon;
        this.core = new Duplex((block, st) => poseidon(block, st, 12), 8, [0n, 0n, 0n, 0n], (out) => out.slice(0, 4));
        if (poseidon.absorbChain) this.core.chain = (blocks, st) => poseidon.absorbChain(blocks, st);     // the drop-in poseidon.js has it
    }
    get state() { return this.core.state; }
    put(a) { absorbAll(this.core, a, BigInt); }
    getFields1() { return this.core.next(); }
    getField() { return [this.core.next(), this.core.next(), this.core.next()]; }
    getState() { return this.core.settle(); }
    getPermutations(n, nBits) { return indicesFromFields(() => this.core.next(), n, nBits, 63); }
};
