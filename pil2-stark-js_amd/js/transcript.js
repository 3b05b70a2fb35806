// Drop-in for src/helpers/transcript/transcript.js (:1-85), over the device permutation of ./poseidon.js
"use strict";
class Transcript {
    constructor(poseidon) {
        this.poseidon = poseidon;
        this.state = [0n, 0n, 0n, 0n];
        this.pending = [];
        this.out = [];
    }
    getState() { if (this.pending.length > 0) this.updateState(); return this.state; }
    getField() { return [this.getFields1(), this.getFields1(), this.getFields1()]; }
    getFields1() { if (this.out.length == 0) this.updateState(); return this.out.shift(); }
    put(a) { if (Array.isArray(a)) { for (let i = 0; i < a.length; i++) this.put(a[i]); } else this._add1(a); }
    updateState() {
        while (this.pending.length < 8) this.pending.push(0n);
        this.out = this.poseidon(this.pending, this.state, 12);
        this.pending = [];
        this.state = this.out.slice(0, 4);
    }
    _add1(a) {
        this.out = [];
        this.pending.push(BigInt(a));
        if (this.pending.length == 8) {
            this.out = this.poseidon(this.pending, this.state, 12);
            this.pending = [];
            this.state = this.out.slice(0, 4);
        }
    }
    getPermutations(n, nBits) {
        const res = [];
        const totalBits = n * nBits;
        const NFields = Math.floor((totalBits - 1) / 63) + 1;
        const fields = [];
        for (let i = 0; i < NFields; i++) fields[i] = this.getFields1();
        let curField = 0, curBit = 0n;
        for (let i = 0; i < n; i++) {
            let a = 0;
            for (let j = 0; j < nBits; j++) {
                if ((fields[curField] >> curBit) & 1n) a = a + (1 << j);
                curBit++;
                if (curBit == 63n) { curBit = 0n; curField++; }
            }
            res.push(a);
        }
        return res;
    }
}
module.exports = Transcript;
