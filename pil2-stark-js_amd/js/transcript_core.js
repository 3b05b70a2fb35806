// One duplex sponge behind both Fiat-Shamir transcripts of the reference (transcript.js for Goldilocks, transcript.bn128.js
// for BN128).  They differ only in the permutation call, the rate, what part of the output is carried as state and how
// many bits of an output feed the query indices, so those are parameters here and the two public classes are thin shells.
//
// Behaviour that must not change (it fixes every challenge of a proof):
//   * absorbing a value invalidates outputs not yet handed out, and a full inbox is permuted at once;
//   * asking for an output with none left zero-pads the inbox and permutes;
//   * outputs are handed out first to last (for Goldilocks that starts with the four state words themselves).
"use strict";

class Duplex {
    // permute(inputs[rate], state) -> outputs[];  carry(outputs) -> next state;  onPermute(): hook run after each permutation
    constructor(permute, rate, zeroState, carry, onPermute) {
        Object.assign(this, { permute, rate, carry, onPermute, state: zeroState, inbox: [], outbox: [] });
    }
    absorb(v) {
        this.outbox = [];
        this.inbox.push(v);
        if (this.inbox.length === this.rate) this.run();
    }
    run() {
        const block = this.inbox.concat(new Array(this.rate - this.inbox.length).fill(0n));
        this.inbox = [];
        this.outbox = this.permute(block, this.state);
        this.state = this.carry(this.outbox);
        if (this.onPermute) this.onPermute();
    }
    next() {
        if (this.outbox.length === 0) this.run();
        return this.outbox.shift();
    }
    settle() { if (this.inbox.length > 0) this.run(); return this.state; }
}

// n indices of nBits bits each, little-endian, read from successive field outputs of which `usable` low bits count
function indicesFromFields(nextField, n, nBits, usable) {
    const words = [];
    for (let need = Math.floor((n * nBits - 1) / usable) + 1; need > 0; need--) words.push(nextField());
    const out = [];
    let w = 0, bit = 0n;
    const lim = BigInt(usable);
    for (let i = 0; i < n; i++) {
        let idx = 0;
        for (let j = 0; j < nBits; j++) {
            if ((words[w] >> bit) & 1n) idx += 1 << j;
            if (++bit === lim) { bit = 0n; w++; }
        }
        out.push(idx);
    }
    return out;
}
// A (nested) list is absorbed block-wise: with a `chain(blocks, state) -> outputs` callback on the core (the Goldilocks
// transcript has one: a single device call for all the full blocks the list completes) the state and outbox end up exactly
// as element-by-element absorbs leave them; without it, element by element.
function absorbAll(core, a, convert) {
    if (!core.chain) { if (Array.isArray(a)) for (const x of a) absorbAll(core, x, convert); else core.absorb(convert(a)); return; }
    const flat = [];
    (function walk(v) { if (Array.isArray(v)) for (const x of v) walk(x); else flat.push(convert(v)); })(a);
    let i = 0;
    while (i < flat.length) {
        const need = core.rate - core.inbox.length;
        if (flat.length - i < need) { for (; i < flat.length; i++) core.inbox.push(flat[i]); core.outbox = []; return; }
        const nFull = 1 + Math.floor((flat.length - i - need) / core.rate), take = need + core.rate * (nFull - 1);
        const blocks = core.inbox.concat(flat.slice(i, i + take));
        core.inbox = [];
        core.outbox = nFull === 1 ? core.permute(blocks, core.state) : core.chain(blocks, core.state);
        core.state = core.carry(core.outbox);
        if (core.onPermute) for (let k = 0; k < nFull; k++) core.onPermute();
        i += take;
    }
}
module.exports = { Duplex, indicesFromFields, absorbAll };
