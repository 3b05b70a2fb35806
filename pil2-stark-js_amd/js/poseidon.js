// Drop-in for src/helpers/hash/poseidon/poseidon.js: getPoseidon() -> poseidon(inputs[8], capacity[4]?, nOuts=4)
// returning BigInt[] (poseidon.js:57-108), one permutation on the GPU per call.
"use strict";
const { addon } = require("./native.js");
const P = 0xFFFFFFFF00000001n;
const e = (a) => { let v = BigInt(a) % P; if (v < 0n) v += P; return v; };        // F.e(), f3g.js:277-293

function poseidon(inputs, capacity, nOuts) {
    nOuts = nOuts || 4;
    if (inputs.length !== 8) throw new Error("Invalid Input size (must be 8)");
    if (capacity && capacity.length !== 4) throw new Error("Invalid Capacity size (must be 4)");
    const i = BigUint64Array.from(inputs, e);
    const c = capacity ? BigUint64Array.from(capacity, e) : null;
    const out = new BigUint64Array(nOuts);
    addon.poseidon(i, c, 1, nOuts, out);
    return Array.from(out);
}
// not in the reference: a list of full blocks absorbed one after the other in one device call (each permutation takes the
// previous one's first four outputs as capacity), for js/transcript.js; -> the last permutation's twelve outputs
poseidon.absorbChain = function absorbChain(blocks, capacity) {
    if (blocks.length === 0 || blocks.length % 8 !== 0) throw new Error("Invalid Input size (must be a multiple of 8)");
    const out = new BigUint64Array(12);
    addon.spongeAbsorb(BigUint64Array.from(blocks, e), blocks.length / 8, BigUint64Array.from(capacity || [0n, 0n, 0n, 0n], e), out);
    return Array.from(out);
};
module.exports = function getPoseidon() { return poseidon; };
