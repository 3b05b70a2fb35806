// Drop-in for src/stark/fri.js: new FRI(starkStruct, MH); fold(step, pol, challenge) -> {pol, tree, proof};
// proofQueries(proof, trees, friQueries); verify(friChallenges, friQueries, proof, checkQuery) (fri.js:7-175).  `pol` is an
// array of [a,b,c] BigInt triples as in the reference; the fold and the transposition run on the GPU, and the verifier
// checks a layer's openings and folds its groups for all queries in one device call each.
"use strict";
const { addon, isDev, DevBuffer } = require("./native.js");
const P = 0xFFFFFFFF00000001n;

function powmod(b, e) { let r = 1n; b %= P; while (e > 0n) { if (e & 1n) r = r * b % P; b = b * b % P; e >>= 1n; } return r; }
function pack(pol) { const a = new BigUint64Array(pol.length * 3); for (let i = 0; i < pol.length; i++) { a[3 * i] = pol[i][0]; a[3 * i + 1] = pol[i][1]; a[3 * i + 2] = pol[i][2]; } return a; }
function unpack(a) { const r = new Array(a.length / 3); for (let i = 0; i < r.length; i++) r[i] = [a[3 * i], a[3 * i + 1], a[3 * i + 2]]; return r; }
const W32 = 7277203076849721926n;                     // f3g.js:40; w[k] = w[k+1]^2 (fft.js:45-50)
function rootOfUnity(bits) { let w = W32; for (let i = bits; i < 32; i++) w = w * w % P; return w; }
function log2(n) { let b = 0; while ((1 << b) < n) b++; return b; }

class FRI {
    constructor(starkStruct, MH) {
        if (!starkStruct) throw new Error("stark struct not defined");
        this.inNBits = starkStruct.nBitsExt;
        this.maxDegNBits = starkStruct.nBits;
        this.nQueries = starkStruct.nQueries;
        this.steps = starkStruct.steps;
        this.MH = MH;
    }

    async fold(step, pol, challenge) {
        if (isDev(pol)) return this.foldDev(step, pol, challenge);
        const polBits = log2(pol.length);
        if (step === 0) { if (polBits !== this.inNBits) throw new Error("Invalid polynomial size"); }
        else if ((1 << polBits) !== pol.length) throw new Error("Invalid polynomial size");
        let shiftInv = powmod(7n, P - 2n);                                    // fri.js:31-36
        if (step > 0) for (let j = 0; j < this.steps[0].nBits - this.steps[step - 1].nBits; j++) shiftInv = shiftInv * shiftInv % P;
        const outBits = this.steps[step].nBits;
        let pol2_e, flat;
        if (step === 0) { pol2_e = pol; flat = null; }                        // fri.js:48-49
        else {
            flat = new BigUint64Array(3 * 2 ** outBits);
            addon.friFold(pack(pol), polBits, outBits, shiftInv, BigUint64Array.from(challenge, BigInt), flat);
            pol2_e = unpack(flat);
        }
        let tree, proof;
        if (step !== this.steps.length - 1) {                                 // fri.js:64-71
            const nGroups = 1 << this.steps[step + 1].nBits;
            const groupSize = (1 << this.steps[step].nBits) / nGroups;
            const tb = new BigUint64Array(3 * pol2_e.length);
            addon.friTranspose(flat || pack(pol2_e), outBits, this.steps[step + 1].nBits, tb);
            tree = await this.MH.merkelize(tb, 3 * groupSize, nGroups);
            proof = { root: this.MH.root(tree) };
        } else {
            proof = pol2_e.slice();
        }
        return { pol: pol2_e, tree, proof };
    }

    // the same step on a device-resident polynomial (3 * 2^bits words): folds, transposes and commits without leaving HBM;
    // only the last step's polynomial is returned as the reference's array of triples
    async foldDev(step, pol, challenge) {
        const polBits = log2(pol.length / 3);
        if (3 * 2 ** polBits !== pol.length || (step === 0 && polBits !== this.inNBits)) throw new Error("Invalid polynomial size");
        let shiftInv = powmod(7n, P - 2n);
        if (step > 0) for (let j = 0; j < this.steps[0].nBits - this.steps[step - 1].nBits; j++) shiftInv = shiftInv * shiftInv % P;
        const outBits = this.steps[step].nBits;
        let pol2 = pol;
        if (step > 0) {
            pol2 = new DevBuffer(3 * 2 ** outBits);
            addon.friFoldDev(pol.ptr, polBits, outBits, shiftInv, BigUint64Array.from(challenge, BigInt), pol2.ptr);
        }
        if (step !== this.steps.length - 1) {
            const nGroups = 1 << this.steps[step + 1].nBits, groupSize = (1 << outBits) / nGroups;
            const tb = new DevBuffer(3 * 2 ** outBits);
            addon.friTransposeDev(pol2.ptr, outBits, this.steps[step + 1].nBits, tb.ptr);
            const tree = await this.MH.merkelize(tb, 3 * groupSize, nGroups);
            return { pol: pol2, tree, proof: { root: this.MH.root(tree) } };
        }
        const last = unpack(pol2.toHost());
        return { pol: last, tree: undefined, proof: last.slice() };
    }

    // fri.js:107-174.  checkQuery(polQuery, idx) -> the step-0 group of a query (an array of triples) or a false value.
    // friQueries is reduced in place, as the reference does.
    verify(friChallenges, friQueries, proof, checkQuery) {
        if (proof.length !== this.steps.length + 1) throw new Error("Invalid proof size");
        const nQ = this.nQueries;
        let polBits = this.inNBits, shift = 7n;
        for (let si = 0; si < this.steps.length; si++) {
            const item = proof[si], reductionBits = polBits - this.steps[si].nBits;
            let groups;
            if (si === 0) {
                groups = [];
                for (let i = 0; i < nQ; i++) {
                    const g = checkQuery(item.polQueries[i], friQueries[i]);
                    if (!g) return false;
                    groups.push(pack(g));
                }
            } else {
                const pq = item.polQueries.slice(0, nQ);
                if (!this.MH.verifyGroupProofs(item.root, pq, friQueries.slice(0, nQ))) return false;
                groups = pq.map((q) => BigUint64Array.from(q[0], BigInt));          // split3 (fri.js:179-185) is a reshape
            }
            const nX = groups[0].length / 3, foldBits = log2(nX);
            if ((1 << foldBits) !== nX || groups.some((g) => g.length !== 3 * nX)) throw new Error("Invalid group size");
            const gt = new BigUint64Array(3 * nX * nQ);                             // row i = element i of every query's group
            for (let q = 0; q < nQ; q++) for (let i = 0; i < nX; i++) gt.set(groups[q].subarray(3 * i, 3 * i + 3), 3 * (i * nQ + q));
            const w = rootOfUnity(polBits), sinv = new BigUint64Array(nQ);
            for (let q = 0; q < nQ; q++) sinv[q] = powmod(shift * powmod(w, BigInt(friQueries[q])) % P, P - 2n);   // fri.js:126
            const ev = new BigUint64Array(3 * nQ);
            addon.friVerifyFold(gt, foldBits, nQ, sinv, BigUint64Array.from(friChallenges[si], BigInt), ev);
            for (let q = 0; q < nQ; q++) {
                let nxt;
                if (si < this.steps.length - 1) {
                    const groupIdx = Math.floor(friQueries[q] / (1 << this.steps[si + 1].nBits));
                    nxt = proof[si + 1].polQueries[q][0].slice(3 * groupIdx, 3 * groupIdx + 3);
                } else nxt = proof[si + 1][friQueries[q]];
                for (let k = 0; k < 3; k++) if (BigInt(nxt[k]) % P !== ev[3 * q + k]) return false;
            }
            polBits = this.steps[si].nBits;
            for (let j = 0; j < reductionBits; j++) shift = shift * shift % P;
            if (si < this.steps.length - 1) for (let i = 0; i < friQueries.length; i++) friQueries[i] = friQueries[i] % (1 << this.steps[si + 1].nBits);
        }
        const last = proof[proof.length - 1];
        const maxDeg = (polBits - (this.inNBits - this.maxDegNBits)) < 0 ? 0 : 1 << (polBits - (this.inNBits - this.maxDegNBits));
        if (last.length > 1) {                   // the extension iNTT is the base-field one on each of the three coordinates
            const coef = new BigUint64Array(3 * last.length);
            addon.ifft(pack(last.map((e) => e.map(BigInt))), 3, log2(last.length), coef);
            for (let i = 3 * (maxDeg + 1); i < coef.length; i++) if (coef[i] !== 0n) return false;   // fri.js:166: no division by the shift needed
        }
        return true;
    }

    proofQueries(proof, trees, friQueries) {                                  // fri.js:83-105
        // all queries of a tree in one call where the MerkleHash offers it (same values as a getGroupProof per query)
        const open = (tree, idxs) => (this.MH.getGroupProofs ? this.MH.getGroupProofs(tree, idxs) : idxs.map((i) => this.MH.getGroupProof(tree, i)));
        for (let step = 0; step < this.steps.length; step++) {
            proof[step].polQueries = [];
            if (step === 0) {
                const perTree = trees[step].map((t) => open(t, friQueries));
                for (let i = 0; i < friQueries.length; i++) proof[step].polQueries.push(perTree.map((p) => p[i]));
            } else {
                for (let i = 0; i < friQueries.length; i++) friQueries[i] = friQueries[i] % (1 << this.steps[step].nBits);
                proof[step].polQueries = open(trees[step], friQueries);
            }
        }
    }
}
module.exports = FRI;
