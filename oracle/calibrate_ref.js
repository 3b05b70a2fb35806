// Times the reference's own dependency-free JS (BigInt) twins of the hot path on this machine, for the calibration
// SURVEY.md 8(d) asks for: per-column extendPol (polutils.js:18-37 over fft/fft.js) and the JS Poseidon (poseidon.js:57-108).
// The reference's production path (glwasm.js WASM + worker pool) cannot be loaded here (absent third-party packages), so
// these single-thread numbers are a lower bound of the reference, stated as such.  Run only in the build container:
//   node oracle/calibrate_ref.js        (reads /root/reference or $PIL2_REFERENCE; prints one JSON line)
"use strict";
const path = require("path");
const ref = process.env.PIL2_REFERENCE || "/root/reference";
const F3g = require(path.join(ref, "src/helpers/f3g.js"));
const { extendPol } = require(path.join(ref, "src/helpers/polutils.js"));
const getPoseidon = require(path.join(ref, "src/helpers/hash/poseidon/poseidon.js"));
const F = new F3g(), poseidon = getPoseidon();
const nBits = Number(process.argv[2] || 14), nPerm = Number(process.argv[3] || 4000);
const col = []; for (let i = 0; i < (1 << nBits); i++) col.push(BigInt(i) * 0x9E3779B97F4A7C15n % F.p);
let t0 = process.hrtime.bigint();
const ext = extendPol(F, col, 3);
const tExt = Number(process.hrtime.bigint() - t0) / 1e9;
const st = []; for (let i = 0; i < 8; i++) st.push(BigInt(i + 1));
let cap = [0n, 0n, 0n, 0n];
t0 = process.hrtime.bigint();
for (let i = 0; i < nPerm; i++) cap = poseidon(st, cap);
const tPos = Number(process.hrtime.bigint() - t0) / 1e9;
console.log(JSON.stringify({ nBits, extendPol_s: tExt, extendPol_check: ext[5].toString(16), nPerm, poseidon_s: tPos, poseidon_check: cap[0].toString(16) }));
