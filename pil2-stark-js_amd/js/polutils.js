// Drop-ins for the data-parallel functions of src/helpers/polutils.js, same names and argument lists:
//   buildZhInv(buffTo, offset, F, nBits, nBitsExt, stark)                                   polutils.js:39-55
//   buildOneRowZerofierInv(buffTo, offset, F, buffZhInv, nBits, nBitsExt, rowIndex, stark)  polutils.js:57-71
//   buildFrameZerofierInv(buffTo, offset, F, buffZhInv, nBits, nBitsExt, frame, stark)      polutils.js:74-102
//   calculateZ(F, num, den), calculateS(F, num, den), calculateH1H2(F, f, t)                polutils.js:105-164
// `F` and `buffZhInv` are accepted and ignored (the field is Goldilocks, the device rebuilds what it needs).
// Only the stark = true forms exist (coset 7 * <w>): the plain-subgroup forms invert zero and are not used by the prover.
// One deviation, on purpose: buildZhInv honours `offset` when it fills rows >= 2^extendBits; the reference's second loop
// (polutils.js:51-53) ignores it and overwrites the table at offset 0 (harmless there because everyRow is boundary 0).
"use strict";
const { addon, isFlat, isDev, download } = require("./native.js");

function setRange(buffTo, offset, tmp) {
    if (isFlat(buffTo)) buffTo.set(tmp, offset); else buffTo.set(tmp, offset);
}
function withOut(n, fn, buffTo, offset) {
    if (isDev(buffTo)) { fn(buffTo.addr(offset)); return; }         // resident table: written in place
    const d = addon.devAlloc(n);
    try {
        fn(d);
        const CH = 1 << 24;
        for (let o = 0; o < n; o += CH) {
            const tmp = new BigUint64Array(Math.min(CH, n - o));
            addon.devDownload(tmp, d, o);
            setRange(buffTo, offset + o, tmp);
        }
    } finally { addon.devFree(d); }
}
function needStark(stark) { if (!stark) throw new Error("only the stark (coset) form is implemented"); }

module.exports.buildZhInv = function buildZhInv(buffTo, offset, F, nBits, nBitsExt, stark) {
    needStark(stark);
    withOut(1 << nBitsExt, (d) => addon.buildZhInvDev(nBits, nBitsExt, d), buffTo, offset);
};
module.exports.buildOneRowZerofierInv = function buildOneRowZerofierInv(buffTo, offset, F, buffZhInv, nBits, nBitsExt, rowIndex, stark) {
    needStark(stark);
    withOut(1 << nBitsExt, (d) => addon.buildOneRowZerofierInvDev(nBits, nBitsExt, rowIndex, d), buffTo, offset);
};
module.exports.buildFrameZerofierInv = function buildFrameZerofierInv(buffTo, offset, F, buffZhInv, nBits, nBitsExt, frame, stark) {
    needStark(stark);
    withOut(1 << nBitsExt, (d) => addon.buildFrameZerofierDev(nBits, nBitsExt, frame.offsetMin, frame.offsetMax, d), buffTo, offset);
};

// columns as the reference passes them: arrays of BigInt (dim 1) or of [a,b,c] (dim 3)
function pack(col) {
    const dim = Array.isArray(col[0]) ? 3 : 1;
    const a = new BigUint64Array(col.length * dim);
    if (dim === 1) for (let i = 0; i < col.length; i++) a[i] = BigInt(col[i]);
    else for (let i = 0; i < col.length; i++) { a[3 * i] = BigInt(col[i][0]); a[3 * i + 1] = BigInt(col[i][1]); a[3 * i + 2] = BigInt(col[i][2]); }
    return { a, dim };
}
function unpack(a, dim) {
    const n = a.length / dim, out = new Array(n);
    if (dim === 1) for (let i = 0; i < n; i++) out[i] = a[i];
    else for (let i = 0; i < n; i++) out[i] = [a[3 * i], a[3 * i + 1], a[3 * i + 2]];
    return out;
}
function onDevice(arrays, nOutWords, fn) {
    const ptrs = [];
    try {
        for (const a of arrays) { const d = addon.devAlloc(a.length); ptrs.push(d); addon.devUpload(d, 0, a); }
        const outs = nOutWords.map((n) => { const d = addon.devAlloc(n); ptrs.push(d); return d; });
        fn(ptrs.slice(0, arrays.length), outs);
        return outs.map((d, i) => { const r = new BigUint64Array(nOutWords[i]); addon.devDownload(r, d, 0); return r; });
    } finally { for (const d of ptrs) addon.devFree(d); }
}

module.exports.calculateZ = async function calculateZ(F, num, den) {
    const N = den.length, pn = pack(num), pd = pack(den), dimOut = Math.max(pn.dim, pd.dim);
    const [z] = onDevice([pn.a, pd.a], [N * dimOut], ([dn, dd], [dz]) => addon.gprodDev(dn, pn.dim, dd, pd.dim, N, dz));
    return unpack(z, dimOut);
};
module.exports.calculateS = async function calculateS(F, num, den) {
    const N = den.length, pn = pack([num]), pd = pack(den), dimOut = Math.max(pn.dim, pd.dim);
    const [s] = onDevice([pn.a, pd.a], [N * dimOut], ([dn, dd], [ds]) => addon.gsumDev(dn, pn.dim, dd, pd.dim, N, ds));
    return unpack(s, dimOut);
};
module.exports.calculateH1H2 = function calculateH1H2(F, f, t) {
    const pf = pack(f), pt = pack(t);
    if (pf.dim !== pt.dim || f.length !== t.length) throw new Error("calculateH1H2: f and t must have the same shape");
    const n = t.length;
    const [h1, h2] = onDevice([pf.a, pt.a], [n * pt.dim, n * pt.dim], ([df, dt], [d1, d2]) => addon.h1h2Dev(df, dt, n, pt.dim, d1, d2));
    return [unpack(h1, pt.dim), unpack(h2, pt.dim)];
};
