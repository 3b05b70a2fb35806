// Loads the N-API addon (../addon/pil2gl.node -> ../lib/libpil2gl.so).  No fallback: if the addon is
// not built or no MI355X is present, requiring this module / calling into it throws.
"use strict";
const path = require("path");
const addon = require(path.join(__dirname, "..", "addon", "pil2gl.node"));

const CHUNK = 1 << 24;          // words per staging transfer for chunked containers (pilcom BigBuffer)

function isFlat(b) { return b instanceof BigUint64Array; }

// A BigBuffer-shaped container that lives in HBM: hand one of these to any of the drop-in modules (as ctx.cm1_ext,
// tree.elements, a FRI polynomial ...) and that module works on it in place -- nothing is staged through the JS heap.
// It also answers the BigBuffer calls the reference makes (length, getElement, setElement, slice, set), each as a
// small transfer, so untouched reference code keeps working on it.
class DevBuffer {
    constructor(nWords, ptr) { this.length = nWords; this.owned = ptr === undefined; this.ptr = this.owned ? addon.devAlloc(Math.max(1, nWords)) : ptr; }
    static from(arr) { const d = new DevBuffer(arr.length); upload(d.ptr, arr, arr.length); return d; }
    addr(offWords = 0) { return this.ptr + 8n * BigInt(offWords); }
    view(offWords, nWords) { return new DevBuffer(nWords, this.addr(offWords)); }      // no ownership
    getElement(i) { const t = new BigUint64Array(1); addon.devDownload(t, this.ptr, i); return t[0]; }
    setElement(i, v) { addon.devUpload(this.ptr, i, BigUint64Array.of(BigInt(v))); }
    slice(a, b) { if (a < 0) a += this.length; if (b === undefined) b = this.length; const t = new BigUint64Array(Math.max(0, b - a)); if (t.length) addon.devDownload(t, this.ptr, a); return t; }
    set(arr, off = 0) { if (arr instanceof DevBuffer) arr = arr.slice(0, arr.length); addon.devUpload(this.ptr, off, arr); }
    zero() { addon.devZero(this.ptr, 0, this.length); }
    toHost() { return this.slice(0, this.length); }
    free() { if (this.owned && this.ptr !== null) { addon.devFree(this.ptr); this.ptr = null; } }
}
function isDev(b) { return b instanceof DevBuffer; }

// Containers: BigUint64Array, or anything with pilcom.BigBuffer's surface {length, slice(a,b) -> BigUint64Array, set(arr, off)}
// (used by the reference at fft_p.js:28-29,89,116; merklehash_p.js:70; stark_gen_helpers.js:104-137).
function upload(dptr, buf, nWords) {
    if (isFlat(buf)) { addon.devUpload(dptr, 0, nWords === buf.length ? buf : buf.subarray(0, nWords)); return; }
    for (let o = 0; o < nWords; o += CHUNK) addon.devUpload(dptr, o, buf.slice(o, Math.min(nWords, o + CHUNK)));
}
function download(buf, dptr, nWords) {
    if (isFlat(buf)) { addon.devDownload(nWords === buf.length ? buf : buf.subarray(0, nWords), dptr, 0); return; }
    for (let o = 0; o < nWords; o += CHUNK) {
        const tmp = new BigUint64Array(Math.min(CHUNK, nWords - o));
        addon.devDownload(tmp, dptr, o);
        buf.set(tmp, o);
    }
}
// run fn(dIn, dOut) with device staging buffers for containers that are not one flat array
function staged(src, nIn, dst, nOut, fn) {
    const dIn = addon.devAlloc(nIn);
    let dOut;
    try {
        dOut = addon.devAlloc(nOut);
        upload(dIn, src, nIn);
        fn(dIn, dOut);
        download(dst, dOut, nOut);
    } finally {
        addon.devFree(dIn);
        if (dOut !== undefined) addon.devFree(dOut);
    }
}

module.exports = { addon, isFlat, isDev, DevBuffer, upload, download, staged, CHUNK };
