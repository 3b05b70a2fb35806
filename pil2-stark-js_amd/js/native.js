// Loads the N-API addon (../addon/pil2gl.node -> ../lib/libpil2gl.so).  No fallback: if the addon is
// not built or no MI355X is present, requiring this module / calling into it throws.
"use strict";
const path = require("path");
const addon = require(path.join(__dirname, "..", "addon", "pil2gl.node"));

const CHUNK = 1 << 24;          // words per staging transfer for chunked containers (pilcom BigBuffer)

function isFlat(b) { return b instanceof BigUint64Array; }

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

module.exports = { addon, isFlat, upload, download, staged, CHUNK };
