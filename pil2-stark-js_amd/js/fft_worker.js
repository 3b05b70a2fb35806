// Drop-in for src/helpers/fft/fft_worker.js of pil2-stark-js: the two operators fft_p.js hands to its worker pool
// (pool.exec("fft_block", ...) fft_p.js:162, pool.exec("interpolatePrepareBlock", ...) fft_p.js:93), same names, same arguments,
// same return value (the block itself, transformed in place), computed by libpil2gl on the MI355X.
//
// A caller that swaps fft_p.js for ./fft_p.js never reaches these: there the whole transform -- bit reversal, every round of
// blocks, the transposes between rounds -- is one call.  They are for a caller that keeps the reference's block loop.
//   buff: BigUint64Array (staged through HBM and written back) or a DevBuffer (in place, nothing staged)
"use strict";
const { addon, isDev, upload, download } = require("./native.js");

function onDevice(buff, fn) {
    if (isDev(buff)) { fn(buff.ptr); return buff; }
    const n = buff.length;
    if (n === 0) return buff;
    const d = addon.devAlloc(n);
    try { upload(d, buff, n); fn(d); download(buff, d, n); } finally { addon.devFree(d); }
    return buff;
}

// fft_worker.js:6-19   row i of the (buff.length / width) x width block is multiplied by start * inc^i
function interpolatePrepareBlock(buff, width, start, inc, st_i, st_n) {
    const height = Math.floor(buff.length / width);
    return onDevice(buff, (d) => addon.interpolatePrepareBlockDev(d, width, height, BigInt(start), BigInt(inc)));
}

// fft_worker.js:62-67  `layers` butterfly stages, ending at stage s of a 2^nBits-row transform, on the 2^blockBits rows of buff that
// stand at row start_pos of that transform
function fft_block(buff, start_pos, nPols, nBits, s, blockBits, layers) {
    if (buff.length < nPols * 2 ** blockBits) throw new Error(`fft_block: ${buff.length} words for a block of 2^${blockBits} rows x ${nPols}`);
    return onDevice(buff, (d) => addon.fftBlockDev(d, start_pos, nPols, nBits, s, blockBits, layers));
}

module.exports.fft_block = fft_block;
module.exports.interpolatePrepareBlock = interpolatePrepareBlock;
