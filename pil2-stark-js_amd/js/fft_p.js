// Drop-in for src/helpers/fft/fft_p.js of pil2-stark-js: same exports, same signatures
// (interpolate/fft/ifft(buffSrc, nPols, nBits, buffDst[, nBitsExt]) -> Promise<void>, dst caller-allocated,
// fft_p.js:178-302), computed by libpil2gl on the MI355X instead of workerpool threads.
"use strict";
const { addon, isFlat, isDev, staged } = require("./native.js");

async function interpolate(buffSrc, nPols, nBits, buffDst, nBitsExt) {
    const nIn = nPols * 2 ** nBits, nOut = nPols * 2 ** nBitsExt;
    if (isDev(buffSrc) && isDev(buffDst)) addon.interpolateDev(buffSrc.ptr, nPols, nBits, buffDst.ptr, nBitsExt);    // resident: no staging
    else if (isFlat(buffSrc) && isFlat(buffDst)) addon.interpolate(buffSrc, nPols, nBits, buffDst, nBitsExt);
    else staged(buffSrc, nIn, buffDst, nOut, (dIn, dOut) => addon.interpolateDev(dIn, nPols, nBits, dOut, nBitsExt));
}
async function fft(buffSrc, nPols, nBits, buffDst) {
    const n = nPols * 2 ** nBits;
    if (isDev(buffSrc) && isDev(buffDst)) addon.fftDev(buffSrc.ptr, nPols, nBits, buffDst.ptr);
    else if (isFlat(buffSrc) && isFlat(buffDst)) addon.fft(buffSrc, nPols, nBits, buffDst);
    else staged(buffSrc, n, buffDst, n, (dIn, dOut) => addon.fftDev(dIn, nPols, nBits, dOut));
}
async function ifft(buffSrc, nPols, nBits, buffDst) {
    const n = nPols * 2 ** nBits;
    if (isDev(buffSrc) && isDev(buffDst)) addon.ifftDev(buffSrc.ptr, nPols, nBits, buffDst.ptr);
    else if (isFlat(buffSrc) && isFlat(buffDst)) addon.ifft(buffSrc, nPols, nBits, buffDst);
    else staged(buffSrc, n, buffDst, n, (dIn, dOut) => addon.ifftDev(dIn, nPols, nBits, dOut));
}

module.exports.fft = fft;
module.exports.ifft = ifft;
module.exports.interpolate = interpolate;
