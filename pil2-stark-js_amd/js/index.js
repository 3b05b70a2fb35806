// Entry point: the MI355X implementations of the reference's L2 operator modules, under their own names.
"use strict";
module.exports = {
    native: require("./native.js").addon,
    DevBuffer: require("./native.js").DevBuffer,
    fft_p: require("./fft_p.js"),
    fft_worker: require("./fft_worker.js"),
    buildMerkleHash: require("./merklehash_p.js"),
    buildPoseidon: require("./poseidon.js"),
    buildMerkleHashBN128: require("./merklehash_bn128_p.js"),
    TranscriptBN128: require("./transcript_bn128.js"),
    FRI: require("./fri.js"),
    prover_helpers: require("./prover_helpers.js"),
    stark_gen_helpers: require("./stark_gen_helpers.js"),
    polutils: require("./polutils.js"),
    Transcript: require("./transcript.js"),
    starkVerify: require("./stark_verify.js"),
};
