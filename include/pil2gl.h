/*
 * pil2gl.h -- C ABI of the MI355X (gfx950) STARK proving hot path for pil2-stark-js.
 *
 * One shared library (pil2-stark-js_amd/lib/libpil2gl.so), plain pointers and
 * sizes, no C++/torch types.  Each entry point replaces one JavaScript-level
 * operator of the reference (cited per function, paths relative to the
 * reference checkout); the Node.js addon (pil2-stark-js_amd/addon) and the
 * Python ctypes binding (pil2-stark-js_amd/python/pil2gl) are thin wrappers
 * over exactly these symbols.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *  - Field elements are canonical little-endian u64 in [0,p), p = 2^64-2^32+1.
 *    Outputs are always canonical.  (The reference's WASM kernel may leave a
 *    digest word in [p,2^64), equal mod p -- src/helpers/glwasm.js:196-210;
 *    its JS twin src/helpers/hash/poseidon/poseidon.js is canonical, and so is this library.)
 *  - Matrices are row-major: element (row r, col c) at r*nCols + c
 *    (src/witness/witnessCalculator.js:113-141, src/helpers/fft/fft_worker.js:53-56).
 *  - Extension-field elements are 3 consecutive u64 (src/helpers/f3g.js:5-9).
 *  - Functions without suffix take HOST pointers (copy in, compute on the GPU,
 *    copy out): drop-in for the reference's BigUint64Array/BigBuffer calls.
 *    Functions with suffix _dev take DEVICE pointers (hipMalloc / pil2gl_dev_alloc /
 *    torch tensor.data_ptr()) plus a hipStream_t passed as void* (NULL = default
 *    stream), so buffers can stay resident in HBM across the prover's steps.
 *    Most of them only ENQUEUE work on that stream and return:
 *      interpolate / interpolate_cosets[_ws] / extend_cosets_unshifted / extend_coefs_brev[_cosets] / fft / ifft, linear_hash_rows, merkelize,
 *      merkelize_level, merkelize_digests, poseidon, fri_fold, fri_verify_fold, fri_transpose, build_x, geometric,
 *      x_div_x_sub_xi[_cosets], gprod, gsum, dev_zero, and their bn128_ twins.
 *    The following _dev calls BLOCK until their work on the stream has finished, because they hand a result to the host or
 *    stage host-side tables in a scratch slot the next call reuses:
 *      eval_program (op-list and scalar pool are host temporaries), rows_dot_ext / rows_dot_ext_multi / cols_dot_ext /
 *      cols_dot_ext_multi / fri_combine / fri_combine_order (host-side weights), compute_evals (returns the evaluations),
 *      build_zhinv, build_one_row_zerofier_inv, build_frame_zerofier, compute_q_split[_brev], build_lev (small host tables),
 *      h1h2, synth_fibonacci, group_proof / group_proofs and bn128_group_proof (openings copied to host memory).
 *    A whole config-3 proof keeps the GPU busy 99.3 % of its wall time with these.
 *  - Every function returns 0 on success, a negative PIL2GL_E* code otherwise;
 *    pil2gl_last_error() describes the failure (the reference throws Error /
 *    rejects the Promise; the addon converts the code back into a JS exception).
 *  - The library has no CPU fallback: without a HIP device every compute entry
 *    point fails with PIL2GL_ENODEV.
 *  - Calls are not thread-safe against each other (the reference issues them
 *    sequentially from one JS thread: src/prover/prover.js, `await` on every step).
 */
#pragma once
#include <stdint.h>
#include "pil2gl_expr.h"
#ifdef __cplusplus
extern "C" {
#endif

#define PIL2GL_OK        0
#define PIL2GL_EINVAL   -1      /* bad argument (the reference would throw / assert) */
#define PIL2GL_ENODEV   -2      /* no usable HIP device */
#define PIL2GL_ENOMEM   -3      /* device or host allocation failed */
#define PIL2GL_EHIP     -4      /* a HIP runtime call failed */

/* ---- lifecycle ---------------------------------------------------------- */
int         pil2gl_init(int device);            /* select device, upload tables; idempotent */
void        pil2gl_shutdown(void);
const char *pil2gl_last_error(void);
int         pil2gl_version(void);
int         pil2gl_device_info(char *name, uint32_t nameLen, uint32_t *numCUs, uint64_t *totalMem);

/* ---- device buffers (for hosts without their own HIP allocator, e.g. Node) ---- */
int pil2gl_dev_alloc(uint64_t nWords, uint64_t **out);
int pil2gl_dev_free(uint64_t *p);
int pil2gl_dev_zero(uint64_t *p, uint64_t nWords, void *stream);
int pil2gl_dev_upload(uint64_t *dst, const uint64_t *hostSrc, uint64_t nWords);
int pil2gl_dev_download(uint64_t *hostDst, const uint64_t *src, uint64_t nWords);
int pil2gl_sync(void *stream);

/* ---- NTT / LDE: src/helpers/fft/fft_p.js -------------------------------- */
/* interpolate(buffSrc,nPols,nBits,buffDst,nBitsExt)  fft_p.js:187-297:
 * per column, coefficients = iNTT_N(col), c_k *= 7^k, zero-pad to 2^nBitsExt, NTT ->
 * evaluations on the coset 7*<w_E> in natural order (== extendPol, polutils.js:18-30). */
int pil2gl_interpolate(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt);
int pil2gl_interpolate_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt, void *stream);
/* The same extension restricted to cosets j in [cosetBegin, cosetBegin+cosetCount) of the 2^(nBitsExt-nBits): dst is
 * 2^nBits x (cosetCount*nPols), element (pos, j - cosetBegin, c) = interpolate()'s row (pos << b) + j, column c.
 * One slice per GPU is the multi-GPU partition of extendAndMerkelize; the full range equals interpolate. */
int pil2gl_interpolate_cosets_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                  uint32_t cosetBegin, uint32_t cosetCount, void *stream);
/* The slice [cosetBegin, cosetBegin+cosetCount) of the PLAIN extension of the columns: src holds their evaluations on the
 * size-2^nBits subgroup, dst row (pos, j - cosetBegin) = row (pos << b) + j of fft(nBitsExt) applied to their zero-padded
 * coefficients (no coset shift).  This is how a rank extends its part of the split quotient (stark_gen_helpers.js:192). */
int pil2gl_extend_cosets_unshifted_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                       uint32_t cosetBegin, uint32_t cosetCount, void *stream);
/* The whole plain extension from COEFFICIENTS: coefBrev is 2^nBits x nPols with coefficient m of every column at row bitrev(m)
 * (pil2gl_compute_q_split_brev_dev writes that order); dst = fft(nBitsExt) of the zero-padded coefficient matrix, natural order
 * (stark_gen_helpers.js:192) -- without the padded 2^nBitsExt-row input and its first nBitsExt - nBits stages. */
int pil2gl_extend_coefs_brev_dev(const uint64_t *coefBrev, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt, void *stream);
/* The slice [cosetBegin, cosetBegin+cosetCount) of that extension, local-slice order (row (pos, j - cosetBegin) = row (pos << b) + j): how a
 * rank of a coset-split proof extends its part of the split quotient from the coefficients every rank holds (stark_gen_helpers.js:179-192),
 * without first turning them into evaluations and back. */
int pil2gl_extend_coefs_brev_cosets_dev(const uint64_t *coefBrev, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                        uint32_t cosetBegin, uint32_t cosetCount, void *stream);
/* Same with a caller-provided workspace of 2^nBits x nPols words for the coefficient matrix instead of the library's own
 * scratch; workspace == src is allowed (src is then overwritten): at config 5 a rank holds the 107 GB trace and its
 * 107 GB coset slice and nothing else. */
int pil2gl_interpolate_cosets_ws_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, uint32_t nBitsExt,
                                     uint32_t cosetBegin, uint32_t cosetCount, uint64_t *workspace, void *stream);
/* fft / ifft (buffSrc,nPols,nBits,buffDst)  fft_p.js:178-184: in-order multi-column NTT / iNTT,
 * root F.w[nBits]; ifft = fft, index j -> (n-j) mod n, times 1/n (fft/fft.js:165-174). src may equal dst. */
int pil2gl_fft(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst);
int pil2gl_ifft(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst);
int pil2gl_fft_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, void *stream);
int pil2gl_ifft_dev(const uint64_t *src, uint64_t nPols, uint32_t nBits, uint64_t *dst, void *stream);
/* The reference's worker-level operators, for a caller that keeps fft_p.js's own block loop (pool.exec sites fft_p.js:93,162,166);
 * the calls above replace that whole loop and are the product path.
 * fft_block(buff,start_pos,nPols,nBits,s,blockBits,layers)  fft_worker.js:21-67: buf = the 2^blockBits x nPols block standing at
 * row start_pos of a 2^nBits-row transform; `layers` butterfly stages ending at stage s, in place.  layers <= blockBits.
 * interpolatePrepareBlock(buff,width,start,inc,st_i,st_n)  fft_worker.js:6-19: row i of the height x width block times start*inc^i. */
int pil2gl_fft_block_dev(uint64_t *buf, uint64_t start_pos, uint64_t nPols, uint32_t nBits, uint32_t s, uint32_t blockBits, uint32_t layers, void *stream);
int pil2gl_interpolate_prepare_block_dev(uint64_t *buf, uint64_t width, uint64_t height, uint64_t start, uint64_t inc, void *stream);

/* ---- Poseidon / linear hash / Merkle tree -------------------------------- */
/* WASM export poseidon(pIn,nIn,pCap,nCap,pOut,nOut)  src/helpers/glwasm.js:216-426;
 * JS twin poseidon(inputs[8],capacity[4],nOuts)  hash/poseidon/poseidon.js:57.
 * `count` independent permutations: in = count x 8, cap = count x 4 (NULL = zeros), out = count x nOut. */
int pil2gl_poseidon(const uint64_t *in, const uint64_t *cap, uint64_t count, uint32_t nOut, uint64_t *out);
int pil2gl_poseidon_dev(const uint64_t *in, const uint64_t *cap, uint64_t count, uint32_t nOut, uint64_t *out, void *stream);
/* Transcript.put of a list (transcript.js:49-66): nBlocks full blocks of 8 elements absorbed one after the other, the
 * first with capacity hostCap, each next one with the previous output's first four words; hostOut12 = the last
 * permutation's twelve outputs.  One launch for the whole chain (twelve lanes share each permutation). */
int pil2gl_sponge_absorb(const uint64_t *hostBlocks, uint64_t nBlocks, const uint64_t hostCap[4], uint64_t hostOut12[12]);
/* worker linearHash(buffIn,width,st_i,st_n,splitLinearHash)  merklehash_worker.js:37-82
 * (= WASM multiLinearHash glwasm.js:1124-1218 / multiLinearHashGPU :1089-1122, with the
 * width<=4 raw copy of merklehash_worker.js:42-49): out = height x 4 digests. */
int pil2gl_linear_hash_rows(const uint64_t *in, uint64_t width, uint64_t height, int split, uint64_t *out);
int pil2gl_linear_hash_rows_dev(const uint64_t *in, uint64_t width, uint64_t height, int split, uint64_t *out, void *stream);
/* WASM merkelizeLevel(pIn,nOps,pOut)  glwasm.js:1220-1254: out[i] = Poseidon(in[8i..8i+7], cap=0)[0..3] */
int pil2gl_merkelize_level(const uint64_t *in, uint64_t nOps, uint64_t *out);
int pil2gl_merkelize_level_dev(const uint64_t *in, uint64_t nOps, uint64_t *out, void *stream);
/* MerkleHash._getNNodes(height*4)  merklehash_p.js:28-42: u64 words of tree.nodes */
uint64_t pil2gl_merkle_num_nodes(uint64_t height);
/* MerkleHash.merkelize(buff,width,height)  merklehash_p.js:44-133: nodes[0..4*height) = leaf digests,
 * then each level padded to an even node count with zero digests; root = last 4 words (:224-226). */
int pil2gl_merkelize(const uint64_t *elems, uint64_t width, uint64_t height, int split, uint64_t *nodes);
int pil2gl_merkelize_dev(const uint64_t *elems, uint64_t width, uint64_t height, int split, uint64_t *nodes, void *stream);
/* The upper part of merkelize (merklehash_p.js:87-103) alone: nodes[0..4*height) already holds the leaf digests (e.g.
 * gathered from the GPUs that hashed their own cosets); fills every higher level up to the root, zero padding included. */
int pil2gl_merkelize_digests_dev(uint64_t *nodes, uint64_t height, void *stream);
/* MerkleHash.getGroupProof(tree,idx)  merklehash_p.js:142-168: copies row idx (width words) to hostVals
 * and the sibling digest of every level (nLevels x 4 words) to hostSiblings; returns nLevels in *nLevels.
 * elems/nodes are DEVICE pointers (the tree stays in HBM); synchronises. */
int pil2gl_group_proof_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height,
                           uint64_t idx, uint64_t *hostVals, uint64_t *hostSiblings, uint32_t *nLevels);

/* the same for nIdx rows at once (fri.js:83-105 opens every tree at every query): hostOut receives, per index,
 * `width` row words followed by nLevels x 4 sibling words; one gather kernel, one device-to-host copy. */
int pil2gl_group_proofs_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height,
                            const uint64_t *hostIdxs, uint32_t nIdx, uint64_t *hostOut, uint32_t *nLevels);
/* verifier side: MerkleHash.calculateRootFromGroupProof  merklehash_p.js:169-203 for a batch of openings in
 * the packed layout pil2gl_group_proofs_dev writes (per opening: width values, then `levels` x 4 sibling words):
 * hostRoots[q] = the root the path of leaf hostIdxs[q] leads to; verifyGroupProof :212-215 is the comparison with the root. */
int pil2gl_roots_from_group_proofs(const uint64_t *hostProofs, uint64_t width, uint32_t levels, const uint64_t *hostIdxs, uint32_t nIdx,
                                   int splitLinearHash, uint64_t *hostRoots /* nIdx x 4 */);

/* ---- FRI: src/stark/fri.js ------------------------------------------------ */
/* FRI.fold(step>0, pol, challenge)  fri.js:22-61: pol has 2^polBits extension elements, out 2^outBits;
 * shiftInv = (1/7)^(2^(steps[0].nBits - polBits)) (fri.js:31-36, computed by the caller). */
int pil2gl_fri_fold(const uint64_t *pol, uint32_t polBits, uint32_t outBits, uint64_t shiftInv,
                    const uint64_t challenge[3], uint64_t *out);
int pil2gl_fri_fold_dev(const uint64_t *pol, uint32_t polBits, uint32_t outBits, uint64_t shiftInv,
                        const uint64_t challenge[3], uint64_t *out, void *stream);
/* FRI.verify, the per-query step of one layer for all queries at once  fri.js:121-127:
 * groups = 2^foldBits x (nQueries*3) row-major (row i = element i of every query's opened group),
 * sinv[q] = 1 / (shift * w_polBits^idx_q); out[q] = evalPol(ifft(group_q), challenge * sinv[q]). */
int pil2gl_fri_verify_fold(const uint64_t *groups, uint32_t foldBits, uint32_t nQueries, const uint64_t *sinv,
                           const uint64_t challenge[3], uint64_t *out /* nQueries x 3 */);
int pil2gl_fri_verify_fold_dev(const uint64_t *groups, uint32_t foldBits, uint32_t nQueries, const uint64_t *sinv,
                               const uint64_t challenge[3], uint64_t *out, void *stream);
/* getTransposedBuffer(pol, trasposeBits)  fri.js:187-202 */
int pil2gl_fri_transpose(const uint64_t *pol, uint32_t polBits, uint32_t transposeBits, uint64_t *out);
int pil2gl_fri_transpose_dev(const uint64_t *pol, uint32_t polBits, uint32_t transposeBits, uint64_t *out, void *stream);

/* ---- STARK step helpers: src/stark/stark_gen_helpers.js, src/helpers/polutils.js ---- */
/* x_n / x_ext tables  stark_gen_helpers.js:111-116,139-144: x[i] = shift * w[nBits]^i */
int pil2gl_build_x_dev(uint32_t nBits, uint64_t shift, uint64_t *x, void *stream);
/* out[i] = first * ratio^i, i < n: one coset's rows of x_ext (first = 7 w_E^j, ratio = w_N; stark_gen_helpers.js:139-144) or any other
 * table of powers, without building the whole 2^nBitsExt-row table (a rank of a coset-sharded proof holds its own cosets only) */
int pil2gl_geometric_dev(uint64_t first, uint64_t ratio, uint64_t n, uint64_t *out, void *stream);
/* buildZhInv(stark=true)  polutils.js:39-55 */
int pil2gl_build_zhinv_dev(uint32_t nBits, uint32_t nBitsExt, uint64_t *out, void *stream);
/* buildOneRowZerofierInv(stark=true)  polutils.js:57-71 */
int pil2gl_build_one_row_zerofier_inv_dev(uint32_t nBits, uint32_t nBitsExt, uint64_t rowIndex, uint64_t *out, void *stream);
/* buildFrameZerofierInv(stark=true)  polutils.js:74-102 (product of (x-root), not inverted) */
int pil2gl_build_frame_zerofier_dev(uint32_t nBits, uint32_t nBitsExt, uint64_t offsetMin, uint64_t offsetMax, uint64_t *out, void *stream);
/* computeQStark split/scale  stark_gen_helpers.js:179-190: qq2[i][p*qDim+k] = qq1[p*N+i][k] * (7^-N)^p, rows >= N zero */
int pil2gl_compute_q_split_dev(const uint64_t *qq1, uint32_t nBits, uint32_t nBitsExt, uint32_t qDim, uint32_t qDeg, uint64_t *qq2, void *stream);
/* the same pieces as the 2^nBits-row coefficient matrix alone, row bitrev(i) = coefficient i: coefBrev[bitrev(i)][p*qDim+k] = qq1[p*N+i][k] * (7^-N)^p */
int pil2gl_compute_q_split_brev_dev(const uint64_t *qq1, uint32_t nBits, uint32_t nBitsExt, uint32_t qDim, uint32_t qDeg, uint64_t *coefBrev, void *stream);
/* computeFRIStark xDivXSubXi  stark_gen_helpers.js:293-322: out[3*(k*nOpen+iOpen)+c] = (x_k / (x_k - xi))_c */
int pil2gl_x_div_x_sub_xi_dev(uint32_t nBitsExt, const uint64_t xi[3], uint64_t nOpen, uint64_t iOpen, uint64_t *out, void *stream);
/* the rows of cosets [cosetBegin, cosetBegin + cosetCount) of the 2^extBits only (one rank's slice of a coset-sharded proof), in
 * slice order: row pos * cosetCount + jl stands for extended row (pos << extBits) + cosetBegin + jl; cosetCount a power of two. */
int pil2gl_x_div_x_sub_xi_cosets_dev(uint32_t nBitsExt, uint32_t extBits, const uint64_t xi[3], uint64_t nOpen, uint64_t iOpen,
                                     uint32_t cosetBegin, uint32_t cosetCount, uint64_t *out, void *stream);
/* computeEvalsStark  stark_gen_helpers.js:216-264: lev = ifft_N(xi^k) (extension, N x 3);
 * evals[e] = sum_k v_e[k << extendBits] * lev[k] for nEvals columns described by (buffer, width, offset, dim). */
int pil2gl_build_lev_dev(uint32_t nBits, const uint64_t xi[3], uint64_t *lev, void *stream);
typedef struct { const uint64_t *buf; uint64_t width; uint64_t offset; uint32_t dim; uint32_t levIndex; } pil2gl_eval_desc;
int pil2gl_compute_evals_dev(const pil2gl_eval_desc *descs, uint32_t nEvals, uint32_t nBits, uint32_t extendBits,
                             const uint64_t *const *levs, uint32_t nLevs, uint64_t *hostEvals /* nEvals x 3 */, void *stream);

/* ---- extension-weighted sums: the two steps that are matrix-vector products (csrc/dot.hip) ----
 * acc[r][o] (+)= sum_c buf[r][c] * coef[o][c]   (coef: host array nOut x width x 3, one extension constant per base
 * column; acc: device nRows x nOut x 3).  With coef = powers of vf2 this is the inner sum of the FRI polynomial
 * (friPolinomial.js:26-36) for every opening at once. */
int pil2gl_rows_dot_ext_dev(const uint64_t *buf, uint64_t width, uint64_t nRows, const uint64_t *hostCoef, uint32_t nOut,
                            uint64_t *acc, int accumulate, void *stream);
/* the same over nBufs matrices with the same rows -- the stage matrices and the constants the FRI polynomial walks
 * (friPolinomial.js:26-50): acc[r][o] (+)= sum_k sum_c bufs[k][r][c] * hostCoefs[k][o][c].  On the matrix cores: one pass over
 * all of them when their columns fit the kernel's staged row side by side (<= 112 columns in all, an odd width counted as the
 * next even one, <= 4 matrices); wider inputs in column windows packed into accumulating passes; three or four outputs as
 * two sweeps of two; matrices left with under 32 columns, and PIL2GL_ROWS_DOT_MFMA=0, on the vector kernels.  nOut: 1..4. */
int pil2gl_rows_dot_ext_multi_dev(const uint64_t *const *bufs, const uint64_t *widths, uint32_t nBufs, uint64_t nRows,
                                  const uint64_t *const *hostCoefs, uint32_t nOut, uint64_t *acc, int accumulate, void *stream);
/* f[r] = Horner in vf1 over the openings of (acc[r][o] - K_o) * xDivXSubXi[r][o]   (friPolinomial.js:38-50);
 * hostK: nOpen x 3 (K_o = sum_j ev_j vf2^(n_o - j)). */
int pil2gl_fri_combine_dev(const uint64_t *acc, const uint64_t *hostK, const uint64_t vf1[3], const uint64_t *xDivXSubXi,
                           uint32_t nOpen, uint64_t nRows, uint64_t *f, void *stream);
/* the same with the Horner order given: the k-th term is opening order[k] (host, a permutation of 0..nOpen-1; acc, hostK and
 * xDivXSubXi stay indexed by the opening's position in openingPoints).  The reference generates the terms in the order of
 * Object.keys(friExps) (friPolinomial.js:42-50): non-negative openings ascending, then negative ones as they first appear in
 * evMap -- with a previous-row opening ([-1, 0, 1]) that is 0, 1, -1, not the order of openingPoints. */
int pil2gl_fri_combine_order_dev(const uint64_t *acc, const uint64_t *hostK, const uint64_t vf1[3], const uint64_t *xDivXSubXi,
                                 uint32_t nOpen, const uint32_t *order, uint64_t nRows, uint64_t *f, void *stream);
/* hostOut[l][c] = sum_k buf[k*rowStep][c] * levs[l][k]   (stark_gen_helpers.js:250-264 for every column of a buffer
 * and every opening at once; hostOut: nLev x width x 3, levs[l]: device nRows x 3). */
int pil2gl_cols_dot_ext_dev(const uint64_t *buf, uint64_t width, uint64_t nRows, uint64_t rowStep, const uint64_t *const *levs,
                            uint32_t nLev, uint64_t *hostOut, void *stream);
/* the same over nBufs (<= 8) matrices with the same rows in one sweep of the weights (computeEvalsStark walks every committed
 * stage and the constants, stark_gen_helpers.js:233-264): hostOuts[k] receives nLev x widths[k] x 3.  nLev: 1..64 (a sweep
 * weighs four opening points; more of them take more sweeps inside the call). */
int pil2gl_cols_dot_ext_multi_dev(const uint64_t *const *bufs, const uint64_t *widths, uint32_t nBufs, uint64_t nRows, uint64_t rowStep,
                                  const uint64_t *const *levs, uint32_t nLev, uint64_t *const *hostOuts, void *stream);
/* the same over the columns [colBegin[k], colBegin[k] + widths[k]) of matrices whose rows are strides[k] words long (colBegin null: from
 * column 0): a rank of a coset-sharded proof evaluates its share of the columns, and only those cells are read. */
int pil2gl_cols_dot_ext_range_dev(const uint64_t *const *bufs, const uint64_t *strides, const uint64_t *colBegin, const uint64_t *widths, uint32_t nBufs,
                                  uint64_t nRows, uint64_t rowStep, const uint64_t *const *levs, uint32_t nLev, uint64_t *const *hostOuts, void *stream);

/* ---- expression evaluator: src/prover/prover_helpers.js:23-259 ------------- */
/* callCalculateExps / calculateExps: run the op-list on every row of the domain.  Section pointers in
 * ctx are DEVICE pointers; prog/ctx structs themselves are host memory (copied at launch). */
int pil2gl_eval_program_dev(const glx_program *prog, const glx_ctx *ctx, void *stream);
/* calculateExps with debug = true (prover_helpers.js:46-70: a constraint evaluated on the rows [first, last) of its boundary, stopping
 * at the first row whose value is not zero): after the constraint's program has written its value to a column of `dim` (1 or 3) words
 * per row, *hostRow = the smallest such row (UINT64_MAX if the constraint holds on the whole range) and hostVal[0..dim) its value.
 * Blocks until the answer is on the host. */
int pil2gl_first_nonzero_row_dev(const uint64_t *col, uint32_t dim, uint64_t first, uint64_t last, uint64_t *hostRow, uint64_t *hostVal, void *stream);

/* ---- stage-2 witness hints (hints_helpers.js:91-114) ------------------------------------------------------------
 * calculateZ(F,num,den)  polutils.js:128-143: out[0] = 1, out[i] = out[i-1]*num[i-1]/den[i-1]   (num, den: n rows)
 * calculateS(F,num,den)  polutils.js:145-164: out[i] = out[i-1] + num/den[i]                    (num: ONE element)
 * dimNum/dimDen in {1,3} (base or cubic-extension columns, row-major); out has dimension 3 if either has, else 1.
 * A zero denominator yields 0 for that ratio (the reference's batchInverse would poison the whole column). */
int pil2gl_gprod_dev(const uint64_t *num, uint32_t dimNum, const uint64_t *den, uint32_t dimDen, uint64_t n, uint64_t *out, void *stream);
int pil2gl_gsum_dev(const uint64_t *num, uint32_t dimNum, const uint64_t *den, uint32_t dimDen, uint64_t n, uint64_t *out, void *stream);
/* calculateH1H2(F,f,t)  polutils.js:105-126: the multiset f (every value must occur in t) merged into t; h1[i], h2[i] =
 * entries 2i, 2i+1 of the merged sequence.  f, t, h1, h2: n rows of dimension dim (1 or 3).  Returns PIL2GL_EINVAL with
 * the reference's "Number not included" message when some f[j] is missing from t (synchronises the stream). */
int pil2gl_h1h2_dev(const uint64_t *f, const uint64_t *t, uint64_t n, uint32_t dim, uint64_t *h1, uint64_t *h2, void *stream);

/* ---- BN128 (BN254 scalar field) Merkle commitment: merklehash_bn128_p.js, merklehash_bn128_worker.js -------------
 * Field elements are 4 little-endian u64 words.  tree.nodes and leaf digests are in MONTGOMERY form (R = 2^256), exactly
 * what the reference's WASM leaves in memory (frm_toMontgomery, merklehash_bn128_worker.js:49,67,82), so files written
 * by writeToFile are interchangeable; poseidon / group proofs / roots cross the boundary in normal form like the JS
 * objects do (F.toObject, merklehash_bn128_p.js:167,241).  arity in {2,4,8,16}; Poseidon parameters for every
 * t = 2..17 are generated on first use (csrc/bn128.hip). */
/* circomlibjs poseidon(inputs[nIn], initState, nOut) -> out[nOut] for `count` independent calls (init may be NULL = 0) */
int pil2gl_bn128_poseidon(const uint64_t *in, const uint64_t *init, uint64_t count, uint32_t nIn, uint32_t nOut, uint64_t *out);
/* TranscriptBN128.put of a list (transcript.bn128.js:56-83): nBlocks full blocks of nIn elements (normal form, 4 words each)
 * absorbed one after the other, state element 0 = hostInit, then each permutation's output 0; hostOut = the nIn+1 outputs
 * of the last permutation.  One launch for the chain, 3*(nIn+1) lanes sharing each permutation. */
int pil2gl_bn128_sponge_absorb(const uint64_t *hostBlocks, uint64_t nBlocks, uint32_t nIn, const uint64_t hostInit[4], uint64_t *hostOut);
int pil2gl_bn128_poseidon_dev(const uint64_t *in, const uint64_t *init, uint64_t count, uint32_t nIn, uint32_t nOut, uint64_t *out, void *stream);
/* worker linearHash(buffIn,width,st_i,st_n,arity,custom)  merklehash_bn128_worker.js:13-100 -> height x 4 words */
int pil2gl_bn128_linear_hash_rows(const uint64_t *in, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *out);
int pil2gl_bn128_linear_hash_rows_dev(const uint64_t *in, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *out, void *stream);
/* worker merkelizeLevel(buffIn,st_i,st_n,arity)  merklehash_bn128_worker.js:104-144: arity*4 words in -> 4 words out per op */
int pil2gl_bn128_merkelize_level_dev(const uint64_t *in, uint64_t nOps, uint32_t arity, uint64_t *out, void *stream);
/* MerkleHash._getNNodes(height)  merklehash_bn128_p.js:31-45 (in nodes; tree.nodes has 4x as many u64 words) */
uint64_t pil2gl_bn128_merkle_num_nodes(uint64_t height, uint32_t arity);
/* MerkleHash.merkelize(buff,width,height)  merklehash_bn128_p.js:47-129 */
int pil2gl_bn128_merkelize(const uint64_t *elems, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *nodes);
int pil2gl_bn128_merkelize_dev(const uint64_t *elems, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *nodes, void *stream);
/* MerkleHash.getGroupProof(tree,idx)  merklehash_bn128_p.js:142-182: hostVals[width], hostSiblings[nLevels][arity][4] (normal form) */
int pil2gl_bn128_group_proof_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height, uint32_t arity,
                                 uint64_t idx, uint64_t *hostVals, uint64_t *hostSiblings, uint32_t *nLevels);
/* The same for a batch of rows in one launch (fri.js:83-105 opens every tree at every query): hostVals nIdx x width values, hostSiblings
 * nIdx x levels x arity field elements (4 words each, normal form), *nLevels = levels. */
int pil2gl_bn128_group_proofs_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height, uint32_t arity,
                                  const uint64_t *hostIdxs, uint32_t nIdx, uint64_t *hostVals, uint64_t *hostSiblings, uint32_t *nLevels);
/* n elements between normal and Montgomery form (F.e / F.toObject); the host form is plain host arithmetic */
int pil2gl_bn128_convert(const uint64_t *in, uint64_t n, int toMontgomery, uint64_t *out);
int pil2gl_bn128_convert_dev(const uint64_t *in, uint64_t n, int toMontgomery, uint64_t *out, void *stream);

/* ---- synthetic workload for bench.py / tests (not a reference operator) ----
 * witness of nPairs independent Fibonacci machines (test/state_machines/sm_fibonacci/sm_fibonacci.js:12-23):
 * cm is 2^nBits x (2*nPairs) row-major (l1_k, l2_k), hostInit = 2*nPairs canonical start values (host pointer). */
int pil2gl_synth_fibonacci_dev(uint32_t nBits, uint32_t nPairs, const uint64_t *hostInit, uint64_t *cm, void *stream);

/* ---- the WASM module's scalar exports (glwasm.js:47-96,1269-1275: add, mul, square of field elements) ------------
 * Host arithmetic, one element per call (the reference calls them from JS for twiddles and shifts); no device needed. */
uint64_t pil2gl_add(uint64_t a, uint64_t b);
uint64_t pil2gl_mul(uint64_t a, uint64_t b);
uint64_t pil2gl_square(uint64_t a);

/* ---- diagnostics used by the parity tests ---------------------------------- */
/* element-wise a*b, a+b, a-b on the device (n elements, host pointers) */
int pil2gl_selftest_field(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *mul, uint64_t *add, uint64_t *sub);
/* host-only: the op-list after validation, value numbering (each distinct cell loaded once, common sub-expressions
 * merged) and live-range renumbering of temporaries, as the kernel runs it (without Horner-chain fusion).
 * outOps must hold 2*nOps+16 entries; outInfo[0] = temporaries needed, outInfo[1] = ops written. */
int pil2gl_debug_compact_program(const glx_program *prog, glx_op *outOps, uint32_t *outInfo);
/* host-only: run the optimiser, generate the straight-line kernel source of the program and compile it with hiprtc
 * (the path long programs take at run time); reports the code object size and the number of fused Horner terms. */
int pil2gl_debug_jit_compile(const glx_program *prog, const glx_ctx *ctx, uint64_t *codeBytes, uint32_t *fusedOps);
/* the two hand-written Goldilocks products on n pairs of arbitrary u64 operands (host pointers): x[i] = a*b by the exact form
 * the transform kernels use (gl_field.cuh mul_lazy_x), pb[i] = a*b by the flagged form of the S-boxes (mul_lazy_b), both canonical;
 * flag[i] != 0 where the flagged form asks to be recomputed (its last subtraction borrowed: probability ~2^-32 on random operands) */
int pil2gl_selftest_products(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *x, uint64_t *pb, uint64_t *flag);
/* `layers` consecutive Poseidon MDS layers (glwasm.js:428-440 matrix) applied to n 12-element states (host pointers,
 * any u64 representatives in, canonical out): mfma = 1 the matrix-core layer the hash kernels use, 0 the vector-ALU one */
int pil2gl_selftest_mds(const uint64_t *states, uint64_t n, uint32_t layers, int mfma, uint64_t *out);
/* the Poseidon-12 permutation (glwasm.js:216-426) of n 12-element states (host pointers, any u64 representatives in, canonical
 * out) in each of the library's statements of it: what = 0 the hash kernels' form (matrix-core MDS, rounds 4..25 four to a
 * linear layer), 1 matrix-core MDS with one layer per round, 2 vector ALU only; 3 / 4 = rounds 4..25 ALONE (folded-constant
 * form: lane 0 + c, x^7, MDS) in the forms of 0 / 1 -- arbitrary states reach the partial rounds' recombination that way */
int pil2gl_selftest_poseidon(const uint64_t *states, uint64_t n, int what, uint64_t *out);
/* diagnostics: the shader clock (MHz) under the Poseidon kernels' own load -- every workgroup of a chip-filling launch runs `iters`
   permutations between two readings of the shader-clock and the 100 MHz counters; mhz[3] = median, 5th, 95th percentile.
   (sysfs reports the nominal DPM level; under this load the MI355X runs near 1.9 GHz, which is what issue-cycle figures need) */
int pil2gl_selftest_clock(uint32_t iters, double *mhz);
/* extension a*b and 1/a on the device (n triples) */
int pil2gl_selftest_ext(const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t *mul, uint64_t *inv);

#ifdef __cplusplus
}
#endif
