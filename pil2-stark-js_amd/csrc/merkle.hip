// Poseidon linear hash of rows and Merkle tree construction (Goldilocks), gfx950.
//
// Replaces src/helpers/hash/merklehash/merklehash_p.js:44-133 (merkelize), its worker
// merklehash_worker.js:37-117 and the WASM exports multiLinearHash / multiLinearHashGPU /
// merkelizeLevel / poseidon of src/helpers/glwasm.js:216-426,879-1254.
// Leaf rule: src/helpers/hash/linearhash/linearhash.js:22-41; split rule: linearhash_gpu.js:30-66.
//
// One permutation per lane: the work is integer-ALU bound (about 600 64-bit modular
// multiplication equivalents per permutation against 96 bytes of traffic), so the kernels only
// need "not stupid" memory access: a row's 8-element chunk is 64 contiguous bytes per lane.
// The MDS layers run on the matrix cores (poseidon_mds_mfma.cuh), which takes the operands of all 64 lanes
// in one instruction: no lane leaves early -- out-of-range lanes hash a clamped (valid) index and skip the store.
#include "common.h"
#include <vector>
#include <string.h>
#include "poseidon_gl.cuh"
#include <algorithm>

using namespace gl;

namespace {

// sponge over `width` consecutive words (linearhash.js:29-40); width > 4
__device__ __forceinline__ void sponge(const u64 *__restrict__ v, u32 width, u64 digest[4], const MdsMfma &m) {
    u64 st[12];
    st[8] = st[9] = st[10] = st[11] = 0;
    for (u32 i = 0; i < width; i += 8) {
        const u32 n = min(8u, width - i);
#pragma unroll
        for (u32 j = 0; j < 8; j++) st[j] = j < n ? v[i + j] : 0;
        poseidon_perm<0, 4>(st, m);                 // lazy: only the digest is canonicalised, below; four outputs are read
        st[8] = st[0]; st[9] = st[1]; st[10] = st[2]; st[11] = st[3];
    }
    digest[0] = canon(st[8]); digest[1] = canon(st[9]); digest[2] = canon(st[10]); digest[3] = canon(st[11]);
}
__device__ __forceinline__ void linear_hash_plain(const u64 *__restrict__ v, u32 width, u64 digest[4], const MdsMfma &m) {
    if (width <= 4) {                               // linearhash.js:22-28, merklehash_worker.js:42-49
#pragma unroll
        for (u32 j = 0; j < 4; j++) digest[j] = j < width ? v[j] : 0;
        return;
    }
    sponge(v, width, digest, m);
}

// FOUR waves per SIMD at 124 registers and no scratch (the accumulator bias of every matrix-instruction chain is an inline constant and the
// operands of every layer come from the LDS table: poseidon_blocks.cuh); five waves spill 26 registers and are 8 % slower.
__global__ void __launch_bounds__(256, 4) linear_hash_kernel(const u64 *__restrict__ in, u64 width, u64 height, u64 *__restrict__ out) {
    const u64 row0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = row0 < height;
    const u64 row = live ? row0 : height - 1;       // width is uniform: every lane takes the same path
    const u64 *v = in + row * width;
    MdsMfma m;
    poseidon_init(m);
    u64 d[4];
    linear_hash_plain(v, (u32)width, d, m);
    if (!live) return;
    u64 *o = out + 4 * row;
    o[0] = d[0]; o[1] = d[1]; o[2] = d[2]; o[3] = d[3];
}

// The split form (linearhash_gpu.js:30-66, glwasm.js:879-1087): up to four batches of max(8, ceil(width / 4)) columns hashed on their
// own, then their digests hashed.  Nothing but the row pointer lives in registers across a permutation: a batch digest waits in LDS
// (eight words per lane) and the second sponge runs as soon as its chunk is there -- P(d0 | d1, 0), then P(d2 | d3, capacity) -- instead
// of sixteen digest words held through eighteen permutations (round 3: 128 registers + 172 bytes of scratch per lane).
// 512-thread workgroups: two per CU share 160 KB of LDS (31 KB operand table + 32 KB of digests each), four waves per SIMD as above.
constexpr u32 SPLIT_BLOCK = 512;
__global__ void __launch_bounds__(SPLIT_BLOCK, 4) linear_hash_split_kernel(const u64 *__restrict__ in, u64 width, u64 height, u64 *__restrict__ out) {
    __shared__ u64 park[8][SPLIT_BLOCK];
    const u32 tid = threadIdx.x;
    const u64 row0 = (u64)blockIdx.x * blockDim.x + tid;
    const bool live = row0 < height;
    const u64 row = live ? row0 : height - 1;
    const u64 *v = in + row * width;
    MdsMfma m;
    poseidon_init(m);
    const u32 w = (u32)width, batch = max(8u, (w + 3) / 4), nb = (w + batch - 1) / batch;   // 2 <= nb <= 4 (width > 8), uniform
    // ONE permutation call site (several inlined copies cost the register allocation of all of them: 444 bytes of scratch per lane):
    // batch b absorbs its chunks; the batch that completes a pair (odd b, or the last one) then also runs the pair's second-level
    // permutation P(d_even | d_odd, capacity of the pair before)
    u64 st[12];
    for (u32 b = 0; b < nb; b++) {
        const u32 bw = min(batch, w - b * batch);
        const u64 *vb = v + (u64)b * batch;
        const u32 nChunks = bw <= 4 ? 0 : (bw + 7) / 8;                          // a batch of <= 4 columns is its own digest (linearhash.js:22-28)
        const bool closes = (b & 1) || b == nb - 1;
        for (u32 i = 0; i < nChunks + (closes ? 1 : 0); i++) {
            if (i < nChunks) {
                const u32 n = min(8u, bw - 8 * i);
#pragma unroll
                for (u32 j = 0; j < 8; j++) st[j] = j < n ? vb[8 * i + j] : 0;
                if (i == 0) st[8] = st[9] = st[10] = st[11] = 0;
            } else {
                u64 d[4];
#pragma unroll
                for (u32 j = 0; j < 4; j++) d[j] = nChunks ? canon(st[8 + j]) : (j < bw ? vb[j] : 0);
#pragma unroll
                for (u32 j = 0; j < 4; j++) {
                    st[j] = (b & 1) ? park[j][tid] : d[j];
                    st[4 + j] = (b & 1) ? d[j] : 0;
                    st[8 + j] = b >= 2 ? park[4 + j][tid] : 0;
                }
            }
            poseidon_perm<0, 4>(st, m);
            st[8] = st[0]; st[9] = st[1]; st[10] = st[2]; st[11] = st[3];
        }
        if (!closes) {                                                          // an even batch with a partner to come: its digest waits in LDS
#pragma unroll
            for (u32 j = 0; j < 4; j++) park[j][tid] = nChunks ? canon(st[8 + j]) : (j < bw ? vb[j] : 0);
        } else if (b < nb - 1) {                                                // the first pair's output is the second pair's capacity (lazy, as sponge() keeps it)
#pragma unroll
            for (u32 j = 0; j < 4; j++) park[4 + j][tid] = st[8 + j];
        }
    }
    if (!live) return;
    u64 *o = out + 4 * row;
    o[0] = canon(st[8]); o[1] = canon(st[9]); o[2] = canon(st[10]); o[3] = canon(st[11]);
}

// glwasm.js:1220-1254: out[i] = Poseidon(in[8i..8i+7], capacity 0)[0..3]
__global__ void __launch_bounds__(256, 4) merkle_level_kernel(const u64 *__restrict__ in, u64 nOps, u64 *__restrict__ out) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < nOps;
    const u64 i = live ? i0 : nOps - 1;
    MdsMfma m;
    poseidon_init(m);
    u64 st[12];
#pragma unroll
    for (int j = 0; j < 8; j++) st[j] = in[8 * i + j];
    st[8] = st[9] = st[10] = st[11] = 0;
    poseidon_perm<4, 4>(st, m);
    if (!live) return;
    u64 *o = out + 4 * i;
    o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3];
}

// merklehash_p.js:187-203 (calculateRootFromGroupProof, after the leaf hash): one lane walks one path; level l hashes
// (cur, sibling) or (sibling, cur) by bit l of the leaf index
__global__ void __launch_bounds__(256, 2) merkle_path_roots_kernel(const u64 *__restrict__ leaf, const u64 *__restrict__ sib, const u64 *__restrict__ idx,
                                                                  u64 count, u32 levels, u64 *__restrict__ roots) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < count;
    const u64 i = live ? i0 : count - 1;
    MdsMfma m;
    poseidon_init(m);
    u64 cur[4];
#pragma unroll
    for (int j = 0; j < 4; j++) cur[j] = canon(leaf[4 * i + j]);
    u64 pos = idx[i];
    for (u32 l = 0; l < levels; l++) {
        const u64 *s = sib + (i * levels + l) * 4;
        const bool right = pos & 1;                  // this node is the right child: the sibling goes first
        u64 st[12];
#pragma unroll
        for (int j = 0; j < 4; j++) { const u64 sv = canon(s[j]); st[j] = right ? sv : cur[j]; st[4 + j] = right ? cur[j] : sv; }
        st[8] = st[9] = st[10] = st[11] = 0;
        poseidon_perm(st, m);
#pragma unroll
        for (int j = 0; j < 4; j++) cur[j] = st[j];
        pos >>= 1;
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 4; j++) roots[4 * i + j] = cur[j];
}

// A chain of dependent permutations -- the Fiat-Shamir transcript absorbing a list (transcript.js:49-66: every full block of 8
// is permuted with the previous output's first four words as capacity).  One permutation per lane would leave the chain at
// one wave-alone permutation (~50 us) per block; here twelve lanes hold one state word each: S-boxes in parallel (lane 0
// alone in the partial rounds), the MDS row of a lane from the twelve words exchanged through LDS, textbook round schedule
// (constants of round r added to every word).  out12 = the last permutation's twelve outputs.
__global__ void __launch_bounds__(64) sponge_chain_kernel(const u64 *__restrict__ blocks, u64 nBlocks, const u64 *__restrict__ cap, u64 *__restrict__ out12) {
    constexpr u32 MC[12] = { 17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20 };
    __shared__ u64 sh[12];
    const u32 lane = threadIdx.x;
    const bool act = lane < 12;
    const u32 l = act ? lane : 0;
    u32 mrow[12];                                    // row `l` of the MDS matrix: circ(MC) + 8 at (0,0)
#pragma unroll
    for (int j = 0; j < 12; j++) mrow[j] = MC[(j + 12 - l) % 12] + ((l == 0 && j == 0) ? 8u : 0u);
    u64 carry = (act && l >= 8) ? canon(cap[l - 8]) : 0;          // lanes 8..11 hold the capacity words
    u64 s = 0;
    for (u64 b = 0; b < nBlocks; b++) {
        s = l < 8 ? canon(blocks[8 * b + l]) : carry;
        for (int r = 0; r < 30; r++) {
            s = add_lazy_canon(s, POSEIDON_GL_RC[r * 12 + l]);
            const bool full = r < 4 || r >= 26;
            if (full || l == 0) s = pow7_lazy(s);
            if (act) sh[l] = s;
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the writes of this wave have landed
            __builtin_amdgcn_wave_barrier();
            u64 al = 0, ah = 0;                      // sum of low halves * m and of high halves * m (each < 2^42)
#pragma unroll
            for (int j = 0; j < 12; j++) { const u64 v = sh[j]; al += (u64)(u32)v * mrow[j]; ah += (u64)(u32)(v >> 32) * mrow[j]; }
            __builtin_amdgcn_wave_barrier();
            const u64 lo = al + (ah << 32);
            const u64 hi = (ah >> 32) + (lo < al ? 1 : 0);
            s = reduce128_lazy(lo, hi);
        }
        s = canon(s);
        if (act) sh[l] = s;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        carry = (l >= 8) ? sh[l - 8] : 0;            // next capacity = outputs 0..3
        __builtin_amdgcn_wave_barrier();
    }
    if (act) out12[l] = s;
}

// batch of independent permutations (glwasm.js:216 `poseidon`, hash/poseidon/poseidon.js:57)
// (four waves per SIMD like the leaf and tree kernels: 124 registers, no scratch, since the permutation is forced inline)
__global__ void __launch_bounds__(256, 4) poseidon_batch_kernel(const u64 *__restrict__ in, const u64 *__restrict__ cap, u64 count, u32 nOut, u64 *__restrict__ out) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < count;
    const u64 i = live ? i0 : count - 1;
    MdsMfma m;
    poseidon_init(m);
    u64 st[12];
#pragma unroll
    for (int j = 0; j < 8; j++) st[j] = canon(in[8 * i + j]);          // F.e(): poseidon.js:65-67
#pragma unroll
    for (int j = 0; j < 4; j++) st[8 + j] = cap ? canon(cap[4 * i + j]) : 0;
    poseidon_perm(st, m);
    if (!live) return;
    for (u32 j = 0; j < nOut; j++) out[(u64)nOut * i + j] = st[j];
}

// diagnostics: MDS layers alone (both implementations), whole waves (n is padded by clamping)
__global__ void __launch_bounds__(256, 2) mds_selftest_kernel(const u64 *__restrict__ in, u64 n, u32 layers, int mfma, u64 *__restrict__ out) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < n;
    const u64 i = live ? i0 : n - 1;
    MdsMfma m;
    mds_mfma_init(m);
    u64 st[12];
#pragma unroll
    for (int j = 0; j < 12; j++) st[j] = in[12 * i + j];
    for (u32 l = 0; l < layers; l++) { if (mfma) mds_layer_mfma(st, m); else mds_layer(st); }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 12; j++) out[12 * i + j] = canon(st[j]);
}

// diagnostics: the permutation in its three statements, and rounds 4..25 alone in their two
__global__ void __launch_bounds__(256, 2) poseidon_selftest_kernel(const u64 *__restrict__ in, u64 n, int what, u64 *__restrict__ out) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < n;
    const u64 i = live ? i0 : n - 1;
    MdsMfma m;
    mds_mfma_init(m);
    poseidon_init(m);
    u64 st[12];
#pragma unroll
    for (int j = 0; j < 12; j++) st[j] = in[12 * i + j];
    if (what == 0) poseidon_perm(st, m);
    else if (what == 1) poseidon_perm_single(st, m);
    else if (what == 2) poseidon_perm(st);
    else poseidon_partial_rounds(st, m, what == 3 ? 0 : 1);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 12; j++) out[12 * i + j] = canon(st[j]);
}

// diagnostics: the shader clock under the hashing kernels' own load.  Every workgroup runs `iters` permutations on all its lanes; its
// first lane reads the shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around them.
__global__ void __launch_bounds__(256, 4) clock_probe_kernel(u64 *__restrict__ sink, int iters, u64 *__restrict__ clk) {
    MdsMfma m;
    poseidon_init(m);
    u64 st[12];
    const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 12; j++) st[j] = ((id * 12 + j) * 0x9E3779B97F4A7C15ull) >> 1;
    const u64 c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) poseidon_perm<0>(st, m);
    const u64 c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    u64 x = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) x ^= st[j];
    sink[id] = x;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// gather of query openings (merklehash_p.js:142-168) for many indices at once: one block per query copies the row
// and the sibling digest of every level into a packed staging area
__global__ void group_proofs_kernel(const u64 *__restrict__ elems, const u64 *__restrict__ nodes, u64 width, u64 height,
                                    const u64 *__restrict__ idxs, u32 nLevels, u64 *__restrict__ out) {
    const u64 stride = width + 4ull * nLevels;
    u64 *o = out + (u64)blockIdx.x * stride;
    u64 idx = idxs[blockIdx.x];
    for (u64 c = threadIdx.x; c < width; c += blockDim.x) o[c] = elems[idx * width + c];
    if (threadIdx.x < 4) {
        u64 offset = 0, n = height * 4;
        for (u32 l = 0; l < nLevels; l++) {
            o[width + 4 * l + threadIdx.x] = nodes[offset + (idx ^ 1) * 4 + threadIdx.x];
            const u64 nextN = ((n - 1) / 8 + 1) * 4;
            offset += nextN * 2; n = nextN; idx >>= 1;
        }
    }
}

}  // namespace

using namespace pil2gl;

extern "C" {

uint64_t pil2gl_merkle_num_nodes(uint64_t height) {        // merklehash_p.js:28-42 with n = height*4
    if (height == 0) return 0;
    uint64_t n = height * 4;
    uint64_t nextN = ((n - 1) / 8 + 1) * 4;
    uint64_t acc = nextN * 2;
    while (n > 4) {
        n = nextN;
        nextN = ((n - 1) / 8 + 1) * 4;
        if (n > 4) acc += nextN * 2; else acc += 4;
    }
    return acc;
}

int pil2gl_linear_hash_rows_dev(const uint64_t *in, uint64_t width, uint64_t height, int split, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (height == 0) return PIL2GL_OK;
    if (!out || (!in && width)) return fail(PIL2GL_EINVAL, "null buffer");
    if (width >= (1ull << 31)) return fail(PIL2GL_EINVAL, "row width too large");
    u64 blocks = (height + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");
    if (split && width > 8) linear_hash_split_kernel<<<(unsigned)((height + SPLIT_BLOCK - 1) / SPLIT_BLOCK), SPLIT_BLOCK, 0, as_stream(stream)>>>(in, width, height, out);
    else linear_hash_kernel<<<(unsigned)blocks, 256, 0, as_stream(stream)>>>(in, width, height, out);     // one batch: the split form IS the plain one
    KERNEL_CHECK();
    return PIL2GL_OK;
}

int pil2gl_merkelize_level_dev(const uint64_t *in, uint64_t nOps, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (nOps == 0) return PIL2GL_OK;
    if (!in || !out) return fail(PIL2GL_EINVAL, "null buffer");
    if (nOps > (0x7fffffffull << 8)) return fail(PIL2GL_EINVAL, "grid too large");
    merkle_level_kernel<<<(unsigned)((nOps + 255) / 256), 256, 0, as_stream(stream)>>>(in, nOps, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

int pil2gl_poseidon_dev(const uint64_t *in, const uint64_t *cap, uint64_t count, uint32_t nOut, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (count == 0) return PIL2GL_OK;
    if (!in || !out) return fail(PIL2GL_EINVAL, "null buffer");
    if (nOut < 1 || nOut > 12) return fail(PIL2GL_EINVAL, "nOut must be 1..12");
    if (count > (0x7fffffffull << 8)) return fail(PIL2GL_EINVAL, "grid too large");
    poseidon_batch_kernel<<<(unsigned)((count + 255) / 256), 256, 0, as_stream(stream)>>>(in, cap, count, nOut, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

int pil2gl_merkelize_digests_dev(uint64_t *nodes, uint64_t height, void *stream) {
    P2_TRY(ensure_init());
    if (height == 0) return fail(PIL2GL_EINVAL, "height must be > 0");
    if (!nodes) return fail(PIL2GL_EINVAL, "null buffer");
    hipStream_t st = as_stream(stream);
    // merklehash_p.js:49: nodes is a fresh (zeroed) BigUint64Array; the zero padding of odd levels relies on it
    u64 total = pil2gl_merkle_num_nodes(height);
    if (total > height * 4) HIP_TRY(hipMemsetAsync(nodes + height * 4, 0, (total - height * 4) * 8, st));
    // merklehash_p.js:87-103 (offsets in u64 words instead of bytes)
    u64 pIn = 0, n64 = height * 4;
    u64 nextN64 = ((n64 - 1) / 8 + 1) * 4;
    u64 pOut = pIn + nextN64 * 2;
    while (n64 > 4) {
        P2_TRY(pil2gl_merkelize_level_dev(nodes + pIn, nextN64 / 4, nodes + pOut, stream));
        n64 = nextN64;
        nextN64 = ((n64 - 1) / 8 + 1) * 4;
        pIn = pOut;
        pOut = pIn + nextN64 * 2;
    }
    return PIL2GL_OK;
}

int pil2gl_merkelize_dev(const uint64_t *elems, uint64_t width, uint64_t height, int split, uint64_t *nodes, void *stream) {
    P2_TRY(ensure_init());
    if (height == 0) return fail(PIL2GL_EINVAL, "height must be > 0");
    if (!nodes || (!elems && width)) return fail(PIL2GL_EINVAL, "null buffer");
    P2_TRY(pil2gl_linear_hash_rows_dev(elems, width, height, split, nodes, stream));
    return pil2gl_merkelize_digests_dev(nodes, height, stream);
}

int pil2gl_group_proof_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height,
                           uint64_t idx, uint64_t *hostVals, uint64_t *hostSiblings, uint32_t *nLevels) {
    P2_TRY(ensure_init());
    if (idx >= height) return fail(PIL2GL_EINVAL, "Out of range");      // merklehash_p.js:143
    if (!nodes || !hostSiblings || (width && (!elems || !hostVals))) return fail(PIL2GL_EINVAL, "null buffer");
    if (width) HIP_TRY(hipMemcpy(hostVals, elems + idx * width, width * 8, hipMemcpyDeviceToHost));
    u64 offset = 0, n = height * 4;
    uint32_t lvl = 0;
    while (n > 4) {                                                      // merklehash_p.js:154-167
        u64 si = (idx ^ 1) * 4;
        HIP_TRY(hipMemcpy(hostSiblings + 4 * lvl, nodes + offset + si, 32, hipMemcpyDeviceToHost));
        u64 nextN = ((n - 1) / 8 + 1) * 4;
        offset += nextN * 2; n = nextN; idx >>= 1; lvl++;
    }
    if (nLevels) *nLevels = lvl;
    return PIL2GL_OK;
}

int pil2gl_group_proofs_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height,
                            const uint64_t *hostIdxs, uint32_t nIdx, uint64_t *hostOut, uint32_t *nLevels) {
    P2_TRY(ensure_init());
    if (!nIdx) return PIL2GL_OK;
    if (!nodes || !hostIdxs || !hostOut || (!elems && width)) return fail(PIL2GL_EINVAL, "null buffer");
    u32 lv = 0;
    for (u64 n = height * 4; n > 4; n = ((n - 1) / 8 + 1) * 4) lv++;
    for (u32 i = 0; i < nIdx; i++) if (hostIdxs[i] >= height) return fail(PIL2GL_EINVAL, "Out of range");     // merklehash_p.js:143
    const u64 stride = width + 4ull * lv;
    u64 *d;
    P2_TRY(scratch(6, (u64)nIdx * (stride + 1), &d));
    u64 *dIdx = d + (u64)nIdx * stride;
    HIP_TRY(hipMemcpy(dIdx, hostIdxs, (u64)nIdx * 8, hipMemcpyHostToDevice));
    group_proofs_kernel<<<nIdx, 64>>>(elems, nodes, width, height, dIdx, lv, d);
    KERNEL_CHECK();
    HIP_TRY(hipMemcpy(hostOut, d, (u64)nIdx * stride * 8, hipMemcpyDeviceToHost));
    if (nLevels) *nLevels = lv;
    return PIL2GL_OK;
}

// Verifier side: calculateRootFromGroupProof (merklehash_p.js:169-203) for a batch of openings in the
// packed layout pil2gl_group_proofs_dev writes: per opening `width` values, then `levels` sibling digests.
int pil2gl_roots_from_group_proofs(const uint64_t *hostProofs, uint64_t width, uint32_t levels, const uint64_t *hostIdxs, uint32_t nIdx,
                                   int splitLinearHash, uint64_t *hostRoots) {
    P2_TRY(ensure_init());
    if (!nIdx) return PIL2GL_OK;
    if (!hostProofs || !hostIdxs || !hostRoots) return fail(PIL2GL_EINVAL, "null buffer");
    if (levels > 64) return fail(PIL2GL_EINVAL, "too many levels");
    const u64 stride = width + 4ull * levels, nv = (u64)nIdx * width, ns = (u64)nIdx * levels * 4;
    std::vector<u64> h(nv + ns + nIdx + 1);
    for (u32 q = 0; q < nIdx; q++) {
        memcpy(h.data() + (u64)q * width, hostProofs + q * stride, width * 8);
        memcpy(h.data() + nv + (u64)q * levels * 4, hostProofs + q * stride + width, 4ull * levels * 8);
        h[nv + ns + q] = hostIdxs[q];
    }
    u64 *d;
    P2_TRY(scratch(6, nv + ns + nIdx + 8ull * nIdx + 1, &d));
    u64 *dVals = d, *dSib = d + nv, *dIdx = dSib + ns, *dLeaf = dIdx + nIdx, *dRoots = dLeaf + 4ull * nIdx;
    HIP_TRY(hipMemcpy(d, h.data(), (nv + ns + nIdx) * 8, hipMemcpyHostToDevice));
    P2_TRY(pil2gl_linear_hash_rows_dev(dVals, width, nIdx, splitLinearHash, dLeaf, nullptr));
    merkle_path_roots_kernel<<<(nIdx + 255) / 256, 256>>>(dLeaf, dSib, dIdx, nIdx, levels, dRoots);
    KERNEL_CHECK();
    HIP_TRY(hipMemcpy(hostRoots, dRoots, 4ull * nIdx * 8, hipMemcpyDeviceToHost));
    return PIL2GL_OK;
}

// transcript.js:49-66 for a list: nBlocks full blocks of 8 absorbed one after the other, starting from capacity cap[4]
int pil2gl_sponge_absorb(const uint64_t *hostBlocks, uint64_t nBlocks, const uint64_t hostCap[4], uint64_t hostOut12[12]) {
    P2_TRY(ensure_init());
    if (!hostBlocks || !hostCap || !hostOut12) return fail(PIL2GL_EINVAL, "null buffer");
    if (nBlocks == 0) return fail(PIL2GL_EINVAL, "nothing to absorb");
    u64 *d; bool owned;
    P2_TRY(stage_acquire(8 * nBlocks + 4 + 12, &d, &owned));
    int rc = PIL2GL_OK;
    hipError_t e = hipMemcpy(d, hostBlocks, 8 * nBlocks * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 8 * nBlocks, hostCap, 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) {
        sponge_chain_kernel<<<1, 64>>>(d, nBlocks, d + 8 * nBlocks, d + 8 * nBlocks + 4);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(hostOut12, d + 8 * nBlocks + 4, 96, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "sponge_chain_kernel");
    }
    stage_release(d, owned);
    return rc;
}

// ---- host-pointer forms ----
static int with_dev(const uint64_t *hIn, u64 nIn, const uint64_t *hIn2, u64 nIn2, uint64_t *hOut, u64 nOut,
                    int (*fn)(const u64 *, const u64 *, u64 *, void *), void *arg) {
    P2_TRY(ensure_init());
    u64 *d = nullptr; bool owned = false;
    P2_TRY(stage_acquire(nIn + nIn2 + nOut, &d, &owned));
    int rc = PIL2GL_OK;
    hipError_t e = hipSuccess;
    if (nIn) e = hipMemcpy(d, hIn, nIn * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && nIn2) e = hipMemcpy(d + nIn, hIn2, nIn2 * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) rc = fn(d, nIn2 ? d + nIn : nullptr, d + nIn + nIn2, arg);
    if (rc == PIL2GL_OK && nOut) { e = hipMemcpy(hOut, d + nIn + nIn2, nOut * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H"); }
    stage_release(d, owned);
    return rc;
}
struct LhArgs { u64 width, height; int split; };
struct PsArgs { u64 count; u32 nOut; };

int pil2gl_linear_hash_rows(const uint64_t *in, uint64_t width, uint64_t height, int split, uint64_t *out) {
    LhArgs a = { width, height, split };
    return with_dev(in, width * height, nullptr, 0, out, height * 4,
                    [](const u64 *i, const u64 *, u64 *o, void *p) { LhArgs *a = (LhArgs *)p; return pil2gl_linear_hash_rows_dev(i, a->width, a->height, a->split, o, nullptr); }, &a);
}
int pil2gl_merkelize_level(const uint64_t *in, uint64_t nOps, uint64_t *out) {
    return with_dev(in, nOps * 8, nullptr, 0, out, nOps * 4,
                    [](const u64 *i, const u64 *, u64 *o, void *p) { return pil2gl_merkelize_level_dev(i, *(u64 *)p, o, nullptr); }, &nOps);
}
int pil2gl_poseidon(const uint64_t *in, const uint64_t *cap, uint64_t count, uint32_t nOut, uint64_t *out) {
    if (nOut < 1 || nOut > 12) return fail(PIL2GL_EINVAL, "nOut must be 1..12");
    PsArgs a = { count, nOut };
    return with_dev(in, count * 8, cap, cap ? count * 4 : 0, out, count * nOut,
                    [](const u64 *i, const u64 *c, u64 *o, void *p) { PsArgs *a = (PsArgs *)p; return pil2gl_poseidon_dev(i, c, a->count, a->nOut, o, nullptr); }, &a);
}
struct MdsArgs { u64 n; u32 layers; int mfma; };
int pil2gl_selftest_mds(const uint64_t *states, uint64_t n, uint32_t layers, int mfma, uint64_t *out) {
    if (n == 0) return PIL2GL_OK;
    MdsArgs a = { n, layers, mfma };
    return with_dev(states, n * 12, nullptr, 0, out, n * 12,
                    [](const u64 *i, const u64 *, u64 *o, void *p) {
                        MdsArgs *a = (MdsArgs *)p;
                        mds_selftest_kernel<<<(unsigned)((a->n + 255) / 256), 256>>>(i, a->n, a->layers, a->mfma, o);
                        KERNEL_CHECK();
                        return (int)PIL2GL_OK;
                    }, &a);
}
int pil2gl_selftest_poseidon(const uint64_t *states, uint64_t n, int what, uint64_t *out) {
    if (n == 0) return PIL2GL_OK;
    if (what < 0 || what > 4) return fail(PIL2GL_EINVAL, "what must be 0..4");
    MdsArgs a = { n, 0, what };
    return with_dev(states, n * 12, nullptr, 0, out, n * 12,
                    [](const u64 *i, const u64 *, u64 *o, void *p) {
                        MdsArgs *a = (MdsArgs *)p;
                        poseidon_selftest_kernel<<<(unsigned)((a->n + 255) / 256), 256>>>(i, a->n, a->mfma, o);
                        KERNEL_CHECK();
                        return (int)PIL2GL_OK;
                    }, &a);
}
int pil2gl_selftest_clock(uint32_t iters, double *mhz /* [3]: median, 5th and 95th percentile over the workgroups */) {
    P2_TRY(ensure_init());
    if (!mhz || iters == 0 || iters > 4096) return fail(PIL2GL_EINVAL, "iters must be 1..4096");
    const unsigned blocks = 256 * 4 * 8;                        // eight rounds of four workgroups per CU
    u64 *d;
    P2_TRY(scratch(5, (u64)blocks * 256 + 2ull * blocks, &d));
    clock_probe_kernel<<<blocks, 256>>>(d, (int)iters, d + (u64)blocks * 256);
    KERNEL_CHECK();
    std::vector<u64> h(2 * blocks);
    HIP_TRY(hipMemcpy(h.data(), d + (u64)blocks * 256, 16ull * blocks, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (unsigned i = 0; i < blocks; i++) if (h[2 * i + 1]) v.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
    if (v.empty()) return fail(PIL2GL_EHIP, "no clock samples");
    std::sort(v.begin(), v.end());
    mhz[0] = v[v.size() / 2]; mhz[1] = v[v.size() / 20]; mhz[2] = v[v.size() * 19 / 20];
    return PIL2GL_OK;
}
int pil2gl_merkelize(const uint64_t *elems, uint64_t width, uint64_t height, int split, uint64_t *nodes) {
    if (height == 0) return fail(PIL2GL_EINVAL, "height must be > 0");
    LhArgs a = { width, height, split };
    return with_dev(elems, width * height, nullptr, 0, nodes, pil2gl_merkle_num_nodes(height),
                    [](const u64 *i, const u64 *, u64 *o, void *p) { LhArgs *a = (LhArgs *)p; return pil2gl_merkelize_dev(i, a->width, a->height, a->split, o, nullptr); }, &a);
}

}  // extern "C"
