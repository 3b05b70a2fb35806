// BN128 (BN254 scalar field) Poseidon linear hash and arity-ary Merkle tree, gfx950.
//
// Replaces src/helpers/hash/merklehash/merklehash_bn128_p.js:28-182 (merkelize, _getNNodes, getGroupProof), its worker
// merklehash_bn128_worker.js:13-144 (linearHash, merkelizeLevel) and the third-party kernels those call
// (circomlibjs@0.1.7 buildPoseidonWasm `poseidon`, wasmcurves@0.1.5 F1m `frm_toMontgomery`; neither is under the
// reference tree).  The permutation follows the in-tree statement circuits.bn128/custom/poseidon.circom:6-45
// (t = nInputs+1, RF = 8, RP = N_ROUNDS_P[t-2], x^5, dense round constants and MDS); its parameters are produced on the
// host by the published Poseidon parameter generation (Grain LFSR) and agree with every constant set the reference holds
// (tests/test_bn128_oracle.py pins the same generator; tests/test_gpu_bn128.py compares this one with it).
//
// The partial rounds run in the sparse form (derivation below, next to derive_sparse()): 2t-1 products per round instead
// of t^2; outputs are the same field elements as the dense statement (the dense form is kept for tests).
//
// Layout: one permutation per lane; the t state elements (8 limbs each, Montgomery form) live in LDS as
// [element][limb][lane] (conflict-free, element index may be a run-time value); a dense layer keeps its new rows in
// registers until all are done (dense_mul).  A row is accumulated unreduced in 17 limbs (t products) and reduced once.
// Nodes are stored as the reference stores them: 4 little-endian u64 words of the Montgomery form.
#include "common.h"
#include "bn_field.cuh"
#include "bn_mfma.cuh"
#include "bn_field29.cuh"
#include <mutex>
#include <vector>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

using namespace pil2gl;
using bn::u32;

// The S-box of the matrix-core pipeline in radix 2^29 (bn_field29.cuh: no carry instructions; its output is the state form times 2^-20,
// which the tiles reading it take back).  -DBN_SBOX29=0: the 32-bit-limb products (A/B builds).
#ifndef BN_SBOX29
#define BN_SBOX29 1
#endif
// State elements kept in LDS (the rest in private memory: "Where the state lives" below) and the batch in which the partial rounds fetch the others'
// operands (partial_rounds_mfma_impl::rows): both shape the order of the tile stream the host writes (mfma_partial_tables).
#ifndef BN_LDS_ELEMS
#define BN_LDS_ELEMS 10
#endif
#ifndef BN_SMALL_T
#define BN_SMALL_T 4                    // widths up to this run the permutation round by round with the state and the layer's tiles in registers (perm_small)
#endif
#ifndef BN_HI_BATCH
#define BN_HI_BATCH 4
#endif

namespace {

constexpr int BN_BLOCK = 64;                         // lanes of a wave = permutations a wave carries
// Waves per workgroup.  Every wave works alone on its own 64 permutations and its own LDS slice; what the waves of a workgroup share is
// TIME: a barrier at the start of every matrix phase (dense layer, rows on y, column update) keeps them on the same operand tiles, so that
// of the workgroup's loads of a tile one goes to the L2 and the others hit the CU's L1.
#ifndef BN_WG_WAVES
#define BN_WG_WAVES 4
#endif
constexpr int BN_THREADS = BN_BLOCK * BN_WG_WAVES;
#if BN_WG_WAVES > 1
#define BN_SYNC() __syncthreads()
#else
#define BN_SYNC()
#endif
constexpr int N_ROUNDS_F = 8;
const int N_ROUNDS_P[16] = { 56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68 };   // poseidon.circom:8

// ------------------------------------------------------------------------------------------ host 256-bit arithmetic
struct U256 { u64 w[4]; };
const U256 HR = { { 0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull } };
const U256 HR2 = { { 0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull } };
const u64 HN0 = 0xc2e1f593efffffffull;
typedef unsigned __int128 u128;

bool h_ge(const U256 &a, const U256 &b) { for (int i = 3; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; return true; }
U256 h_sub(const U256 &a, const U256 &b) { U256 r; u64 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a.w[i] - b.w[i] - br; r.w[i] = (u64)d; br = (u64)(d >> 64) & 1; } return r; }
U256 h_addmod(const U256 &a, const U256 &b) {
    U256 r; u64 c = 0;
    for (int i = 0; i < 4; i++) { u128 s = (u128)a.w[i] + b.w[i] + c; r.w[i] = (u64)s; c = (u64)(s >> 64); }
    if (c || h_ge(r, HR)) r = h_sub(r, HR);
    return r;
}
U256 h_addraw(const U256 &a, const U256 &b) { U256 r; u64 c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)a.w[i] + b.w[i] + c; r.w[i] = (u64)s; c = (u64)(s >> 64); } return r; }
U256 h_mont(const U256 &a, const U256 &b) {           // a*b/2^256 mod r
    u64 t[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        u64 c = 0;
        for (int j = 0; j < 4; j++) { u128 x = (u128)a.w[j] * b.w[i] + t[j] + c; t[j] = (u64)x; c = (u64)(x >> 64); }
        u128 x = (u128)t[4] + c; t[4] = (u64)x; t[5] = (u64)(x >> 64);
        u64 m = t[0] * HN0;
        c = (u64)(((u128)m * HR.w[0] + t[0]) >> 64);
        for (int j = 1; j < 4; j++) { u128 y = (u128)m * HR.w[j] + t[j] + c; t[j - 1] = (u64)y; c = (u64)(y >> 64); }
        x = (u128)t[4] + c; t[3] = (u64)x; t[4] = t[5] + (u64)(x >> 64);
    }
    U256 r = { { t[0], t[1], t[2], t[3] } };
    if (t[4] || h_ge(r, HR)) r = h_sub(r, HR);
    return r;
}
U256 h_submod(const U256 &a, const U256 &b) { return h_ge(a, b) ? h_sub(a, b) : h_sub(h_addraw(a, HR), b); }   // a, b < r < 2^254
bool h_is_zero(const U256 &a) { return !(a.w[0] | a.w[1] | a.w[2] | a.w[3]); }
U256 h_to_mont(const U256 &a) { return h_mont(a, HR2); }
U256 h_from_mont(const U256 &a) { U256 one = { { 1, 0, 0, 0 } }; return h_mont(a, one); }
U256 h_inv_mont(const U256 &a) {                      // a^(r-2), Montgomery in and out
    U256 e = HR; e.w[0] -= 2;
    U256 acc = h_to_mont(U256{ { 1, 0, 0, 0 } });
    for (int i = 255; i >= 0; i--) {
        acc = h_mont(acc, acc);
        if ((e.w[i / 64] >> (i % 64)) & 1) acc = h_mont(acc, a);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------ Poseidon parameters
// Grain LFSR parameter stream of the Poseidon paper's reference generator: 80-bit register initialised with
// field=1 (2 bits), sbox=0 (4), n=254 (12), t (12), RF (10), RP (10), thirty ones; 160 warm-up steps; output bits are
// self-shrunk (a 1 passes the next bit, a 0 drops it); field elements = 254 bits MSB first, rejected when >= r (round
// constants) or reduced mod r (the 2t Cauchy points); M[i][j] = 1/(x_i + y_j).
struct Grain {
    uint8_t b[80]; int p = 0;
    int step() { int nb = b[(p + 62) % 80] ^ b[(p + 51) % 80] ^ b[(p + 38) % 80] ^ b[(p + 23) % 80] ^ b[(p + 13) % 80] ^ b[p]; b[p] = (uint8_t)nb; p = (p + 1) % 80; return nb; }
    int next() { int nb = step(); while (!nb) { step(); nb = step(); } return step(); }
    U256 rnd() { U256 v = { { 0, 0, 0, 0 } }; for (int i = 0; i < 254; i++) { for (int k = 3; k > 0; k--) v.w[k] = (v.w[k] << 1) | (v.w[k - 1] >> 63); v.w[0] = (v.w[0] << 1) | (u64)next(); } return v; }
    Grain(int t, int rp) {
        int n = 0;
        auto put = [&](unsigned v, int w) { for (int i = w - 1; i >= 0; i--) b[n++] = (v >> i) & 1; };
        put(1, 2); put(0, 4); put(254, 12); put((unsigned)t, 12); put(N_ROUNDS_F, 10); put((unsigned)rp, 10);
        while (n < 80) b[n++] = 1;
        for (int i = 0; i < 160; i++) step();
    }
};

// device tables of one state width t, Montgomery form, 8 limbs per element, one allocation:
//   C8[8][t]  constants of the 4+4 full rounds (the first of the second half also carries what the partial rounds pushed out)
//   M[t][t]   dense MDS;  D[t-1][t-1] = Mhat^RP;  S[RP] scalar constants;  V[RP][t-1], W[RP][t-1] sparse rows / columns
//   Cd[(8+RP)][t] the original constants, for the dense (test) form
//   Mt / Dt   the same two dense layers as matrix-core operand tiles (bn_mfma.cuh), MK / DK their per-row constants;
//   Pt        the tile stream of the blocked partial rounds, KR / KU its row constants (mfma_partial_tables)
struct Params { int t = 0, rp = 0; u32 *base = nullptr, *C8, *M, *D, *S, *V, *W, *Cd; u32 m00[8];
                const bnm::v4i *Mt = nullptr, *Dt = nullptr, *Pt = nullptr; const u32 *MK = nullptr, *DK = nullptr, *KR = nullptr, *KU = nullptr;
                const bnm::v4i *Mt0 = nullptr; const u32 *MK0 = nullptr, *C0p = nullptr;        // the first layer for inputs S-boxed as plain integers (plain_sbox_store)
                const bnm::v4i *St = nullptr; const u32 *SK = nullptr; };                         // widths <= BN_SMALL_T: the round-by-round form in registers (perm_small)
Params g_params[18];
std::mutex g_mu;

typedef std::vector<U256> Vec;
Vec mat_vec(const Vec &A, const Vec &x, int n) {        // A (n x n) * x
    Vec y((size_t)n);
    for (int i = 0; i < n; i++) { U256 a = { { 0, 0, 0, 0 } }; for (int j = 0; j < n; j++) a = h_addmod(a, h_mont(A[(size_t)i * n + j], x[j])); y[i] = a; }
    return y;
}
int mat_inv(Vec &A, int n) {                             // Gauss-Jordan in place (Montgomery form)
    const U256 one = h_to_mont(U256{ { 1, 0, 0, 0 } });
    Vec I((size_t)n * n, U256{ { 0, 0, 0, 0 } });
    for (int i = 0; i < n; i++) I[(size_t)i * n + i] = one;
    for (int c = 0; c < n; c++) {
        int p = c;
        while (p < n && h_is_zero(A[(size_t)p * n + c])) p++;
        if (p == n) return fail(PIL2GL_EINVAL, "singular MDS sub-matrix");
        if (p != c) for (int j = 0; j < n; j++) { std::swap(A[(size_t)p * n + j], A[(size_t)c * n + j]); std::swap(I[(size_t)p * n + j], I[(size_t)c * n + j]); }
        const U256 iv = h_inv_mont(A[(size_t)c * n + c]);
        for (int j = 0; j < n; j++) { A[(size_t)c * n + j] = h_mont(A[(size_t)c * n + j], iv); I[(size_t)c * n + j] = h_mont(I[(size_t)c * n + j], iv); }
        for (int r = 0; r < n; r++) {
            if (r == c || h_is_zero(A[(size_t)r * n + c])) continue;
            const U256 f = A[(size_t)r * n + c];
            for (int j = 0; j < n; j++) {
                A[(size_t)r * n + j] = h_submod(A[(size_t)r * n + j], h_mont(f, A[(size_t)c * n + j]));
                I[(size_t)r * n + j] = h_submod(I[(size_t)r * n + j], h_mont(f, I[(size_t)c * n + j]));
            }
        }
    }
    A = I;
    return PIL2GL_OK;
}

// Sparse form of the RP partial rounds.  Dense statement: x_{k+1} = M * sigma(x_k + c_k), sigma = x^5 on element 0 only.
//  (1) constants: with e_0 = c_0, s_k = e_k[0], e_{k+1} = c_{k+1} + M*(0, e_k[1:]), the sequence y_{k+1} = M*sigma'(y_k + s_k e0)
//      satisfies x_k + c_k = y_k + e_k; what is left, f = M*(0, e_{RP-1}[1:]), joins the next full round's constants.
//  (2) matrices: M = [[m00, v],[w, Mh]].  With D_k = diag(1, Mh^k), M*D_k = D_{k+1} * [[m00, v*Mh^k],[Mh^-(k+1) w, I]], and
//      D_k commutes with sigma', so y_k = D_k u_k with u_{k+1} = [[m00, V_k],[W_k, I]] * sigma'(u_k): 2t-1 products;
//      one dense multiplication by D_RP = diag(1, Mh^RP) closes the sequence.
int derive_sparse(int t, int rp, const Vec &C, const Vec &M, Vec &C8, Vec &D, Vec &S, Vec &V, Vec &W) {
    const int n = t - 1;
    const U256 zero = { { 0, 0, 0, 0 } };
    Vec Mh((size_t)n * n), v((size_t)n), w((size_t)n);
    for (int i = 0; i < n; i++) { v[i] = M[(size_t)1 + i]; w[i] = M[(size_t)(i + 1) * t]; for (int j = 0; j < n; j++) Mh[(size_t)i * n + j] = M[(size_t)(i + 1) * t + 1 + j]; }
    Vec Mhi = Mh;
    P2_TRY(mat_inv(Mhi, n));
    S.resize((size_t)rp); V.resize((size_t)rp * n); W.resize((size_t)rp * n); C8.resize((size_t)8 * t);
    Vec e(C.begin() + (size_t)4 * t, C.begin() + (size_t)5 * t), f;
    for (int k = 0; k < rp; k++) {
        S[k] = e[0];
        Vec et = e; et[0] = zero;
        Vec Me = mat_vec(M, et, t);
        if (k + 1 < rp) for (int j = 0; j < t; j++) e[j] = h_addmod(C[(size_t)(5 + k) * t + j], Me[j]);
        else f = Me;
    }
    Vec vk = v, wk = mat_vec(Mhi, w, n);
    for (int k = 0; k < rp; k++) {
        for (int j = 0; j < n; j++) { V[(size_t)k * n + j] = vk[j]; W[(size_t)k * n + j] = wk[j]; }
        Vec nv((size_t)n);
        for (int j = 0; j < n; j++) { U256 a = zero; for (int i = 0; i < n; i++) a = h_addmod(a, h_mont(vk[i], Mh[(size_t)i * n + j])); nv[j] = a; }
        vk = nv;
        wk = mat_vec(Mhi, wk, n);
    }
    D.assign((size_t)n * n, zero);
    const U256 one = h_to_mont(U256{ { 1, 0, 0, 0 } });
    for (int i = 0; i < n; i++) D[(size_t)i * n + i] = one;
    for (int k = 0; k < rp; k++) {
        Vec nd((size_t)n * n);
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { U256 a = zero; for (int q = 0; q < n; q++) a = h_addmod(a, h_mont(Mh[(size_t)i * n + q], D[(size_t)q * n + j])); nd[(size_t)i * n + j] = a; }
        D = nd;
    }
    for (int r = 0; r < 4; r++) for (int j = 0; j < t; j++) {
        C8[(size_t)r * t + j] = C[(size_t)r * t + j];
        C8[(size_t)(4 + r) * t + j] = r == 0 ? h_addmod(C[(size_t)(4 + rp) * t + j], f[j]) : C[(size_t)(4 + rp + r) * t + j];
    }
    return PIL2GL_OK;
}

// Operand tiles and row constants for the matrix cores (layout and derivation: bn_mfma.cuh).
struct MfmaConsts {
    U256 P[32];                                      // 2^(8b+32) mod r, plain
    U256 off;                                        // sum_k ACC_BIAS 256^k mod r
    MfmaConsts() {
        U256 v = { { 1, 0, 0, 0 } };
        for (int e = 0; e < 32; e++) v = h_addmod(v, v);
        for (int b = 0; b < 32; b++) { P[b] = v; for (int e = 0; e < 8; e++) v = h_addmod(v, v); }
        off = U256{ { 0, 0, 0, 0 } };
        v = U256{ { (u64)bnm::ACC_BIAS, 0, 0, 0 } };
        for (int k = 0; k < 32; k++) { off = h_addmod(off, v); for (int e = 0; e < 8; e++) v = h_addmod(v, v); }
    }
};
// one tile (1 KB, lane order) of the coefficient a0 (Montgomery form); tot += the sum of its 32 constants.  sboxed: the operand this tile
// multiplies comes straight out of the S-box, i.e. (bn_field29.cuh) carries a factor 2^-20: the coefficient takes it back
void mfma_tile(const MfmaConsts &mc, const U256 &a0, int8_t *tile, U256 &tot, bool sboxed, const U256 *extra = nullptr) {
    U256 a = sboxed && BN_SBOX29 ? h_mont(a0, h_to_mont(U256{ { 1ull << 20, 0, 0, 0 } })) : a0;
    if (extra) a = h_mont(a, *extra);                // (a further factor in Montgomery form: the plain-input first layer)
    for (int b = 0; b < 32; b++) {
        const U256 c = h_mont(a, mc.P[b]);           // a 2^(8b+32) mod r as a plain integer
        tot = h_addmod(tot, c);
        int d[32], carry = 0;
        for (int k = 0; k < 32; k++) {
            int v = (int)((c.w[k / 8] >> (8 * (k % 8))) & 255) + carry;
            carry = v >= 128;
            d[k] = carry ? v - 256 : v;
        }                                             // c < 2^254: the top digit takes the last carry
        const int g = b / 16, sl = b % 16;
        for (int m = 0; m < 32; m++) {
            const int pos = 16 * ((m / 4) % 2) + 4 * (m / 8) + m % 4;
            tile[(size_t)(g * 32 + m) * 16 + sl] = (int8_t)d[pos];
        }
    }
}
// the row constant: (128 tot - nAcc sum_k ACC_BIAS 256^k) / 2^32 + fold mod r   (nAcc accumulations started at the bias make up the row)
U256 mfma_row_const(const MfmaConsts &mc, U256 tot, int nAcc, const U256 &fold) {
    const U256 inv32 = { { 0, 0, 0, 1ull << 32 } };  // 2^224: h_mont(a, 2^224) = a / 2^32
    for (int e = 0; e < 7; e++) tot = h_addmod(tot, tot);
    for (int e = 0; e < nAcc; e++) tot = h_submod(tot, mc.off);
    return h_addmod(h_mont(tot, inv32), fold);
}
// A: rows x cols entries in Montgomery form.  tiles: rows*cols KB; K: rows plain integers mod r (callers add what follows the layer).
// colFactor: columns >= 1 carry this further factor (Montgomery form)
void mfma_layer_tables(const Vec &A, int rows, int cols, std::vector<int8_t> &tiles, Vec &K, bool sboxed, const U256 *colFactor = nullptr) {
    const MfmaConsts mc;
    const U256 zero = { { 0, 0, 0, 0 } };
    tiles.assign((size_t)rows * cols * 1024, 0);
    K.resize((size_t)rows);
    for (int i = 0; i < rows; i++) {
        U256 tot = zero;
        for (int j = 0; j < cols; j++) mfma_tile(mc, A[(size_t)i * cols + j], tiles.data() + ((size_t)i * cols + j) * 1024, tot, sboxed, j >= 1 ? colFactor : nullptr);
        K[i] = mfma_row_const(mc, tot, 1, zero);
    }
}
// The partial rounds four to a block, blocks two to a SUPER-BLOCK (partial_rounds_mfma): the tile stream in the order the kernel consumes it and
// the row constants.  Block b (rounds k0 = 4b .. k0+3), z_i = the S-box output of round k0+i, y = elements 1..n at the start of b's SUPER-BLOCK:
//   x0 after round k0+i = m00 z_i + sum_j V[k0+i][j] y_j + sum_{i'<i} (V[k0+i] . W[k0+i']) z_i'  [+ for the second block of a super-block the same
//   cross terms with the four z of the first];   y_j after the super-block = y_j + sum over its rounds of W[k][j] z_k  (one column update per 8 rounds).
// Stream per block, in two passes (rows 0-1, then rows 2-3): n x 2 tiles V[k0+i][j] (j outer), second block: 2 x 4 tiles (V[k0+i] . W[k0-4+s]);  then the block's own
// cross terms, round i = 0..3: i + 1 tiles (z_0 .. z_i of the block; the last is m00).  Per super-block after its blocks: n x (1 + 8) tiles (1, W[k][j]; four zero tiles when
// the super-block has one block).
// KR[k]: round k's row constant with S[k+1] folded in while round k+1 is one of these; KU[sb][j]: the column constants.
void mfma_partial_tables(int t, int rp, const Vec &S, const Vec &V, const Vec &W, const U256 &m00, std::vector<int8_t> &tiles, Vec &KR, Vec &KU) {
    const MfmaConsts mc;
    const U256 zero = { { 0, 0, 0, 0 } }, one = h_to_mont(U256{ { 1, 0, 0, 0 } });
    const int n = t - 1, nb = rp / 4, nsb = (nb + 1) / 2;
    tiles.assign(((size_t)nb * (4 * n + 10) + (size_t)nsb * 9 * n + (size_t)(nb / 2) * 16) * 1024, 0);      // per block its rows and rounds, per super-block the columns (1 + 8 tiles each), 16 more for a second block's rows
    KR.assign((size_t)nb * 4, zero); KU.assign((size_t)nsb * n, zero);
    auto dot = [&](int ka, int kb) { U256 c = zero; for (int j = 0; j < n; j++) c = h_addmod(c, h_mont(V[(size_t)ka * n + j], W[(size_t)kb * n + j])); return c; };
    int8_t *tp = tiles.data();
    for (int sb = 0; sb < nsb; sb++) {
        const int halves = nb - 2 * sb >= 2 ? 2 : 1;
        for (int h = 0; h < halves; h++) {
            const int k0 = 4 * (2 * sb + h);
            U256 tot[4] = { zero, zero, zero, zero };
            // the kernel takes the rows two at a time (partial_rounds_mfma_impl::rows): the first pass walks the columns in order, the second starts with the last
            // batch of the columns that live in private memory (still in its registers), then the first batch, then the columns in LDS
            const int nlo = n + 1 <= BN_LDS_ELEMS ? n : BN_LDS_ELEMS - 1, nhi = n - nlo, hb2 = nhi < BN_HI_BATCH ? nhi : BN_HI_BATCH, spl = nhi - hb2;
            for (int pass = 0; pass < 2; pass++) {
                std::vector<int> order;
                if (pass == 0) for (int j = 0; j < n; j++) order.push_back(j);
                else {
                    for (int q = spl; q < nhi; q++) order.push_back(nlo + q);
                    for (int q = 0; q < spl; q++) order.push_back(nlo + q);
                    for (int j = 0; j < nlo; j++) order.push_back(j);
                }
                for (int j : order) for (int i = 2 * pass; i < 2 * pass + 2; i++, tp += 1024) mfma_tile(mc, V[(size_t)(k0 + i) * n + j], tp, tot[i], false);
                if (h == 1) for (int i = 2 * pass; i < 2 * pass + 2; i++) for (int s = 0; s < 4; s++, tp += 1024) mfma_tile(mc, dot(k0 + i, k0 - 4 + s), tp, tot[i], true);
            }
            for (int i = 0; i < 4; i++) {
                for (int ip = 0; ip <= i; ip++, tp += 1024)                                    // round i: its i + 1 cross terms, z_0 .. z_i of the block
                    mfma_tile(mc, ip < i ? dot(k0 + i, k0 + ip) : m00, tp, tot[i], true);      // a z: the S-box's output
                KR[(size_t)k0 + i] = mfma_row_const(mc, tot[i], 2, k0 + i + 1 < 4 * nb ? S[(size_t)k0 + i + 1] : zero);
            }
        }
        for (int j = 0; j < n; j++) {
            U256 tu = zero;
            mfma_tile(mc, one, tp, tu, false); tp += 1024;
            if (halves == 1) tp += 4 * 1024;       // (zero tiles where the kernel multiplies the absent first block's operands: one form of the column)
            for (int s = 0; s < 4 * halves; s++, tp += 1024) mfma_tile(mc, W[(size_t)(8 * sb + s) * n + j], tp, tu, true);
            KU[(size_t)sb * n + j] = mfma_row_const(mc, tu, 1, zero);
        }
    }
}

int get_params(int t, const Params **out) {
    if (t < 2 || t > 17) return fail(PIL2GL_EINVAL, "BN128 Poseidon takes 1..16 inputs (t=%d)", t);
    std::lock_guard<std::mutex> lk(g_mu);
    Params &P = g_params[t];
    if (!P.t) {
        const int rp = N_ROUNDS_P[t - 2], nC = (N_ROUNDS_F + rp) * t, n = t - 1;
        Grain g(t, rp);
        Vec C((size_t)nC), M((size_t)t * t), xy((size_t)2 * t);
        for (int i = 0; i < nC; i++) { U256 v = g.rnd(); while (h_ge(v, HR)) v = g.rnd(); C[i] = h_to_mont(v); }
        for (int i = 0; i < 2 * t; i++) { U256 v = g.rnd(); while (h_ge(v, HR)) v = h_sub(v, HR); xy[i] = h_to_mont(v); }
        for (int i = 0; i < t; i++) for (int j = 0; j < t; j++) M[(size_t)i * t + j] = h_inv_mont(h_addmod(xy[i], xy[t + j]));
        Vec C8, D, S, V, W;
        P2_TRY(derive_sparse(t, rp, C, M, C8, D, S, V, W));
        Vec all;
        auto put = [&](const Vec &x) { size_t o = all.size(); all.insert(all.end(), x.begin(), x.end()); return o; };
        const size_t oC8 = put(C8), oM = put(M), oD = put(D), oS = put(S), oV = put(V), oW = put(W), oCd = put(C);
        u32 *d = nullptr;
        HIP_TRY(hipMalloc((void **)&d, all.size() * 32));
        HIP_TRY(hipMemcpy(d, all.data(), all.size() * 32, hipMemcpyHostToDevice));
        P.base = d; P.C8 = d + oC8 * 8; P.M = d + oM * 8; P.D = d + oD * 8; P.S = d + oS * 8; P.V = d + oV * 8; P.W = d + oW * 8; P.Cd = d + oCd * 8;
        memcpy(P.m00, M[0].w, 32);
        {
            std::vector<int8_t> tm, td, tpr, tm0; Vec km0, km, kd, kr, ku, km0p, c0p;
            mfma_layer_tables(M, t, t, tm, km0, true);     // every dense layer follows an S-box layer
            // The first layer once more for inputs that went through the S-box as PLAIN integers (leaf kernel: v + c instead of (v + c) 2^256 mod r,
            // no conversion product): bn29::pow5 then returns the state form's value times 2^-1280 (five missing factors 2^256), which columns 1..t-1 of
            // this copy take back.  c0p: the first round's constants as plain integers.
            {
                U256 f = { { 1, 0, 0, 0 } };
                for (int e = 0; e < 1280; e++) f = h_addmod(f, f);
                const U256 fm = h_to_mont(f);
                mfma_layer_tables(M, t, t, tm0, km0p, true, &fm);
                c0p.resize((size_t)t);
                for (int i = 0; i < t; i++) c0p[i] = h_mont(C8[i], U256{ { 1, 0, 0, 0 } });
            }
            mfma_layer_tables(D, n, n, td, kd, false);    // the closing layer reads the columns the blocks left
            mfma_partial_tables(t, rp, S, V, W, M[0], tpr, kr, ku);
            // What follows a layer is added by its row constants (the values in between are lazy representatives, no other addition
            // is left): the next full round's constants C8; after the fourth full round S[0] on element 0; after the closing layer
            // C8[4] on elements 1..n -- element 0 gets C8[4][0] from the last partial round's row when that round is one of the
            // blocked ones (rp % 4 == 0), from the vector code otherwise.  MK: one set per dense layer of the permutation, 8 x t.
            km.resize((size_t)8 * t);
            const bool nofold = getenv("PIL2GL_BN128_NOFOLD") && atoi(getenv("PIL2GL_BN128_NOFOLD"));
            for (int inst = 0; inst < 8; inst++) for (int i = 0; i < t; i++) {
                U256 f = { { 0, 0, 0, 0 } };
                if (nofold) {}
                else if (inst == 3) { if (i == 0) f = S[0]; }
                else if (inst < 7) f = C8[(size_t)(inst + 1) * t + i];
                km[(size_t)inst * t + i] = h_addmod(km0[i], f);
            }
            for (int i = 0; i < t; i++) km0p[i] = h_addmod(km0p[i], C8[(size_t)t + i]);      // (used without PIL2GL_BN128_NOFOLD only)
            if (!nofold) for (int i = 0; i < n; i++) kd[i] = h_addmod(kd[i], C8[(size_t)4 * t + 1 + i]);
            if (!nofold && rp % 4 == 0 && rp >= 4) kr[(size_t)rp - 1] = h_addmod(kr[(size_t)rp - 1], C8[(size_t)4 * t]);
            // Small widths (t <= BN_SMALL_T): poseidon.circom:22-44 as written, every round one t x t layer of the SAME matrix -- two tile sets (after a full
            // round every column comes out of the S-box, after a partial round only column 0) that stay in registers, and one row constant per round and row
            // (the layer's own + the next round's constants).  No tile stream, no sparse blocks: a width-3 permutation is a chain of 65 short rounds whose
            // latency, not its work, was the cost (perm_small).
            std::vector<int8_t> ts; Vec sk;
            if (t <= BN_SMALL_T) {
                const MfmaConsts mc;
                const U256 zero = { { 0, 0, 0, 0 } };
                const int R = N_ROUNDS_F + rp;
                ts.assign((size_t)2 * t * t * 1024, 0);
                Vec kset[2]; kset[0].resize((size_t)t); kset[1].resize((size_t)t);
                for (int set = 0; set < 2; set++)
                    for (int i = 0; i < t; i++) {
                        U256 tot = zero;
                        for (int j = 0; j < t; j++) mfma_tile(mc, M[(size_t)i * t + j], ts.data() + ((size_t)(set * t + i) * t + j) * 1024, tot, set == 0 || j == 0);
                        kset[set][i] = mfma_row_const(mc, tot, 1, zero);
                    }
                sk.resize((size_t)R * t);
                for (int r = 0; r < R; r++) {
                    const bool full = r < N_ROUNDS_F / 2 || r >= N_ROUNDS_F / 2 + rp;
                    for (int i = 0; i < t; i++) sk[(size_t)r * t + i] = h_addmod(kset[full ? 0 : 1][i], r + 1 < R ? C[(size_t)(r + 1) * t + i] : zero);
                }
            }
            const size_t spare = 16 * 1024;          // the tiles the read-ahead touches past the end of a table (MFMA_AHEAD)
            int8_t *dt = nullptr; u32 *dk = nullptr;
            HIP_TRY(hipMalloc((void **)&dt, tm.size() + td.size() + tpr.size() + tm0.size() + ts.size() + 5 * spare));
            HIP_TRY(hipMemset(dt, 0, tm.size() + td.size() + tpr.size() + tm0.size() + ts.size() + 5 * spare));
            if (!ts.empty()) HIP_TRY(hipMemcpy(dt + tm.size() + td.size() + tpr.size() + tm0.size() + 4 * spare, ts.data(), ts.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dt + tm.size() + td.size() + tpr.size() + 3 * spare, tm0.data(), tm0.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dt, tm.data(), tm.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dt + tm.size() + spare, td.data(), td.size(), hipMemcpyHostToDevice));
            if (!tpr.empty()) HIP_TRY(hipMemcpy(dt + tm.size() + td.size() + 2 * spare, tpr.data(), tpr.size(), hipMemcpyHostToDevice));
            Vec kall;
            kall.insert(kall.end(), km.begin(), km.end()); kall.insert(kall.end(), kd.begin(), kd.end());
            kall.insert(kall.end(), kr.begin(), kr.end()); kall.insert(kall.end(), ku.begin(), ku.end());
            kall.insert(kall.end(), km0p.begin(), km0p.end()); kall.insert(kall.end(), c0p.begin(), c0p.end());
            kall.insert(kall.end(), sk.begin(), sk.end());
            HIP_TRY(hipMalloc((void **)&dk, kall.size() * 32));
            HIP_TRY(hipMemcpy(dk, kall.data(), kall.size() * 32, hipMemcpyHostToDevice));
            P.Mt = (const bnm::v4i *)dt; P.Dt = (const bnm::v4i *)(dt + tm.size() + spare); P.Pt = (const bnm::v4i *)(dt + tm.size() + td.size() + 2 * spare);
            P.MK = dk; P.DK = dk + km.size() * 8; P.KR = P.DK + kd.size() * 8; P.KU = P.KR + kr.size() * 8;
            P.Mt0 = (const bnm::v4i *)(dt + tm.size() + td.size() + tpr.size() + 3 * spare); P.MK0 = P.KU + ku.size() * 8; P.C0p = P.MK0 + km0p.size() * 8;
            if (!ts.empty()) { P.St = (const bnm::v4i *)(dt + tm.size() + td.size() + tpr.size() + tm0.size() + 4 * spare); P.SK = P.C0p + c0p.size() * 8; }
        }
        P.rp = rp; P.t = t;
    }
    *out = &P;
    return PIL2GL_OK;
}

// ------------------------------------------------------------------------------------------ device side
// -DBN_STAMPS: a diagnostic build that sums, per phase of the matrix-core permutation, the shader cycles a wave spends in it
// (s_memtime around each phase, lane 0 adds into g_bn_stamps at the end of a permutation; pil2gl_bn128_debug_stamps reads them).
// No stamp executes in the product build.
#ifdef BN_STAMPS
__device__ unsigned long long g_bn_stamps[16];
__device__ __forceinline__ unsigned long long bn_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define BN_STAMP(slot, expr) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t0_ = bn_now(); __builtin_amdgcn_sched_barrier(0); expr; \
    __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = bn_now(); __builtin_amdgcn_sched_barrier(0); if (st.lane == 0) atomicAdd(&g_bn_stamps[slot], t1_ - t0_); }
#else
#define BN_STAMP(slot, expr) { expr; }
#endif
// -DBN_PRIO_MFMA=n: the wave raises its issue priority to n while it feeds the matrix pipe (dense layers, rows on y, column updates) and drops it
// to 0 for its vector phases, so that of a SIMD's two waves the one in a matrix phase is served first (A/B builds; 0 = no s_setprio at all)
#ifndef BN_PRIO_MFMA
#define BN_PRIO_MFMA 0
#endif
#if BN_PRIO_MFMA
#define BN_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define BN_PRIO(n)
#endif
struct PermArgs { const u32 *C8, *M, *D, *S, *V, *W, *Cd; int t, rp, dense; u32 m00[8];
                  const bnm::v4i *Mt, *Dt, *Pt; const u32 *MK, *DK, *KR, *KU; int mfma, nofold;
                  const bnm::v4i *Mt0; const u32 *MK0, *C0p; int plain;
                  int nout1;         // nout1: the caller reads element 0 of the result only (sponges, tree nodes): the last layer computes ONE row
                  const bnm::v4i *St; const u32 *SK; int small_; };      // small_: the width runs perm_small      // plain: elements 1..t-1 arrive S-boxed from the absorb (leaf kernel, width 17)

// Where the state lives.  Elements [0, BN_LDS_ELEMS) in LDS as [element][limb][lane]; the elements above -- only the states
// wider than BN_LDS_ELEMS have any: t = 11..17 -- in the lane's own private (scratch) memory, which the hardware swizzles so
// that a wave's access to one word is one contiguous 256-byte piece.  10 elements = 20 KB of LDS per wave: EIGHT waves per CU
// (two per SIMD: all 160 KB) instead of the four that a 17-element state in LDS allows.  The private half is the kernel's costliest
// traffic (the per-XCD working set of 256 waves' private state plus the tile table passes the 4 MB L2: it travels through the fabric,
// and the card is power-limited): every phase touches it as few times as its registers allow, and never inside a loop that runs a
// tile ring (partial_rounds_mfma_impl).  The element index is static wherever that matters.
typedef u32 __attribute__((address_space(5))) *priv_u32;
typedef u32 __attribute__((address_space(3))) *lds_u32;
struct St { lds_u32 S; priv_u32 hi; int tmax, lane; };
__host__ __device__ constexpr int lds_words(int tmax) { return (tmax < BN_LDS_ELEMS ? tmax : BN_LDS_ELEMS) * 8 * BN_BLOCK; }   // one wave's slice
#define S_LDS(st, j, l) (st).S[(((j) * 8 + (l)) * BN_BLOCK) + (st).lane]

__device__ __forceinline__ void lds_load(const St st, int j, u32 x[8]) {
#ifdef BN_ABLATE_SCRATCH
    if (j >= BN_LDS_ELEMS) j -= 17 - BN_LDS_ELEMS;   // timing experiments only: the upper elements aliased onto LDS slots, results meaningless
#endif
    if (j < BN_LDS_ELEMS) {
#pragma unroll
        for (int l = 0; l < 8; l++) x[l] = S_LDS(st, j, l);
    } else {
        priv_u32 hp = st.hi;
        asm volatile("" : "+v"(hp));                  // opaque: ONE address register and immediate offsets (hipcc would hoist an address per element into registers of its own)
#pragma unroll
        for (int l = 0; l < 8; l++) x[l] = hp[(j - BN_LDS_ELEMS) * 8 + l];
    }
}
__device__ __forceinline__ void lds_store(const St st, int j, const u32 x[8]) {
#ifdef BN_ABLATE_SCRATCH
    if (j >= BN_LDS_ELEMS) j -= 17 - BN_LDS_ELEMS;
#endif
    if (j < BN_LDS_ELEMS) {
#pragma unroll
        for (int l = 0; l < 8; l++) S_LDS(st, j, l) = x[l];
    } else {
        priv_u32 hp = st.hi;
        asm volatile("" : "+v"(hp));
#pragma unroll
        for (int l = 0; l < 8; l++) hp[(j - BN_LDS_ELEMS) * 8 + l] = x[l];
    }
}
// An element kept in OPERAND form (the two v4i of bnm::b_prep: bytes signed, halves swapped between the wave's halves) in its usual slot: the y
// of the partial rounds live so from the layer before them to the layer after them -- every rows' pass and column update then LOADS its operand
// where it prepared it (twelve instructions per column and pass: 704 preparations per width-17 permutation become 160)
__device__ __forceinline__ void lds_load_b(const St st, int j, bnm::v4i &b0, bnm::v4i &b1) {
    u32 x[8];
    lds_load(st, j, x);
#pragma unroll
    for (int q = 0; q < 4; q++) { b0[q] = (int)x[q]; b1[q] = (int)x[4 + q]; }
}
__device__ __forceinline__ void lds_store_b(const St st, int j, const bnm::v4i &b0, const bnm::v4i &b1) {
    u32 x[8];
#pragma unroll
    for (int q = 0; q < 4; q++) { x[q] = (u32)b0[q]; x[4 + q] = (u32)b1[q]; }
    lds_store(st, j, x);
}
// wave-uniform address.  WIDE: the tables live in global memory; saying so (the pointers reach the out-of-line helpers as
// generic ones) turns the flat loads into global loads, whose counter is separate from the LDS one and in order, so that a
// request for the next term can stay in flight across the current multiply
typedef const u32 __attribute__((address_space(1))) *gconst_u32;
template <bool WIDE>
__device__ __forceinline__ void load_const(const u32 *p, size_t idx, u32 c[8]) {
    if constexpr (WIDE) {
        gconst_u32 q = (gconst_u32)(p + idx * 8);
#pragma unroll
        for (int l = 0; l < 8; l++) c[l] = q[l];
    } else {
#pragma unroll
        for (int l = 0; l < 8; l++) c[l] = p[idx * 8 + l];
    }
}
__device__ __forceinline__ void pow5(u32 x[8]) {
    u32 x2[8], x4[8];
    bn::fr_mul(x2, x, x); bn::fr_mul(x4, x2, x2); bn::fr_mul(x, x4, x);
}
// the same on lazy representatives: x < 0.69 * 2^256 in (a layer's output < 2^255, plus a round constant at most), x^5 < 0.63 * 2^256
// out -- any 256-bit value will do for the matrix operand that reads it (bn_field.cuh fr_mul_nr)
__device__ __forceinline__ void pow5_lazy(u32 x[8]) {
#ifdef BN_ABLATE_SBOX
    return;                                          // timing experiments only (tools): the S-box left out, results meaningless
#endif
#if BN_SBOX29
    bn29::pow5(x);                                   // x^5 in the state's form times 2^-20 (the next layer's tiles carry 2^20)
#else
    u32 x2[8], x4[8];
    bn::fr_mul_nr(x2, x, x); bn::fr_mul_nr(x4, x2, x2); bn::fr_mul_nr(x, x4, x);
#endif
}
// x + c for a lazy x < 2^255 and a constant c < r: < 0.69 * 2^256, left as it is (the S-box that follows takes it, pow5_lazy)
__device__ __forceinline__ void add_lazy(u32 x[8], const u32 c[8]) { bnm::add_chain8(x, c); }
// The S-box layer of a full round in the matrix-core pipeline: the round's constants arrive with the previous layer's rows
// (except the first round's, C != nullptr); lazy in, lazy out
// (element j + 1 is requested before element j is worked on: more than half of a wide state lives in private memory, whose L2 latency --
// a microsecond -- would otherwise be paid in full by every element; the same in the two loops of partial_rounds_mfma)
__device__ __forceinline__ void sbox_lazy_impl(const St st, int t, const u32 *C) {
    u32 xn[8];
    lds_load(st, 0, xn);
    for (int j = 0; j < t; j++) {
        u32 x[8];
#pragma unroll
        for (int l = 0; l < 8; l++) x[l] = xn[l];
        if (j + 1 < t) lds_load(st, j + 1, xn);
        if (C) {
            u32 c[8];
            load_const<true>(C, (size_t)j, c);
            add_lazy(x, c);
        }
        pow5_lazy(x);
        lds_store(st, j, x);
    }
}
__device__ __noinline__ void sbox_lazy(const St st, int t, const u32 *C) { sbox_lazy_impl(st, t, C); }
__device__ __noinline__ void canon_state(const St st, int t) {
    for (int j = 0; j < t; j++) {
        u32 x[8];
        lds_load(st, j, x);
        bnm::canon(x);
        lds_store(st, j, x);
    }
}

// x^5 on elements [0, nSbox) after adding constants C[0..t), then the dense n x n matrix A applied to elements
// [first, first+n) of buffer cur into buffer cur^1 (elements below `first` are copied)
template <bool WIDE>
__device__ __noinline__ void add_sbox(const St st, int cur, int t, const u32 *C, size_t cOff, int nSbox) {
    for (int j = 0; j < t; j++) {
        u32 x[8], c[8];
        lds_load(st, j, x);
        load_const<WIDE>(C, cOff + j, c);
        bn::fr_add(x, c);
        if (j < nSbox) pow5(x);
        lds_store(st, j, x);
    }
}
// In place: every row reads the whole old state, so the n new elements wait in a per-lane private array (scratch memory,
// 17 x 32 B, a few KB of traffic per permutation against ~10^5 multiply steps) until all rows are done; one LDS buffer per
// wave then suffices (4 waves per CU at t = 17 instead of 2).
template <bool WIDE>
__device__ __noinline__ void dense_mul(const St st, int cur, const u32 *A, int n, int first) {
    u32 nw[17 * 8];
    for (int i = 0; i < n; i++) {
        u32 acc[17];
#pragma unroll
        for (int l = 0; l < 17; l++) acc[l] = 0;
        if constexpr (WIDE) {
            // operands of term j+1 are requested before term j is multiplied: the LDS read and the (wave-uniform) table
            // load then overlap the ~600 issue cycles of a multiply-accumulate instead of stalling the wave, which at this
            // width is alone on its SIMD.  One multiply in the loop body: the permutation has to stay in the instruction cache.
            u32 y[8], m[8];
            lds_load(st, first, y);
            load_const<true>(A, (size_t)i * n, m);
            for (int j = 0; j < n; j++) {
                u32 yn[8], mn[8];
                if (j + 1 < n) {
                    lds_load(st, first + j + 1, yn);
                    load_const<true>(A, (size_t)i * n + j + 1, mn);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the multiply (the scheduler sinks them otherwise)
                bn::mac17(acc, y, m);
#pragma unroll
                for (int l = 0; l < 8; l++) { y[l] = yn[l]; m[l] = mn[l]; }
            }
        } else {
            for (int j = 0; j < n; j++) {
                u32 y[8], m[8];
                lds_load(st, first + j, y);
                load_const<false>(A, (size_t)i * n + j, m);
                bn::mac17(acc, y, m);
            }
        }
        u32 o[8];
        bn::redc17(o, acc);
#pragma unroll
        for (int l = 0; l < 8; l++) nw[i * 8 + l] = o[l];
    }
    for (int i = 0; i < n; i++) lds_store(st, first + i, &nw[i * 8]);
}

// The same layer on the matrix cores (bn_mfma.cuh): the N operand pairs are made once and stay in registers, a row is N pairs of
// MFMAs and one short finish; no 32x32 product of the state is left.  The operand tiles come from the L2 as ONE linear stream
// (row after row, tile after tile) read MFMA_AHEAD tiles ahead of their use -- a load per tile and lane, the oldest awaited
// alone -- so the table carries MFMA_AHEAD spare tiles after its last one.
constexpr int MFMA_AHEAD = 8;
#ifndef BN_DENSE_AHEAD_SBOX
#define BN_DENSE_AHEAD_SBOX 6
#endif
// SBOX: the NEXT round's S-box is applied to every finished row before it is stored (its constant came with the row): the separate S-box pass over
// the state -- a load and a store of every element, seven of them in private memory -- disappears for that round.
// STORE_B: the rows of elements 1.. are stored in operand form (the layer before the partial rounds); LOAD_B: the inputs are in that form (the layer after).
// nrows: the leading rows that are computed (the permutation's last layer when only element 0 of the result is read: one of N)
template <int N, bool SBOX = false, bool STORE_B = false, bool LOAD_B = false>
__device__ __forceinline__ void dense_mfma_impl(const St st, const bnm::v4i *tiles, const u32 *kc, int first, int nrows = N) {
    bnm::v4i B0[N], B1[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        if constexpr (LOAD_B) lds_load_b(st, first + j, B0[j], B1[j]);
        else {
            u32 x[8];
            lds_load(st, first + j, x);
            bnm::b_prep(x, B0[j], B1[j]);
        }
    }
    const bnm::Sh sh = bnm::sh_init();
    bnm::gtile tp = (bnm::gtile)tiles + st.lane;
    constexpr int AHEAD = SBOX ? BN_DENSE_AHEAD_SBOX : MFMA_AHEAD;      // (the S-box in the row loop needs some of the ring's registers: 4 / 6 / 8 tiles 27.15 / 26.9 / 26.9 ms)
    bnm::v4i q[AHEAD];
    BN_SYNC();                                       // the workgroup's waves start the layer's tile stream together
#pragma unroll
    for (int k = 0; k < AHEAD; k++) q[k] = tp[(size_t)k * 64];
#ifdef BN_STAMPS
    unsigned long long sBurst = 0, sCarry = 0, sFinish = 0;
#endif
    for (int i = 0; i < nrows; i++) {
        u32 k[8], o[8];
#ifdef BN_STAMPS
        __builtin_amdgcn_sched_barrier(0); const unsigned long long tr0 = bn_now(); __builtin_amdgcn_sched_barrier(0);
#endif
        load_const<true>(kc, (size_t)i, k);          // asked for ahead of the row's tiles: an in-order counter waits for everything older than what it wants
        bnm::v16i a0, a1;
        BN_PRIO(BN_PRIO_MFMA);
#pragma unroll
        for (int j = 0; j < N; j++) {
            const bnm::v4i a = q[0];
#pragma unroll
            for (int k = 0; k + 1 < AHEAD; k++) q[k] = q[k + 1];
#ifdef BN_ABLATE_TILEADDR
            q[AHEAD - 1] = ((bnm::gtile)tiles + st.lane)[(size_t)((j + AHEAD) & 7) * 64];      // timing experiments only: every tile from one 8 KB window
#else
            q[AHEAD - 1] = tp[(size_t)(j + AHEAD) * 64];
#endif
            if (j == 0) bnm::mfma_first(a, B0[0], B1[0], a0, a1);
            else {
                a0 = bnm::mfma(a, B0[j], a0);
                a1 = bnm::mfma(a, B1[j], a1);
            }
        }
        tp += (size_t)N * 64;
        BN_PRIO(0);
#ifdef BN_STAMPS
        // (the burst ends when the first accumulator can be read: the stamp after an instruction that depends on both)
        u32 w[10];
        __builtin_amdgcn_sched_barrier(0); const unsigned long long tr1 = bn_now(); __builtin_amdgcn_sched_barrier(0);
        bnm::carry_pair(a0, a1, w, sh);
        __builtin_amdgcn_sched_barrier(0); const unsigned long long tr2 = bn_now(); __builtin_amdgcn_sched_barrier(0);
        bnm::finish_words(w, k, o);
        if constexpr (SBOX) pow5_lazy(o);
        if (STORE_B && first + i >= 1) { bnm::v4i ob0, ob1; bnm::b_prep(o, ob0, ob1); lds_store_b(st, first + i, ob0, ob1); }
        else lds_store(st, first + i, o);
        __builtin_amdgcn_sched_barrier(0); const unsigned long long tr3 = bn_now(); __builtin_amdgcn_sched_barrier(0);
        sBurst += tr1 - tr0; sCarry += tr2 - tr1; sFinish += tr3 - tr2;
#else
        bnm::finish_row(a0, a1, k, o, sh);
        if constexpr (SBOX) pow5_lazy(o);
        if (STORE_B && first + i >= 1) { bnm::v4i ob0, ob1; bnm::b_prep(o, ob0, ob1); lds_store_b(st, first + i, ob0, ob1); }
        else lds_store(st, first + i, o);            // the old state is in B0 / B1: the new row can go straight to its place
#endif
    }
#ifdef BN_STAMPS
    if (st.lane == 0) { atomicAdd(&g_bn_stamps[13], sBurst); atomicAdd(&g_bn_stamps[14], sCarry); atomicAdd(&g_bn_stamps[15], sFinish); atomicAdd(&g_bn_stamps[6], (unsigned long long)nrows); }
#endif
}
template <int N>
__device__ __noinline__ void dense_mfma_n(const St st, const bnm::v4i *tiles, const u32 *kc, int first, int nrows) { dense_mfma_impl<N>(st, tiles, kc, first, nrows < N ? nrows : N); }
__device__ __forceinline__ void dense_mfma(const St st, const bnm::v4i *tiles, const u32 *kc, int n, int first, int nrows = 17) {
#ifdef BN_ABLATE_DENSE
    return;                                          // timing experiments only: the layer left out, results meaningless
#endif
    switch (n) {
    case 1: dense_mfma_n<1>(st, tiles, kc, first, nrows); break;
    case 2: dense_mfma_n<2>(st, tiles, kc, first, nrows); break;
    case 3: dense_mfma_n<3>(st, tiles, kc, first, nrows); break;
    case 4: dense_mfma_n<4>(st, tiles, kc, first, nrows); break;
    case 5: dense_mfma_n<5>(st, tiles, kc, first, nrows); break;
    case 6: dense_mfma_n<6>(st, tiles, kc, first, nrows); break;
    case 7: dense_mfma_n<7>(st, tiles, kc, first, nrows); break;
    case 8: dense_mfma_n<8>(st, tiles, kc, first, nrows); break;
    case 9: dense_mfma_n<9>(st, tiles, kc, first, nrows); break;
    case 10: dense_mfma_n<10>(st, tiles, kc, first, nrows); break;
    case 11: dense_mfma_n<11>(st, tiles, kc, first, nrows); break;
    case 12: dense_mfma_n<12>(st, tiles, kc, first, nrows); break;
    case 13: dense_mfma_n<13>(st, tiles, kc, first, nrows); break;
    case 14: dense_mfma_n<14>(st, tiles, kc, first, nrows); break;
    case 15: dense_mfma_n<15>(st, tiles, kc, first, nrows); break;
    case 16: dense_mfma_n<16>(st, tiles, kc, first, nrows); break;
    default: dense_mfma_n<17>(st, tiles, kc, first, nrows); break;
    }
}

// partial rounds, sparse form, in place: element 0 stays in registers.  WIDE: the next term's requests are pinned ahead of
// the current term's two products (see dense_mul)
template <bool WIDE>
__device__ __noinline__ void partial_rounds(const St st, int cur, const PermArgs &A, int kFirst) {
    const int t = A.t;
    u32 x0[8], m00[8];
    lds_load(st, 0, x0);
#pragma unroll
    for (int l = 0; l < 8; l++) m00[l] = A.m00[l];
    const int n = t - 1;
    for (int k = kFirst; k < A.rp; k++) {
        u32 c[8];
        load_const<WIDE>(A.S, (size_t)k, c);
        bn::fr_add(x0, c);
        pow5(x0);
        u32 acc[17];
#pragma unroll
        for (int l = 0; l < 17; l++) acc[l] = 0;
        bn::mac17(acc, x0, m00);
        if constexpr (WIDE) {
            u32 y[8], vv[8], ww[8];
            lds_load(st, 1, y);
            load_const<true>(A.V, (size_t)k * n, vv);
            load_const<true>(A.W, (size_t)k * n, ww);
            for (int j = 0; j < n; j++) {
                u32 yn[8], vn[8], wn[8], p[8];
                if (j + 1 < n) {
                    lds_load(st, 2 + j, yn);
                    load_const<true>(A.V, (size_t)k * n + j + 1, vn);
                    load_const<true>(A.W, (size_t)k * n + j + 1, wn);
                }
                __builtin_amdgcn_sched_barrier(0);
                bn::mac17_and_fr_mul(acc, y, vv, p, x0, ww);   // row 0: m00*x0 + sum V_kj * y_j;  column: y_j + W_kj * x0
                bn::fr_add(y, p);
                lds_store(st, 1 + j, y);
#pragma unroll
                for (int l = 0; l < 8; l++) { y[l] = yn[l]; vv[l] = vn[l]; ww[l] = wn[l]; }
            }
        } else {
            for (int j = 0; j < n; j++) {
                u32 y[8], vv[8], ww[8], p[8];
                lds_load(st, 1 + j, y);
                load_const<false>(A.V, (size_t)k * n + j, vv);
                bn::mac17(acc, y, vv);               // row 0:   m00*x0 + sum V_kj * y_j
                load_const<false>(A.W, (size_t)k * n + j, ww);
                bn::fr_mul(p, x0, ww);               // column:  y_j + W_kj * x0
                bn::fr_add(y, p);
                lds_store(st, 1 + j, y);
            }
        }
        bn::redc17(x0, acc);
    }
    lds_store(st, 0, x0);
}

// The partial rounds on the matrix cores, four to a block, two blocks to a super-block (tables: mfma_partial_tables).  Per block: the four
// rows' parts on y (n x 4 pairs of MFMAs -- for a super-block's second block on the y of the super-block's START, plus 16 pairs on the first
// block's S-box outputs -- carried to ten words each), then the four rounds: S-box on the vector ALU, its output z_i made an operand, the
// cross terms of row i (<= 4 pairs) added to the row's stored part, one short finish = the next x0.  Per super-block, once: the n columns
// y_j + sum_k W z_k (1 + 8 pairs and one finish each -- the costliest phase, hence every eight rounds, not four).  No 32x32 product is left
// but the S-box's.  The tiles are ONE linear stream in consumption order, read PR_AHEAD tiles ahead.
#ifndef BN_KR_LATE
#define BN_KR_LATE 0
#endif
#ifndef BN_PR_AHEAD
#define BN_PR_AHEAD 4
#endif
constexpr int PR_AHEAD = BN_PR_AHEAD;
struct TileStream {
    bnm::gtile p;
    bnm::v4i q[PR_AHEAD];
#ifdef BN_ABLATE_TILEADDR
    bnm::gtile base; unsigned cnt = 0;
#endif
    __device__ __forceinline__ void start(const bnm::v4i *tiles, int lane) {
        p = (bnm::gtile)tiles + lane;
#ifdef BN_ABLATE_TILEADDR
        base = p;
#endif
#pragma unroll
        for (int k = 0; k < PR_AHEAD; k++) q[k] = p[(size_t)k * 64];
    }
    // (the request for tile k + PR_AHEAD stays where tile k is taken: left to itself hipcc's scheduler, short of registers, sinks every request to
    // just before its use and the ring is one or two tiles deep)
    __device__ __forceinline__ bnm::v4i next() {
        const bnm::v4i a = q[0];
#pragma unroll
        for (int k = 0; k + 1 < PR_AHEAD; k++) q[k] = q[k + 1];
        __builtin_amdgcn_sched_barrier(0);
#ifdef BN_ABLATE_TILEADDR
        q[PR_AHEAD - 1] = base[(size_t)(cnt++ & 7) * 64];
#else
        q[PR_AHEAD - 1] = p[(size_t)PR_AHEAD * 64];
#endif
        __builtin_amdgcn_sched_barrier(0);
        p += 64;
        return a;
    }
};
// N = t - 1 columns.  Everything that indexes the state is unrolled, so that which columns live in LDS (elements below BN_LDS_ELEMS) and which in
// private memory is known statically, and NO private-memory access sits inside a loop that runs the tile ring: hipcc answers a mix of scratch and
// global accesses in flight -- or a branch around one -- with `s_waitcnt vmcnt(0)`, which drains the ring on every column and exposes the L2
// latency of every tile (round 5's form: 2 465 cycles per column of eight matrix instructions).  The upper columns are therefore fetched in ONE batch:
// before the rows' pass as matrix operands (kept for both half-passes), before / after the column update as words.  The rows on y are taken two at
// a time (two passes over the columns: four accumulators instead of eight leave the registers for the batch; the lower columns are read from LDS twice).
// BFORM: the y (elements 1..N) arrive, live and leave in operand form (lds_load_b): the fast path of the width-17 permutation.
template <int N, bool BFORM = false>
__device__ __forceinline__ void partial_rounds_mfma_impl(const St st, const bnm::v4i *Pt, const u32 *KR, const u32 *KU, int rp) {
    constexpr int NLO = N + 1 <= BN_LDS_ELEMS ? N : BN_LDS_ELEMS - 1;     // columns j whose element 1 + j lives in LDS
    constexpr int NHI = N - NLO, NHA = NHI ? NHI : 1;
    const int nb = rp / 4, nsb = (nb + 1) / 2;
    const bnm::Sh sh = bnm::sh_init();
    TileStream ts;
    BN_SYNC();
    ts.start(Pt, st.lane);
    // x0 (S[0] came with the row of the layer before) stays in its LDS slot outside the rounds: the rows' pass and the columns need the registers
    // BFORM: the upper columns as the column update leaves them (operand form) stay in registers into the first rows' pass of the NEXT super-block
    // (the update's own operands are dead by then): that pass reads no private memory at all
    u32 yh[NHA][8];
    if constexpr (BFORM) {
#pragma unroll
        for (int q = 0; q < NHI; q++) lds_load(st, 1 + NLO + q, yh[q]);
    }
    for (int sb = 0; sb < nsb; sb++) {
        const int halves = nb - 2 * sb >= 2 ? 2 : 1;
        bnm::v4i zbA0[4], zbA1[4];                    // the first block's z operands, for the second block's rows and the column update
        bnm::v4i zb0[4], zb1[4];                      // the current block's: z_(i-3) .. z_i
#pragma unroll
        for (int s = 0; s < 4; s++) { zbA0[s] = bnm::v4i{ 0, 0, 0, 0 }; zbA1[s] = bnm::v4i{ 0, 0, 0, 0 }; }
        u32 pc[4][10];
        // the four rows' parts on y of one block; H = 1: the second block of a super-block (two instances: no branch inside the ring's straight line)
        auto rows = [&](auto Hc) {
            constexpr int H = decltype(Hc)::value;
            BN_SYNC();
            BN_PRIO(BN_PRIO_MFMA);
            // The upper columns as operands, in at most two batches of private-memory elements fetched just before their columns (all seven kept through a
            // pass do not fit beside the accumulators: hipcc spilled them, ~700 spill stores per wave).  The SECOND batch (columns SPL.., at most
            // BN_HI_BATCH of them) is still in its registers when the second pass begins: that pass takes it first, re-reads only the first batch,
            // and ends on the lower columns (mfma_partial_tables writes the tiles in this order).
            constexpr int HB2 = NHI < BN_HI_BATCH ? NHI : BN_HI_BATCH, SPL = NHI - HB2, HBA = HB2 ? HB2 : 1;
            static_assert(SPL <= HBA, "two batches of BN_HI_BATCH elements must cover the upper columns");
            bnm::v4i hb0[HBA], hb1[HBA];
            auto fetch = [&](int q0, int cnt) {
#pragma unroll
                for (int e = 0; e < HBA; e++)
                    if (e < cnt) {
                        if constexpr (BFORM) lds_load_b(st, 1 + NLO + q0 + e, hb0[e], hb1[e]);
                        else {
                            u32 y[8];
                            lds_load(st, 1 + NLO + q0 + e, y);
                            bnm::b_prep(y, hb0[e], hb1[e]);
                        }
                    }
            };
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                bnm::v16i P0[2], P1[2];
                bool first = true;
                auto products = [&](const bnm::v4i &b0, const bnm::v4i &b1) {
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const bnm::v4i a = ts.next();
                        if (first) bnm::mfma_first(a, b0, b1, P0[r], P1[r]);
                        else {
                            P0[r] = bnm::mfma(a, b0, P0[r]);
                            P1[r] = bnm::mfma(a, b1, P1[r]);
                        }
                    }
                    first = false;
                };
                auto lower = [&]() {
                    u32 yn[8];
                    if (NLO) lds_load(st, 1, yn);
#pragma unroll
                    for (int j = 0; j < NLO; j++) {
                        bnm::v4i b0, b1;
                        u32 y[8];
#pragma unroll
                        for (int l = 0; l < 8; l++) y[l] = yn[l];
                        if (j + 1 < NLO) lds_load(st, 2 + j, yn);         // the next column's words are on their way while this one's products run
                        if constexpr (BFORM) {
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) { b0[q4] = (int)y[q4]; b1[q4] = (int)y[4 + q4]; }
                        } else bnm::b_prep(y, b0, b1);
                        products(b0, b1);
                    }
                };
                auto upper = [&](int q0, int cnt) {
#pragma unroll
                    for (int e = 0; e < HBA; e++)
                        if (e < cnt) products(hb0[e], hb1[e]);
                };
                auto carried = [&](int q0, int cnt) {                  // the upper columns the last column update left in registers
#pragma unroll
                    for (int q = 0; q < NHA; q++)
                        if (q >= q0 && q < q0 + cnt) {
                            bnm::v4i b0, b1;
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) { b0[q4] = (int)yh[q][q4]; b1[q4] = (int)yh[q][4 + q4]; }
                            products(b0, b1);
                        }
                };
                constexpr bool CARRIED = BFORM && H == 0;
                if (pass == 0) {
                    lower();
                    if constexpr (CARRIED) carried(0, NHI);
                    else {
                        if (SPL) { fetch(0, SPL); upper(0, SPL); }
                        if (HB2) { fetch(SPL, HB2); upper(SPL, HB2); }
                    }
                } else {
                    if constexpr (CARRIED) { carried(SPL, HB2); carried(0, SPL); }
                    else {
                        if (HB2) upper(SPL, HB2);                      // (in registers since the first pass)
                        if (SPL) { fetch(0, SPL); upper(0, SPL); }
                    }
                    lower();
                }
                if constexpr (H == 1) {                               // the rows of the second block see the first block's z through cross terms of their own
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            const bnm::v4i a = ts.next();
                            P0[r] = bnm::mfma(a, zbA0[s], P0[r]);
                            P1[r] = bnm::mfma(a, zbA1[s], P1[r]);
                        }
                }
#pragma unroll
                for (int r = 0; r < 2; r++) bnm::carry_pair(P0[r], P1[r], pc[2 * pass + r], sh);
            }
            BN_PRIO(0);
        };
#ifdef BN_STAMPS
        unsigned long long tp0 = bn_now();
#endif
        rows(std::integral_constant<int, 0>{});          // (outside the loop over the blocks: the carried columns must not be live around its back edge)
        for (int h = 0; h < halves; h++) {
            const int b = 2 * sb + h;
            if (h == 1) {
#ifdef BN_STAMPS
                tp0 = bn_now();
#endif
#pragma unroll
                for (int s = 0; s < 4; s++) { zbA0[s] = zb0[s]; zbA1[s] = zb1[s]; }
                rows(std::integral_constant<int, 1>{});
            }
#ifdef BN_STAMPS
            { const unsigned long long t_ = bn_now(); if (st.lane == 0) atomicAdd(&g_bn_stamps[3], t_ - tp0); tp0 = t_; }
#endif
            u32 x0[8];
            lds_load(st, 0, x0);
            // the four rounds, unrolled: round i's cross terms are i + 1 tiles (z_0 .. z_i of this block), its row's part pc[i] and its operand slot
            // are static -- no zero tiles multiplied, no operand or row shifted along
#pragma unroll
            for (int i = 0; i < 4; i++) {
                u32 k[8];
#if !BN_KR_LATE
                load_const<true>(KR, (size_t)(4 * b + i), k);         // (long before its use: the S-box hides it)
#endif
                pow5_lazy(x0);
#if BN_KR_LATE
                load_const<true>(KR, (size_t)(4 * b + i), k);         // (behind the S-box, whose registers it would otherwise take: the cross terms' products hide it)
#endif
                bnm::b_prep(x0, zb0[i], zb1[i]);
                bnm::v16i c0, c1;
#pragma unroll
                for (int s = 0; s <= i; s++) {
                    const bnm::v4i a = ts.next();
                    if (s == 0) bnm::mfma_first(a, zb0[0], zb1[0], c0, c1);
                    else {
                        c0 = bnm::mfma(a, zb0[s], c0);
                        c1 = bnm::mfma(a, zb1[s], c1);
                    }
                }
                u32 w[10];
                bnm::carry_pair(c0, c1, w, sh);
                bnm::add_pair(w, pc[i]);
                bnm::finish_words(w, k, x0);
            }
            lds_store(st, 0, x0);
#ifdef BN_STAMPS
            { const unsigned long long t_ = bn_now(); if (st.lane == 0) atomicAdd(&g_bn_stamps[4], t_ - tp0); }
#endif
        }
#ifdef BN_STAMPS
        unsigned long long tu0 = bn_now();
#endif
        // the columns, once per super-block: y_j + sum over its rounds of W z -- 1 + 8 tiles and one finish each (a last super-block of one block
        // has four zero tiles for the first block's operands, which are zero: ONE form of the column, mfma_partial_tables)
        auto column = [&](u32 y[8], int j) {
            u32 k[8];
            load_const<true>(KU, (size_t)sb * N + j, k);              // (asked for early: used after the products)
            bnm::v4i b0, b1;
            if constexpr (BFORM) {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) { b0[q4] = (int)y[q4]; b1[q4] = (int)y[4 + q4]; }
            } else bnm::b_prep(y, b0, b1);
            BN_PRIO(BN_PRIO_MFMA);
            bnm::v4i a = ts.next();
            bnm::v16i c0, c1;
            bnm::mfma_first(a, b0, b1, c0, c1);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                a = ts.next();
                c0 = bnm::mfma(a, zbA0[s], c0);
                c1 = bnm::mfma(a, zbA1[s], c1);
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                a = ts.next();
                c0 = bnm::mfma(a, zb0[s], c0);
                c1 = bnm::mfma(a, zb1[s], c1);
            }
            BN_PRIO(0);
            bnm::finish_row(c0, c1, k, y, sh);
            if constexpr (BFORM) {                                    // back to the form it is kept in
                bnm::v4i nb0, nb1;
                bnm::b_prep(y, nb0, nb1);
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) { y[q4] = (u32)nb0[q4]; y[4 + q4] = (u32)nb1[q4]; }
            }
        };
        if (halves == 1) {
#pragma unroll
            for (int s = 0; s < 4; s++) { zbA0[s] = bnm::v4i{ 0, 0, 0, 0 }; zbA1[s] = bnm::v4i{ 0, 0, 0, 0 }; }
        }
        BN_SYNC();
        if (NLO) {                                                    // the lower columns: a loop (LDS takes a run-time index; the ring turns by three tiles per column)
            u32 yn[8];
            lds_load(st, 1, yn);
#pragma unroll 1
            for (int j = 0; j < NLO; j++) {
                u32 y[8];
#pragma unroll
                for (int l = 0; l < 8; l++) y[l] = yn[l];
                if (j + 1 < NLO) lds_load(st, 2 + j, yn);
                column(y, j);
                lds_store(st, 1 + j, y);
            }
        }
        {                                                             // the upper columns: one batch in, one batch out (and kept: see above)
#pragma unroll
            for (int q = 0; q < NHI; q++) lds_load(st, 1 + NLO + q, yh[q]);
#pragma unroll
            for (int q = 0; q < NHI; q++) column(yh[q], NLO + q);
#pragma unroll
            for (int q = 0; q < NHI; q++) lds_store(st, 1 + NLO + q, yh[q]);
        }
#ifdef BN_STAMPS
        { const unsigned long long t_ = bn_now(); if (st.lane == 0) atomicAdd(&g_bn_stamps[5], t_ - tu0); }
#endif
    }
}
template <int N>
__device__ __noinline__ void partial_rounds_mfma_n(const St st, const bnm::v4i *Pt, const u32 *KR, const u32 *KU, int rp) { partial_rounds_mfma_impl<N>(st, Pt, KR, KU, rp); }
__device__ __forceinline__ void partial_rounds_mfma(const St st, const PermArgs &A) {
    switch (A.t - 1) {
    case 1: partial_rounds_mfma_n<1>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 2: partial_rounds_mfma_n<2>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 3: partial_rounds_mfma_n<3>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 4: partial_rounds_mfma_n<4>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 5: partial_rounds_mfma_n<5>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 6: partial_rounds_mfma_n<6>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 7: partial_rounds_mfma_n<7>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 8: partial_rounds_mfma_n<8>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 9: partial_rounds_mfma_n<9>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 10: partial_rounds_mfma_n<10>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 11: partial_rounds_mfma_n<11>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 12: partial_rounds_mfma_n<12>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 13: partial_rounds_mfma_n<13>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 14: partial_rounds_mfma_n<14>(st, A.Pt, A.KR, A.KU, A.rp); break;
    case 15: partial_rounds_mfma_n<15>(st, A.Pt, A.KR, A.KU, A.rp); break;
    default: partial_rounds_mfma_n<16>(st, A.Pt, A.KR, A.KU, A.rp); break;
    }
}

// Widths up to BN_SMALL_T: the permutation round by round (poseidon.circom:22-44 as written) with the state, the layer's T x T tiles and the operands in
// REGISTERS: every round adds its constants (they arrive with the previous layer's rows), takes the S-box (all elements / element 0) and multiplies by the
// same matrix.  Two tile sets (which columns carry the S-box's 2^-20), swapped twice per permutation; nothing streams, nothing goes through LDS between
// the rounds.  The blocked pipeline is built for seventeen elements: at three its phases are a few matrix instructions each and the wave spends its time
// between them (a width-3 permutation took 1.08 M cycles per wave at an UNCAPPED 2.38 GHz, twice its S-boxes' issue time).
template <int T>
__device__ __noinline__ void perm_small(const St st, const PermArgs &A) {
    u32 x[T][8];
#pragma unroll
    for (int j = 0; j < T; j++) {
        u32 c[8];
        lds_load(st, j, x[j]);
        load_const<true>(A.Cd, (size_t)j, c);
        add_lazy(x[j], c);
    }
    const bnm::Sh sh = bnm::sh_init();
    const int R = N_ROUNDS_F + A.rp;
    bnm::v4i tl[T * T];
    bnm::gtile tp = (bnm::gtile)A.St + st.lane;
#pragma unroll
    for (int q = 0; q < T * T; q++) tl[q] = tp[(size_t)q * 64];
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        const bool full = r < N_ROUNDS_F / 2 || r >= N_ROUNDS_F / 2 + A.rp;
        if (r == N_ROUNDS_F / 2 || r == N_ROUNDS_F / 2 + A.rp) {          // the other tile set from here on
            const size_t off = (size_t)(r == N_ROUNDS_F / 2 ? T * T : 0) * 64;
#pragma unroll
            for (int q = 0; q < T * T; q++) tl[q] = tp[off + (size_t)q * 64];
        }
        u32 k[T][8];
#pragma unroll
        for (int i = 0; i < T; i++) load_const<true>(A.SK, (size_t)r * T + i, k[i]);      // (asked for ahead of the S-boxes, which hide them)
        pow5_lazy(x[0]);
        if (full) {
#pragma unroll
            for (int j = 1; j < T; j++) pow5_lazy(x[j]);
        }
        bnm::v4i B0[T], B1[T];
#pragma unroll
        for (int j = 0; j < T; j++) bnm::b_prep(x[j], B0[j], B1[j]);
        const int rows = r == R - 1 && A.nout1 ? 1 : T;
#pragma unroll
        for (int i = 0; i < T; i++) {
            if (i < rows) {
                bnm::v16i a0, a1;
                bnm::mfma_first(tl[i * T], B0[0], B1[0], a0, a1);
#pragma unroll
                for (int j = 1; j < T; j++) { a0 = bnm::mfma(tl[i * T + j], B0[j], a0); a1 = bnm::mfma(tl[i * T + j], B1[j], a1); }
                bnm::finish_row(a0, a1, k[i], x[i], sh);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < T; j++) lds_store(st, j, x[j]);
}

// permutation of the t elements in buffer `cur`; returns the buffer holding the result
template <bool WIDE>
__device__ __noinline__ int bn_perm(const St st, int cur, const PermArgs &A) {
    const int t = A.t;
    if (A.small_) {
        if (t == 2) perm_small<2>(st, A);
        else if (t == 3) perm_small<3>(st, A);
        else perm_small<4>(st, A);
        return cur;
    }
    if (A.dense) {                                   // poseidon.circom:22-44 as written (tests)
        for (int r = 0; r < N_ROUNDS_F + A.rp; r++) {
            const bool full = r < N_ROUNDS_F / 2 || r >= N_ROUNDS_F / 2 + A.rp;
            add_sbox<WIDE>(st, cur, t, A.Cd, (size_t)r * t, full ? t : 1);
            dense_mul<WIDE>(st, cur, A.M, t, 0);
        }
        return cur;
    }
    if (A.mfma && !A.nofold && t == 17) {
        // The width of the arity-16 trees (config 4) in ONE function body: every phase called out of line saves and restores the callee-saved
        // half of its 256 registers in private memory (92-112 words per lane and call, ~480 KB per wave and permutation -- a third of the
        // kernel's private-memory traffic, which the chip pays for in power: tools/power_probe.py).  One copy of each phase: the partial rounds
        // and the closing layer sit at the top of the fifth full round (rp = 68 = 17 blocks of four, nothing left over).
        // The S-boxes of rounds 1-3 and 5-7 ride on the rows of the layer before them, round 4's on the closing layer's rows (its element 0, which comes
        // out of the partial rounds, alone afterwards); only round 0's is a pass of its own.  Two copies of the wide layer (with / without the S-box).
        // A.plain (leaf kernel): elements 1..16 went through round 0's S-box where they were absorbed, as plain integers (plain_sbox_store): only element 0 is left,
        // and the first layer reads its own copy of the tiles
        if (A.plain) { u32 x[8], c[8]; lds_load(st, 0, x); load_const<true>(A.C8, 0, c); add_lazy(x, c); pow5_lazy(x); lds_store(st, 0, x); }
        else BN_STAMP(0, sbox_lazy_impl(st, 17, A.C8));
        for (int r = 0; r < 8; r++) {                                         // (one copy of each form of the layer)
            const bnm::v4i *Mt = r == 0 && A.plain ? A.Mt0 : A.Mt;
            const u32 *MK = r == 0 && A.plain ? A.MK0 : A.MK + (size_t)r * 17 * 8;
            if (r == 3) BN_STAMP(1, (dense_mfma_impl<17, false, true>(st, Mt, MK, 0)))      // its rows 1..16: the partial rounds' y, in operand form
            else if (r == 7) BN_STAMP(1, dense_mfma_impl<17>(st, Mt, MK, 0, A.nout1 ? 1 : 17))
            else BN_STAMP(1, (dense_mfma_impl<17, true>(st, Mt, MK, 0)))
            if (r == 3) {
                BN_STAMP(2, (partial_rounds_mfma_impl<16, true>(st, A.Pt, A.KR, A.KU, A.rp)));
                BN_STAMP(1, (dense_mfma_impl<16, true, false, true>(st, A.Dt, A.DK, 1)));  // diag(1, Mh^RP) on the y as they are, then round 4's S-box on elements 1..16
                u32 x[8]; lds_load(st, 0, x); pow5_lazy(x); lds_store(st, 0, x);
            }
        }
        return cur;
    }
    if (A.mfma) {
        // the linear layers on the matrix cores (bn_mfma.cuh); between them the state is lazy (< 2^255), every constant but the first
        // round's arrives with a layer's rows, and the only 32x32 products left are the S-boxes'
        for (int r = 0; r < 4; r++) {
            BN_STAMP(0, sbox_lazy(st, t, r == 0 || A.nofold ? A.C8 + (size_t)r * t * 8 : nullptr));
            BN_STAMP(1, dense_mfma(st, A.Mt, A.MK + (size_t)r * t * 8, t, 0));
        }
        if (A.nofold) { u32 x[8], c[8]; lds_load(st, 0, x); load_const<true>(A.S, 0, c); add_lazy(x, c); lds_store(st, 0, x); }
        if (A.rp >= 4) BN_STAMP(2, partial_rounds_mfma(st, A));
        if (A.rp % 4 || A.rp < 4) {                  // the rounds left over, one by one on canonical values
            canon_state(st, t);
            partial_rounds<WIDE>(st, cur, A, A.rp & ~3);
            if (!A.nofold) {
                u32 x[8], c[8];
                lds_load(st, 0, x);
                load_const<true>(A.C8, (size_t)4 * t, c);
                bn::fr_add(x, c);
                lds_store(st, 0, x);
            }
        }
        BN_STAMP(1, dense_mfma(st, A.Dt, A.DK, t - 1, 1));        // diag(1, Mh^RP)
        for (int r = 4; r < 8; r++) {
            BN_STAMP(0, sbox_lazy(st, t, A.nofold ? A.C8 + (size_t)r * t * 8 : nullptr));
            BN_STAMP(1, dense_mfma(st, A.Mt, A.MK + (size_t)r * t * 8, t, 0, r == 7 && A.nout1 ? 1 : 17));
        }
        return cur;
    }
    for (int r = 0; r < 4; r++) {
        add_sbox<WIDE>(st, cur, t, A.C8, (size_t)r * t, t);
        dense_mul<WIDE>(st, cur, A.M, t, 0);
    }
    partial_rounds<WIDE>(st, cur, A, 0);
    dense_mul<WIDE>(st, cur, A.D, t - 1, 1);         // diag(1, Mh^RP)
    for (int r = 4; r < 8; r++) {
        add_sbox<WIDE>(st, cur, t, A.C8, (size_t)r * t, t);
        dense_mul<WIDE>(st, cur, A.M, t, 0);
    }
    return cur;
}

__device__ __forceinline__ void to_mont_store(const St st, int j, const u64 w[4]) {
    u32 x[8], r2[8], o[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { x[2 * k] = (u32)w[k]; x[2 * k + 1] = (u32)(w[k] >> 32); }
#pragma unroll
    for (int l = 0; l < 8; l++) r2[l] = bn::r2_limb(l);
    bn::fr_mul(o, x, r2);                            // frm_toMontgomery: x * 2^256 mod r (x < 2^256)
    lds_store(st, j, o);
}
// the same input for a permutation with A.plain: round 0's S-box on the PLAIN integer x + c (c = the round constant as a plain integer; x < 2^192: no reduction,
// no conversion product); bn29::pow5's result is then 2^-1280 times what the state's form would give, which the first layer's own tiles take back
__device__ __forceinline__ void plain_sbox_store(const St st, int j, const u64 w[4], const u32 *C0p) {
    u32 x[8], c[8];
#pragma unroll
    for (int k = 0; k < 4; k++) { x[2 * k] = (u32)w[k]; x[2 * k + 1] = (u32)(w[k] >> 32); }
    load_const<true>(C0p, (size_t)j, c);
    bnm::add_chain8(x, c);
    pow5_lazy(x);
    lds_store(st, j, x);
}
__device__ __forceinline__ void zero_store(const St st, int j) {
    const u32 z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    lds_store(st, j, z);
}
__device__ __forceinline__ void digest_out(const St st, int j, u64 *o) {
    u32 x[8];
    lds_load(st, j, x);
    bnm::canon(x);                                   // the matrix-core pipeline leaves lazy representatives
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = (u64)x[2 * k] | ((u64)x[2 * k + 1] << 32);
}

// leaf digests (merklehash_bn128_worker.js:42-98): one row per lane
template <bool WIDE>
__global__ void __launch_bounds__(BN_THREADS) __attribute__((amdgpu_waves_per_eu(2))) bn_linear_hash_kernel(const u64 *__restrict__ in, u64 width, u64 height, int arity, int custom,
                                                                    PermArgs full, PermArgs last, u64 *__restrict__ out) {
    extern __shared__ u32 S[];
    const int lane = threadIdx.x % BN_BLOCK, wv = threadIdx.x / BN_BLOCK, tmax = arity + 1;
    u32 hi_arr[(17 - BN_LDS_ELEMS) * 8];
    const St st = { (lds_u32)S + wv * lds_words(tmax), (priv_u32)hi_arr, tmax, lane };
    const u64 row0 = ((u64)blockIdx.x * BN_WG_WAVES + wv) * BN_BLOCK + lane;
    const bool live = row0 < height;
    const u64 *v = in + (live ? row0 : height - 1) * width;
    int cur = 0;
#ifdef BN_STAMPS
    const unsigned long long tk0 = bn_now(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (width <= 4) {                                // :45-50: up to four words taken as one 256-bit integer
        u64 w[4] = { 0, 0, 0, 0 };
        for (u64 k = 0; k < width; k++) w[k] = v[k];
        to_mont_store(st, 0, w);
    } else {
        zero_store(st, 0);             // st = 0
        const u64 nEl = (width + 2) / 3;             // 3 Goldilocks words per field element (:54-67)
        u64 e = 0;
        while (e < nEl) {
            const u64 n = nEl - e < (u64)arity ? nEl - e : (u64)arity;
            const bool plain = full.plain && (n == (u64)arity || custom);      // the chunk goes to `full`, whose first S-box rides on the absorb
            for (u64 k = 0; k < n; k++) {
                u64 w[4] = { 0, 0, 0, 0 };
                for (int q = 0; q < 3; q++) { const u64 idx = 3 * (e + k) + q; if (idx < width) w[q] = v[idx]; }
                if (plain) plain_sbox_store(st, 1 + (int)k, w, full.C0p);
                else to_mont_store(st, 1 + (int)k, w);
            }
            if (n == (u64)arity) cur = bn_perm<WIDE>(st, cur, full);
            else if (custom) {                       // :87-93: zero-pad the last chunk to `arity` inputs
                const u64 z[4] = { 0, 0, 0, 0 };
                for (u64 k = n; k < (u64)arity; k++) { if (plain) plain_sbox_store(st, 1 + (int)k, z, full.C0p); else zero_store(st, 1 + (int)k); }
                cur = bn_perm<WIDE>(st, cur, full);
            } else cur = bn_perm<WIDE>(st, cur, last);      // :85-86: t = nLast + 1
            e += n;
        }
    }
    if (live) digest_out(st, 0, out + 4 * row0);
#ifdef BN_STAMPS
    if (lane == 0) { atomicAdd(&g_bn_stamps[7], bn_now() - tk0); atomicAdd(&g_bn_stamps[8], 1ull); atomicAdd(&g_bn_stamps[12], __builtin_amdgcn_s_memrealtime() - tr0); }
#endif
}

// parents (merklehash_bn128_worker.js:104-144): out[i] = Poseidon(0; in[arity*i .. arity*i+arity-1])[0]
template <bool WIDE>
__global__ void __launch_bounds__(BN_THREADS) __attribute__((amdgpu_waves_per_eu(2))) bn_merkle_level_kernel(const u64 *__restrict__ in, u64 nOps, int arity, PermArgs full, u64 *__restrict__ out) {
    extern __shared__ u32 S[];
    const int lane = threadIdx.x % BN_BLOCK, wv = threadIdx.x / BN_BLOCK, tmax = arity + 1;
    u32 hi_arr[(17 - BN_LDS_ELEMS) * 8];
    const St st = { (lds_u32)S + wv * lds_words(tmax), (priv_u32)hi_arr, tmax, lane };
    const u64 i0 = ((u64)blockIdx.x * BN_WG_WAVES + wv) * BN_BLOCK + lane;
    const bool live = i0 < nOps;
    const u64 *v = in + (live ? i0 : nOps - 1) * (u64)arity * 4;
    zero_store(st, 0);
    for (int k = 0; k < arity; k++) {                // children are already in Montgomery form
        u32 x[8];
#pragma unroll
        for (int q = 0; q < 4; q++) { const u64 w = v[4 * k + q]; x[2 * q] = (u32)w; x[2 * q + 1] = (u32)(w >> 32); }
        lds_store(st, 1 + k, x);
    }
    (void)bn_perm<WIDE>(st, 0, full);
    if (live) digest_out(st, 0, out + 4 * i0);
}

// circomlibjs poseidon(inputs, initState, nOut): normal-form words in and out (transcript, verification, tests)
template <bool WIDE>
__global__ void __launch_bounds__(BN_THREADS) __attribute__((amdgpu_waves_per_eu(2))) bn_poseidon_kernel(const u64 *__restrict__ in, const u64 *__restrict__ init, u64 count, int nIn, int nOut,
                                                                 PermArgs full, u64 *__restrict__ out) {
    extern __shared__ u32 S[];
    const int lane = threadIdx.x % BN_BLOCK, wv = threadIdx.x / BN_BLOCK, tmax = nIn + 1;
    u32 hi_arr[(17 - BN_LDS_ELEMS) * 8];
    const St st = { (lds_u32)S + wv * lds_words(tmax), (priv_u32)hi_arr, tmax, lane };
    const u64 i0 = ((u64)blockIdx.x * BN_WG_WAVES + wv) * BN_BLOCK + lane;
    const bool live = i0 < count;
    const u64 i = live ? i0 : count - 1;
    u64 w[4] = { 0, 0, 0, 0 };
    if (init) for (int q = 0; q < 4; q++) w[q] = init[4 * i + q];
    to_mont_store(st, 0, w);
    for (int k = 0; k < nIn; k++) {
        for (int q = 0; q < 4; q++) w[q] = in[(i * nIn + k) * 4 + q];
        to_mont_store(st, 1 + k, w);
    }
    (void)bn_perm<WIDE>(st, 0, full);
    if (!live) return;
    for (int k = 0; k < nOut; k++) {                 // out of Montgomery form: multiply by 1
        u32 x[8], one[8] = { 1, 0, 0, 0, 0, 0, 0, 0 }, o[8];
        lds_load(st, k, x);
        bn::fr_mul(o, x, one);
        for (int q = 0; q < 4; q++) out[(i * nOut + k) * 4 + q] = (u64)o[2 * q] | ((u64)o[2 * q + 1] << 32);
    }
}

// A chain of dependent permutations (transcript.bn128.js:56-66 absorbing a list: each full block of nIn elements is permuted
// with the previous output 0 as state element 0).  One permutation per lane leaves such a chain at one wave-alone
// permutation (~3 ms) per block; here the wave shares each permutation and runs the round function as poseidon.circom:22-44
// states it: lane (l, s), l < t, s < 3, holds state element l (the three copies stay equal); constants and S-boxes in
// parallel (element 0 alone in the partial rounds); of row l of the dense MDS product, lane (l, s) accumulates the terms
// j = s, s+3, ... unreduced in 17 limbs, the three partial sums are added through LDS and reduced once.  Normal-form words
// in and out.  Measured at t = 17: 3.1 ms per permutation for a call per block, 0.76 ms with one lane per row (issue-bound
// on its 17 multiply-accumulates per round; requesting operands a term ahead changes nothing), 0.49 ms with the row split.
constexpr int CHAIN_SUB = 3;
// One workgroup (one wave) per chain: chain g reads nBlocks*nIn elements at blocks + g*nBlocks*nIn*4, state element 0 from
// init + 4g (zero if init is null), and writes its first nOut outputs at out + g*nOut*4 -- with nBlocks = 1 this is a batch of
// independent permutations, the faster form while there are fewer of them than SIMDs to give a whole wave each.
__global__ void __launch_bounds__(64) bn_sponge_chain_kernel(const u64 *__restrict__ blocks, u64 nBlocks, int nIn, const u64 *__restrict__ init,
                                                            PermArgs A, int nOut, int mont, u64 *__restrict__ out) {
    __shared__ u32 sh[17 * 8];
    __shared__ u32 part[CHAIN_SUB * 17 * 17];
    const int t = nIn + 1, lane = threadIdx.x;
    blocks += (u64)blockIdx.x * nBlocks * nIn * 4;
    out += (u64)blockIdx.x * nOut * 4;
    const u64 zero4[4] = { 0, 0, 0, 0 };
    if (init) init += (u64)blockIdx.x * 4;
    const bool act = lane < CHAIN_SUB * t;
    const int sub = act ? lane / t : 0, l = act ? lane - sub * t : 0;
    u32 r2[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r2[i] = bn::r2_limb(i);
    auto load_mont = [&](const u64 *w) {             // frm_toMontgomery; mont: the words are in Montgomery form already (tree nodes)
        u32 v[8];
#pragma unroll
        for (int q = 0; q < 4; q++) { v[2 * q] = (u32)w[q]; v[2 * q + 1] = (u32)(w[q] >> 32); }
        if (mont) {
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = v[i];
        } else bn::fr_mul(x, v, r2);
    };
    if (l == 0) load_mont(init ? init : zero4);
    const int nRounds = N_ROUNDS_F + A.rp;
    for (u64 b = 0; b < nBlocks; b++) {
        if (l > 0) load_mont(blocks + (b * nIn + (l - 1)) * 4);
        for (int r = 0; r < nRounds; r++) {
            u32 c[8];
            load_const<true>(A.Cd, (size_t)r * t + l, c);
            bn::fr_add(x, c);
            const bool full = r < N_ROUNDS_F / 2 || r >= N_ROUNDS_F / 2 + A.rp;
            if (full || l == 0) pow5(x);
            if (act && sub == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) sh[l * 8 + i] = x[i];
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
            u32 acc[17];
#pragma unroll
            for (int i = 0; i < 17; i++) acc[i] = 0;
            for (int j = sub; j < t; j += CHAIN_SUB) {
                u32 y[8], m[8];
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = sh[j * 8 + i];
                load_const<true>(A.M, (size_t)l * t + j, m);
                bn::mac17(acc, y, m);
            }
            if (act) {
#pragma unroll
                for (int i = 0; i < 17; i++) part[(sub * 17 + l) * 17 + i] = acc[i];
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // every copy adds the three partial sums (together at most t products: they fit the 17 limbs as one row did)
#pragma unroll
            for (int i = 0; i < 17; i++) acc[i] = part[l * 17 + i];
#pragma unroll
            for (int q = 1; q < CHAIN_SUB; q++) {
                u64 cy = 0;
#pragma unroll
                for (int i = 0; i < 17; i++) {
                    const u64 v = (u64)acc[i] + part[(q * 17 + l) * 17 + i] + cy;
                    acc[i] = (u32)v; cy = v >> 32;
                }
            }
            __builtin_amdgcn_wave_barrier();
            bn::redc17(x, acc);
        }
    }
    if (!act || sub != 0 || l >= nOut) return;
    u32 one[8] = { 1, 0, 0, 0, 0, 0, 0, 0 }, o[8];
    if (mont) {
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = x[i];
    } else bn::fr_mul(o, x, one);                    // out of Montgomery form
#pragma unroll
    for (int q = 0; q < 4; q++) out[l * 4 + q] = (u64)o[2 * q] | ((u64)o[2 * q + 1] << 32);
}

// Montgomery <-> normal form of n elements (frm_toMontgomery / F.toObject)
__global__ void bn_convert_kernel(const u64 *__restrict__ in, u64 n, int toMont, u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 x[8], k[8], o[8];
    for (int q = 0; q < 4; q++) { x[2 * q] = (u32)in[4 * i + q]; x[2 * q + 1] = (u32)(in[4 * i + q] >> 32); }
    for (int l = 0; l < 8; l++) k[l] = toMont ? bn::r2_limb(l) : (l == 0 ? 1u : 0u);
    bn::fr_mul(o, x, k);
    for (int q = 0; q < 4; q++) out[4 * i + q] = (u64)o[2 * q] | ((u64)o[2 * q + 1] << 32);
}

// A batch of openings (fri.js:83-105 opens every tree at every query): block q gathers row idxs[q] and, per level, the `arity` nodes of its group
// (Montgomery words as stored; nodes beyond the level's count read as zero, merklehash_bn128_p.js:165-170) into out[q] = [width values | levels x arity x 4 words]
struct BnLevels { u64 off[40], n[40]; u32 levels; };
__global__ void bn_group_proofs_kernel(const u64 *__restrict__ elems, const u64 *__restrict__ nodes, u64 width, int arity, int abits,
                                       const u64 *__restrict__ idxs, BnLevels L, u64 *__restrict__ out) {
    const u64 stride = width + (u64)L.levels * arity * 4;
    u64 *o = out + blockIdx.x * stride;
    const u64 idx = idxs[blockIdx.x];
    for (u64 c = threadIdx.x; c < width; c += blockDim.x) o[c] = elems[idx * width + c];
    u64 id = idx;
    for (u32 l = 0; l < L.levels; l++) {
        const u64 si = id ^ (id & (u64)(arity - 1));
        for (u32 k = threadIdx.x; k < (u32)arity * 4; k += blockDim.x)
            o[width + ((u64)l * arity) * 4 + k] = si + k / 4 < L.n[l] ? nodes[(L.off[l] + si) * 4 + k] : 0;
        id >>= abits;
    }
}

size_t lds_bytes(int tmax) {                         // the elements above BN_LDS_ELEMS live in private memory
    static const size_t pad = getenv("PIL2GL_BN128_LDS_PAD") ? (size_t)atol(getenv("PIL2GL_BN128_LDS_PAD")) : 0;   // occupancy experiments: extra bytes per workgroup
    return (size_t)lds_words(tmax) * 4 * BN_WG_WAVES + pad;
}

PermArgs perm_args(const Params *P, bool plainInputs = false, bool firstOnly = false) {
    PermArgs a;
    a.C8 = P->C8; a.M = P->M; a.D = P->D; a.S = P->S; a.V = P->V; a.W = P->W; a.Cd = P->Cd; a.t = P->t; a.rp = P->rp;
    static const bool dense = getenv("PIL2GL_BN128_DENSE") && atoi(getenv("PIL2GL_BN128_DENSE"));
    a.dense = dense ? 1 : 0;
    static const bool mfma = !(getenv("PIL2GL_BN128_MFMA") && !atoi(getenv("PIL2GL_BN128_MFMA")));   // =0: the layers on the vector ALU (A/B runs)
    a.mfma = mfma ? 1 : 0;
    static const bool nofold = getenv("PIL2GL_BN128_NOFOLD") && atoi(getenv("PIL2GL_BN128_NOFOLD"));
    a.nofold = nofold ? 1 : 0;
    a.Mt = P->Mt; a.Dt = P->Dt; a.Pt = P->Pt; a.MK = P->MK; a.DK = P->DK; a.KR = P->KR; a.KU = P->KU;
    memcpy(a.m00, P->m00, 32);
    a.Mt0 = P->Mt0; a.MK0 = P->MK0; a.C0p = P->C0p;
    a.St = P->St; a.SK = P->SK;
    static const bool nosmall = getenv("PIL2GL_BN128_SMALL") && !atoi(getenv("PIL2GL_BN128_SMALL"));       // =0: small widths through the blocked pipeline as well (A/B runs)
    a.small_ = P->St && !nosmall && a.mfma && !a.nofold && !a.dense && BN_SBOX29 ? 1 : 0;
    static const bool noplain = getenv("PIL2GL_BN128_PLAIN") && !atoi(getenv("PIL2GL_BN128_PLAIN"));   // =0: inputs converted and S-boxed by the permutation (A/B runs)
    static const bool allrows = getenv("PIL2GL_BN128_ALLROWS") && atoi(getenv("PIL2GL_BN128_ALLROWS"));   // =1: every row of the last layer whatever the caller reads (A/B runs)
    a.nout1 = firstOnly && !allrows ? 1 : 0;
    a.plain = plainInputs && !noplain && a.mfma && !a.nofold && !a.dense && BN_SBOX29 && P->t == 17 ? 1 : 0;
    return a;
}

template <typename K>
int set_lds_attr(K kernel, size_t bytes) {
    HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PIL2GL_OK;
}

// The kernel instances that request the next term's operands before multiplying the current one (WIDE) were the faster ones
// while a wide state (t >= 10) kept its whole state in LDS and ran ONE wave per SIMD.  With the upper elements in private memory
// (BN_LDS_ELEMS) two waves fit, the second wave covers the operand latency, and the kernels turn out to be bound by their vector
// instruction COUNT (5.3e10 per 2^20 x 100 arity-16 commit = 0.85 of the vector issue slots of its 110 ms): the plain instances,
// which do not rotate prefetched operands through registers, are 12 % faster (121.1 -> 106.6 ms).  PIL2GL_BN128_WIDE=1 selects
// the prefetching instances for an A/B run.
bool wide_state(int t) {
    static const int force = getenv("PIL2GL_BN128_WIDE") ? atoi(getenv("PIL2GL_BN128_WIDE")) : 0;
    (void)t;
    return force != 0;
}

int check_arity(uint32_t arity) {
    if (arity < 2 || arity > 16 || (arity & (arity - 1))) return fail(PIL2GL_EINVAL, "arity must be 2, 4, 8 or 16 (got %u)", arity);
    return PIL2GL_OK;
}

}  // namespace

extern "C" {

#ifdef BN_STAMPS
int pil2gl_bn128_debug_stamps(uint64_t *host16, int reset) {
    HIP_TRY(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_bn_stamps), 128));
    if (reset) { uint64_t z[16] = { 0 }; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_bn_stamps), z, 128)); }
    return PIL2GL_OK;
}
#endif

uint64_t pil2gl_bn128_merkle_num_nodes(uint64_t height, uint32_t arity) {      // merklehash_bn128_p.js:31-45, in nodes
    if (height == 0 || arity < 2) return 0;
    uint64_t n = height, nextN = (n - 1) / arity + 1, acc = nextN * arity;
    while (n > 1) {
        n = nextN;
        nextN = (n - 1) / arity + 1;
        acc += n > 1 ? nextN * arity : 1;
    }
    return acc;
}

int pil2gl_bn128_linear_hash_rows_dev(const uint64_t *in, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (height == 0) return PIL2GL_OK;
    P2_TRY(check_arity(arity));
    if (!out || (!in && width)) return fail(PIL2GL_EINVAL, "null buffer");
    const Params *pf, *pl;
    P2_TRY(get_params((int)arity + 1, &pf));
    pl = pf;
    const uint64_t nEl = (width + 2) / 3, nLast = nEl % arity;
    if (width > 4 && !custom && nLast) P2_TRY(get_params((int)nLast + 1, &pl));
    const size_t lds = lds_bytes((int)arity + 1);
    const uint64_t blocks = (height + BN_THREADS - 1) / BN_THREADS;
    if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");
    if (wide_state((int)arity + 1)) {
        P2_TRY(set_lds_attr(bn_linear_hash_kernel<true>, lds));
        bn_linear_hash_kernel<true><<<(unsigned)blocks, BN_THREADS, lds, as_stream(stream)>>>(in, width, height, (int)arity, custom ? 1 : 0, perm_args(pf, true, true), perm_args(pl, false, true), out);
    } else {
        P2_TRY(set_lds_attr(bn_linear_hash_kernel<false>, lds));
        bn_linear_hash_kernel<false><<<(unsigned)blocks, BN_THREADS, lds, as_stream(stream)>>>(in, width, height, (int)arity, custom ? 1 : 0, perm_args(pf, true, true), perm_args(pl, false, true), out);
    }
    KERNEL_CHECK();
    return PIL2GL_OK;
}

static long wave_per_perm_max() {
    static const long v = getenv("PIL2GL_BN128_WAVE_PER_PERM_MAX") ? atol(getenv("PIL2GL_BN128_WAVE_PER_PERM_MAX")) : 2048;
    return v;
}
int pil2gl_bn128_merkelize_level_dev(const uint64_t *in, uint64_t nOps, uint32_t arity, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (nOps == 0) return PIL2GL_OK;
    P2_TRY(check_arity(arity));
    if (!in || !out) return fail(PIL2GL_EINVAL, "null buffer");
    const Params *pf;
    P2_TRY(get_params((int)arity + 1, &pf));
    const size_t lds = lds_bytes((int)arity + 1);
    const uint64_t blocks = (nOps + BN_THREADS - 1) / BN_THREADS;
    if (blocks > 0x7fffffffull) return fail(PIL2GL_EINVAL, "grid too large");
    if ((long)nOps <= wave_per_perm_max()) {          // the levels near the root: a wave per parent (see pil2gl_bn128_poseidon_dev)
        bn_sponge_chain_kernel<<<(unsigned)nOps, 64, 0, as_stream(stream)>>>(in, 1, (int)arity, nullptr, perm_args(pf), 1, 1, out);
        KERNEL_CHECK();
        return PIL2GL_OK;
    }
    if (wide_state((int)arity + 1)) {
        P2_TRY(set_lds_attr(bn_merkle_level_kernel<true>, lds));
        bn_merkle_level_kernel<true><<<(unsigned)blocks, BN_THREADS, lds, as_stream(stream)>>>(in, nOps, (int)arity, perm_args(pf, false, true), out);
    } else {
        P2_TRY(set_lds_attr(bn_merkle_level_kernel<false>, lds));
        bn_merkle_level_kernel<false><<<(unsigned)blocks, BN_THREADS, lds, as_stream(stream)>>>(in, nOps, (int)arity, perm_args(pf, false, true), out);
    }
    KERNEL_CHECK();
    return PIL2GL_OK;
}

int pil2gl_bn128_merkelize_dev(const uint64_t *elems, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *nodes, void *stream) {
    P2_TRY(ensure_init());
    if (height == 0) return fail(PIL2GL_EINVAL, "height must be > 0");
    P2_TRY(check_arity(arity));
    if (!nodes || (!elems && width)) return fail(PIL2GL_EINVAL, "null buffer");
    hipStream_t st = as_stream(stream);
    // merklehash_bn128_p.js:51: nodes is a fresh (zeroed) array; the zero padding of short levels relies on it
    HIP_TRY(hipMemsetAsync(nodes, 0, pil2gl_bn128_merkle_num_nodes(height, arity) * 32, st));
    P2_TRY(pil2gl_bn128_linear_hash_rows_dev(elems, width, height, arity, custom, nodes, stream));
    uint64_t pIn = 0, n = height, nextN = (n - 1) / arity + 1, pOut = pIn + nextN * arity * 4;   // :89-101, in u64 words
    while (n > 1) {
        P2_TRY(pil2gl_bn128_merkelize_level_dev(nodes + pIn, nextN, arity, nodes + pOut, stream));
        n = nextN;
        nextN = (n - 1) / arity + 1;
        pIn = pOut;
        pOut = pIn + nextN * arity * 4;
    }
    return PIL2GL_OK;
}

int pil2gl_bn128_poseidon_dev(const uint64_t *in, const uint64_t *init, uint64_t count, uint32_t nIn, uint32_t nOut, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (count == 0) return PIL2GL_OK;
    if (nIn < 1 || nIn > 16) return fail(PIL2GL_EINVAL, "BN128 Poseidon takes 1..16 inputs (got %u)", nIn);
    if (nOut < 1 || nOut > nIn + 1) return fail(PIL2GL_EINVAL, "nOut must be 1..nInputs+1");
    if (!in || !out) return fail(PIL2GL_EINVAL, "null buffer");
    const Params *pf;
    P2_TRY(get_params((int)nIn + 1, &pf));
    const size_t lds = lds_bytes((int)nIn + 1);
    const unsigned pblocks = (unsigned)((count + BN_THREADS - 1) / BN_THREADS);
    // few permutations (a transcript squeeze, the levels of a handful of Merkle paths): a lane each would leave them at the
    // latency of one wave working alone (~3 ms at t = 17); a wave each runs them in ~0.5 ms while the SIMDs outnumber them
    if ((long)count <= wave_per_perm_max()) {
        bn_sponge_chain_kernel<<<(unsigned)count, 64, 0, as_stream(stream)>>>(in, 1, (int)nIn, init, perm_args(pf), (int)nOut, 0, out);
        KERNEL_CHECK();
        return PIL2GL_OK;
    }
    if (wide_state((int)nIn + 1)) {
        P2_TRY(set_lds_attr(bn_poseidon_kernel<true>, lds));
        bn_poseidon_kernel<true><<<pblocks, BN_THREADS, lds, as_stream(stream)>>>(in, init, count, (int)nIn, (int)nOut, perm_args(pf), out);
    } else {
        P2_TRY(set_lds_attr(bn_poseidon_kernel<false>, lds));
        bn_poseidon_kernel<false><<<pblocks, BN_THREADS, lds, as_stream(stream)>>>(in, init, count, (int)nIn, (int)nOut, perm_args(pf), out);
    }
    KERNEL_CHECK();
    return PIL2GL_OK;
}

int pil2gl_bn128_convert_dev(const uint64_t *in, uint64_t n, int toMontgomery, uint64_t *out, void *stream) {
    P2_TRY(ensure_init());
    if (n == 0) return PIL2GL_OK;
    if (!in || !out) return fail(PIL2GL_EINVAL, "null buffer");
    bn_convert_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(in, n, toMontgomery, out);
    KERNEL_CHECK();
    return PIL2GL_OK;
}

// getGroupProof (merklehash_bn128_p.js:142-182): row values + all `arity` nodes of idx's group at every level, normal form
int pil2gl_bn128_group_proof_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height, uint32_t arity,
                                 uint64_t idx, uint64_t *hostVals, uint64_t *hostSiblings, uint32_t *nLevels) {
    P2_TRY(ensure_init());
    if (idx >= height) return fail(PIL2GL_EINVAL, "Out of range");               // :145
    P2_TRY(check_arity(arity));
    if (!nodes || !hostSiblings || !nLevels || (width && (!elems || !hostVals))) return fail(PIL2GL_EINVAL, "null buffer");
    if (width) HIP_TRY(hipMemcpy(hostVals, elems + idx * width, width * 8, hipMemcpyDeviceToHost));
    uint32_t nbits = 0; while ((1u << nbits) < arity) nbits++;
    uint64_t offset = 0, n = height, id = idx; uint32_t lv = 0;
    std::vector<uint64_t> mont;
    while (n > 1) {
        const uint64_t si = id ^ (id & (arity - 1));
        mont.resize((size_t)(lv + 1) * arity * 4);
        HIP_TRY(hipMemcpy(mont.data() + (size_t)lv * arity * 4, nodes + (offset + si) * 4, (size_t)arity * 32, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < arity; i++) if (i >= n) memset(mont.data() + ((size_t)lv * arity + i) * 4, 0, 32);   // :165-170
        const uint64_t nextN = (n - 1) / arity + 1;
        offset += nextN * arity; n = nextN; id >>= nbits; lv++;
    }
    for (size_t k = 0; k < mont.size() / 4; k++) {
        U256 v = { { mont[4 * k], mont[4 * k + 1], mont[4 * k + 2], mont[4 * k + 3] } };
        v = h_from_mont(v);
        memcpy(hostSiblings + 4 * k, v.w, 32);
    }
    *nLevels = lv;
    return PIL2GL_OK;
}

// The same for a batch of rows in one launch and one copy each way (a proof opens 64 rows of every tree: one call per row is seven small
// synchronous copies per row and tree).  hostVals: nIdx x width; hostSiblings: nIdx x levels x arity x 4 words, normal form.
int pil2gl_bn128_group_proofs_dev(const uint64_t *elems, const uint64_t *nodes, uint64_t width, uint64_t height, uint32_t arity,
                                  const uint64_t *hostIdxs, uint32_t nIdx, uint64_t *hostVals, uint64_t *hostSiblings, uint32_t *nLevels) {
    P2_TRY(ensure_init());
    P2_TRY(check_arity(arity));
    if (!nIdx) return PIL2GL_OK;
    if (!nodes || !hostIdxs || !hostSiblings || !nLevels || (width && (!elems || !hostVals))) return fail(PIL2GL_EINVAL, "null buffer");
    for (uint32_t i = 0; i < nIdx; i++) if (hostIdxs[i] >= height) return fail(PIL2GL_EINVAL, "Out of range");       // merklehash_bn128_p.js:145
    uint32_t nbits = 0; while ((1u << nbits) < arity) nbits++;
    BnLevels L; L.levels = 0;
    uint64_t offset = 0, n = height;
    while (n > 1) {
        if (L.levels >= 40) return fail(PIL2GL_EINVAL, "too many levels");
        L.off[L.levels] = offset; L.n[L.levels] = n; L.levels++;
        const uint64_t nextN = (n - 1) / arity + 1;
        offset += nextN * arity; n = nextN;
    }
    const u64 stride = width + (u64)L.levels * arity * 4;
    u64 *d;
    P2_TRY(scratch(6, (u64)nIdx * (stride + 1), &d));
    u64 *dIdx = d + (u64)nIdx * stride;
    HIP_TRY(hipMemcpy(dIdx, hostIdxs, (u64)nIdx * 8, hipMemcpyHostToDevice));
    bn_group_proofs_kernel<<<nIdx, 64>>>(elems, nodes, width, (int)arity, (int)nbits, dIdx, L, d);
    KERNEL_CHECK();
    std::vector<u64> h((size_t)nIdx * stride);
    HIP_TRY(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    const u64 per = (u64)L.levels * arity * 4;
    for (uint32_t q = 0; q < nIdx; q++) {
        if (width) memcpy(hostVals + (u64)q * width, h.data() + (u64)q * stride, width * 8);
        for (u64 k = 0; k < per / 4; k++) {
            const u64 *w = h.data() + (u64)q * stride + width + 4 * k;
            U256 v = { { w[0], w[1], w[2], w[3] } };
            v = h_from_mont(v);
            memcpy(hostSiblings + (u64)q * per + 4 * k, v.w, 32);
        }
    }
    *nLevels = L.levels;
    return PIL2GL_OK;
}

// ---- host-pointer forms ----
int pil2gl_bn128_poseidon(const uint64_t *in, const uint64_t *init, uint64_t count, uint32_t nIn, uint32_t nOut, uint64_t *out) {
    P2_TRY(ensure_init());
    if (count == 0) return PIL2GL_OK;
    if (!in || !out) return fail(PIL2GL_EINVAL, "null buffer");
    u64 *d = nullptr;
    const u64 nI = count * nIn * 4, nS = init ? count * 4 : 0, nO = count * nOut * 4;
    bool owned = false;
    P2_TRY(stage_acquire(nI + nS + nO, &d, &owned));
    int rc = PIL2GL_OK;
    hipError_t e = hipMemcpy(d, in, nI * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && nS) e = hipMemcpy(d + nI, init, nS * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) rc = pil2gl_bn128_poseidon_dev(d, nS ? d + nI : nullptr, count, nIn, nOut, d + nI + nS, nullptr);
    if (rc == PIL2GL_OK) { e = hipMemcpy(out, d + nI + nS, nO * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H"); }
    stage_release(d, owned);
    return rc;
}

// transcript.bn128.js:56-66 for a list: nBlocks full blocks of nIn elements absorbed one after the other; state element 0
// starts as hostInit and is then each permutation's output 0; hostOut = the nIn+1 outputs of the last permutation
int pil2gl_bn128_sponge_absorb(const uint64_t *hostBlocks, uint64_t nBlocks, uint32_t nIn, const uint64_t hostInit[4], uint64_t *hostOut) {
    P2_TRY(ensure_init());
    if (!hostBlocks || !hostInit || !hostOut) return fail(PIL2GL_EINVAL, "null buffer");
    if (nBlocks == 0) return fail(PIL2GL_EINVAL, "nothing to absorb");
    if (nIn < 1 || nIn > 16) return fail(PIL2GL_EINVAL, "BN128 Poseidon takes 1..16 inputs (got %u)", nIn);
    const Params *pf;
    P2_TRY(get_params((int)nIn + 1, &pf));
    const u64 nB = nBlocks * nIn * 4, nO = (u64)(nIn + 1) * 4;
    u64 *d = nullptr; bool owned = false;
    P2_TRY(stage_acquire(nB + 4 + nO, &d, &owned));
    int rc = PIL2GL_OK;
    hipError_t e = hipMemcpy(d, hostBlocks, nB * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + nB, hostInit, 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) {
        bn_sponge_chain_kernel<<<1, 64>>>(d, nBlocks, (int)nIn, d + nB, perm_args(pf), (int)nIn + 1, 0, d + nB + 4);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(hostOut, d + nB + 4, nO * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "bn_sponge_chain_kernel");
    }
    stage_release(d, owned);
    return rc;
}

int pil2gl_bn128_merkelize(const uint64_t *elems, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *nodes) {
    P2_TRY(ensure_init());
    if (height == 0) return fail(PIL2GL_EINVAL, "height must be > 0");
    P2_TRY(check_arity(arity));
    const u64 nE = width * height, nN = pil2gl_bn128_merkle_num_nodes(height, arity) * 4;
    u64 *d = nullptr;
    bool owned = false;
    P2_TRY(stage_acquire(nE + nN, &d, &owned));
    int rc = PIL2GL_OK;
    hipError_t e = nE ? hipMemcpy(d, elems, nE * 8, hipMemcpyHostToDevice) : hipSuccess;
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) rc = pil2gl_bn128_merkelize_dev(d, width, height, arity, custom, d + nE, nullptr);
    if (rc == PIL2GL_OK) { e = hipMemcpy(nodes, d + nE, nN * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H"); }
    stage_release(d, owned);
    return rc;
}

int pil2gl_bn128_linear_hash_rows(const uint64_t *in, uint64_t width, uint64_t height, uint32_t arity, int custom, uint64_t *out) {
    P2_TRY(ensure_init());
    if (height == 0) return PIL2GL_OK;
    const u64 nE = width * height, nO = height * 4;
    u64 *d = nullptr;
    bool owned = false;
    P2_TRY(stage_acquire(nE + nO, &d, &owned));
    int rc = PIL2GL_OK;
    hipError_t e = nE ? hipMemcpy(d, in, nE * 8, hipMemcpyHostToDevice) : hipSuccess;
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy H2D");
    if (rc == PIL2GL_OK) rc = pil2gl_bn128_linear_hash_rows_dev(d, width, height, arity, custom, d + nE, nullptr);
    if (rc == PIL2GL_OK) { e = hipMemcpy(out, d + nE, nO * 8, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H"); }
    stage_release(d, owned);
    return rc;
}

int pil2gl_bn128_convert(const uint64_t *in, uint64_t n, int toMontgomery, uint64_t *out) {     // host-only arithmetic (a few values: roots, proofs)
    if (n && (!in || !out)) return fail(PIL2GL_EINVAL, "null buffer");
    for (uint64_t k = 0; k < n; k++) {
        U256 v = { { in[4 * k], in[4 * k + 1], in[4 * k + 2], in[4 * k + 3] } };
        v = toMontgomery ? h_to_mont(v) : h_from_mont(v);
        memcpy(out + 4 * k, v.w, 32);
    }
    return PIL2GL_OK;
}

}  // extern "C"
