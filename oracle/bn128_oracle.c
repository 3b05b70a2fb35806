/* CPU restatement in C of the BN254-Poseidon linear hash and Merkle tree of the reference's BN128 MerkleHash -- test
 * infrastructure and the `cpu_baseline` of bench.py's config-4 line (never part of the product).  It restates, in 4 x 64-bit
 * Montgomery arithmetic, exactly what oracle/bn128_oracle.py states on Python integers (which is the pinned statement: see its
 * header), and tests/test_bn128_oracle.py checks the two against each other:
 *   permutation      circuits.bn128/custom/poseidon.circom:6-45 (t = nInputs + 1, RF = 8, RP by t, x^5, dense constants and MDS)
 *   leaf rule        merklehash_bn128_worker.js:42-98  (three 64-bit words per element, chunks of `arity`, chained capacity)
 *   tree / layout    merklehash_bn128_p.js:31-45, 87-129 (levels padded to a multiple of arity, the root alone at the end)
 * The round constants and the MDS matrix of each width are handed in by the Python side (its Grain-LFSR generator). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef struct { u64 w[4]; } fr;

static const fr R_ = { { 0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull } };
static const fr R2 = { { 0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull } };   /* 2^512 mod r */
static const u64 N0 = 0xc2e1f593efffffffull;                                                                                  /* -r^-1 mod 2^64 */

static int ge(const fr *a, const fr *b) { for (int i = 3; i >= 0; i--) if (a->w[i] != b->w[i]) return a->w[i] > b->w[i]; return 1; }
static void sub_(fr *r, const fr *a, const fr *b) { u64 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a->w[i] - b->w[i] - br; r->w[i] = (u64)d; br = (u64)(d >> 64) & 1; } }
static void addmod(fr *r, const fr *a, const fr *b) {
    u64 c = 0; fr t;
    for (int i = 0; i < 4; i++) { u128 s = (u128)a->w[i] + b->w[i] + c; t.w[i] = (u64)s; c = (u64)(s >> 64); }
    if (c || ge(&t, &R_)) sub_(&t, &t, &R_);
    *r = t;
}
static void mont(fr *r, const fr *a, const fr *b) {       /* a b / 2^256 mod r (CIOS) */
    u64 t[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        u64 c = 0;
        for (int j = 0; j < 4; j++) { u128 x = (u128)a->w[j] * b->w[i] + t[j] + c; t[j] = (u64)x; c = (u64)(x >> 64); }
        u128 x = (u128)t[4] + c; t[4] = (u64)x; t[5] = (u64)(x >> 64);
        u64 m = t[0] * N0;
        c = (u64)(((u128)m * R_.w[0] + t[0]) >> 64);
        for (int j = 1; j < 4; j++) { u128 y = (u128)m * R_.w[j] + t[j] + c; t[j - 1] = (u64)y; c = (u64)(y >> 64); }
        x = (u128)t[4] + c; t[3] = (u64)x; t[4] = t[5] + (u64)(x >> 64);
    }
    fr o = { { t[0], t[1], t[2], t[3] } };
    if (t[4] || ge(&o, &R_)) sub_(&o, &o, &R_);
    *r = o;
}
static void to_mont(fr *r, const fr *a) { mont(r, a, &R2); }
static void from_mont(fr *r, const fr *a) { fr one = { { 1, 0, 0, 0 } }; mont(r, a, &one); }

#define MAXT 17
static struct { int t, rp; fr *C, *M; } P_[MAXT + 1];
static const int N_ROUNDS_P[16] = { 56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68 };

/* constants of width t in normal form: C[(8+rp)*t][4 words], M[t*t][4 words] */
int bn_set_params(int t, const u64 *C, const u64 *M) {
    if (t < 2 || t > MAXT) return 1;
    int rp = N_ROUNDS_P[t - 2], nc = (8 + rp) * t;
    free(P_[t].C); free(P_[t].M);
    P_[t].C = (fr *)malloc(sizeof(fr) * nc); P_[t].M = (fr *)malloc(sizeof(fr) * t * t);
    for (int i = 0; i < nc; i++) { fr v; memcpy(&v, C + 4 * i, 32); to_mont(&P_[t].C[i], &v); }
    for (int i = 0; i < t * t; i++) { fr v; memcpy(&v, M + 4 * i, 32); to_mont(&P_[t].M[i], &v); }
    P_[t].t = t; P_[t].rp = rp;
    return 0;
}
int bn_have_params(int t) { return t >= 2 && t <= MAXT && P_[t].t == t; }

static void pow5(fr *x) { fr x2, x4; mont(&x2, x, x); mont(&x4, &x2, &x2); mont(x, &x4, x); }
/* st[t] in Montgomery form, in place */
static void perm(int t, fr *st) {
    const fr *C = P_[t].C, *M = P_[t].M;
    int rp = P_[t].rp;
    fr nx[MAXT];
    for (int r = 0; r < 8 + rp; r++) {
        for (int j = 0; j < t; j++) addmod(&st[j], &st[j], &C[t * r + j]);
        if (r < 4 || r >= 4 + rp) { for (int j = 0; j < t; j++) pow5(&st[j]); } else pow5(&st[0]);
        for (int i = 0; i < t; i++) {
            fr acc = { { 0, 0, 0, 0 } }, p;
            for (int j = 0; j < t; j++) { mont(&p, &M[i * t + j], &st[j]); addmod(&acc, &acc, &p); }
            nx[i] = acc;
        }
        memcpy(st, nx, sizeof(fr) * t);
    }
}
/* poseidon(inputs[n], initState) -> first output; everything Montgomery */
static void hash_chunk(fr *out, const fr *in, int n, const fr *init) {
    fr st[MAXT];
    st[0] = *init;
    memcpy(st + 1, in, sizeof(fr) * n);
    perm(n + 1, st);
    *out = st[0];
}
/* normal-form interface for tests: state of t elements (4 words each), nOut outputs */
int bn_poseidon(int nInputs, const u64 *inputs, const u64 *initState, int nOut, u64 *out) {
    int t = nInputs + 1;
    if (!bn_have_params(t)) return 1;
    fr st[MAXT], v;
    memcpy(&v, initState, 32); to_mont(&st[0], &v);
    for (int i = 0; i < nInputs; i++) { memcpy(&v, inputs + 4 * i, 32); to_mont(&st[i + 1], &v); }
    perm(t, st);
    for (int i = 0; i < nOut; i++) { from_mont(&v, &st[i]); memcpy(out + 4 * i, &v, 32); }
    return 0;
}
/* leaf digest of one row (merklehash_bn128_worker.js:42-98), Montgomery out */
static void leaf(fr *out, const u64 *row, u64 width, int arity, int custom) {
    fr v = { { 0, 0, 0, 0 } };
    if (width <= 4) { for (u64 i = 0; i < width; i++) v.w[i] = row[i]; if (ge(&v, &R_)) { /* reduce: value < 2^256 < 6r */ while (ge(&v, &R_)) sub_(&v, &v, &R_); } to_mont(out, &v); return; }
    u64 nEl = (width + 2) / 3;
    fr st = { { 0, 0, 0, 0 } }, chunk[16];
    for (u64 e0 = 0; e0 < nEl; e0 += (u64)arity) {
        int n = (int)(nEl - e0 < (u64)arity ? nEl - e0 : (u64)arity);
        for (int k = 0; k < n; k++) {
            fr e = { { 0, 0, 0, 0 } };
            for (int q = 0; q < 3; q++) { u64 c = 3 * (e0 + k) + q; if (c < width) e.w[q] = row[c]; }
            to_mont(&chunk[k], &e);
        }
        if (n < arity && custom) { for (int k = n; k < arity; k++) memset(&chunk[k], 0, sizeof(fr)); n = arity; }
        fr o; hash_chunk(&o, chunk, n, &st); st = o;
    }
    *out = st;
}
u64 bn_merkle_num_nodes(u64 height, int arity) {
    u64 n = height, nxt = (n - 1) / arity + 1, acc = nxt * arity;
    while (n > 1) { n = nxt; nxt = (n - 1) / arity + 1; acc += n > 1 ? nxt * arity : 1; }
    return acc;
}
/* the whole tree; nodes: bn_merkle_num_nodes(height) x 4 words, MONTGOMERY form (what tree.nodes holds in the reference);
 * returns 1 when the constants of a needed width have not been set */
int bn_merkelize(const u64 *rows, u64 width, u64 height, int arity, int custom, u64 *nodes) {
    u64 total = bn_merkle_num_nodes(height, arity);
    memset(nodes, 0, total * 32);
    int bad = 0;
    u64 nEl = (width + 2) / 3;
    if (width > 4) {
        for (u64 e0 = 0; e0 < nEl; e0 += (u64)arity) { u64 n = nEl - e0 < (u64)arity ? nEl - e0 : (u64)arity; if (custom) n = (u64)arity; if (!bn_have_params((int)n + 1)) bad = 1; }
    }
    if (height > 1 && !bn_have_params(arity + 1)) bad = 1;
    if (bad) return 1;
#pragma omp parallel for schedule(static)
    for (u64 i = 0; i < height; i++) { fr d; leaf(&d, rows + i * width, width, arity, custom); memcpy(nodes + 4 * i, &d, 32); }
    u64 pIn = 0, n = height, nxt = (n - 1) / arity + 1, pOut = pIn + nxt * arity;
    fr zero = { { 0, 0, 0, 0 } };
    while (n > 1) {
#pragma omp parallel for schedule(static)
        for (u64 i = 0; i < nxt; i++) { fr o; hash_chunk(&o, (const fr *)(nodes + 4 * (pIn + i * arity)), arity, &zero); memcpy(nodes + 4 * (pOut + i), &o, 32); }
        n = nxt; nxt = (n - 1) / arity + 1; pIn = pOut; pOut = pIn + nxt * arity;
    }
    return 0;
}
