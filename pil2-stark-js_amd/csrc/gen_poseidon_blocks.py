#!/usr/bin/env python3
"""Tables for the blocked partial rounds of the Goldilocks Poseidon-12 permutation (poseidon_blocks.cuh).

Spec of the permutation: the un-optimised 30-round form of the reference's WASM kernel, src/helpers/glwasm.js:216-426
(round constants :535-627, MDS :428-440), as poseidon_gl.cuh states it.  This script only re-arranges the LINEAR part
of the 22 partial rounds; it reads the round constants from poseidon_gl_constants.inc (generated from the reference's
table by oracle/gen_constants.js) and writes poseidon_gl_blocks.inc.

The re-arrangement.  A partial round is  x0 <- S(x0 + c);  x <- M x.  Take K = 4 rounds as one block: with
y = (S(x0 + c), x1..x11) the state after the first S-box and d_j = S(t_j + c_j) - t_j the increment of the j-th later S-box,
    t_r   = (M^r y)[0] + sum_{1<=j<r} d_j (M^(r-j))[0][0]          r = 1..3      (the S-box inputs)
    x_out = M^4 y + sum_{j=1..3} d_j M^(4-j) e0
M^4 has 30-bit entries, so "M^4 y" is the same byte-plane product the single MDS layer is (poseidon_mds_mfma.cuh) with
4 signed base-256 digits per coefficient: 7 planes per 32-bit word instead of 4, ONE recombination per element per FOUR
rounds, 54 matrix instructions instead of 72.  The matrix-instruction operands (A: coefficient digits per lane) are
tabulated here in the exact register layout; the affine constants every biased accumulator leaves behind are pushed
through the linear maps and folded into the round constants (S-box addends) and into round 26's constants, so the device
adds nothing at run time.

`python gen_poseidon_blocks.py` writes the .inc; `--check N` also runs the integer model of the device arithmetic
(byte planes, biased accumulators, the recombination exactly as the kernel does it, range assertions) against the plain
permutation on N random states and the reference's known answers (tests/golden/poseidon.json).
"""
import os
import re
import sys
import random

P = 0xFFFFFFFF00000001
MC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
M = [[MC[(j - i) % 12] + (8 if i == 0 and j == 0 else 0) for j in range(12)] for i in range(12)]
HERE = os.path.dirname(os.path.abspath(__file__))
K = 4                       # rounds per block
NBLK = 5                    # blocks; the remaining 22 - 20 rounds keep the single-layer form
ACC_BIAS = 0x40000000       # every accumulator starts at 2^30 (the inline constant 2.0f's bit pattern)


def matmul(a, b):
    return [[sum(a[i][k] * b[k][j] for k in range(12)) for j in range(12)] for i in range(12)]


MP = [[[int(i == j) for j in range(12)] for i in range(12)]]
for _ in range(K):
    MP.append(matmul(MP[-1], M))           # exact integers: M^4 < 2^31


def read_constants():
    txt = open(os.path.join(HERE, "poseidon_gl_constants.inc")).read()
    out = {}
    for name in ("POSEIDON_GL_RC", "POSEIDON_GL_PARTIAL_C0", "POSEIDON_GL_RC26F"):
        m = re.search(name + r"\[\d+\] = \{(.*?)\};", txt, re.S)
        out[name] = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", m.group(1))]
    return out


CONST = read_constants()
RC, PC0, RC26F = CONST["POSEIDON_GL_RC"], CONST["POSEIDON_GL_PARTIAL_C0"], CONST["POSEIDON_GL_RC26F"]


def sbox(x):
    return pow(x % P, 7, P)


def mds(v):
    return [sum(M[i][j] * v[j] for j in range(12)) % P for i in range(12)]


def perm_plain(st):
    """the permutation as poseidon_gl.cuh's vector form runs it (folded partial constants)"""
    st = [v % P for v in st]
    for r in range(4):
        st = mds([sbox(st[i] + RC[12 * r + i]) for i in range(12)])
    for r in range(22):
        st[0] = sbox(st[0] + PC0[r])
        st = mds(st)
    for r in range(26, 30):
        rc = RC26F if r == 26 else RC[12 * r:12 * r + 12]
        st = mds([sbox(st[i] + rc[i]) for i in range(12)])
    return st


def perm_textbook(st):
    st = [v % P for v in st]
    for r in range(30):
        st = [(st[i] + RC[12 * r + i]) % P for i in range(12)]
        if r < 4 or r >= 26:
            st = [sbox(v) for v in st]
        else:
            st[0] = sbox(st[0])
        st = mds(st)
    return st


def signed_digits(v, nd):
    d = []
    for _ in range(nd):
        b = v & 255
        if b >= 128:
            b -= 256
        d.append(b)
        v = (v - b) >> 8
    assert v == 0, "coefficient does not fit its digits"
    return d


# ---- operand tables -------------------------------------------------------------------------------------------------
# A matrix instruction takes, per lane, 16 coefficient bytes (operand A) and 16 data bytes (operand B) and leaves 16 i32
# accumulators per lane.  Seen from ONE lane (poseidon_mds_mfma.cuh explains why a lane only sees its own data):
#     acc[v] += sum_{slot<16} A_v[slot] * sbyte(B[slot]),   v = 0..15,   slot = 4*e + b  (byte b of the e-th word of B)
# A "row set" is one accumulator vector; its logical rows v are (output element, plane) pairs.
# Row sets 0..5: outputs 2s, 2s+1; v = 8*ii + p, planes p = 0..6 (v = 7, 15 idle).
# Row set 6: the S-box inputs u_r = (M^r y)[0]: u1 planes 0..3 at v = 0..3, u2 planes 0..4 at v = 4..8, u3 planes 0..5 at v = 9..14.
# K groups 0..2: state elements 4t..4t+3; K group 3 (row sets 0..5 only): the increments d_1..d_3 (slot e = j-1).
def logical_rows():
    """per row set: list of 16 entries (coefficient row of which matrix power, which output row, plane) or None"""
    sets = []
    for s in range(6):
        rows = [None] * 16
        for ii in range(2):
            for p in range(K + 3):
                rows[8 * ii + p] = (K, 2 * s + ii, p)
        sets.append(rows)
    rows = [None] * 16
    v = 0
    for r in range(1, K):
        for p in range(r + 3):
            rows[v] = (r, 0, p)
            v += 1
    sets.append(rows)
    return sets


ROWSETS = logical_rows()


def a_row(rowdesc, t):
    """the 16 coefficient bytes of one logical row for K group t"""
    out = [0] * 16
    if rowdesc is None:
        return out
    power, o, p = rowdesc
    for e in range(4):
        if t < 3:
            coef, nd = MP[power][o][4 * t + e], power
        else:
            j = e + 1                                  # increment d_j multiplies column 0 of M^(K-j)
            if power != K or j >= K:
                continue
            coef, nd = MP[K - j][o][0], K - j
        dg = signed_digits(coef, nd)
        for b in range(4):
            if 0 <= p - b < nd:
                out[4 * e + b] = dg[p - b]
    return out


def lane_operand(s, t, lane):
    """operand A of lane `lane` (0..63) for (row set s, K group t): 4 dwords.
    v_mfma_i32_32x32x32_i8: lane l holds row i = l%32 of A, K slots 16*(l/32)..+15; the result rows of lane-half h are
    {8q + 4h + r}: logical row v = 4q + r of half h is matrix row i = 8q + 4h + r, which must only see K group h."""
    i, g = lane & 31, lane >> 5
    h = (i >> 2) & 1
    if h != g:
        return [0, 0, 0, 0]
    v = 4 * (i >> 3) + (i & 3)
    row = a_row(ROWSETS[s][v], t)
    return [sum((row[4 * e + b] & 255) << (8 * b) for b in range(4)) for e in range(4)]


def table_index():
    """(row set, K group) in table order"""
    idx = []
    for s in range(7):
        for t in range(3):
            idx.append((s, t))
    for s in range(6):
        idx.append((s, 3))
    return idx


# Single layers (the eight full rounds and the partial rounds left over by the blocks) on the same biased accumulators:
# row set s = outputs 4s..4s+3, logical row v = 4*ii + b (plane b of output 4s+ii), K group t = elements 4t..4t+3; the
# circulant makes the operand depend on (t - s) mod 3 only, except M[0][0]'s +8 (s = t = 0): operands L0, L1, L2, L00.
def layer_row(s, t, v):
    ii, b = v >> 2, v & 3
    out = [0] * 16
    for e in range(4):
        out[4 * e + b] = M[4 * s + ii][4 * t + e]
    return out


def layer_operand(d, special, lane):
    """operand A of a single layer for (t - s) mod 3 = d (special: s = t = 0, with the +8), lane's 4 dwords"""
    i, g = lane & 31, lane >> 5
    if ((i >> 2) & 1) != g:
        return [0, 0, 0, 0]
    v = 4 * (i >> 3) + (i & 3)
    row = layer_row(0, d, v)                  # s = 0, t = d: coefficients MC[(4d + e - ii) mod 12] (+8 at [0][0])
    if not special and d == 0 and v < 4:      # the plain (t - s) = 0 operand has no +8
        row = [c - (8 if (k >> 2) == 0 and (k & 3) == (v & 3) and (v >> 2) == 0 else 0) for k, c in enumerate(row)]
    return [sum((row[4 * e + b] & 255) << (8 * b) for b in range(4)) for e in range(4)]


def layer_device(x):
    """one MDS layer on device representatives (any u64), biased accumulators: device words out"""
    lo, hi = [v & 0xFFFFFFFF for v in x], [v >> 32 for v in x]
    out = [0] * 12
    for s in range(3):
        accs = []
        for words in (lo, hi):
            acc = [ACC_BIAS] * 16
            for t in range(3):
                data = []
                for e in range(4):
                    data += sbytes(words[4 * t + e])
                for v in range(16):
                    row = layer_row(s, t, v)
                    acc[v] += sum(row[k] * data[k] for k in range(16))
            accs.append([a & 0xFFFFFFFF for a in acc])
        for ii in range(4):
            xs, ys = recombine_xy(accs[0][4 * ii:4 * ii + 4], accs[1][4 * ii:4 * ii + 4], 4)
            out[4 * s + ii] = finish(xs, ys)
    return out


def layer_error():
    """what a biased single layer adds to every true output (a constant per output)"""
    z = layer_device([0] * 12)
    return [v % P for v in z]


# ---- integer model of the device arithmetic ----------------------------------------------------------------------------
def sbytes(word):
    return [((word >> (8 * b)) & 255) - 128 for b in range(4)]


def mfma_rowset(s, words_lo_or_hi, inj_words):
    """accumulators of row set s for one 32-bit half: words = 12 state words, inj_words = 4 increment words (or None)"""
    acc = [ACC_BIAS] * 16
    for t in range(3 if inj_words is None else 4):
        data = []
        for e in range(4):
            w = words_lo_or_hi[4 * t + e] if t < 3 else inj_words[e]
            data += sbytes(w)
        for v in range(16):
            row = a_row(ROWSETS[s][v], t)
            q = sum(row[k] * data[k] for k in range(16))
            acc[v] += q
    for v in range(16):
        assert abs(acc[v] - ACC_BIAS) <= (1 << 21), "plane out of the range the recombination assumes"
    return [a & 0xFFFFFFFF for a in acc]


def recombine_xy(ql, qh, npl):
    """planes (biased, u32) of the low and high words -> (X signed, Y unsigned) with value = X + 2^32 Y, as the kernel forms them"""
    def pairs(q):
        a = []
        for i in range(0, npl, 2):
            if i + 1 < npl:
                a.append((q[i] + (q[i + 1] << 8)) & 0xFFFFFFFF)
            else:
                a.append(q[i])
        return a + [0] * (4 - len(a))
    al, ah = pairs(ql), pairs(qh)
    c = [al[0], al[1], al[2] + ah[0], al[3] + ah[1], ah[2], ah[3]]
    e2, e3 = c[2] + c[4], c[3] + c[5]
    assert e2 < (1 << 32) and e3 < (1 << 32)
    d0, d1 = c[0] - c[4], c[1] - c[5]
    assert abs(d0) < (1 << 31) and abs(d1) < (1 << 31)
    x = d0 + 65536 * d1
    y = e2 + 65536 * e3
    return x, y


def finish(x, y):
    """X + 2^32 Y -> a 64-bit representative the way the kernel does it: tt = Y_hi (2^32-1) + X, then + Y_lo 2^32 with one wrap"""
    tt = (y >> 32) * 0xFFFFFFFF + x
    assert 0 <= tt < (1 << 64)
    r = tt + ((y & 0xFFFFFFFF) << 32)
    if r >> 64:
        r = (r & ((1 << 64) - 1)) + 0xFFFFFFFF
        assert r < (1 << 64)
    return r


def block_device(x, chat, track=None):
    """one block of K partial rounds on device representatives x[12] (any u64); chat = the K S-box addends.
    Returns the device's output words."""
    x = list(x)
    s0 = sbox(x[0] + chat[0])
    y = [s0] + x[1:]
    lo, hi = [v & 0xFFFFFFFF for v in y], [v >> 32 for v in y]
    ul, uh = mfma_rowset(6, lo, None), mfma_rowset(6, hi, None)
    d = [None] * K
    v0 = 0
    for r in range(1, K):
        npl = r + 3
        xs, ys = recombine_xy(ul[v0:v0 + npl], uh[v0:v0 + npl], npl)
        v0 += npl
        for j in range(1, r):
            tco = MP[r - j][0][0]
            xs += (d[j] & 0xFFFFFFFF) * tco
            ys += (d[j] >> 32) * tco
        t = finish(xs, ys)
        if track is not None:
            track.append(t)
        sr = sbox(t + chat[r])
        d[r] = (sr - t) % P
    injl = [d[j] & 0xFFFFFFFF for j in range(1, K)] + [0]
    injh = [d[j] >> 32 for j in range(1, K)] + [0]
    out = [0] * 12
    for s in range(6):
        ql, qh = mfma_rowset(s, lo, injl), mfma_rowset(s, hi, injh)
        for ii in range(2):
            xs, ys = recombine_xy(ql[8 * ii:8 * ii + K + 3], qh[8 * ii:8 * ii + K + 3], K + 3)
            out[2 * s + ii] = finish(xs, ys)
    return out


def block_true(x, c):
    x = [v % P for v in x]
    ts = []
    for r in range(K):
        ts.append(x[0])
        x[0] = sbox(x[0] + c[r])
        x = mds(x)
    return x, ts


def derive_affine():
    """the constants the biased accumulators leave behind.  With every S-box fed its true input, the device block is
    affine in (y, d_1..d_3):  t_dev_r = t_true_r + EU[r] (r = 1..3), x_dev = x_true + EX + linear terms of the input's own error.
    EU / EX are read off the model at the all-zero point."""
    zero = [0] * 12
    lo = hi = zero
    ul, uh = mfma_rowset(6, lo, None), mfma_rowset(6, hi, None)
    eu, v0 = [0] * K, 0
    for r in range(1, K):
        npl = r + 3
        xs, ys = recombine_xy(ul[v0:v0 + npl], uh[v0:v0 + npl], npl)
        eu[r] = (xs + (ys << 32)) % P
        v0 += npl
    ex = [0] * 12
    for s in range(6):
        ql, qh = mfma_rowset(s, lo, [0, 0, 0, 0]), mfma_rowset(s, hi, [0, 0, 0, 0])
        for ii in range(2):
            xs, ys = recombine_xy(ql[8 * ii:8 * ii + K + 3], qh[8 * ii:8 * ii + K + 3], K + 3)
            ex[2 * s + ii] = (xs + (ys << 32)) % P
    return eu, ex


def fold_constants():
    """Round constants for the device schedule.  err = device representative - true value of the state, a known vector at every
    point: every biased layer adds layer_error(), every block its own constants; the S-box addends absorb it.
    Schedule: full rounds 0..3 (biased layers), NBLK blocks of K partial rounds, the remaining partial rounds as biased single
    layers, full rounds 26..28 (biased layers), round 29 with the EXACT layer (its outputs leave the permutation).
    Returns (rcf[8][12] for rounds 0-3, 26-29; chat[22])."""
    eu, ex = derive_affine()
    el = layer_error()
    rcf = [[RC[i] for i in range(12)]]
    err = list(el)
    for r in range(1, 4):
        rcf.append([(RC[12 * r + i] - err[i]) % P for i in range(12)])
        err = list(el)                                        # all twelve lanes pass an S-box: only the layer's own constant remains
    chat = []
    r = 0
    for _ in range(NBLK):
        c = PC0[r:r + K]
        ch = [(c[0] - err[0]) % P]
        e_y = [0] + err[1:]                                  # y = (true S-box output, x1..x11 with their errors)
        # t_dev_j = t_true_j + (M^j e_y)[0] + EU[j] + sum_{i<j} e_d[i] (M^(j-i))[0][0];  d_dev = s - t_dev: e_d[j] = -that
        e_d = [0] * K
        for j in range(1, K):
            e_t = (sum(MP[j][0][k] * e_y[k] for k in range(12)) + eu[j] + sum(e_d[i] * MP[j - i][0][0] for i in range(1, j))) % P
            ch.append((c[j] - e_t) % P)
            e_d[j] = (-e_t) % P
        err = [(sum(MP[K][o][k] * e_y[k] for k in range(12)) + ex[o] + sum(e_d[j] * MP[K - j][o][0] for j in range(1, K))) % P
               for o in range(12)]
        chat += ch
        r += K
    while r < 22:                                             # single biased layers: the error goes through M and gains the layer's own
        chat.append((PC0[r] - err[0]) % P)
        err = [(a + b) % P for a, b in zip(mds([0] + err[1:]), el)]
        r += 1
    rcf.append([(RC26F[i] - err[i]) % P for i in range(12)])
    for r in range(27, 30):
        rcf.append([(RC[12 * r + i] - el[i]) % P for i in range(12)])
    fold_constants.err_in, fold_constants.err_out = list(el), list(err)      # what rounds 4..25 expect on their input / leave on their output
    return rcf, chat


def perm_device(st, rcf, chat):
    st = [v % P for v in st]
    for r in range(4):
        st = layer_device([sbox(st[i] + rcf[r][i]) for i in range(12)])
    r = 0
    for _ in range(NBLK):
        st = block_device(st, chat[r:r + K])
        r += K
    while r < 22:
        st[0] = sbox(st[0] + chat[r])
        st = layer_device(st)
        r += 1
    for k in range(4, 7):
        st = layer_device([sbox(st[i] + rcf[k][i]) for i in range(12)])
    return mds([sbox(st[i] + rcf[7][i]) for i in range(12)])


def write_inc(rcf, chat):
    idx = table_index()
    words = []
    for (s, t) in idx:
        for lane in range(64):
            words += lane_operand(s, t, lane)
    for d, special in ((0, False), (1, False), (2, False), (0, True)):       # single-layer operands: (t - s) mod 3 = 0, 1, 2, then s = t = 0
        for lane in range(64):
            words += layer_operand(d, special, lane)
    tco = [MP[m][0][0] for m in range(1, K - 1)]              # (M^1)[0][0], (M^2)[0][0]
    with open(os.path.join(HERE, "poseidon_gl_blocks.inc"), "w") as f:
        f.write("// GENERATED by gen_poseidon_blocks.py -- operands and folded constants of the blocked partial rounds (poseidon_blocks.cuh)\n")
        f.write("#define POSEIDON_BLK_K %d\n#define POSEIDON_BLK_N %d\n#define POSEIDON_BLK_OPERANDS %d\n" % (K, NBLK, len(idx) + 4))
        f.write("#define POSEIDON_BLK_LAYER_OPERAND %d      // single-layer operands: +0, +1, +2 by (t - s) mod 3, +3 for s = t = 0\n" % len(idx))
        f.write("// (M^m)[0][0], m = 1..%d\n" % (K - 2))
        f.write("#define POSEIDON_BLK_T1 %du\n#define POSEIDON_BLK_T2 %du\n" % (tco[0], tco[1]))
        f.write("// operand A per (row set, K group) and lane: [operand][lane][4 dwords]; order: row sets 0..6 x state groups 0..2, then row sets 0..5 x increments, then the single layer's four\n")
        f.write("POSEIDON_GL_RC_QUAL const uint32_t POSEIDON_BLK_A[%d] = {\n" % len(words))
        for i in range(0, len(words), 16):
            f.write("    " + ", ".join("0x%08xu" % w for w in words[i:i + 16]) + ",\n")
        f.write("};\n// S-box addends of the 22 partial rounds for the blocked schedule, and the constants of the full rounds 0..3, 26..29\n")
        f.write("POSEIDON_GL_RC_QUAL const uint64_t POSEIDON_BLK_C0[22] = {\n    " + ", ".join("0x%016xull" % v for v in chat) + ",\n};\n")
        for name, vec in (("POSEIDON_BLK_ERR_IN", fold_constants.err_in), ("POSEIDON_BLK_ERR_OUT", fold_constants.err_out)):
            f.write("POSEIDON_GL_RC_QUAL const uint64_t %s[12] = {\n    " % name + ", ".join("0x%016xull" % v for v in vec) + ",\n};\n")
        f.write("POSEIDON_GL_RC_QUAL const uint64_t POSEIDON_BLK_RCF[96] = {\n")
        for row in rcf:
            f.write("    " + ", ".join("0x%016xull" % v for v in row) + ",\n")
        f.write("};\n")
        # the same addends by WHERE the kernel adds them: biased layer k (rounds 0..3, 24, 25, 26..28) adds the addends of the S-boxes that follow it
        assert NBLK * K == 20
        lc = [rcf[1], rcf[2], rcf[3], [chat[0]] + [0] * 11, [chat[21]] + [0] * 11, rcf[4], rcf[5], rcf[6], rcf[7], [0] * 12]
        f.write("// what biased layer k adds to its outputs: layers of rounds 0..3, 24, 25, 26..28, then a row of zeros (the parity tests' partial-rounds-only entry)\n")
        f.write("POSEIDON_GL_RC_QUAL const uint64_t POSEIDON_BLK_LC[%d] = {\n" % (12 * len(lc)))
        for row in lc:
            f.write("    " + ", ".join("0x%016xull" % v for v in row) + ",\n")
        f.write("};\n")


def main():
    for m in range(1, K + 1):
        mx = max(max(r) for r in MP[m])
        assert mx < (1 << (8 * m - 1)), "M^%d does not fit %d signed digits" % (m, m)
    # worst case of every plane over ALL inputs (|signed byte| <= 128): the recombination adds up to three pair sums
    # a = 2^30 + q0 + 256 q1 in 32 bits and takes differences of two as i32, which holds while 257 max|q| < 2^28
    worst = 0
    for s in range(7):
        for v in range(16):
            if ROWSETS[s][v] is not None:
                worst = max(worst, 128 * sum(abs(a) for t in range(4 if s < 6 else 3) for a in a_row(ROWSETS[s][v], t)))
    assert 257 * worst < (1 << 28), "plane bound too large for the 32-bit pair sums"
    rcf, chat = fold_constants()
    write_inc(rcf, chat)
    n = 0
    if "--check" in sys.argv:
        n = int(sys.argv[sys.argv.index("--check") + 1])
    if n:
        import json
        gold = json.load(open(os.path.join(HERE, "..", "..", "tests", "golden", "poseidon.json")))
        rng = random.Random(7)
        cases = [([int(v, 16) for v in g[0]] + [int(v, 16) for v in g[1]], [int(v, 16) for v in g[2]]) for g in gold[:6]]
        for _ in range(n):
            st = [rng.randrange(P) for _ in range(12)]
            cases.append((st, None))
        edge = [0, 1, P - 1, 0xFFFFFFFF, 0xFFFFFFFF00000000, 0x8080808080808080 % P, 0x7F7F7F7F7F7F7F7F]
        for e in edge:
            cases.append(([e] * 12, None))
        for st, want in cases:
            a, b, c = perm_plain(st), perm_textbook(st), [v % P for v in perm_device(st, rcf, chat)]
            assert a == b, "folded-constant form differs from the textbook form"
            assert want is None or a == want, "plain permutation differs from the reference vector"
            assert c == a, "blocked form differs"
        print("blocked partial rounds: %d states identical to the plain permutation (6 reference vectors among them)" % len(cases))
    print("wrote poseidon_gl_blocks.inc")


if __name__ == "__main__":
    main()
