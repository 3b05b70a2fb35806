#!/usr/bin/env python3
"""Integer model of the BN254 Poseidon linear layers as byte-digit matrix products (the form csrc/bn_mfma.cuh runs on
v_mfma_i32_32x32x32_i8), checked against the plain modular statement.  Development aid: python tools/bn_mfma_model.py

A layer out_i = sum_j A_ij x_j (mod r) on Montgomery-form states x~ = x * 2^256:
    c[i][j][b] = A_ij * 256^b * 2^32 mod r      (plain integers, then signed base-256 digits d_k in [-128, 127], k < 32)
    S[i][k]    = 2^25 + sum_{j,b} d_k(c[i][j][b]) * (byte_b(x~_j) - 128)          one i32 accumulator per byte position
    V          = sum_k 256^k S[i][k]                                              < 2^274.01
    out~_i     = (V + m r) / 2^32 + K_i,  m = -V r^-1 mod 2^32,  K_i = (128 sum_{j,b} c[i][j][b] - sum_k 2^25 256^k) 2^-32 mod r
               < 2^242.01 + 2 r; two conditional subtractions make it canonical.
"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import bn128_oracle as O

R = O.R
MONT = 1 << 256
N0INV = (-pow(R, -1, 1 << 32)) % (1 << 32)
BIAS = 1 << 30
OFF = sum(BIAS << (8 * k) for k in range(32))


def signed_digits(c):
    d, carry = [], 0
    for k in range(32):
        v = ((c >> (8 * k)) & 255) + carry
        carry = 0
        if v >= 128:
            v -= 256
            carry = 1
        d.append(v)
    assert carry == 0 and sum(x << (8 * k) for k, x in enumerate(d)) == c
    return d


def layer_tables(A):
    """A: rows x cols of plain field elements -> (digits[i][j][b][k], K[i])"""
    rows, cols = len(A), len(A[0])
    dig = [[[signed_digits(A[i][j] * pow(256, b, R) * (1 << 32) % R) for b in range(32)] for j in range(cols)] for i in range(rows)]
    inv32 = pow(1 << 32, -1, R)
    K = []
    for i in range(rows):
        tot = sum(A[i][j] * pow(256, b, R) * (1 << 32) % R for j in range(cols) for b in range(32))
        K.append((128 * tot - OFF) * inv32 % R)
    return dig, K


def layer_apply(dig, K, x):
    """x: Montgomery-form values (any representative < 2^256) -> canonical Montgomery-form outputs"""
    out = []
    for i in range(len(dig)):
        S = [BIAS] * 32
        for j, xj in enumerate(x):
            assert 0 <= xj < MONT
            for b in range(32):
                s = ((xj >> (8 * b)) & 255) - 128
                row = dig[i][j][b]
                for k in range(32):
                    S[k] += row[k] * s
        assert all(0 < v < (1 << 31) for v in S), (min(S), max(S))
        V = sum(v << (8 * k) for k, v in enumerate(S))
        m = (V % (1 << 32)) * N0INV % (1 << 32)
        t = (V + m * R) >> 32
        assert (V + m * R) % (1 << 32) == 0
        t += K[i]
        assert t < (1 << 255)
        for _ in range(2):
            if t >= R:
                t -= R
        assert t < R
        out.append(t)
    return out


def main():
    rnd = random.Random(5)
    for t in (3, 5, 17):
        C, M = O.poseidon_constants(t)
        dig, K = layer_tables(M)
        for trial in range(3):
            x = [rnd.randrange(R) for _ in range(t)]
            if trial == 1:
                x = [R - 1] * t
            if trial == 2:
                x = [(1 << 256) - 1 - rnd.randrange(1 << 20) for _ in range(t)]      # lazy representatives, every byte 255
            xm = [v * MONT % R if trial < 2 else v for v in x]
            want = [sum(M[i][j] * xm[j] for j in range(t)) % R for i in range(t)]
            got = layer_apply(dig, K, xm)
            assert got == want, (t, trial)
        print("t = %2d: dense layer by byte-digit products == plain statement" % t)


if __name__ == "__main__":
    main()


# ------------------------------------------------------------------ the partial rounds, four to a block, rows as digit products
def derive_sparse(t):
    """bn128.hip::derive_sparse on plain integers: (C8[8][t], D[(t-1)^2], S[rp], V[rp][t-1], W[rp][t-1], m00)"""
    C, M = O.poseidon_constants(t)
    rp, n = O.N_ROUNDS_P[t - 2], t - 1
    Mh = [[M[i + 1][j + 1] for j in range(n)] for i in range(n)]
    v = [M[0][1 + j] for j in range(n)]
    w = [M[i + 1][0] for i in range(n)]

    def inv_mat(A):
        A = [row[:] + [int(i == k) for k in range(n)] for i, row in enumerate(A)]
        for c in range(n):
            p = next(r for r in range(c, n) if A[r][c])
            A[c], A[p] = A[p], A[c]
            iv = pow(A[c][c], -1, R)
            A[c] = [x * iv % R for x in A[c]]
            for r in range(n):
                if r != c and A[r][c]:
                    f = A[r][c]
                    A[r] = [(x - f * y) % R for x, y in zip(A[r], A[c])]
        return [row[n:] for row in A]
    Mhi = inv_mat(Mh)
    mv = lambda A, x: [sum(a * b for a, b in zip(row, x)) % R for row in A]
    S, V, W = [], [], []
    e = C[4 * t:5 * t]
    f = None
    for k in range(rp):
        S.append(e[0])
        Me = mv(M, [0] + e[1:])
        if k + 1 < rp:
            e = [(C[(5 + k) * t + j] + Me[j]) % R for j in range(t)]
        else:
            f = Me
    vk, wk = v, mv(Mhi, w)
    for k in range(rp):
        V.append(vk); W.append(wk)
        vk = [sum(vk[i] * Mh[i][j] for i in range(n)) % R for j in range(n)]
        wk = mv(Mhi, wk)
    D = [[int(i == j) for j in range(n)] for i in range(n)]
    for k in range(rp):
        D = [[sum(Mh[i][q] * D[q][j] for q in range(n)) % R for j in range(n)] for i in range(n)]
    C8 = [C[r * t:(r + 1) * t] for r in range(4)]
    C8 += [[(C[(4 + rp) * t + j] + f[j]) % R for j in range(t)]] + [C[(4 + rp + r) * t:(5 + rp + r) * t] for r in range(1, 4)]
    return C8, D, S, V, W, M[0][0], M


def row_tables(coefs, n_acc=1, fold=0):
    """one row: plain coefficients, one per operand -> (digits[j][b][k], K); fold: a Montgomery-form constant added to the row"""
    dig = [[signed_digits(a * pow(256, b, R) * (1 << 32) % R) for b in range(32)] for a in coefs]
    tot = sum(a * pow(256, b, R) * (1 << 32) % R for a in coefs for b in range(32))
    K = ((128 * tot - n_acc * OFF) * pow(1 << 32, -1, R) + fold) % R
    return dig, K


def row_positions(dig, x):
    S = [BIAS] * 32
    for j, xj in enumerate(x):
        assert 0 <= xj < MONT
        for b in range(32):
            s = ((xj >> (8 * b)) & 255) - 128
            for k in range(32):
                S[k] += dig[j][b][k] * s
    assert all(0 < v < (1 << 31) for v in S)
    return S


def row_finish(pos_lists, K):
    V = sum(sum(v << (8 * k) for k, v in enumerate(S)) for S in pos_lists)
    m = (V % (1 << 32)) * N0INV % (1 << 32)
    t = ((V + m * R) >> 32) + K
    assert t < (1 << 255)
    for _ in range(2):
        if t >= R:
            t -= R
    assert t < R
    return t


def sbox(x):           # Montgomery form in and out
    return pow(x, 5, R) * pow(MONT, -4, R) % R


def partial_rounds_blocked(t, x, sp):
    """x: Montgomery-form state entering the partial rounds -> state after them (before the closing D layer).
    Blocks of four rounds in SUPER-BLOCKS of two: the columns y are updated once per super-block (eight rounds); the second block's rows are
    taken on the y of the super-block's start and get the first block's S-box outputs through cross terms like their own."""
    C8, D, S, V, W, m00, M = sp
    rp, n = len(S), t - 1
    toM = lambda a: a * MONT % R
    nb = rp // 4
    x0, y = x[0], x[1:]
    x0 = (x0 + toM(S[0])) % R
    inv32 = pow(1 << 32, -1, R)
    for sb in range((nb + 1) // 2):
        halves = min(2, nb - 2 * sb)
        z = []
        for h in range(halves):
            k0 = 4 * (2 * sb + h)
            P, Ktile = [], []
            for i in range(4):                                                           # the block's rows on y at the SUPER-BLOCK's start,
                coefs = list(V[k0 + i]) + [sum(V[k0 + i][j] * W[k0 - 4 + ip][j] for j in range(n)) % R for ip in range(4 * h)]   # + the first block's z (second block)
                dig, _ = row_tables(coefs)
                P.append(row_positions(dig, y + z[:4 * h]))
                Ktile.append(128 * sum(a * pow(256, b, R) * (1 << 32) % R for a in coefs for b in range(32)) * inv32 % R)
            for i in range(4):
                z.append(sbox(x0))
                own = z[4 * h:]
                coefs = [sum(V[k0 + i][j] * W[k0 + ip][j] for j in range(n)) % R for ip in range(i)] + [m00]
                fold = toM(S[k0 + i + 1]) if k0 + i + 1 < 4 * nb else 0
                dig, K = row_tables(coefs, 2, fold)
                x0 = row_finish([P[i], row_positions(dig, own)], (K + Ktile[i]) % R)
        k0 = 8 * sb
        for j in range(n):
            dig, K = row_tables([1] + [W[k0 + i][j] for i in range(4 * halves)])
            y[j] = row_finish([row_positions(dig, [y[j]] + z)], K)
    for k in range(4 * nb, rp):                                                       # the rounds left over, as they are
        x0 = (x0 + toM(S[k])) % R
        zz = sbox(x0)
        x0 = (m00 * zz + sum(V[k][j] * y[j] for j in range(n))) % R
        y = [(y[j] + W[k][j] * zz) % R for j in range(n)]
    return [x0] + y


def check_blocked():
    rnd = random.Random(9)
    for t in (3, 6, 9, 17):                                # 57 = 14 blocks + 1 round; 60 = 15 blocks (odd: a half super-block); 63; 68 = 17 blocks
        sp = derive_sparse(t)
        C8, D, S, V, W, m00, M = sp
        n = t - 1
        toM = lambda a: a * MONT % R
        ins = [rnd.randrange(R) for _ in range(t)]
        st = [toM(a) for a in ins]
        for r in range(4):
            st = [sbox((a + toM(c)) % R) for a, c in zip(st, C8[r])]
            st = [sum(M[i][j] * st[j] for j in range(t)) % R for i in range(t)]
        st = partial_rounds_blocked(t, st, sp)
        st = [st[0]] + [sum(D[i][j] * st[1 + j] for j in range(n)) % R for i in range(n)]
        for r in range(4, 8):
            st = [sbox((a + toM(c)) % R) for a, c in zip(st, C8[r])]
            st = [sum(M[i][j] * st[j] for j in range(t)) % R for i in range(t)]
        got = [a * pow(MONT, -1, R) % R for a in st]
        assert got == O.poseidon(ins[1:], ins[0], t), t
        print("t = %2d: permutation with blocked partial rounds (rows as digit products) == oracle" % t)


if __name__ == "__main__":
    check_blocked()
