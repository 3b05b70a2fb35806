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
BIAS = 1 << 25
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
        assert all(0 < v < (1 << 26) for v in S), (min(S), max(S))
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
