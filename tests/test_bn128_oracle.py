"""BN128 Merkle path on the CPU oracle: generated Poseidon parameters against every constant set the
reference tree holds, and every Merkle opening of a proof the reference prover wrote (test/final/verifier.proof.zkin.json,
arity 4, non-custom: t = 5 and the generated t = 4) against its roots."""
import hashlib
import json
import os

import pytest

from conftest import GOLDEN

import bn128_oracle as bn

ROOT_C = 4996530440829051953383356903338638570209318620783152464040752714611525657817   # test/final/verifier.circom:3299


def _digest(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def test_generated_constants_match_reference_sets():
    g = json.load(open(os.path.join(GOLDEN, "poseidon_bn128_constants.json")))
    checked = 0
    for t, e in g["circom"]["C"].items():
        C, _ = bn.poseidon_constants(int(t))
        assert len(C) == e["n"] and _digest(C) == e["sha256_le32"] and hex(C[0]) == e["first"] and hex(C[-1]) == e["last"]
        if "all" in e:
            assert [hex(v) for v in C] == e["all"]
        checked += 1
    for t, e in g["circom"]["M"].items():
        _, M = bn.poseidon_constants(int(t))
        flat = [x for row in M for x in row]
        assert len(flat) == e["n"] and _digest(flat) == e["sha256_le32"]
        checked += 1
    # src/final/poseidon_constants.js keys its sets loosely (C: 4,7,8,16 / M: 4,6,8,16): take t from the set's size
    for e in g["final_js"]["C"].values():
        t = next(t for t in range(2, 18) if (bn.N_ROUNDS_F + bn.N_ROUNDS_P[t - 2]) * t == e["n"])
        C, _ = bn.poseidon_constants(t)
        assert _digest(C) == e["sha256_le32"]
        checked += 1
    for e in g["final_js"]["M"].values():
        t = int(round(e["n"] ** 0.5))
        _, M = bn.poseidon_constants(t)
        assert _digest([x for row in M for x in row]) == e["sha256_le32"]
        checked += 1
    assert checked == 18


def _walk(vals, siblings, arity, custom):
    """index-free path check: the running value must be one of the group's nodes at every level"""
    value = bn.linear_hash_class(vals, arity, custom)
    idx, shift = 0, 0
    for sibs in siblings:
        group = [int(s) for s in sibs]
        assert value in group, "running hash is not a member of its group"
        idx |= group.index(value) << shift
        shift += (arity - 1).bit_length()
        value = bn.poseidon(group, 0, 1)[0]
    return value, idx


@pytest.fixture(scope="module")
def final_proof():
    return json.load(open(os.path.join(GOLDEN, "ref_final_verifier.proof.zkin.json")))


def test_reference_final_proof_merkle_paths(final_proof):
    p = final_proof
    trees = [("s0_vals1", "s0_siblings1", int(p["root1"])), ("s0_vals3", "s0_siblings3", int(p["root3"])),
             ("s0_valsQ", "s0_siblingsQ", int(p["rootQ"])), ("s0_valsC", "s0_siblingsC", ROOT_C)]
    idx0 = None
    for vk, sk, rt in trees:
        idxs = []
        for q in range(0, 32, 3):                      # 11 of the 32 queries per tree (pure-Python Poseidon)
            r, idx = _walk(p[vk][q], p[sk][q], 4, False)
            assert r == rt, (vk, q)
            idxs.append(idx)
        assert idx0 is None or idxs == idx0            # the same positions open all four stage-0 trees
        idx0 = idxs
    for s, bits in ((1, 14), (2, 11), (3, 7), (4, 4)):   # tree heights: test/final/verifier.circom:3449-3467
        for q in range(0, 32, 6):
            r, idx = _walk(p["s%d_vals" % s][q], p["s%d_siblings" % s][q], 4, False)
            assert r == int(p["s%d_root" % s]), (s, q)
            # FRI: the step tree is opened at the query position modulo its height (fri.js:96-104)
            assert idx == idx0[q // 3] % (1 << bits)


def test_reference_final_proof_transcript_gives_the_opened_positions(final_proof):
    """Fiat-Shamir replay in the order of test/final/verifier.circom:43-235 (TranscriptBN128 with 16 inputs): the query
    positions it yields are the positions at which the proof's Merkle paths were found to open"""
    p = final_proof
    T = bn.TranscriptBN128(16)
    T.put([int(x) for x in p["publics"]]); T.put(int(p["root1"])); T.getField(); T.getField()
    T.put(int(p["root2"])); T.getField(); T.getField()
    T.put(int(p["root3"])); T.getField()
    T.put(int(p["rootQ"])); T.getField()
    T.put([[int(x) for x in e] for e in p["evals"]]); T.getField(); T.getField(); T.getField()
    for s in range(1, 5):
        T.put(int(p["s%d_root" % s])); T.getField()
    T.put([[int(x) for x in e] for e in p["finalPol"]])
    ys = T.getPermutations(32, 17)
    for q in (0, 7, 31):
        _, idx = _walk(p["s0_vals1"][q], p["s0_siblings1"][q], 4, False)
        assert ys[q] == idx


def test_merkelize_layout_and_proofs():
    for arity, custom, h, w in ((16, False, 33, 9), (4, True, 20, 7), (8, False, 9, 3), (4, False, 1, 5), (16, False, 17, 100)):
        rows = [[(i * 1000 + j * 7 + 1) % bn.GL_P for j in range(w)] for i in range(h)]
        nodes = bn.merkelize(rows, arity, custom)
        assert len(nodes) == bn.merkle_num_nodes(h, arity)
        for idx in (0, h - 1, h // 2):
            mp = bn.group_proof(nodes, h, arity, idx)
            if h > 1:
                assert bn.root_from_group_proof(mp, idx, rows[idx], arity, custom) == bn.root(nodes)
    # Montgomery words round trip (tree.nodes representation)
    x = 0x123456789ABCDEF0123456789ABCDEF0123456789ABCDEF
    assert bn.from_montgomery_words(bn.to_montgomery_words(x)) == x


def test_worker_and_class_leaf_rules():
    # <= 3 values: both pack into one element; 4 values: the worker takes one 256-bit integer, the class hashes two elements
    v = [5, 6, 7]
    assert bn.linear_hash_worker(v, 16, False) == bn.linear_hash_class(v, 16, False) == 5 + (6 << 64) + (7 << 128)
    v4 = [1, 2, 3, 4]
    assert bn.linear_hash_worker(v4, 16, False) == (1 + (2 << 64) + (3 << 128) + (4 << 192)) % bn.R
    assert bn.linear_hash_class(v4, 16, False) == bn.poseidon([1 + (2 << 64) + (3 << 128), 4], 0, 1)[0]
    v = list(range(1, 60))
    assert bn.linear_hash_worker(v, 16, False) == bn.linear_hash_class(v, 16, False)
    assert bn.linear_hash_worker(v, 16, True) == bn.linear_hash_class(v, 16, True)
    assert bn.linear_hash_worker(v, 16, True) != bn.linear_hash_worker(v, 16, False)


def test_c_port_equals_python_oracle():
    """oracle/bn128_oracle.c (the config-4 cpu_baseline of bench.py) against the Python-integer statement, which the reference's
    constants and its `test/final` proof pin: permutations of several widths, leaf rules, ragged trees, custom mode"""
    import numpy as np
    import bn128_oracle as b
    rng = np.random.default_rng(11)
    for n_in in (1, 2, 3, 4, 8, 16):
        ins = [int(x) for x in rng.integers(0, 1 << 63, n_in)]
        assert b.c_poseidon(ins, 7, 3 if n_in > 1 else 2) == b.poseidon(ins, 7, 3 if n_in > 1 else 2), n_in
    big = [b.R - 1, b.R - 2, (1 << 253) + 12345]
    assert b.c_poseidon(big, b.R - 1, 4) == b.poseidon(big, b.R - 1, 4)
    for (h, w, ar, cu) in ((1, 9, 16, False), (5, 3, 16, False), (7, 4, 4, False), (40, 9, 16, False), (33, 100, 16, False),
                           (20, 21, 4, True), (17, 13, 8, False), (65, 50, 16, True), (4, 1, 2, False)):
        rows = rng.integers(0, 0xFFFFFFFF00000001, (h, w), dtype=np.uint64)
        assert b.c_merkelize(rows, ar, cu) == b.merkelize([[int(v) for v in r] for r in rows], ar, cu), (h, w, ar, cu)
    rows = np.full((3, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)                    # width <= 4: one 256-bit integer above r
    assert b.c_merkelize(rows, 16, False) == b.merkelize([[int(v) for v in r] for r in rows], 16, False)


def test_matrix_core_layers_integer_model():
    """tools/bn_mfma_model.py: the byte-digit form of the BN254 linear layers that csrc/bn_mfma.cuh runs on the matrix cores (constants folded mod r,
    signed digits, accumulator bias, one 32-bit Montgomery step, blocks of four partial rounds two to a super-block) on Python integers against the
    plain statement and against the oracle's permutation -- the arithmetic the device kernels were written from"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bn_mfma_model.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert out.stdout.count("== plain statement") == 3 and out.stdout.count("== oracle") == 4, out.stdout
