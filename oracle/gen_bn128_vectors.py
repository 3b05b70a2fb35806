#!/usr/bin/env python3
"""Writes tests/golden/bn128_merkle.json: roots / leaf digests / one opening of the test/merklehash_bn128_p.test.js shapes
(pols[i][j] = i + 1000 j), computed by oracle/bn128_oracle.py (itself pinned by the reference's constants and final proof).
Used by the Node.js boundary test, which has no Python oracle at hand."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bn128_oracle as bn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
for arity, custom, N, nPols, idx in ((16, False, 256, 3, 3), (4, True, 256, 3, 3), (16, False, 256, 9, 3), (8, False, 33, 9, 32), (4, False, 20, 21, 7)):
    rows = [[i + 1000 * j for j in range(nPols)] for i in range(N)]
    nodes = bn.merkelize(rows, arity, custom)
    out.append({"arity": arity, "custom": custom, "N": N, "nPols": nPols, "idx": idx, "root": str(nodes[-1]),
                "leaf0": str(nodes[0]), "leafIdx": str(nodes[idx]),
                "proof": [[str(x) for x in lvl] for lvl in bn.group_proof(nodes, N, arity, idx)]})
pos = {"in": [str(i + 1) for i in range(16)], "init": "7", "out17": [str(x) for x in bn.poseidon(list(range(1, 17)), 7, 17)],
       "out_t3": str(bn.poseidon([1, 2], 0, 1)[0])}
json.dump({"trees": out, "poseidon": pos}, open(os.path.join(ROOT, "tests/golden/bn128_merkle.json"), "w"), indent=0)
print("ok")
