#!/usr/bin/env python3
"""Writes tests/golden/fib_flow.json: the synthetic Fibonacci AIR in the reference's pilInfo / expressionsInfo shape,
its witness, and the proof the CPU checker backend produces for it -- the expected output of tests/js/prove_flow.js,
which drives the same proof from Node through the JS drop-in modules."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "pil2-stark-js_amd", "python")]
import gl_oracle
gl_oracle.build()
from pil2gl import stark
from stark_backend import OracleBackend


def s(v):
    if isinstance(v, dict):
        return {k: s(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [s(x) for x in v]
    if isinstance(v, int) and not isinstance(v, bool) and (v > 2 ** 31 or v < 0):
        return str(v)
    return v


def strs(v):
    if isinstance(v, dict):
        return {k: strs(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [strs(x) for x in v]
    if isinstance(v, int) and not isinstance(v, bool):
        return str(v)
    return v


def write(name, hash_commits, prev_row=False, im_pols=False, boundaries=False, perm_copies=0):
    n_bits, pairs = 6, 2
    ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 9}, {"nBits": 5}, {"nBits": 2}]}
    if hash_commits:
        ss["hashCommits"] = True                  # the transcript absorbs hashes of publics / evaluations / last polynomial
    if perm_copies:                                   # two witness stages; the stage-2 column comes from a gprod hint in the reference's shape
        info, exprs, vinfo = stark.permutation_air(ss, perm_copies, ref_hints=True)
        cm, consts, publics = stark.permutation_trace(n_bits, copies=perm_copies)
    else:
        info, exprs, vinfo = stark.fibonacci_air(pairs, ss, prev_row, im_pols=im_pols, boundaries=boundaries)
        cm, consts, publics = stark.fibonacci_trace(n_bits, pairs, im_pols=im_pols)  # (im_pols: the witness goes out with those columns empty)
    be = OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)          # (from_host copies: cm itself stays as generated)
    out = {"pilInfo": info, "expressionsInfo": exprs, "cm1": [str(int(v)) for v in cm.reshape(-1)], "consts": [str(int(v)) for v in consts.reshape(-1)],
           "publics": [str(v) for v in publics], "constRoot": [str(v) for v in setup["constRoot"]],
           "proof": json.loads(json.dumps(res["proof"], default=int)), "challenges": res["challenges"], "queries": res["queries"],
           # what starkVerify takes besides the proof (stark_verify.js:8): the two verifier programs, in the reference's shape
           "verifierInfo": {"qVerifier": vinfo["qVerifier"], "queryVerifier": stark.query_verifier_of(info, exprs)}}
    for k in ("proof", "challenges"):
        out[k] = strs(out[k])
    json.dump(out, open(os.path.join(ROOT, "tests/golden", name), "w"))
    print("ok", name, os.path.getsize(os.path.join(ROOT, "tests/golden", name)))


write("fib_flow.json", False)
write("fib_flow_hashcommits.json", True)
# an AIR that also reads the PREVIOUS row: openings [-1, 0, 1], FRI polynomial terms in the reference's key order 0, 1, -1
write("fib_flow_prevrow.json", False, True)
# intermediate polynomials: imPolsCode with destinations of type cm, filled by the prover on the trace domain (prover.js:212-214)
write("fib_flow_impols.json", False, False, True)
# pil2 boundaries (everyFrame, firstRow, lastRow) instead of selector constants: one zerofier column of Zi_ext per boundary
write("fib_flow_boundaries.json", False, False, True, True)
# two witness stages: expressionsInfo.hintsInfo in the reference's shape, numerator / denominator as expressions (hints_helpers.js:21-33,102-113)
# boundaries alone (no intermediate polynomials)
write("fib_flow_boundaries_only.json", False, False, False, True)
write("perm_flow_hints.json", False, perm_copies=2)
