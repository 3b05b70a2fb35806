#!/usr/bin/env python3
"""The device verifier (pil2gl.stark.stark_verify: transcript, evaluation identity, batched Merkle paths, the FRI polynomial at the
query points through the device evaluator, FRI.verify) on random proofs: every proof of the GPU prover must be ACCEPTED, and the same
proof with ONE word altered anywhere -- a root, an evaluation, an opened value, a sibling, a FRI layer, the last polynomial -- must
be REJECTED (an accepted alteration would be a word the verifier does not bind).  gpurun -- python tests/fuzz/fuzz_verify.py [seconds] [seed]"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import numpy as np
from pil2gl import stark

P = 0xFFFFFFFF00000001
BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def leaves(v, path=()):
    """paths of every integer in a nested proof"""
    if isinstance(v, dict):
        for k in v:
            yield from leaves(v[k], path + (k,))
    elif isinstance(v, (list, tuple)):
        for i, x in enumerate(v):
            yield from leaves(x, path + (i,))
    elif isinstance(v, (int, np.integer)) and not isinstance(v, bool):
        yield path


def altered(proof, path):
    p = copy.deepcopy(proof)
    o = p
    for k in path[:-1]:
        if isinstance(o[k], tuple):
            o[k] = list(o[k])
        o = o[k]
    o[path[-1]] = (int(o[path[-1]]) + 1 + int(rng.integers(0, 5))) % P
    return p


t0 = time.time(); n_ok = n_alt = 0; bad = []
while time.time() - t0 < BUDGET:
    nb = int(rng.integers(3, 12)); eb = int(rng.integers(1, 4)); nbe = nb + eb
    steps = [nbe]
    while steps[-1] > 3 and len(steps) < 5:
        nxt = steps[-1] - int(rng.integers(1, 6))
        if nxt < 1:
            break
        steps.append(nxt)
    split = bool(rng.random() < 0.3); air = "perm" if rng.random() < 0.35 else "fib"
    ss = {"nBits": nb, "nBitsExt": nbe, "nQueries": int(rng.integers(1, 12)), "verificationHashType": "GL", "splitLinearHash": split, "steps": [{"nBits": b} for b in steps]}
    if rng.random() < 0.3:
        ss["hashCommits"] = True
    if air == "perm":
        copies = int(rng.integers(1, 3))
        info, exprs, vinfo = stark.permutation_air(ss, copies, ref_hints=[False, True, "result"][int(rng.integers(0, 3))])
        cm, consts, publics = stark.permutation_trace(nb, copies=copies)
    else:
        pairs = int(rng.integers(1, 6)); prev = bool(rng.random() < 0.3); im = bool(rng.random() < 0.3); bd = bool(rng.random() < 0.3)
        info, exprs, vinfo = stark.fibonacci_air(pairs, ss, prev, im_pols=im, boundaries=bd)
        cm, consts, publics = stark.fibonacci_trace(nb, pairs, im_pols=im)
    what = (air, nb, eb, steps, ss["nQueries"], split, bool(ss.get("hashCommits")))
    be = stark.GpuBackend(0, split)
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    ok, why = stark.stark_verify(be, res["proof"], publics, setup["constRoot"], info, exprs, vinfo)
    n_ok += 1
    if not ok:
        bad.append(("REJECTED a valid proof", what, why)); print(bad[-1], flush=True)
        continue
    paths = list(leaves(res["proof"]))
    for k in rng.choice(len(paths), size=min(8, len(paths)), replace=False):
        try:
            ok2, why2 = stark.stark_verify(be, altered(res["proof"], paths[int(k)]), publics, setup["constRoot"], info, exprs, vinfo)
        except Exception as e:                                # a malformed-input error is a rejection too
            ok2, why2 = False, repr(e)
        n_alt += 1
        if ok2:
            bad.append(("ACCEPTED an altered proof", what, paths[int(k)])); print(bad[-1], flush=True)
    if publics:
        wp = list(publics); i = int(rng.integers(0, len(wp))); wp[i] = (wp[i] + 1) % P
        n_alt += 1
        if stark.stark_verify(be, res["proof"], wp, setup["constRoot"], info, exprs, vinfo)[0]:
            bad.append(("ACCEPTED altered publics", what, i)); print(bad[-1], flush=True)
print("fuzz_verify: %d proofs accepted, %d single-word alterations rejected, %d findings, %.0f s" % (n_ok - sum(1 for b in bad if b[0].startswith("REJ")), n_alt - sum(1 for b in bad if b[0].startswith("ACC")), len(bad), time.time() - t0))
sys.exit(1 if bad else 0)
