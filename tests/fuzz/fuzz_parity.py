#!/usr/bin/env python3
"""Randomised differential run of the HIP library (through the C ABI / the Python mirror) against the CPU oracle, on shapes the
parametrised tests do not enumerate: transforms of any row / column / blow-up count, coset slices, leaf hashes of any width (plain and
split), trees of any height with openings, FRI folds of any step, random evaluator programs, hint columns, row sums, BN128 trees, and WHOLE PROOFS of random starkStructs (both synthetic AIRs,
hashCommits, split leaves) whose every field must equal the proof of the same prove loop over the oracle backend.
Test infrastructure (it imports oracle/): not part of the product.   gpurun -- python tests/fuzz/fuzz_parity.py [seconds] [seed]
Prints one line per failing case (and exits 1), a count per operator otherwise."""
import os
import sys
import time
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("oracle", "tests", os.path.join("pil2-stark-js_amd", "python")):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import torch
import pil2gl
import gl_oracle as orc
from pil2gl import stark

P = 0xFFFFFFFF00000001
BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(SEED)
orc.build(); orc.set_threads(16)
pil2gl.init(0)
counts, fails = collections.Counter(), []


def field(shape):
    """uniform canonical elements with the edge values sprinkled in"""
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    edge = np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000], dtype=np.uint64)
    m = rng.random(size=shape) < 0.02
    a[m] = edge[rng.integers(0, len(edge), size=int(m.sum()))]
    return a


def check(name, ok, what):
    counts[name] += 1
    if not ok:
        fails.append((name, what))
        print("FAIL %s %s" % (name, what), flush=True)


def case_transform():
    big = os.environ.get("FUZZ_BIG") == "1"            # FUZZ_BIG=1: 2^12..2^22 rows, up to 2^27 extended elements (three- and four-pass transforms)
    nb = int(rng.integers(12, 23) if big else rng.integers(0, 15)); C = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 64, 100, 129])); eb = int(rng.integers(0, 5))
    while (C << (nb + eb)) > (1 << (27 if big else 23)):
        if eb > 1 and rng.random() < 0.5:
            eb -= 1
        else:
            nb = max(0, nb - 1)
    a = field((1 << nb, C))
    out = np.zeros_like(a)
    pil2gl.fft(a, C, nb, out); check("fft", (out == orc.fft_cols(a, nb)).all(), (nb, C))
    pil2gl.ifft(a, C, nb, out); check("ifft", (out == orc.ifft_cols(a, nb)).all(), (nb, C))
    ext = np.zeros((1 << (nb + eb), C), np.uint64)
    pil2gl.interpolate(a, C, nb, ext, nb + eb)
    want = orc.interpolate(a, nb, nb + eb)
    check("interpolate", (ext == want).all(), (nb, C, eb))
    if eb:
        cc = int(rng.choice([c for c in (1, 2, 4, 8, 16) if c <= (1 << eb)])); cb = int(rng.integers(0, (1 << eb) // cc)) * cc
        src = torch.from_numpy(a.view(np.int64)).cuda().reshape(-1)
        dst = torch.zeros((cc << nb) * C, dtype=torch.int64, device="cuda")
        pil2gl.interpolateCosets(src, C, nb, dst, nb + eb, cb, cc, None)
        got = dst.cpu().numpy().view(np.uint64).reshape(1 << nb, cc, C)
        check("interpolate_cosets", (got == want.reshape(1 << nb, 1 << eb, C)[:, cb:cb + cc, :]).all(), (nb, C, eb, cb, cc))


def case_worker_ops():
    """the reference's worker-level transform operators (fft_worker.js:6-67): random arguments against the restatement (small blocks),
    and fft_p.js's block loop over them (numpy bit reversal / transposes) against the one-call transforms (any size)"""
    import fft_worker_ref as R
    nb = int(rng.integers(1, 10)); bb = int(rng.integers(0, min(nb, 5) + 1)); layers = int(rng.integers(0, bb + 1)); s_ = int(rng.integers(layers, nb + 1)); C = int(rng.integers(1, 5))
    sp = int(rng.integers(0, (1 << nb) - (1 << bb) + 1))
    buf = field((C << bb,))
    got = buf.copy(); pil2gl.fft_block(got, sp, C, nb, s_, bb, layers)
    want = [int(x) for x in buf] if layers == 0 else R.fft_block([int(x) for x in buf], sp, C, nb, s_, bb, layers)
    check("fft_block", [int(x) for x in got] == want, (nb, s_, bb, layers, C, sp))
    h = int(rng.integers(1, 40)); w = int(rng.integers(1, 9)); buf = field((h * w,)); st, inc = int(field((1,))[0]), int(field((1,))[0])
    got = buf.copy(); pil2gl.interpolatePrepareBlock(got, w, st, inc)
    check("interpolatePrepareBlock", [int(x) for x in got] == R.interpolatePrepareBlock([int(x) for x in buf], w, st, inc), (h, w))
    nb = int(rng.integers(1, 13)); C = int(rng.choice([1, 2, 3, 8, 17, 33])); bb = int(rng.integers(1, nb + 1)); n = 1 << nb
    a = field((n, C)); want = np.zeros_like(a); pil2gl.fft(a, C, nb, want)
    br = np.array([int(format(i, "0%db" % nb)[::-1], 2) for i in range(n)])
    cur = torch.from_numpy(a[br].copy().view(np.int64)).cuda()                      # bitReverse (fft_p.js:35-42), then the rounds of :153-173 on the device
    i = 0
    while i < nb:
        sInc = min(bb, nb - i)
        for j in range(n >> bb):
            pil2gl.fft_block(cur[j << bb:(j + 1) << bb].reshape(-1), j << bb, C, nb, i + sInc, bb, sInc)
        if sInc < nb:
            cur = cur.reshape(n >> sInc, 1 << sInc, C).transpose(0, 1).contiguous().reshape(n, C)      # traspose (:20-32)
        i += bb
    check("fft over fft_block", (cur.cpu().numpy().view(np.uint64) == want).all(), (nb, C, bb))


def case_hash_tree():
    w = int(rng.integers(0, 140)); h = int(rng.choice([1, 2, 3, 5, 31, 64, 65, 257, 1000, 4097])); split = bool(rng.integers(0, 2))
    if w == 0:
        w = 1
    a = field((h, w))
    MH = pil2gl.buildMerkleHash(split)
    got = pil2gl.linearHash(a, w, split)
    want = np.array([orc.linear_hash(a[i], split) for i in range(min(h, 64))], dtype=np.uint64)
    check("linear_hash", (np.asarray(got).reshape(-1, 4)[:want.shape[0]] == want).all(), (w, h, split))
    tree = MH.merkelize(a, w, h)
    nodes = orc.merkelize(a, split)
    check("merkelize", (tree["nodes"] == nodes).all(), (w, h, split))
    if h > 1:
        idx = int(rng.integers(0, h))
        vals, mp = MH.getGroupProof(tree, idx)
        check("group_proof", vals == a[idx].tolist() and np.array(mp, dtype=np.uint64).reshape(-1, 4).tolist() == orc.group_proof(nodes, h, idx).tolist()
              and MH.verifyGroupProof(MH.root(tree), mp, idx, vals), (w, h, split, idx))


def case_fri_fold():
    from pil2gl import _lib
    pb = int(rng.integers(1, 17)); ob = int(rng.integers(max(0, pb - 6), pb)); b0 = pb + int(rng.integers(0, 4))
    pol = field((1 << pb, 3)); ch = field(3)
    sinv = orc.fri_shift_inv(b0, pb)
    out = np.zeros((1 << ob, 3), np.uint64)
    _lib.call("pil2gl_fri_fold", pil2gl._ptr(pol), pb, ob, sinv, pil2gl._ptr(ch), pil2gl._ptr(out))
    check("fri_fold", (out == orc.fri_fold(pol, ob, sinv, ch)).all(), (pb, ob, b0))


def case_proof():
    from stark_backend import OracleBackend
    nb = int(rng.integers(3, 11)); eb = int(rng.integers(1, 4)); nbe = nb + eb
    steps = [nbe]
    while steps[-1] > 3 and len(steps) < 5:
        nxt = steps[-1] - int(rng.integers(1, 6))
        if nxt < 1:
            break
        steps.append(nxt)
    air = "perm" if rng.random() < 0.35 else "fib"
    ss = {"nBits": nb, "nBitsExt": nbe, "nQueries": int(rng.integers(1, 20)), "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
    if rng.random() < 0.3:
        ss["hashCommits"] = True
    split = bool(rng.random() < 0.3)
    ss["splitLinearHash"] = split
    if air == "perm":
        copies = int(rng.integers(1, 4))
        info, exprs, _ = stark.permutation_air(ss, copies, ref_hints=[False, True, "result"][int(rng.integers(0, 3))])
        cm, consts, publics = stark.permutation_trace(nb, copies=copies)
    else:
        pairs = int(rng.integers(1, 9)); prev = bool(rng.random() < 0.3); im = bool(rng.random() < 0.3); bd = bool(rng.random() < 0.3)
        info, exprs, _ = stark.fibonacci_air(pairs, ss, prev, im_pols=im, boundaries=bd)
        cm, consts, publics = stark.fibonacci_trace(nb, pairs, im_pols=im)
    what = (air, nb, eb, steps, ss["nQueries"], bool(ss.get("hashCommits")), split, cm.shape[1])
    try:
        res = {}
        for name, be in (("gpu", stark.GpuBackend(0, split)), ("oracle", OracleBackend(split))):
            setup = stark.build_const_tree(be, consts, info)
            res[name] = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
        same = all(res["gpu"][k] == res["oracle"][k] for k in ("challenges", "challengesFRISteps", "queries", "publics")) and res["gpu"]["proof"] == res["oracle"]["proof"]
        check("proof", same, what)
    except Exception as e:                                   # an exception on one side only is a finding too
        check("proof", False, what + (repr(e)[:200],))


def case_evaluator():
    """random straight-line programs (tests/test_gpu_parity.py::_random_program) through the interpreter or the run-time compiled kernel"""
    import ctypes as C
    from pil2gl import _lib
    from test_gpu_parity import _random_program
    n_ops = int(rng.choice([1, 5, 30, 70, 200, 400])); nb = int(rng.integers(2, 18 if n_ops < 100 else 17)); ps = int(rng.integers(0, 4))
    ps = min(ps, max(0, nb - 2))
    widths = [int(rng.integers(1, 20)), int(rng.integers(3, 40)), int(rng.integers(1, 4)), 3]
    secs = [field((1 << nb, w)) for w in widths]; secs[-1][:] = 0
    scalars = field(40)
    ops, n_tmp = _random_program(rng, n_ops, widths, scalars.size, len(widths) - 1)
    ref = [x.copy() for x in secs]
    orc.eval_program(ops, n_tmp, ref, scalars, nb, ps)
    dsecs = [torch.from_numpy(x.view(np.int64)).cuda() for x in secs]
    prog = orc.make_program(ops, n_tmp, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
    csecs = (_lib.GlxSection * len(dsecs))()
    for i, x in enumerate(dsecs):
        csecs[i].ptr = x.data_ptr(); csecs[i].width = widths[i]
    ctx = _lib.GlxCtx(nb, ps, len(dsecs), scalars.size, csecs, scalars.ctypes.data_as(_lib.u64p))
    _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None)
    torch.cuda.synchronize()
    check("eval_program", all((d.cpu().numpy().view(np.uint64).reshape(r.shape) == r).all() for d, r in zip(dsecs, ref)), (n_ops, nb, ps, widths))


def case_hints():
    n = int(rng.choice([1, 2, 7, 64, 255, 2048, 2049, 40000, 100003])); dn = int(rng.choice([1, 3])); dd = int(rng.choice([1, 3]))
    num = field(n * dn); den = field(n * dd); den[den == 0] = 1
    if dd == 3:                                           # an ext denominator is zero only when all three words are: keep it simple
        den.reshape(n, 3)[:, 0] |= np.uint64(1)
    tn, td = torch.from_numpy(num.view(np.int64)).cuda(), torch.from_numpy(den.view(np.int64)).cuda()
    z = pil2gl.calculateZ(tn, td, dn, dd).cpu().numpy().view(np.uint64)
    check("gprod", np.array_equal(z, orc.gprod(num, den, dn, dd)), (n, dn, dd))
    sm = pil2gl.calculateS(tn[:dn].contiguous(), td, dn, dd).cpu().numpy().view(np.uint64)
    check("gsum", np.array_equal(sm, orc.gsum(num[:dn], den, dn, dd)), (n, dn, dd))
    dim = int(rng.choice([1, 3])); distinct = int(rng.integers(1, n + 1))
    vals = field((distinct, dim))
    t = vals[rng.integers(0, distinct, n)]
    f = t[rng.integers(0, n, n)]
    h1, h2 = pil2gl.calculateH1H2(torch.from_numpy(f.reshape(-1).copy().view(np.int64)).cuda(), torch.from_numpy(t.reshape(-1).copy().view(np.int64)).cuda(), dim)
    key = (lambda r: int(r[0])) if dim == 1 else (lambda r: tuple(int(x) for x in r))
    w1, w2 = orc.h1h2([key(r) for r in f], [key(r) for r in t])
    g1 = h1.cpu().numpy().view(np.uint64).reshape(n, dim); g2 = h2.cpu().numpy().view(np.uint64).reshape(n, dim)
    check("h1h2", [key(r) for r in g1] == w1 and [key(r) for r in g2] == w2, (n, dim, distinct))


def case_rows_dot():
    from pil2gl import _lib
    n_rows = int(rng.choice([1, 3, 63, 64, 65, 129, 300, 1030, 4097])); width = int(rng.integers(1, 260)); n_out = int(rng.integers(1, 5)); skew = int(rng.integers(0, 2))
    m = field((n_rows, width)); coef = field((n_out, width, 3))
    store = torch.zeros(n_rows * width + skew, dtype=torch.int64, device="cuda")
    dm = store[skew:]; dm.copy_(torch.from_numpy(m.view(np.int64).reshape(-1)))
    acc = torch.zeros(n_rows * n_out * 3, dtype=torch.int64, device="cuda")
    _lib.call("pil2gl_rows_dot_ext_dev", pil2gl._ptr(dm), width, n_rows, pil2gl._ptr(coef), n_out, pil2gl._ptr(acc), 0, None)
    got = acc.cpu().numpy().view(np.uint64).reshape(n_rows, n_out * 3)
    want = (m.astype(object) @ coef.astype(object).transpose(1, 0, 2).reshape(width, n_out * 3)) % P
    check("rows_dot", (got.astype(object) == want).all(), (n_rows, width, n_out, skew))


def case_bn128():
    import bn128_oracle as bo
    from pil2gl import bn128
    arity = int(rng.choice([2, 4, 8, 16])); custom = bool(rng.integers(0, 2)); w = int(rng.integers(1, 60)); h = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 70]))
    if rng.random() < 0.15:                                 # tall enough that the first tree level is a lane-per-permutation launch too (> 2 048 parents), ragged workgroups
        h = int(rng.choice([2049 * arity + 3, 40000, 4099 * arity]))
    a = field((h, w))
    MH = bn128.buildMerkleHash(arity, custom)
    tree = MH.merkelize(a, w, h)
    want = bo.c_merkelize_words(a, arity, custom)           # the C port of the Python oracle (checked against it in tests/test_bn128_oracle.py)
    got = np.asarray(tree["nodes"]).view(np.uint64).reshape(-1, 4)
    ok = got.shape == want.shape and (got == want).all()
    check("bn128_tree", ok, (arity, custom, w, h))


def case_proof_bn128():
    """whole proofs with BN128 trees and transcript (verificationHashType BN128): random arity / custom mode, small traces (the checker's
    BN254 Poseidon is Python integers)"""
    from stark_backend import OracleBackend
    nb = int(rng.integers(3, 7)); eb = int(rng.integers(1, 4)); nbe = nb + eb
    steps = [nbe]
    while steps[-1] > 2 and len(steps) < 4:
        nxt = steps[-1] - int(rng.integers(1, 5))
        if nxt < 1:
            break
        steps.append(nxt)
    arity = int(rng.choice([2, 4, 8, 16])); custom = bool(rng.integers(0, 2)); pairs = int(rng.integers(1, 4))
    ss = {"nBits": nb, "nBitsExt": nbe, "nQueries": int(rng.integers(1, 6)), "verificationHashType": "BN128", "steps": [{"nBits": b} for b in steps]}
    if rng.random() < 0.3:
        ss["hashCommits"] = True
    info, exprs, vinfo = stark.fibonacci_air(pairs, ss)
    cm, consts, publics = stark.fibonacci_trace(nb, pairs)
    what = ("bn128", nb, eb, steps, ss["nQueries"], arity, custom, pairs, bool(ss.get("hashCommits")))
    try:
        res = {}
        for name, be in (("gpu", stark.GpuBackend(0, False, "BN128", arity, custom)), ("oracle", OracleBackend(False, "BN128", arity, custom))):
            setup = stark.build_const_tree(be, consts, info)
            res[name] = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
            res[name + "_root"] = setup["constRoot"]
        same = res["gpu_root"] == res["oracle_root"] and all(res["gpu"][k] == res["oracle"][k] for k in ("challenges", "challengesFRISteps", "queries")) and res["gpu"]["proof"] == res["oracle"]["proof"]
        check("proof_bn128", same, what)
        # and the device verifier takes it -- except where the REFERENCE's own prover and verifier disagree: a 4-column stage is hashed by
        # the Merkle worker as ONE 256-bit integer (merklehash_bn128_worker.js:45-50) and by LinearHashBN.hash, which the verifier uses,
        # as two elements through Poseidon (linearhash.bn128.js:13-59): such a proof is rejected by the reference too (quirk kept)
        widths = [info["mapSectionsN"].get(k) for k in ("const", "cm1", "cm2", "cm3")]
        if same and 4 not in widths:
            gpu = stark.GpuBackend(0, False, "BN128", arity, custom)
            check("verify_bn128", stark.stark_verify(gpu, res["gpu"]["proof"], publics, res["gpu_root"], info, exprs, vinfo)[0], what)
    except Exception as e:
        check("proof_bn128", False, what + (repr(e)[:200],))


ONLY = os.environ.get("FUZZ_ONLY", "").split(",") if os.environ.get("FUZZ_ONLY") else None
CASES = [(f, w) for f, w in [(case_transform, 4), (case_worker_ops, 1), (case_hash_tree, 4), (case_fri_fold, 2), (case_proof, 3), (case_evaluator, 3), (case_hints, 2), (case_rows_dot, 3), (case_bn128, 1), (case_proof_bn128, 1)]
         if ONLY is None or f.__name__[5:] in ONLY]
t0 = time.time(); last = t0
order = [f for f, w in CASES for _ in range(w)]
i = 0
while time.time() - t0 < BUDGET:
    order[i % len(order)](); i += 1
    if time.time() - last > 30:
        last = time.time()
        print("... %d cases, %d failures, %.0f s" % (sum(counts.values()), len(fails), last - t0), flush=True)
print("fuzz_parity seed %d, %.0f s: %s; failures: %d" % (SEED, time.time() - t0, dict(counts), len(fails)))
sys.exit(1 if fails else 0)
