"""Known-answer test against a proof the reference prover itself wrote:
/root/reference/test/compressor/verifier.proof.zkin.json (copied as a data fixture; sm_all machine,
nBits 10, nBitsExt 11, FRI steps 11/7/3, 8 queries, GL Poseidon, plain linear hash).

The Fiat-Shamir transcript is replayed in the order of
src/stark/calculateTranscriptVerify.js:7-125 (constRoot = the `rootC` constant of
test/compressor/verifier.circom:802; 0/2/2 challenges in stages 1..3, ibid. :90-104).  The query
positions it yields must open every Merkle path in the proof, and folding the opened FRI groups
(src/stark/fri.js:107-150) must reproduce the next layer's values."""
import numpy as np
from conftest import golden, P

ROOT_C = [1586467561057753308, 1229990203770229397, 10559924528244357123, 6072090729782730028]
STEPS = [11, 7, 3]
N_QUERIES = 8


def _ints(v):
    if isinstance(v, dict):
        return {k: _ints(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_ints(x) for x in v]
    return int(v)


def _replay(oracle, z):
    t = oracle.Transcript()
    t.put(ROOT_C)
    t.put(z["publics"])
    ch = {}
    for stage, nch in ((1, 0), (2, 2), (3, 2)):
        ch[stage] = [t.get_field() for _ in range(nch)]
        t.put(z["root%d" % stage])
    ch["q"] = t.get_field()
    t.put(z["root4"])
    ch["xi"] = t.get_field()
    t.put(np.array(z["evals"], dtype=np.uint64).reshape(-1))
    ch["fri"] = [t.get_field(), t.get_field()]
    fri_steps = []
    for step in range(len(STEPS)):
        fri_steps.append(t.get_field())
        if step < len(STEPS) - 1:
            t.put(z["s%d_root" % (step + 1)])
        else:
            t.put(np.array(z["finalPol"], dtype=np.uint64).reshape(-1))
    fri_steps.append(t.get_field())
    tq = oracle.Transcript()
    tq.put(fri_steps[-1])
    queries = tq.get_permutations(N_QUERIES, STEPS[0]).tolist()
    return ch, fri_steps, queries


def test_reference_proof_paths_and_folds(oracle):
    z = _ints(golden("ref_compressor_verifier.proof.zkin.json"))
    ch, fri_steps, queries = _replay(oracle, z)
    # positions found by brute force there; here they come out of the transcript
    assert queries == [891, 1628, 1228, 1991, 1856, 415, 833, 296]

    for q, idx in enumerate(queries):
        for name, root in (("1", z["root1"]), ("2", z["root2"]), ("3", z["root3"]), ("4", z["root4"]), ("C", ROOT_C)):
            vals = np.array(z["s0_vals" + name][q], dtype=np.uint64)
            sib = np.array(z["s0_siblings" + name][q], dtype=np.uint64)
            assert oracle.root_from_proof(vals, idx, sib).tolist() == root, (q, name)

    # FRI layers: tree s holds pol_s transposed into 2^STEPS[s] groups (fri.js:64-71)
    pol_bits = STEPS[0]
    for s in (1, 2):
        out_bits = STEPS[s]
        sinv0 = oracle.fri_shift_inv(STEPS[0], STEPS[s - 1])
        wi = oracle.inv(oracle.root(pol_bits))
        for q, idx0 in enumerate(queries):
            idx = idx0 % (1 << out_bits)
            vals = np.array(z["s%d_vals" % s][q], dtype=np.uint64)
            sib = np.array(z["s%d_siblings" % s][q], dtype=np.uint64)
            assert oracle.root_from_proof(vals, idx, sib).tolist() == z["s%d_root" % s], (s, q)
            # fold this single group exactly as fri.fold does for g = idx (fri.js:45-60):
            # a 2^xBits-point pol whose fold to 1 point uses sinv = shiftInv * wi^g
            group = vals.reshape(-1, 3)
            sinv_g = oracle.mul(sinv0, oracle.exp(wi, idx))
            folded = oracle.fri_fold(group, 0, sinv_g, fri_steps[s])[0].tolist()
            if s + 1 < len(STEPS):
                nxt = np.array(z["s%d_vals" % (s + 1)][q], dtype=np.uint64).reshape(-1, 3)
                grp = idx // (1 << STEPS[s + 1])        # fri.js:129-132
                assert nxt[grp].tolist() == folded, (s, q)
            else:
                assert z["finalPol"][idx] == folded, (s, q)
        pol_bits = out_bits


# ---------------------------------------------------------------------------------------------------------------------------
# The whole verification of that proof, with the two programs its verifier circuit evaluates: the constraint identity at the
# evaluation point and the FRI polynomial at the query points, transliterated op by op from the circuit the reference
# generated for it (test/compressor/verifier.circom:277-679 -> tests/golden/ref_compressor_verifier_programs.json by
# oracle/gen_compressor_verifier_programs.py).  Order of checks: stark_verify.js:8-218.
def _programs():
    g = golden("ref_compressor_verifier_programs.json")
    return g["starkInfo"], g["verifierInfo"], [int(v) for v in g["constRoot"]]


def _challenges(ch, info):
    """the challenges by stage as stark_verify.js keeps them (stage s at index s-1)"""
    n = info["nStages"]
    out = [[] for _ in range(n + 3)]
    ints = lambda c: [int(v) for v in c]
    for st in range(2, n + 1):
        out[st - 1] = [ints(c) for c in ch[st]]
    out[n] = [ints(ch["q"])]; out[n + 1] = [ints(ch["xi"])]; out[n + 2] = [ints(c) for c in ch["fri"]]
    return out


def test_reference_proof_programs_match_the_proof(oracle):
    """the evaluation identity holds for the reference's proof, and the FRI polynomial computed at each query point from the
    opened rows is the value the first FRI layer opens there (stark_verify.js:95-152, :180-215; fri.js:118-136) -- on host
    big integers through the restated interpreter, and through the oracle's evaluator"""
    from pil2gl import stark
    info, vinfo, root_c = _programs()
    assert root_c == ROOT_C and [s["nBits"] for s in info["starkStruct"]["steps"]] == STEPS
    z = _ints(golden("ref_compressor_verifier.proof.zkin.json"))

    def replay3():                                          # stage 3 draws three challenges (verifier.circom:104-106)
        t = oracle.Transcript()
        t.put(ROOT_C); t.put(z["publics"])
        t.put(z["root1"])
        c2 = [t.get_field(), t.get_field()]; t.put(z["root2"])
        c3 = [t.get_field(), t.get_field(), t.get_field()]; t.put(z["root3"])
        return c2, c3
    ch, fri_steps, queries = _replay(oracle, z)
    c2, c3 = replay3()
    assert [list(c) for c in c2] == [list(c) for c in ch[2]] and [list(c) for c in c3[:2]] == [list(c) for c in ch[3]]
    ch[3] = c3
    challenges = _challenges(ch, info)
    nb, nbe = info["starkStruct"]["nBits"], info["starkStruct"]["nBitsExt"]
    xi = [int(v) for v in ch["xi"]]
    xN = stark.ext_pow(xi, 1 << nb)
    Z = stark.ext_inv([(xN[0] - 1) % P, xN[1], xN[2]])
    evals = [[int(v) for v in e] for e in z["evals"]]

    def resolve(r, row=None):
        ty = r["type"]
        if ty == "eval": return evals[r["id"]]
        if ty == "challenge": return [int(v) for v in challenges[r["stage"] - 1][r["stageId"]]]
        if ty == "public": return int(z["publics"][r["id"]])
        if ty == "number": return int(r["value"]) % P
        if ty == "Zi": return list(Z)
        if ty == "const": return int(row["C"][r["id"]])
        if ty == "xDivXSubXi": return row["x"][r["id"]]
        if ty.startswith("tree"):
            v = row[int(ty[4:])]
            return int(v[r["treePos"]]) if r["dim"] == 1 else [int(x) for x in v[r["treePos"]:r["treePos"] + 3]]
        raise ValueError(ty)
    lhs = stark.execute_code(vinfo["qVerifier"]["code"], resolve)
    q_ev = [k for k, e in enumerate(info["evMap"]) if e["type"] == "cm" and info["cmPolsMap"][e["id"]]["stage"] == info["nStages"] + 1]
    q, xAcc = [0, 0, 0], [1, 0, 0]
    for k in q_ev:
        q = [(a + b) % P for a, b in zip(q, stark.ext_mul(xAcc, evals[k]))]; xAcc = stark.ext_mul(xAcc, xN)
    assert lhs == q, "Invalid evaluations"
    keep = evals
    evals = [list(e) for e in keep]; evals[12][1] = (evals[12][1] + 1) % P          # a witness column's evaluation
    assert stark.execute_code(vinfo["qVerifier"]["code"], resolve) != q
    evals = keep
    # the FRI polynomial at the query points
    wN, wE = int(oracle.root(nb)), int(oracle.root(nbe))
    for qi, idx in enumerate(queries):
        x = 7 * pow(wE, idx, P) % P
        xd = []
        for o in info["openingPoints"]:
            w = pow(wN, o, P)
            den = [(x - xi[0] * w) % P, (-xi[1] * w) % P, (-xi[2] * w) % P]
            xd.append([v * x % P for v in stark.ext_inv(den)])
        row = {1: z["s0_vals1"][qi], 2: z["s0_vals2"][qi], 3: z["s0_vals3"][qi], 4: z["s0_vals4"][qi], "C": z["s0_valsC"][qi], "x": xd}
        val = stark.execute_code(vinfo["queryVerifier"]["code"], lambda r: resolve(r, row))
        grp = np.array(z["s1_vals"][qi], dtype=np.uint64).reshape(-1, 3)
        assert [int(v) for v in grp[idx >> STEPS[1]]] == val, qi
    # the same query program through the ORACLE's expression evaluator (gl_oracle.c eval_program -- the checker the device
    # evaluator is compared with everywhere else): pinned here by values the reference prover committed to
    import stark_ref
    from pil2gl import io
    proof = io.zkin2proof(z, info)
    vals = stark.fri_values_at_queries(stark_ref.OracleBackend(), info, None, vinfo, proof, z["publics"], challenges, queries)
    for qi, idx in enumerate(queries):
        grp = np.array(z["s1_vals"][qi], dtype=np.uint64).reshape(-1, 3)
        assert [int(v) for v in grp[idx >> STEPS[1]]] == [int(v) for v in vals[qi]], qi


import pytest


@pytest.mark.gpu
def test_reference_proof_full_device_verification(oracle):
    """pil2gl.stark.stark_verify -- device transcript, evaluation identity, one batched opening check per tree, the query-point
    program on the device evaluator, FRI.verify -- accepts the proof the reference prover wrote, and rejects it once altered"""
    import copy
    from pil2gl import stark, io
    info, vinfo, root_c = _programs()
    z = golden("ref_compressor_verifier.proof.zkin.json")
    proof = io.zkin2proof(z, info)
    publics = [int(v) for v in z["publics"]]
    assert io.proof2zkin(proof, info) == {k: v for k, v in _ints(z).items() if k != "publics"}
    be = stark.GpuBackend(0)
    ok, why = stark.stark_verify(be, proof, publics, root_c, info, None, vinfo)
    assert ok, why

    def tampered(what):
        bad = copy.deepcopy(proof)
        if what == "eval":
            bad["evals"][12][0] = (bad["evals"][12][0] + 1) % P
        elif what == "opened value":
            bad["fri"][0]["polQueries"][3][0][0][2] = (bad["fri"][0]["polQueries"][3][0][0][2] + 1) % P
        elif what == "sibling":
            bad["fri"][0]["polQueries"][2][2][1][4][0] = (bad["fri"][0]["polQueries"][2][2][1][4][0] + 1) % P
        elif what == "fri layer value":
            bad["fri"][1]["polQueries"][5][0][4] = (bad["fri"][1]["polQueries"][5][0][4] + 1) % P
        elif what == "fri layer sibling":
            bad["fri"][2]["polQueries"][0][1][0][0] = (bad["fri"][2]["polQueries"][0][1][0][0] + 1) % P
        elif what == "last polynomial":
            bad["fri"][-1][1][2] = (bad["fri"][-1][1][2] + 1) % P
        elif what == "root":
            bad["root2"][0] = (bad["root2"][0] + 1) % P
        return bad
    for what in ("eval", "opened value", "sibling", "fri layer value", "fri layer sibling", "last polynomial", "root"):
        ok, why = stark.stark_verify(be, tampered(what), publics, root_c, info, None, vinfo)
        assert not ok, what
    ok, _ = stark.stark_verify(be, proof, [publics[0], (publics[1] + 1) % P, publics[2]], root_c, info, None, vinfo)
    assert not ok


# ---------------------------------------------------------------------------------------------------------------------------
# The BN128 twin: test/final/verifier.proof.zkin.json, the only proof in the reference tree written on the BN128 hash family
# (Poseidon-BN254 trees of arity 4, BN128 transcript with 16 inputs; nBits 13, nBitsExt 17, FRI 17/14/11/7/4, 32 queries), with the
# two programs of ITS verifier circuit (test/final/verifier.circom:290-2920 and :2921-3189, the older pil-stark generator:
# 2 567 + 201 ops, `muladd` among them) -> tests/golden/ref_final_verifier_programs.json.gz by oracle/gen_final_verifier_programs.py.
# The older layout differs from calculateTranscriptVerify.js in two places, both handled by handing stark_verify the challenges as
# the reference's own fourth argument does (stark_verify.js:62-91): the constant root is not absorbed, and the query positions come
# out of the main transcript (verifier.circom:18-246), not out of a fresh one.
def _final():
    g = golden("ref_final_verifier_programs.json.gz")
    z = dict(golden("ref_final_verifier.proof.zkin.json"))
    for old, new in (("rootQ", "root4"), ("s0_valsQ", "s0_vals4"), ("s0_siblingsQ", "s0_siblings4")):
        z[new] = z.pop(old)                                   # the quotient is stage nStages+1 in today's naming (proof2zkin.js:16,45-48)
    return g["starkInfo"], g["verifierInfo"], int(g["constRoot"]), z


def _final_transcript(T, z, info):
    """Transcript() of test/final/verifier.circom:18-246 on a TranscriptBN128(16) -> the reference's `challenges` argument"""
    ints = lambda v: [int(x) for x in v]
    T.put(ints(z["publics"])); T.put(int(z["root1"])); c2 = [T.getField(), T.getField()]
    T.put(int(z["root2"])); c3 = [T.getField(), T.getField()]
    T.put(int(z["root3"])); cq = [T.getField()]
    T.put(int(z["root4"])); xi = [T.getField()]
    T.put([ints(e) for e in z["evals"]]); cf = [T.getField(), T.getField()]
    chF = [T.getField()]
    for s in range(1, len(info["starkStruct"]["steps"])):
        T.put(int(z["s%d_root" % s])); chF.append(T.getField())
    T.put([ints(e) for e in z["finalPol"]])
    queries = T.getPermutations(info["starkStruct"]["nQueries"], info["starkStruct"]["steps"][0]["nBits"])
    return {"challenges": [[], [ints(c) for c in c2], [ints(c) for c in c3], [ints(c) for c in cq], [ints(c) for c in xi], [ints(c) for c in cf]],
            "challengesFRISteps": [ints(c) for c in chF], "friQueries": [int(q) for q in queries]}


def test_final_proof_programs_match_the_proof(oracle):
    """host big integers: the constraint identity of the BN128 reference proof holds in the circuit's own form C(z) = Q(z) (z^N - 1)
    and in stark_verify.js's form; the FRI polynomial computed at each of the 32 query points from the opened rows equals the value
    the first FRI layer opens there; the same values through the oracle's expression evaluator"""
    import bn128_oracle as bn
    from pil2gl import stark, io
    info, vinfo, root_c, z = _final()
    ss = info["starkStruct"]
    assert (ss["nBits"], ss["nBitsExt"], ss["nQueries"], ss["merkleTreeArity"]) == (13, 17, 32, 4)
    assert sum(1 for c in vinfo["qVerifier"]["code"] if c["op"] == "muladd") == 129 and len(vinfo["queryVerifier"]["code"]) == 201
    tr = _final_transcript(bn.TranscriptBN128(16), z, info)
    challenges, queries = tr["challenges"], tr["friQueries"]
    nb, nbe = ss["nBits"], ss["nBitsExt"]
    xi = challenges[4][0]
    xN = stark.ext_pow(xi, 1 << nb)
    zh = [(xN[0] - 1) % P, xN[1], xN[2]]
    Z = stark.ext_inv(zh)
    evals = [[int(v) for v in e] for e in z["evals"]]
    publics = [int(v) for v in z["publics"]]

    def resolve(r, row=None):
        ty = r["type"]
        if ty == "eval": return evals[r["id"]]
        if ty == "challenge": return challenges[r["stage"] - 1][r["stageId"]]
        if ty == "public": return publics[r["id"]]
        if ty == "number": return int(r["value"]) % P
        if ty == "Zi": return list(Z)
        if ty == "const": return int(row["C"][r["id"]])
        if ty == "xDivXSubXi": return row["x"][r["id"]]
        if ty.startswith("tree"):
            v = row[int(ty[4:])]
            return int(v[r["treePos"]]) if r["dim"] == 1 else [int(x) for x in v[r["treePos"]:r["treePos"] + 3]]
        raise ValueError(ty)
    code = vinfo["qVerifier"]["code"]
    q_ev = [k for k, e in enumerate(info["evMap"]) if e["type"] == "cm" and info["cmPolsMap"][e["id"]]["stage"] == info["nStages"] + 1]
    assert q_ev == list(range(76, 83))
    q, xAcc = [0, 0, 0], [1, 0, 0]
    for k in q_ev:
        q = [(a + b) % P for a, b in zip(q, stark.ext_mul(xAcc, evals[k]))]; xAcc = stark.ext_mul(xAcc, xN)
    assert stark.execute_code(code[:-1], resolve) == stark.ext_mul(q, zh), "the circuit's form: C(z) = Q(z) Z(z)"
    assert stark.execute_code(code, resolve) == q, "Invalid evaluations"
    keep = evals
    evals = [list(e) for e in keep]; evals[41][2] = (evals[41][2] + 1) % P
    assert stark.execute_code(code, resolve) != q
    evals = keep
    wN, wE = int(oracle.root(nb)), int(oracle.root(nbe))
    step1 = ss["steps"][1]["nBits"]
    for qi, idx in enumerate(queries):
        x = 7 * pow(wE, idx, P) % P
        xd = []
        for o in info["openingPoints"]:
            w = pow(wN, o, P)
            den = [(x - xi[0] * w) % P, (-xi[1] * w) % P, (-xi[2] * w) % P]
            xd.append([v * x % P for v in stark.ext_inv(den)])
        row = {1: z["s0_vals1"][qi], 3: z["s0_vals3"][qi], 4: z["s0_vals4"][qi], "C": z["s0_valsC"][qi], "x": xd}
        val = stark.execute_code(vinfo["queryVerifier"]["code"], lambda r: resolve(r, row))
        grp = np.array([int(v) for v in z["s1_vals"][qi]], dtype=np.uint64).reshape(-1, 3)
        assert [int(v) for v in grp[idx >> step1]] == val, qi
    import stark_ref
    proof = io.zkin2proof(z, info)
    vals = stark.fri_values_at_queries(stark_ref.OracleBackend(), info, None, vinfo, proof, publics, challenges, queries)
    for qi, idx in enumerate(queries):
        grp = np.array([int(v) for v in z["s1_vals"][qi]], dtype=np.uint64).reshape(-1, 3)
        assert [int(v) for v in grp[idx >> step1]] == [int(v) for v in vals[qi]], qi


@pytest.mark.gpu
def test_final_proof_full_device_verification():
    """pil2gl.stark.stark_verify with verificationHashType BN128 accepts the proof the reference prover wrote -- BN128 transcript on the
    device permutation, evaluation identity, every arity-4 Poseidon-BN254 Merkle path in one batch per tree, the 201-op query program on the
    device evaluator, FRI.verify -- and rejects it after each of eight alterations"""
    import copy
    from pil2gl import stark, io
    info, vinfo, root_c, z = _final()
    ss = info["starkStruct"]
    be = stark.GpuBackend(0, hash_type=ss["verificationHashType"], arity=ss["merkleTreeArity"], custom=ss["merkleTreeCustom"])
    proof = io.zkin2proof(z, info)
    publics = [int(v) for v in z["publics"]]
    tr = _final_transcript(be.new_transcript(), z, info)
    ok, why = stark.stark_verify(be, proof, publics, root_c, info, None, vinfo, challenges=tr, legacy_transcript_queries=True)
    assert ok, why

    def tampered(what):
        bad = copy.deepcopy(proof)
        if what == "eval":
            bad["evals"][41][0] = (bad["evals"][41][0] + 1) % P
        elif what == "opened value":
            bad["fri"][0]["polQueries"][3][0][0][2] = (bad["fri"][0]["polQueries"][3][0][0][2] + 1) % P
        elif what == "constant":
            bad["fri"][0]["polQueries"][30][4][0][35] = (bad["fri"][0]["polQueries"][30][4][0][35] + 1) % P
        elif what == "sibling":                                   # (a level of an arity-4 path lists four nodes; the one at the path's own
            slot = ((tr["friQueries"][2] >> 8) % 4 + 1) % 4        #  position is recomputed, not read: alter a true sibling)
            bad["fri"][0]["polQueries"][2][2][1][4][slot] = int(bad["fri"][0]["polQueries"][2][2][1][4][slot]) + 1
        elif what == "fri layer value":
            bad["fri"][1]["polQueries"][5][0][4] = (bad["fri"][1]["polQueries"][5][0][4] + 1) % P
        elif what == "fri layer sibling":
            slot = ((tr["friQueries"][0] % (1 << ss["steps"][2]["nBits"])) % 4 + 1) % 4
            bad["fri"][2]["polQueries"][0][1][0][slot] = int(bad["fri"][2]["polQueries"][0][1][0][slot]) + 1
        elif what == "last polynomial":
            bad["fri"][-1][1][2] = (bad["fri"][-1][1][2] + 1) % P
        elif what == "root":
            bad["root3"] = int(bad["root3"]) + 1
        return bad
    # positions handed in by the caller are not bound to the FRI challenge: refused unless the older layout is asked for by name
    ok, why = stark.stark_verify(be, proof, publics, root_c, info, None, vinfo, challenges=tr)
    assert not ok and "query positions" in why
    for what in ("eval", "opened value", "constant", "sibling", "fri layer value", "fri layer sibling", "last polynomial", "root"):
        ok, why = stark.stark_verify(be, tampered(what), publics, root_c, info, None, vinfo, challenges=tr, legacy_transcript_queries=True)
        assert not ok, what
    # a challenge that is not the transcript's: the identity no longer holds
    bad_tr = dict(tr, challenges=[list(c) for c in tr["challenges"]]); bad_tr["challenges"][3] = [[1, 2, 3]]
    ok, _ = stark.stark_verify(be, proof, publics, root_c, info, None, vinfo, challenges=bad_tr, legacy_transcript_queries=True)
    assert not ok
