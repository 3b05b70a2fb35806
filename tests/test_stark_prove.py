"""End-to-end prove loop (commit -> Q -> evals -> FRI -> queries) on the synthetic Fibonacci AIR:
the orchestration is exercised on the CPU oracle backend here (no GPU), and GPU-vs-oracle proof identity
is asserted in the gpu-marked test."""
import numpy as np
import pytest

from conftest import P


def _setup(n_bits, n_pairs, steps, n_queries=8, prev_row=False):
    from pil2gl import stark
    ss = {"nBits": n_bits, "nBitsExt": steps[0], "nQueries": n_queries, "verificationHashType": "GL",
          "steps": [{"nBits": b} for b in steps]}
    info, exprs, vinfo = stark.fibonacci_air(n_pairs, ss, prev_row)
    cm, consts, publics = stark.fibonacci_trace(n_bits, n_pairs)
    return stark, info, exprs, vinfo, cm, consts, publics


def test_trace_satisfies_air():
    stark, info, exprs, vinfo, cm, consts, publics = _setup(6, 2, [9, 5, 2])
    N = cm.shape[0]
    for i in range(N - 1):
        for k in range(2):
            l1, l2 = int(cm[i, 2 * k]), int(cm[i, 2 * k + 1])
            assert int(cm[i + 1, 2 * k + 1]) == l1 and int(cm[i + 1, 2 * k]) == (l1 * l1 + l2 * l2) % P
    assert publics == [int(cm[0, 1]), int(cm[0, 0]), int(cm[N - 1, 0])]


# (10, 1, [11, 7, 3]) is BASELINE config 1: the reference's Fibonacci starkStruct (fibonacci.starkstruct.gpu.json: nBits 10,
# nBitsExt 11, 8 queries, steps 11/7/3)
@pytest.mark.parametrize("n_bits,n_pairs,steps", [(6, 1, [9, 5, 2]), (8, 3, [11, 7, 3]), (10, 1, [11, 7, 3])])
def test_prove_and_verify_on_oracle_backend(oracle, n_bits, n_pairs, steps):
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _setup(n_bits, n_pairs, steps)
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    # soundness smoke: a corrupted evaluation / witness must be rejected
    bad = {**res, "proof": {**res["proof"], "evals": [list(e) for e in res["proof"]["evals"]]}}
    bad["proof"]["evals"][0][0] ^= 1
    assert not stark_ref.stark_verify(bad, setup["constRoot"], info, vinfo)[0]
    # same corruption with the prover's own challenges re-derived: the evaluation identity itself must fail
    bad2 = {**bad}
    ok_b, why_b = stark_ref.stark_verify({**bad2, "challenges": None, "challengesFRISteps": None, "queries": None}, setup["constRoot"], info, vinfo, check_transcript=False)
    assert not ok_b and why_b == "Invalid evaluations"
    cm2 = cm.copy(); cm2[5, 0] ^= 1
    res2 = stark.stark_gen(be, be.from_host(cm2), setup, info, exprs, publics)
    ok2, why2 = stark_ref.stark_verify(res2, setup["constRoot"], info, vinfo)
    assert not ok2


def test_fri_opening_order_is_the_references_key_order():
    """friPolinomial.js:42-50 folds the openings in the order of Object.keys(friExps): node itself is asked for that order"""
    import json, shutil, subprocess
    from pil2gl import stark
    cases = [[0, 1], [0, 1, -1], [-1, 0, 1], [1, -1, 0, -2], [2, 0, -1, 10, 1], [-2, -1, 0]]
    for primes in cases:
        assert sorted(stark.fri_opening_order({"evMap": [{"prime": p} for p in primes + primes[::-1]]})) == sorted(primes)
    assert stark.fri_opening_order({"evMap": [{"prime": p} for p in [0, 1, -1, 0, 1]]}) == [0, 1, -1]
    assert stark.fri_opening_order({"evMap": [{"prime": p} for p in [-1, 1, 0]]}) == [0, 1, -1]
    assert stark.fri_opening_order({"evMap": [{"prime": p} for p in [-1, 10, -2, 2, 0]]}) == [0, 2, 10, -1, -2]
    node = shutil.which("node")
    if node:
        js = "const r=[];for(const c of %s){const o={};for(const p of c)if(!(p in o))o[p]=1;r.push(Object.keys(o).map(Number));}console.log(JSON.stringify(r))" % json.dumps(cases)
        got = json.loads(subprocess.run([node, "-e", js], capture_output=True, text=True, timeout=60).stdout)
        assert got == [stark.fri_opening_order({"evMap": [{"prime": p} for p in c]}) for c in cases]


def test_previous_row_opening_proof_on_oracle_backend(oracle):
    """an AIR that reads the previous row: openings [-1, 0, 1]; the proof verifies, and the weighted-sum plan of the FRI polynomial
    (what the GPU backend runs instead of the op-list) lists its terms in the op-list's order 0, 1, -1"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _setup(6, 2, [9, 5, 2], prev_row=True)
    assert info["openingPoints"] == [-1, 0, 1] and stark.fri_opening_order(info) == [0, 1, -1]
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    plan = stark.fri_polynomial_plan(info, res["proof"]["evals"], res["challenges"][3][1])
    assert plan is not None and plan[3] == [1, 2, 0]
    cm2 = cm.copy(); cm2[9, 1] ^= 1                       # breaks only the previous-row identity's neighbourhood
    res2 = stark.stark_gen(be, be.from_host(cm2), setup, info, exprs, publics)
    assert not stark_ref.stark_verify(res2, setup["constRoot"], info, vinfo)[0]


def _im_case(n_bits, n_pairs, steps, prev_row=False, im_pols=True, boundaries=False):
    from pil2gl import stark
    ss = {"nBits": n_bits, "nBitsExt": steps[0], "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
    info, exprs, vinfo = stark.fibonacci_air(n_pairs, ss, prev_row, im_pols=im_pols, boundaries=boundaries)
    cm, consts, publics = stark.fibonacci_trace(n_bits, n_pairs, im_pols=im_pols)
    return stark, info, exprs, vinfo, cm, consts, publics


def test_boundary_constraints_on_oracle_backend(oracle):
    """pil2 boundaries (constraintPolynomial.js:21-45): transitions on everyFrame{0, 1}, inputs on firstRow, output on lastRow -- Zi_ext carries
    one zerofier column per boundary (stark_gen_helpers.js:146-160) and the verifier one value per boundary (stark_verify.js:99-136);
    a violation on the first row, on the last row or in between is rejected"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _im_case(6, 2, [9, 5, 2], im_pols=False, boundaries=True)
    assert [b["name"] for b in info["boundaries"]] == ["everyRow", "everyFrame", "firstRow", "lastRow"]
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    for r, c in ((0, 1), (63, 0), (7, 0)):
        cm2 = cm.copy(); cm2[r, c] ^= 1
        res2 = stark.stark_gen(be, be.from_host(cm2), setup, info, exprs, publics)
        assert not stark_ref.stark_verify(res2, setup["constRoot"], info, vinfo)[0], (r, c)
    # the zerofier values of the verifier are the table columns' own polynomials: Zi(firstRow) * (x - 1) = Zh on the extended domain
    zfr = be.build_one_row_zerofier_inv(6, 9, 0); zh_inv = be.build_zhinv(6, 9); x = be.build_x(9, stark.SHIFT)
    for i in (0, 1, 77, 511):
        assert int(zfr[i]) * ((int(x[i]) - 1) % P) % P * int(zh_inv[i]) % P == 1


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,n_pairs,steps,im", [(6, 2, [9, 5, 2], False), (13, 9, [16, 11, 6], True)])
def test_gpu_proof_with_boundary_constraints_is_identical_to_oracle_proof(oracle, n_bits, n_pairs, steps, im):
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _im_case(n_bits, n_pairs, steps, im_pols=im, boundaries=True)
    gpu = stark.GpuBackend(0)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend()
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], publics, s_gpu["constRoot"], info, exprs, vinfo)
    assert ok, why
    bad = {**r_gpu["proof"], "evals": [list(e) for e in r_gpu["proof"]["evals"]]}
    bad["evals"][0][0] ^= 1
    assert not stark.stark_verify(gpu, bad, publics, s_gpu["constRoot"], info, exprs, vinfo)[0]


def test_intermediate_polynomials_are_computed_by_the_prover_on_oracle_backend(oracle):
    """expressionsInfo.imPolsCode (prover.js:212-214): the last witness stage's intermediate polynomials are op-lists with destinations
    of type cm, run on the trace domain before the stage is extended; the witness arrives with those columns empty"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _im_case(6, 2, [9, 5, 2])
    assert info["mapSectionsN"]["cm1"] == 6 and len(exprs["imPolsCode"][0]["code"]) == 6 and not cm[:, 4:].any()
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    w = be.from_host(cm)
    res = stark.stark_gen(be, w, setup, info, exprs, publics)
    wm = w.reshape(-1, 6)
    for r in (0, 1, 63):
        for i in range(2):
            assert int(wm[r, 4 + i]) == (int(wm[r, 2 * i]) ** 2 + int(wm[r, 2 * i + 1]) ** 2) % P
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    # without the prover's own computation the columns stay empty and the binding identity im - (l1^2 + l2^2) = 0 fails
    res2 = stark.stark_gen(be, be.from_host(cm), setup, info, {**exprs, "imPolsCode": [{"tmpUsed": 0, "code": []}]}, publics)
    assert not stark_ref.stark_verify(res2, setup["constRoot"], info, vinfo)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,n_pairs,steps,prev_row", [(6, 2, [9, 5, 2], False), (12, 7, [15, 10, 5], True), (16, 33, [19, 14, 9, 4], False)])
def test_gpu_proof_with_intermediate_polynomials_is_identical_to_oracle_proof(oracle, n_bits, n_pairs, steps, prev_row):
    """the same through the device evaluator (the 2^16 x 99 case: the run-time compiled kernel writes 33 columns of a stage buffer)"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _im_case(n_bits, n_pairs, steps, prev_row)
    gpu = stark.GpuBackend(0)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend()
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], publics, s_gpu["constRoot"], info, exprs, vinfo)
    assert ok, why


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,n_pairs,steps", [(6, 2, [9, 5, 2]), (12, 5, [15, 10, 5])])
def test_gpu_previous_row_opening_proof_is_identical_to_oracle_proof(oracle, n_bits, n_pairs, steps):
    """the GPU backend computes the FRI polynomial as weighted row sums + one combine kernel, the oracle backend runs the op-list:
    with openings [-1, 0, 1] the two agree only if the combine takes its terms in the reference's order (0, 1, -1)"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _setup(n_bits, n_pairs, steps, prev_row=True)
    gpu = stark.GpuBackend(0)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend()
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark_ref.stark_verify(r_gpu, s_gpu["constRoot"], info, vinfo)
    assert ok, why


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,n_pairs,steps,split,jit", [(8, 2, [11, 7, 3], False, "0"), (10, 1, [11, 7, 3], False, "0"), (12, 5, [15, 11, 7, 3], False, "1"), (10, 4, [13, 9, 4], True, "1"), (13, 40, [16, 11, 6], False, "0")])
def test_gpu_proof_is_bit_identical_to_oracle_proof(oracle, n_bits, n_pairs, steps, split, jit, monkeypatch):
    monkeypatch.setenv("PIL2GL_EXPR_JIT", jit)
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _setup(n_bits, n_pairs, steps)
    gpu = stark.GpuBackend(0, split)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend(split)
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert s_gpu["constRoot"] == s_cpu["constRoot"]
    assert r_gpu["challenges"] == r_cpu["challenges"] and r_gpu["queries"] == r_cpu["queries"]
    assert r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark_ref.stark_verify(r_gpu, s_gpu["constRoot"], info, vinfo, split)
    assert ok, why


def _tampered(res, what):
    import copy
    bad = copy.deepcopy(res["proof"])
    if what == "eval":
        bad["evals"][1][0] = (bad["evals"][1][0] + 1) % P
    elif what == "opened value":
        bad["fri"][0]["polQueries"][3][0][0][0] = (int(bad["fri"][0]["polQueries"][3][0][0][0]) + 1) % P
    elif what == "sibling":
        sib = bad["fri"][0]["polQueries"][2][1][1]
        sib[0][0] = (int(sib[0][0]) + 1) % P
    elif what == "fri layer value":
        v = bad["fri"][1]["polQueries"][-1][0]; v[4] = (int(v[4]) + 1) % P
    elif what == "fri layer sibling":
        sib = bad["fri"][2]["polQueries"][0][1]; sib[0][0] = (int(sib[0][0]) + 1) % P
    elif what == "last polynomial":
        bad["fri"][-1][1][2] = (bad["fri"][-1][1][2] + 1) % P
    elif what == "root":
        bad["root2"] = list(bad["root2"]); bad["root2"][0] = (int(bad["root2"][0]) + 1) % P
    return bad


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,n_pairs,steps,split,nq", [(8, 2, [11, 7, 3], False, 8), (10, 4, [13, 9, 4], True, 13), (12, 5, [15, 11, 7, 3], False, 32)])
def test_device_verifier_agrees_with_restated_verifier(oracle, n_bits, n_pairs, steps, split, nq):
    """pil2gl.stark.stark_verify (batched openings, device evaluator on the query rows, FRI.verify) accepts what the
    restated stark_verify.js accepts and rejects each kind of tampering it rejects"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _setup(n_bits, n_pairs, steps, n_queries=nq)
    gpu = stark.GpuBackend(0, split)
    setup = stark.build_const_tree(gpu, consts, info)
    res = stark.stark_gen(gpu, gpu.from_host(cm), setup, info, exprs, publics)
    ok, why = stark.stark_verify(gpu, res["proof"], publics, setup["constRoot"], info, exprs, vinfo)
    assert ok, why
    assert stark_ref.stark_verify(res, setup["constRoot"], info, vinfo, split)[0]
    for what in ("eval", "opened value", "sibling", "fri layer value", "fri layer sibling", "last polynomial", "root"):
        bad = _tampered(res, what)
        ok_ref = stark_ref.stark_verify({**res, "proof": bad}, setup["constRoot"], info, vinfo, split, check_transcript=False)[0]
        ok_dev, why = stark.stark_verify(gpu, bad, publics, setup["constRoot"], info, exprs, vinfo)
        assert not ok_ref and not ok_dev, (what, why)
    wrong_publics = list(publics); wrong_publics[0] = (wrong_publics[0] + 1) % P
    assert not stark.stark_verify(gpu, res["proof"], wrong_publics, setup["constRoot"], info, exprs, vinfo)[0]


def _query_verifier_of(info, exprs):
    from pil2gl import stark
    return stark.query_verifier_of(info, exprs)


@pytest.mark.gpu
@pytest.mark.parametrize("two_stage", [False, True])
def test_device_verifier_runs_the_references_query_program_shape(oracle, two_stage):
    """the reference's verifier does not re-run the prover's FRI op-list: it runs verifierInfo.queryVerifier, whose witness
    operands are tree<stage> records and whose result is its last temporary (stark_verify.js:185-238, 245-246)"""
    if two_stage:
        stark, info, exprs, vinfo, cm, consts, publics = _perm_case(9, (12, 8, 4))
    else:
        stark, info, exprs, vinfo, cm, consts, publics = _setup(9, 3, [12, 8, 4], n_queries=9)
    gpu = stark.GpuBackend(0, False)
    setup = stark.build_const_tree(gpu, consts, info)
    res = stark.stark_gen(gpu, gpu.from_host(cm), setup, info, exprs, publics)
    qv = _query_verifier_of(info, exprs)
    assert any(r["type"].startswith("tree") for c in qv["code"] for r in c["src"])
    vq = {**vinfo, "queryVerifier": qv}
    ok, why = stark.stark_verify(gpu, res["proof"], publics, setup["constRoot"], info, None, vq)   # no prover expressions needed
    assert ok, why
    for what in ("opened value", "eval", "fri layer value"):
        assert not stark.stark_verify(gpu, _tampered(res, what), publics, setup["constRoot"], info, None, vq)[0], what


@pytest.mark.gpu
def test_fri_verify_fold_batch(oracle):
    """pil2gl_fri_verify_fold: per query evalPol(ifft(group), challenge * sinv) (fri.js:121-127) against the oracle's fold of
    the same group, for group sizes 1..64 and a query count that is not a power of two"""
    import gl_oracle as orc
    import pil2gl
    pil2gl.init(0)
    rng = np.random.default_rng(23)
    from conftest import rand_field
    for fold_bits, nq in ((0, 5), (1, 1), (3, 13), (4, 64), (6, 7)):
        G = rand_field(rng, (nq, 1 << fold_bits, 3)); sinv = rand_field(rng, nq); ch = rand_field(rng, 3)
        out = np.zeros((nq, 3), np.uint64)
        Gt = np.ascontiguousarray(G.transpose(1, 0, 2))
        pil2gl.call("pil2gl_fri_verify_fold", pil2gl._ptr(Gt), fold_bits, nq, pil2gl._ptr(sinv), pil2gl._ptr(ch), pil2gl._ptr(out))
        for q in range(nq):
            want = orc.fri_fold(G[q], 0, int(sinv[q]), ch)[0]
            assert [int(v) for v in out[q]] == [int(v) for v in want], (fold_bits, q)
    with pytest.raises(pil2gl.Pil2glError):
        pil2gl.call("pil2gl_fri_verify_fold", pil2gl._ptr(Gt), 21, 1, pil2gl._ptr(sinv), pil2gl._ptr(ch), pil2gl._ptr(out))


@pytest.mark.gpu
def test_synthetic_trace_kernel_matches_reference_recurrence():
    import ctypes as C
    import torch
    import pil2gl
    pil2gl.init(0)
    n_bits, K = 7, 3
    init = np.array([5, 9, 1 << 40, P - 1, 123456789, 987654321], dtype=np.uint64)
    cm = torch.empty((1 << n_bits) * 2 * K, dtype=torch.int64, device="cuda")
    pil2gl.call("pil2gl_synth_fibonacci_dev", n_bits, K, C.c_void_p(init.ctypes.data), C.c_void_p(cm.data_ptr()), None)
    got = cm.cpu().numpy().view(np.uint64).reshape(1 << n_bits, 2 * K)
    for k in range(K):
        l1, l2 = int(init[2 * k]), int(init[2 * k + 1])
        for i in range(1 << n_bits):
            assert (int(got[i, 2 * k]), int(got[i, 2 * k + 1])) == (l1, l2)
            l1, l2 = (l1 * l1 + l2 * l2) % P, l1


@pytest.mark.gpu
def test_config2_size_proof_verifies(oracle):
    """BASELINE config 2 shape (2^20 rows x 8 cols, blow-up 8): the GPU proof passes the restated verifier"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _setup(20, 4, [23, 18, 13, 8], n_queries=16)
    gpu = stark.GpuBackend(0, False)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    res = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    ok, why = stark_ref.stark_verify(res, s_gpu["constRoot"], info, vinfo)
    assert ok, why


@pytest.mark.gpu
def test_config3_size_proof_verifies(oracle):
    """BASELINE config 3, the bench's default workload (2^24 rows x 100 cols, blow-up 8, FRI 27/22/17/12/7, 64 queries): the
    proof the GPU writes at full size passes the restated verifier (openings, evaluation identity, FRI folds)"""
    import gc
    import torch
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()                                      # what earlier tests left in torch's allocator cache
    if not torch.cuda.is_available() or torch.cuda.mem_get_info()[0] < 200e9:
        pytest.skip("needs ~190 GB of free device memory")
    import stark_ref
    import bench
    from pil2gl import stark
    n_bits, n_cols = 24, 100
    ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": False,
          "steps": [{"nBits": b} for b in (27, 22, 17, 12, 7)]}
    info, exprs, vinfo = stark.fibonacci_air(n_cols // 2, ss)
    gpu = stark.GpuBackend(0, False)
    cm, consts, publics = bench.fibonacci_trace_gpu(torch.device("cuda", 0), n_bits, n_cols // 2, 0)
    setup = stark.build_const_tree(gpu, consts, info)
    res = stark.stark_gen(gpu, cm, setup, info, exprs, publics)
    del cm
    torch.cuda.empty_cache()
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    # and the witness it proves really is K Fibonacci machines ending in the public output
    assert len(res["proof"]["evals"]) > 0 and res["publics"] == publics


@pytest.mark.gpu
def test_config4_size_bn128_proof_verifies(oracle):
    """BASELINE config 4's size with `verificationHashType: "BN128"` (stark_gen_helpers.js:93-101, 388-412): 2^24 rows x 100 columns, every tree a
    BN254-Poseidon arity-16 tree over 2^27 rows (merklehash_bn128_p.js:47-129), BN128 transcript, FRI 27/22/17/12/7, 64 queries -- the whole
    proof (`bench.py --workload c4 --mode prove`) passes the restated verifier, whose paths go through the Python-integer oracle's rule"""
    import gc
    import torch
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    if not torch.cuda.is_available() or torch.cuda.mem_get_info()[0] < 200e9:
        pytest.skip("needs ~190 GB of free device memory")
    import stark_ref
    import bench
    from pil2gl import stark
    n_bits, n_cols = 24, 100
    ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 64, "verificationHashType": "BN128", "merkleTreeArity": 16, "merkleTreeCustom": False,
          "steps": [{"nBits": b} for b in (27, 22, 17, 12, 7)]}
    info, exprs, vinfo = stark.fibonacci_air(n_cols // 2, ss)
    gpu = stark.GpuBackend(0, False, "BN128", 16, False)
    cm, consts, publics = bench.fibonacci_trace_gpu(torch.device("cuda", 0), n_bits, n_cols // 2, 0)
    setup = stark.build_const_tree(gpu, consts, info)
    res = stark.stark_gen(gpu, cm, setup, info, exprs, publics)
    del cm
    torch.cuda.empty_cache()
    assert isinstance(res["proof"]["root1"], int) and len(res["proof"]["fri"][0]["polQueries"][0][0][1][0]) == 16
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo, hash_type="BN128", arity=16, custom=False)
    assert ok, why
    bad = {**res, "proof": {**res["proof"], "root2": res["proof"]["root2"] ^ 1}}
    assert not stark_ref.stark_verify(bad, setup["constRoot"], info, vinfo, hash_type="BN128", arity=16, custom=False)[0]


@pytest.mark.gpu
def test_config3_size_two_stage_proof_verifies(oracle):
    """the two-stage workload of `bench.py --air perm` at config 3's size (2^24 rows, 9 permutation checks: 18 stage-1 and 81
    stage-2 base columns, nine grand-product hints resolved on the device, polutils.js:105-164): the proof passes the restated
    verifier, and stops passing when one b column is not a permutation of its a column (its running product does not close)"""
    import gc
    import torch
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    if not torch.cuda.is_available() or torch.cuda.mem_get_info()[0] < 200e9:
        pytest.skip("needs ~190 GB of free device memory")
    import stark_ref
    import bench
    from pil2gl import stark
    n_bits, copies = 24, 9
    ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": False,
          "steps": [{"nBits": b} for b in (27, 22, 17, 12, 7)]}
    info, exprs, vinfo = stark.permutation_air(ss, copies)
    gpu = stark.GpuBackend(0, False)
    cm, consts, publics = bench.permutation_trace_gpu(torch.device("cuda", 0), n_bits, copies)
    setup = stark.build_const_tree(gpu, consts, info)
    res = stark.stark_gen(gpu, cm, setup, info, exprs, publics)
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    del res
    cm.view(1 << n_bits, 2 * copies)[12345, 7] += 1                     # b_3 is no longer a permutation of a_3
    res = stark.stark_gen(gpu, cm, setup, info, exprs, publics)
    del cm
    torch.cuda.empty_cache()
    ok, _ = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert not ok


def _bn_case(n_bits=5, pairs=1):
    from pil2gl import stark
    steps = [n_bits + 3, 4] if n_bits <= 8 else list(range(n_bits + 3, 4, -5))
    ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 4, "verificationHashType": "BN128", "steps": [{"nBits": b} for b in steps]}
    info, exprs, vinfo = stark.fibonacci_air(pairs, ss)
    cm, consts, publics = stark.fibonacci_trace(n_bits, pairs)
    return stark, info, exprs, vinfo, cm, consts, publics


@pytest.mark.parametrize("arity,custom", [(16, False), (4, True)])
def test_bn128_proof_on_oracle_backend_verifies(oracle, arity, custom):
    """verificationHashType BN128 (stark_gen_helpers.js:95-99): BN128 trees and transcript in the same stage loop"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _bn_case()
    be = stark_ref.OracleBackend(False, "BN128", arity, custom)
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    assert isinstance(res["proof"]["root1"], int) and len(res["proof"]["fri"][0]["polQueries"][0][0][1][0]) == arity
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo, hash_type="BN128", arity=arity, custom=custom)
    assert ok, why
    bad = {**res, "proof": {**res["proof"], "root2": res["proof"]["root2"] ^ 1}}
    assert not stark_ref.stark_verify(bad, setup["constRoot"], info, vinfo, hash_type="BN128", arity=arity, custom=custom)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("arity,custom,n_bits,pairs", [(16, False, 5, 1), (4, True, 5, 1), (8, False, 6, 3), (16, False, 13, 26)])
def test_gpu_bn128_proof_is_identical_to_oracle_proof(oracle, arity, custom, n_bits, pairs):
    """the last case (2^13 rows x 52 columns -> 2^16 extended rows: 18 field elements per leaf = a width-17 and a width-3 permutation, 4 096
    parents in the first tree level) takes the lane-per-permutation matrix-core kernels for the leaves AND the tree levels (the wave-per-
    permutation kernel serves up to 2 048 permutations per call, bn128.hip wave_per_perm_max); its oracle trees come from oracle/bn128_oracle.c"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _bn_case(n_bits, pairs)
    gpu = stark.GpuBackend(0, False, "BN128", arity, custom)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend(False, "BN128", arity, custom)
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert s_gpu["constRoot"] == s_cpu["constRoot"]
    assert r_gpu["challenges"] == r_cpu["challenges"] and r_gpu["queries"] == r_cpu["queries"]
    assert r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark_ref.stark_verify(r_gpu, s_gpu["constRoot"], info, vinfo, hash_type="BN128", arity=arity, custom=custom)
    assert ok, why
    # the device verifier over BN128 trees and transcript
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], publics, s_gpu["constRoot"], info, exprs, vinfo)
    assert ok, why
    for what in ("opened value", "fri layer value", "last polynomial"):
        assert not stark.stark_verify(gpu, _tampered(r_gpu, what), publics, s_gpu["constRoot"], info, exprs, vinfo)[0], what


def _perm_ref_case(n_bits=6, steps=(9, 5, 2), copies=1, ref_hints=True):
    from pil2gl import stark
    ss = {"nBits": n_bits, "nBitsExt": steps[0], "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
    info, exprs, vinfo = stark.permutation_air(ss, copies, ref_hints=ref_hints)
    cm, consts, publics = stark.permutation_trace(n_bits, copies=copies)
    return stark, info, exprs, vinfo, cm, consts, publics


def test_hints_in_the_references_shape_with_expression_fields_on_oracle_backend(oracle):
    """expressionsInfo.hintsInfo as the reference writes it (hints_helpers.js:21-33,102-113): a gprod hint whose numerator and
    denominator are EXPRESSIONS (fields of op tmp, evaluated on the trace domain by calculateExpression) and whose reference is the
    only committed stage-2 column"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _perm_ref_case(6, (9, 5, 2), 2)
    assert info["mapSectionsN"]["cm2"] == 6 and [f["op"] for f in exprs["hintsInfo"][0]["fields"]] == ["tmp", "tmp", "cm"]
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    cm2 = cm.copy(); cm2[3, 1] ^= 1                        # b is no longer a permutation of a: the product does not close
    res2 = stark.stark_gen(be, be.from_host(cm2), setup, info, exprs, publics)
    assert not stark_ref.stark_verify(res2, setup["constRoot"], info, vinfo)[0]
    # the expression column itself: a + gamma on the trace domain
    ctx = {"pilInfo": info, "publics": [], "challenges": [[], [[5, 6, 7]], [], [], []], "evals": []}
    bufs = {"cm1_n": be.from_host(cm), "const_n": be.from_host(consts)}
    col, dim = stark.calculate_expression(be, exprs, 2, bufs, {"cm1_n": 4, "const_n": 2}, 6, ctx)
    assert dim == 3 and col.reshape(-1, 3)[9].tolist() == [(int(cm[9, 0]) + 5) % P, 6, 7]
    with pytest.raises(ValueError, match="not found"):
        stark.calculate_expression(be, exprs, 99, bufs, {"cm1_n": 4, "const_n": 2}, 6, ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,steps,copies", [(6, (9, 5, 2), 1), (14, (17, 12, 7), 5)])
def test_gpu_proof_with_reference_shaped_hints_is_identical_to_oracle_proof(oracle, n_bits, steps, copies):
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _perm_ref_case(n_bits, steps, copies)
    gpu = stark.GpuBackend(0)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend()
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], publics, s_gpu["constRoot"], info, exprs, vinfo)
    assert ok, why


def _perm_case(n_bits=6, steps=(9, 5, 2)):
    from pil2gl import stark
    ss = {"nBits": n_bits, "nBitsExt": steps[0], "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
    info, exprs, vinfo = stark.permutation_air(ss)
    cm, consts, publics = stark.permutation_trace(n_bits)
    return stark, info, exprs, vinfo, cm, consts, publics


def test_two_stage_proof_with_grand_product_hint_on_oracle_backend(oracle):
    """stage 2 after its challenge: stage code on the trace domain, the gprod hint (calculateZ), commitment; then the
    quotient as stage 3.  The proof verifies, and it stops verifying when b is not a permutation of a."""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _perm_case()
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    assert set(res["proof"]) == {"root1", "root2", "root3", "evals", "fri"} and len(res["challenges"]) == 5
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    cm2 = cm.copy(); cm2[7, 1] = (int(cm2[7, 1]) + 1) % P          # b no longer a permutation of a: z does not close
    res2 = stark.stark_gen(be, be.from_host(cm2), setup, info, exprs, publics)
    assert not stark_ref.stark_verify(res2, setup["constRoot"], info, vinfo)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,steps", [(6, (9, 5, 2)), (12, (15, 10, 5))])
def test_gpu_two_stage_proof_is_identical_to_oracle_proof(oracle, n_bits, steps):
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _perm_case(n_bits, steps)
    gpu = stark.GpuBackend(0)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend()
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert r_gpu["challenges"] == r_cpu["challenges"] and r_gpu["queries"] == r_cpu["queries"]
    assert r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark_ref.stark_verify(r_gpu, s_gpu["constRoot"], info, vinfo)
    assert ok, why
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], publics, s_gpu["constRoot"], info, exprs, vinfo)   # stage-2 challenges, three opening points
    assert ok, why
    assert not stark.stark_verify(gpu, _tampered(r_gpu, "eval"), publics, s_gpu["constRoot"], info, exprs, vinfo)[0]


# ---- starkStruct.hashCommits (the recursion starkStructs set it): the transcript absorbs the HASH of the publics, of the
#      evaluations and of the last FRI polynomial instead of the values (prover.js:152-173, stark_gen_helpers.js:267-272,349-354)
def _hc_case(hash_type="GL", n_bits=6, pairs=2, steps=(9, 5, 2)):
    from pil2gl import stark
    ss = {"nBits": n_bits, "nBitsExt": steps[0], "nQueries": 8, "verificationHashType": hash_type, "hashCommits": True,
          "steps": [{"nBits": b} for b in steps]}
    info, exprs, vinfo = stark.fibonacci_air(pairs, ss)
    cm, consts, publics = stark.fibonacci_trace(n_bits, pairs)
    return stark, info, exprs, vinfo, cm, consts, publics


@pytest.mark.parametrize("hash_type,arity", [("GL", 16), ("BN128", 16)])
def test_hash_commits_proof_on_oracle_backend(oracle, hash_type, arity):
    import copy
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _hc_case(hash_type, 5 if hash_type == "BN128" else 6, 1 if hash_type == "BN128" else 2, (8, 4, 2) if hash_type == "BN128" else (9, 5, 2))
    be = stark_ref.OracleBackend(False, hash_type, arity, False)
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    kw = {"hash_type": hash_type, "arity": arity} if hash_type == "BN128" else {}
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo, **kw)
    assert ok, why
    # the challenges differ from the plain transcript's from the first one drawn after the publics
    info2 = copy.deepcopy(info); info2["starkStruct"]["hashCommits"] = False
    plain = stark.stark_gen(be, be.from_host(cm), setup, info2, exprs, publics)
    assert plain["challenges"] != res["challenges"] and plain["proof"]["root1"] == res["proof"]["root1"]
    # and a verifier that replays the plain transcript rejects the proof
    assert not stark_ref.stark_verify(res, setup["constRoot"], info2, vinfo, **kw)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("hash_type,n_bits,pairs,steps", [("GL", 10, 3, (13, 9, 4)), ("BN128", 5, 1, (8, 4, 2))])
def test_gpu_hash_commits_proof_is_identical_to_oracle_proof(oracle, hash_type, n_bits, pairs, steps):
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _hc_case(hash_type, n_bits, pairs, steps)
    gpu = stark.GpuBackend(0, False, hash_type, 16, False)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend(False, hash_type, 16, False)
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert r_gpu["challenges"] == r_cpu["challenges"] and r_gpu["queries"] == r_cpu["queries"] and r_gpu["proof"] == r_cpu["proof"]
    kw = {"hash_type": hash_type, "arity": 16} if hash_type == "BN128" else {}
    ok, why = stark_ref.stark_verify(r_gpu, s_gpu["constRoot"], info, vinfo, **kw)
    assert ok, why
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], publics, s_gpu["constRoot"], info, exprs, vinfo)
    assert ok, why


def _hint_fixture(be, n_bits=6):
    """a fake two-stage context for resolve_hints_info: stage 1 holds f (dim 1), t (dim 1), a (dim 1), d (dim 3); stage 2 receives the hints' columns"""
    from pil2gl import stark
    N = 1 << n_bits
    rng = np.random.default_rng(5)
    t = rng.integers(1, 1 << 40, size=N, dtype=np.uint64)
    f = t[rng.integers(0, N, size=N)]
    a = rng.integers(1, 1 << 62, size=N, dtype=np.uint64)
    d = rng.integers(1, 1 << 62, size=(N, 3), dtype=np.uint64)
    cm1 = np.concatenate([f[:, None], t[:, None], a[:, None], d], axis=1)
    cmap = [{"stage": 1, "dim": 1, "stagePos": 0}, {"stage": 1, "dim": 1, "stagePos": 1}, {"stage": 1, "dim": 1, "stagePos": 2}, {"stage": 1, "dim": 3, "stagePos": 3},
            {"stage": 2, "dim": 1, "stagePos": 0}, {"stage": 2, "dim": 1, "stagePos": 1},            # h1, h2
            {"stage": 2, "dim": 3, "stagePos": 2}, {"stage": 2, "dim": 3, "stagePos": 5}]            # gsum column, gprod column
    info = {"cmPolsMap": cmap, "mapSectionsN": {"cm1": 6, "cm2": 8}}
    # expression 4: a * a + 7 (dim 1) -- a numerator that is not a committed column
    code = {"tmpUsed": 2, "code": [{"op": "mul", "dest": {"type": "tmp", "id": 0, "dim": 1}, "src": [{"type": "cm", "id": 2, "prime": 0, "dim": 1}, {"type": "cm", "id": 2, "prime": 0, "dim": 1}]},
                                   {"op": "add", "dest": {"type": "tmp", "id": 1, "dim": 1}, "src": [{"type": "tmp", "id": 0, "dim": 1}, {"type": "number", "value": "7", "dim": 1}]}]}
    exprs = {"expressionsCode": [None, {"expId": 4, "code": code}], "hintsInfo": [
        {"name": "h1h2", "fields": [{"name": "f", "op": "cm", "id": 0}, {"name": "t", "op": "cm", "id": 1}, {"name": "referenceH1", "op": "cm", "id": 4}, {"name": "referenceH2", "op": "cm", "id": 5}]},
        {"name": "gsum", "fields": [{"name": "numerator", "op": "number", "value": "3"}, {"name": "denominator", "op": "cm", "id": 3}, {"name": "reference", "op": "cm", "id": 6},
                                    {"name": "result", "op": "subproofValue", "id": 1}]},
        {"name": "gprod", "fields": [{"name": "numerator", "op": "tmp", "id": 4}, {"name": "denominator", "op": "cm", "id": 2}, {"name": "reference", "op": "cm", "id": 7},
                                     {"name": "result", "op": "subproofValue", "id": 0}]},
        {"name": "public", "fields": [{"name": "expression", "op": "tmp", "id": 4}, {"name": "row_index", "op": "number", "value": "5"}, {"name": "reference", "op": "public", "id": 2}]},
        {"name": "subproofValue", "fields": [{"name": "expression", "op": "cm", "id": 3}, {"name": "reference", "op": "subproofValue", "id": 2}]}]}
    bufs = {"cm1_n": be.from_host(cm1), "cm2_n": be.zeros(8 * N)}
    widths = {"cm1_n": 6, "cm2_n": 8}
    ctx = {"pilInfo": info, "publics": [11], "challenges": [[], []], "evals": []}
    for stage in (1, 2):                                   # as the stage loop does: the public / subproof value read off stage-1 columns resolve with stage 1
        stark.resolve_hints_info(be, info, exprs, stage, bufs, widths, n_bits, ctx)
        if stage == 1:
            assert ctx["publics"] == [11, 0, (int(a[5]) * int(a[5]) + 7) % P] and not be.to_host(bufs["cm2_n"]).any()
    return cm1, be.to_host(bufs["cm2_n"]).reshape(N, 8), ctx


def _check_hint_fixture(oracle, cm1, cm2, ctx):
    N = cm1.shape[0]
    f, t, a, d = cm1[:, 0], cm1[:, 1], cm1[:, 2], cm1[:, 3:6]
    w1, w2 = oracle.h1h2([int(v) for v in f], [int(v) for v in t])
    assert cm2[:, 0].tolist() == w1 and cm2[:, 1].tolist() == w2                                   # calculateH1H2, polutils.js:105-126
    gs = oracle.gsum(np.array([3], dtype=np.uint64), d.reshape(-1).copy(), 1, 3).reshape(N, 3)
    assert (cm2[:, 2:5] == gs).all()                                                                 # calculateS, :147-164
    num = np.array([(int(v) * int(v) + 7) % P for v in a], dtype=np.uint64)
    gp = oracle.gprod(num, a.copy(), 1, 1).reshape(N, -1)
    assert (cm2[:, 5] == gp[:, 0]).all() and not cm2[:, 6:8].any()                                    # a base result lands in an extension column as [v, 0, 0]
    assert ctx["subproofValues"][0] == int(gp[N - 1, 0]) and ctx["subproofValues"][1] == [int(v) for v in gs[N - 1]]   # `result`: the last row
    assert ctx["publics"] == [11, 0, int(num[5])] and ctx["subproofValues"][2] == [int(v) for v in d[N - 1]]


def test_reference_shaped_hint_kinds_on_oracle_backend(oracle):
    """resolve_hints_info over every hint kind and field kind of hints_helpers.js:21-123: h1h2, gsum with a `result`, gprod whose numerator is
    an expression, a public taken from one row of an expression, a subproof value from a column's last row"""
    import stark_ref
    _check_hint_fixture(oracle, *_hint_fixture(stark_ref.OracleBackend()))


@pytest.mark.gpu
def test_reference_shaped_hint_kinds_on_gpu_backend(oracle):
    from pil2gl import stark
    _check_hint_fixture(oracle, *_hint_fixture(stark.GpuBackend(0)))


def _pub_case(n_bits=6, pairs=2, steps=(9, 5, 2)):
    from pil2gl import stark
    ss = {"nBits": n_bits, "nBitsExt": steps[0], "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
    info, exprs, vinfo = stark.fibonacci_air(pairs, ss, public_hints=True)
    cm, consts, publics = stark.fibonacci_trace(n_bits, pairs)
    return stark, info, exprs, vinfo, cm, consts, publics


def test_publics_read_off_the_witness_by_stage_1_hints_on_oracle_backend(oracle):
    """hints of kind "public" (hints_helpers.js:83-90) resolved with stage 1, before the publics enter the transcript (prover.js:41-52): the
    prover is handed NO publics; the proof and the publics it reports equal those of the proof that was handed them"""
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _pub_case()
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, [0, 0, 0])
    assert res["publics"] == publics
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    given = stark.stark_gen(be, be.from_host(cm), setup, info, {k: v for k, v in exprs.items() if k != "hintsInfo"}, publics)
    assert given["proof"] == res["proof"]


@pytest.mark.gpu
def test_publics_read_off_the_witness_on_gpu_backend(oracle):
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _pub_case(10, 3, (13, 9, 4))
    gpu = stark.GpuBackend(0)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, [0, 0, 0])
    cpu = stark_ref.OracleBackend()
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, [0, 0, 0])
    assert r_gpu["publics"] == publics and r_gpu["proof"] == r_cpu["proof"]
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], r_gpu["publics"], s_gpu["constRoot"], info, exprs, vinfo)
    assert ok, why


def test_subproof_values_travel_in_the_proof_and_are_bound_on_oracle_backend(oracle):
    """a gprod hint's `result` field (hints_helpers.js:109-112) puts the column's last row into ctx.subproofValues, genProofStark (:369)
    into the proof, and the verifier reads it from there (stark_verify.js:19,256); the constraint LLAST * (z - subproofValue) binds it"""
    import copy
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _perm_ref_case(6, (9, 5, 2), 2, ref_hints="result")
    assert info["nSubproofValues"] == 2
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    assert len(res["proof"]["subproofValues"]) == 2 and all(len(v) == 3 for v in res["proof"]["subproofValues"])
    ok, why = stark_ref.stark_verify(res, setup["constRoot"], info, vinfo)
    assert ok, why
    bad = copy.deepcopy(res); bad["proof"]["subproofValues"][1][2] ^= 1
    assert not stark_ref.stark_verify(bad, setup["constRoot"], info, vinfo, check_transcript=False)[0]


@pytest.mark.gpu
def test_subproof_values_on_gpu_backend(oracle):
    import copy
    import stark_ref
    stark, info, exprs, vinfo, cm, consts, publics = _perm_ref_case(11, (14, 9, 4), 3, ref_hints="result")
    gpu = stark.GpuBackend(0)
    s_gpu = stark.build_const_tree(gpu, consts, info)
    r_gpu = stark.stark_gen(gpu, gpu.from_host(cm), s_gpu, info, exprs, publics)
    cpu = stark_ref.OracleBackend()
    s_cpu = stark.build_const_tree(cpu, consts, info)
    r_cpu = stark.stark_gen(cpu, cpu.from_host(cm), s_cpu, info, exprs, publics)
    assert r_gpu["proof"] == r_cpu["proof"] and len(r_gpu["proof"]["subproofValues"]) == 3
    ok, why = stark.stark_verify(gpu, r_gpu["proof"], publics, s_gpu["constRoot"], info, exprs, vinfo)
    assert ok, why
    bad = copy.deepcopy(r_gpu["proof"]); bad["subproofValues"][0][0] ^= 1
    assert not stark.stark_verify(gpu, bad, publics, s_gpu["constRoot"], info, exprs, vinfo)[0]
