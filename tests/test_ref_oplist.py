"""A real op-list of the reference through the encoder and the evaluators.

tests/golden/ref_verify_evals_code.json.gz is `verifierCode.code` of the reference's test data file
test/circuits/bn128/verifyEvals.starkInfo.json (made by oracle/gen_verify_evals_fixture.py): 3 257 ops written by the
reference's own code generator (codegen.js:257-283), with the mixed dim-1 / dim-3 operands, negative `number`s, `copy`s
and the operand kinds tmp / number / eval / challenge / x / public of SURVEY.md Appendix B.  It is executed
  * by the big-integer interpreter of tests/stark_ref.py (stark_verify.js:222-298 restated),
  * by the C oracle's evaluator, and (GPU) by the HIP evaluator, both fed by pil2gl.stark.encode_code -- the same mapping
    js/prover_helpers.js applies to `code.code`,
on random evaluations / challenges / publics and a per-row `x`; all three must agree on every row."""
import gzip
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, P, rand_field


def _load():
    with gzip.open(os.path.join(GOLDEN, "ref_verify_evals_code.json.gz")) as f:
        code = json.load(f)["code"]
    assert len(code) == 3257
    ops = {}
    for c in code:
        ops[c["op"]] = ops.get(c["op"], 0) + 1
    assert ops == {"mul": 1589, "add": 1335, "sub": 297, "copy": 36}
    last = code[-1]["dest"]
    # the program leaves its value in its last temporary (stark_verify.js:297); store it so that it can be observed
    return code + [{"op": "copy", "dest": {"type": "f", "dim": 3}, "src": [{"type": "tmp", "id": last["id"], "dim": last["dim"]}]}]


def _inputs(n_bits, seed=11):
    rng = np.random.default_rng(seed)
    return {"evals": [[int(v) for v in r] for r in rand_field(rng, (108, 3))],
            "challengesFlat": [[int(v) for v in r] for r in rand_field(rng, (5, 3))],
            "publics": [int(v) for v in rand_field(rng, 3)],
            "x": rand_field(rng, (1 << n_bits, 3))}


def _run(be, code, inp, n_bits):
    from pil2gl import stark
    ctx = {"pilInfo": {}, "publics": inp["publics"], "evals": inp["evals"], "challengesFlat": inp["challengesFlat"], "challenges": []}
    ops, n_tmp, secs, scalars = stark.encode_code(code, "ext", ctx)
    assert sorted(secs) == ["f_ext", "x_ext"] and n_tmp == 3257
    bufs = {"x_ext": be.from_host(inp["x"]), "f_ext": be.zeros(3 << n_bits)}
    be.eval_program(ops, n_tmp, [(bufs[s], 3) for s in secs], scalars, n_bits, 0)
    return np.asarray(be.to_host(bufs["f_ext"])).view(np.uint64).reshape(-1, 3)


def _bigint_rows(code, inp, rows):
    import stark_ref

    out = []
    for i in rows:
        def resolve(r, i=i):
            t = r["type"]
            if t == "number": return int(r["value"], 0) % P
            if t == "eval": return list(inp["evals"][r["id"]])
            if t == "challenge": return list(inp["challengesFlat"][r["id"]])
            if t == "public": return inp["publics"][r["id"]]
            if t == "x": return [int(v) for v in inp["x"][i]]
            raise ValueError(t)
        v = stark_ref.exec_code(code[:-1], resolve)
        out.append([int(c) % P for c in (v if isinstance(v, list) else [v, 0, 0])])
    return np.array(out, dtype=np.uint64)


def test_reference_oplist_oracle_vs_bigint(oracle):
    from stark_backend import OracleBackend
    code, n_bits = _load(), 4
    inp = _inputs(n_bits)
    got = _run(OracleBackend(), code, inp, n_bits)
    assert (got == _bigint_rows(code, inp, range(1 << n_bits))).all()
    assert got.any()


@pytest.mark.gpu
def test_reference_oplist_gpu(oracle):
    from pil2gl import stark
    from stark_backend import OracleBackend
    code, n_bits = _load(), 10
    inp = _inputs(n_bits, seed=12)
    got = _run(stark.GpuBackend(0), code, inp, n_bits)
    want = _run(OracleBackend(), code, inp, n_bits)
    assert (got == want).all()
    assert (got[[0, 513, 1023]] == _bigint_rows(code, inp, [0, 513, 1023])).all()


@pytest.mark.gpu
def test_reference_oplist_through_the_js_encoder(oracle, tmp_path):
    """same program through js/prover_helpers.js (encode + callCalculateExps on the addon)"""
    import shutil
    import subprocess
    from conftest import ROOT
    from stark_backend import OracleBackend
    node = shutil.which("node")
    if node is None:
        pytest.skip("node not installed")
    code, n_bits = _load(), 6
    inp = _inputs(n_bits, seed=13)
    want = _run(OracleBackend(), code, inp, n_bits)
    f = tmp_path / "inp.json"
    f.write_text(json.dumps({"nBits": n_bits, "evals": [[str(v) for v in e] for e in inp["evals"]],
                             "challenges": [[str(v) for v in c] for c in inp["challengesFlat"]],
                             "publics": [str(v) for v in inp["publics"]], "x": [str(int(v)) for v in inp["x"].reshape(-1)]}))
    out = subprocess.run([node, os.path.join(ROOT, "tests", "js", "ref_oplist.js"), str(f)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    got = np.array([int(v, 16) for v in json.loads(out.stdout.strip().splitlines()[-1])], dtype=np.uint64).reshape(-1, 3)
    assert (got == want).all()


# ---------------------------------------------------------------------------------------------------------------------------
# `muladd` (verifier code: codegen.js:137-165, executed as F.add(F.mul(a, b), c) at stark_verify.js:234) and `subproofValue`
# operands (prover_helpers.js:214-216, stark_verify.js:258) through both encoders
def _muladd_program():
    ch = lambda s, i: {"type": "challenge", "stage": s, "stageId": i, "dim": 3}
    t = lambda i, d=3: {"type": "tmp", "id": i, "dim": d}
    return [
        {"op": "muladd", "dest": t(0), "src": [ch(2, 0), {"type": "number", "value": "-5", "dim": 1}, {"type": "subproofValue", "id": 1, "dim": 1}]},
        {"op": "muladd", "dest": t(1), "src": [t(0), ch(2, 1), {"type": "eval", "id": 0, "dim": 3}]},
        {"op": "muladd", "dest": t(2, 1), "src": [{"type": "public", "id": 0, "dim": 1}, {"type": "subproofValue", "id": 0, "dim": 1}, {"type": "number", "value": "0x11", "dim": 1}]},
        {"op": "sub", "dest": t(3), "src": [t(2, 1), t(1)]},
        {"op": "muladd", "dest": t(4), "src": [t(3), {"type": "x", "dim": 3}, t(1)]},
        {"op": "copy", "dest": {"type": "f", "dim": 3}, "src": [t(4)]},
    ]


def _muladd_ctx():
    return {"pilInfo": {}, "publics": [123456789], "evals": [[5, 6, 7]], "challenges": [[], [[1, 2, 3], [P - 1, 0, 9]]],
            "subproofValues": [77, P - 2]}


def test_muladd_and_subproof_values_through_the_encoder(oracle):
    from pil2gl import stark
    from stark_backend import OracleBackend
    code, ctx, n_bits = _muladd_program(), _muladd_ctx(), 3
    ops, n_tmp, secs, scalars = stark.encode_code(code, "ext", ctx)
    assert len(ops) == len(code) + 4 and n_tmp == 6                       # each muladd = mul into one shared extra temporary + add
    x = rand_field(np.random.default_rng(5), (1 << n_bits, 3))
    be = OracleBackend()
    bufs = {"x_ext": be.from_host(x), "f_ext": be.zeros(3 << n_bits)}
    be.eval_program(ops, n_tmp, [(bufs[s], 3) for s in secs], scalars, n_bits, 0)
    got = np.asarray(be.to_host(bufs["f_ext"])).view(np.uint64).reshape(-1, 3)
    for i in range(1 << n_bits):
        def resolve(r, i=i):
            ty = r["type"]
            if ty == "number": return int(r["value"], 0) % P
            if ty == "eval": return list(ctx["evals"][r["id"]])
            if ty == "challenge": return list(ctx["challenges"][r["stage"] - 1][r["stageId"]])
            if ty == "public": return ctx["publics"][r["id"]]
            if ty == "subproofValue": return ctx["subproofValues"][r["id"]]
            if ty == "x": return [int(v) for v in x[i]]
            raise ValueError(ty)
        assert [int(v) for v in got[i]] == stark.execute_code(code[:-1], resolve), i
    # the global form indexes [subproofId][id]
    g = dict(ctx, subproofValues=[[77, 2], [3, P - 2]]); g["global"] = True
    where = {1: 1, 0: 0}                                                  # id -> subproofId holding the same value
    code2 = [dict(c, src=[dict(r, subproofId=where[r["id"]]) if r["type"] == "subproofValue" else r for r in c["src"]]) for c in code]
    assert (stark.encode_code(code2, "ext", g)[3] == scalars).all()
    with pytest.raises(ValueError):
        stark.encode_code([{"op": "divide", "dest": code[0]["dest"], "src": code[0]["src"][:2]}], "ext", ctx)


def test_js_and_python_encoders_agree_on_muladd_and_subproof_values():
    """js/prover_helpers.js::encode and pil2gl.stark.encode_code write the same glx_op records and the same scalar pool"""
    import ctypes
    import shutil
    import subprocess
    from conftest import ROOT
    from pil2gl import stark
    node = shutil.which("node")
    if node is None:
        pytest.skip("node not installed")
    code, ctx = _muladd_program(), _muladd_ctx()
    ops, n_tmp, secs, scalars = stark.encode_code(code, "ext", ctx)
    prog = stark.make_c_program(ops, n_tmp)
    want = bytes(ctypes.string_at(prog._keep, ctypes.sizeof(prog._keep)))
    js = ("const ph=require(%r);const code=%s;const c=%s;"
          "const ctx={pilInfo:{},publics:c.publics.map(BigInt),evals:c.evals.map(e=>e.map(BigInt)),challenges:c.challenges.map(s=>s.map(e=>e.map(BigInt))),subproofValues:c.subproofValues.map(BigInt)};"
          "const e=ph.encode(code,'ext',ctx,false);"
          "console.log(JSON.stringify({nOps:e.nOps,nTmp:e.nTmp,ops:Buffer.from(e.ops.buffer).toString('hex'),scalars:Array.from(e.scalars).map(String),secs:e.sections.map(s=>s.name)}))"
          ) % (os.path.join(ROOT, "pil2-stark-js_amd", "js", "prover_helpers.js"), json.dumps(code),
               json.dumps({k: ([[str(v) for v in e] for e in ctx[k]] if k == "evals" else [[[str(v) for v in e] for e in s] for s in ctx[k]] if k == "challenges" else [str(v) for v in ctx[k]])
                           for k in ("publics", "evals", "challenges", "subproofValues")}))
    out = subprocess.run([node, "-e", js], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert (got["nOps"], got["nTmp"], got["secs"]) == (len(ops), n_tmp, secs)
    assert [int(v) for v in got["scalars"]] == [int(v) for v in scalars]
    assert bytes.fromhex(got["ops"]) == want
