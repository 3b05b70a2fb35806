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
