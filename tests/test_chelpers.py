"""the reference's expression bytecode (`.chelpers.bin`, src/stark/chelpers) read back into programs for
the device evaluator.  The file is written by oracle/chelpers_ref.py, a line-by-line restatement of getParserArgs.js /
helpers.js / generateParser.js / binFile.js (the reference's own encoder needs chai and @iden3/binfileutils, absent here:
parity unpinned by execution for this row), and read by tests/chelpers_reader.py.  Both writer and reader are this repository's: the pair
proves nothing about the reference's files, so the reader does not travel with the product; the row stays closed."""
import copy

import pytest



def _airs():
    from pil2gl import stark
    ss = {"nBits": 6, "nBitsExt": 9, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 9}, {"nBits": 5}, {"nBits": 2}]}
    fib = stark.fibonacci_air(3, ss) + stark.fibonacci_trace(6, 3)
    perm = stark.permutation_air(ss) + stark.permutation_trace(6)
    return stark, {"fibonacci": fib, "permutation (two stages, grand-product hint)": perm}


def _through_bytecode(info, exprs, tmp_path):
    """exprs with the constraint and FRI expressions' op-lists replaced by what comes back from the .chelpers.bin"""
    import chelpers_ref
    import chelpers_reader as chelpers
    f = str(tmp_path / "air.chelpers.bin")
    chelpers_ref.write_chelpers(f, chelpers_ref.build_generic_bin_info(info, exprs))
    back = chelpers.read_chelpers_bin(f)
    out = copy.deepcopy(exprs)
    seen = 0
    for e in back["expressions"]:
        code, dest = chelpers.decode_code(e, info)
        assert len(code) == len(e["ops"]) and dest["type"] == "tmp" and dest["dim"] == 3
        if e["expId"] == info["cExpId"]:
            code[-1]["dest"] = {"type": "q", "id": 0, "dim": 3}      # stark_chelpers.js:90-95 redirected it to a temporary; generateCode.js:52-58 names it q / f
        elif e["expId"] == info["friExpId"]:
            code[-1]["dest"] = {"type": "f", "id": 0, "dim": 3}
        else:
            continue
        seen += 1
        out["expressionsCode"][e["expId"]]["code"] = {"tmpUsed": 2 * (e["nTemp1"] + e["nTemp3"]) + 2, "code": code}
    assert seen == 2
    return out, back


def test_operation_table_and_container(tmp_path):
    import chelpers_ref
    import chelpers_reader as chelpers
    t = chelpers.all_operations()
    assert len(t) == len(chelpers_ref.all_operations()) == 2 * 10 + 2 * (16 + 10 + 4) + 5 == 85     # generateParser.js:519-578
    assert t[0] == (None, "commit1", "commit1", "commit1") and t[-1] == (None, "tmp3", "commit3", "eval")
    assert sum(1 for o in t if o[0] == "mul") == 2 * 4 + 1
    stark, airs = _airs()
    info, exprs = airs["fibonacci"][0], airs["fibonacci"][1]
    f = str(tmp_path / "x.bin")
    chelpers_ref.write_chelpers(f, chelpers_ref.build_generic_bin_info(info, exprs))
    raw = open(f, "rb").read()
    assert raw[:4] == b"chps" and int.from_bytes(raw[4:8], "little") == 1 and int.from_bytes(raw[8:12], "little") == 4
    back = chelpers.read_chelpers_bin(f)
    assert [e["expId"] for e in back["expressions"]] == [e["expId"] for e in exprs["expressionsCode"]]
    for e, src in zip(back["expressions"], exprs["expressionsCode"]):
        assert len(e["ops"]) == len(src["code"]["code"]) and max(e["ops"]) < len(t)
        # live-range renumbering (helpers.js:3-98): far fewer slots than tmpUsed
        assert e["nTemp1"] + e["nTemp3"] < src["code"]["tmpUsed"]
    with pytest.raises(ValueError):
        open(f, "wb").write(b"zkey" + raw[4:]); chelpers.read_chelpers_bin(f)


@pytest.mark.parametrize("air", ["fibonacci", "permutation (two stages, grand-product hint)"])
def test_proof_from_bytecode_equals_proof_from_oplists_cpu(oracle, tmp_path, air):
    """the whole proof, with the constraint and FRI expressions taken from the bytecode file, equals the proof from the JSON
    op-lists (checker backend: encoder + evaluator semantics, incl. sub_swap and the split temporaries)"""
    import stark_ref
    stark, airs = _airs()
    info, exprs, vinfo, cm, consts, publics = airs[air]
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    want = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    exprs2, back = _through_bytecode(info, exprs, tmp_path)
    got = stark.stark_gen(be, be.from_host(cm), setup, info, exprs2, publics)
    assert got["proof"] == want["proof"] and got["challenges"] == want["challenges"]
    kinds = {o for e in back["expressions"] for o in e["args"][:1]}
    assert kinds <= {0, 1, 2, 3}


@pytest.mark.gpu
@pytest.mark.parametrize("jit", ["0", "1"])
def test_proof_from_bytecode_on_gpu(oracle, tmp_path, jit, monkeypatch):
    monkeypatch.setenv("PIL2GL_EXPR_JIT", jit)
    from pil2gl import stark
    ss = {"nBits": 10, "nBitsExt": 13, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 13}, {"nBits": 9}, {"nBits": 4}]}
    info, exprs, vinfo = stark.fibonacci_air(20, ss)
    cm, consts, publics = stark.fibonacci_trace(10, 20)
    gpu = stark.GpuBackend(0)
    setup = stark.build_const_tree(gpu, consts, info)
    want = stark.stark_gen(gpu, gpu.from_host(cm), setup, info, exprs, publics)
    exprs2, _ = _through_bytecode(info, exprs, tmp_path)
    got = stark.stark_gen(gpu, gpu.from_host(cm), setup, info, exprs2, publics)
    assert got["proof"] == want["proof"]
