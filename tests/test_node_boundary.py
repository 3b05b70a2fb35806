"""The Node.js side of the boundary: addon loads (CPU) and the drop-in JS modules match the goldens (GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "pil2-stark-js_amd", "addon", "pil2gl.node")


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_addon_loads_and_exports():
    assert os.path.exists(ADDON), "addon not built (make -C pil2-stark-js_amd)"
    js = ("const m=require(%r);const a=m.native;"
          "for (const k of ['interpolate','fft','ifft','merkelize','merkelizeLevel','linearHashRows','poseidon','friFold',"
          "'friTranspose','buildXDev','buildZhInvDev','computeQSplitDev','xDivXSubXiDev','buildLevDev','computeEvalsDev','gprodDev','gsumDev','h1h2Dev','bn128Poseidon','bn128Merkelize','bn128MerkelizeDev','bn128LinearHashRows','bn128Convert','devAlloc','devFree','devUpload','devDownload','interpolateDev','merkelizeDev','groupProofDev','rootsFromGroupProofs','spongeAbsorb','bn128SpongeAbsorb','friVerifyFold','evalProgramDev'])"
          " if (typeof a[k] !== 'function') throw new Error('missing '+k);"
          "if (a.merkleNumNodes(256) !== 2044) throw new Error('merkleNumNodes');"
          "if (a.bn128MerkleNumNodes(256, 16) !== 273) throw new Error('bn128MerkleNumNodes');"
          "for (const k of ['fft','ifft','interpolate']) if (typeof m.fft_p[k] !== 'function') throw new Error(k);"
          "console.log('ok')") % os.path.join(ROOT, "pil2-stark-js_amd", "js", "index.js")
    out = subprocess.run([NODE, "-e", js], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_modules_match_goldens_on_gpu():
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "addon_parity.js")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "addon parity OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_whole_proof_driven_from_node():
    """prover.js's stage order over the JS drop-in modules: the proof equals the CPU checker's proof field by field"""
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "prove_flow.js")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "prove flow OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_config2_proof_from_node_device_resident():
    """config 2's shape proved from Node with HBM-resident buffers: digest of the proof equals the CPU checker's"""
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "prove_c2.js")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "prove c2 OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _constraint_errors(g, flips, points=None):
    """what calculateExps(ctx, constraint, "n", debug = true) of the reference leaves in ctx.errors for every stage-1 constraint of a
    golden AIR (prover_helpers.js:46-70): rows of the boundary one after the other, F3g arithmetic on Python integers, the FIRST row
    whose value is not zero"""
    P_ = 0xFFFFFFFF00000001
    info, N = g["pilInfo"], 1 << g["pilInfo"]["starkStruct"]["nBits"]
    w, nc = info["mapSectionsN"]["cm1"], info["nConstants"]
    cm = [int(v) for v in g["cm1"]]
    for row, col, delta in flips:
        cm[row * w + col] = (cm[row * w + col] + delta) % P_
    consts, pub = [int(v) for v in g["consts"]], [int(v) for v in g["publics"]]
    im = g["expressionsInfo"]["imPolsCode"][0]

    def run(code, i, write):
        tmp = {}

        def get(r):
            t = r["type"]
            if t == "tmp": return tmp[r["id"]]
            if t == "cm": return cm[((i + r.get("prime", 0)) % N) * w + info["cmPolsMap"][r["id"]]["stagePos"]]
            if t == "const": return consts[((i + r.get("prime", 0)) % N) * nc + r["id"]]
            if t == "number": return int(r["value"]) % P_
            if t == "public": return pub[r["id"]]
            raise AssertionError(t)
        last = None
        for c in code:
            a = get(c["src"][0])
            b = get(c["src"][1]) if c["op"] != "copy" else 0
            last = {"add": (a + b) % P_, "sub": (a - b) % P_, "mul": a * b % P_, "copy": a}[c["op"]]
            if c["dest"]["type"] == "tmp": tmp[c["dest"]["id"]] = last
            elif write: cm[i * w + info["cmPolsMap"][c["dest"]["id"]]["stagePos"]] = last
        return last
    for i in range(N):                                  # the intermediate polynomials are the prover's own (prover.js:212-214)
        run(im["code"], i, True)
    errors = []
    for c in g["expressionsInfo"]["constraints"]:
        if c["stage"] != 1:
            continue
        b = c["boundary"]
        first, last = {"everyRow": (0, N), "firstRow": (0, 1), "lastRow": (N - 1, N)}.get(b) or (c["offsetMin"], N - c["offsetMax"])
        for i in range(first, last):
            v = run(c["code"], i, False)
            if v:
                errors.append("%s: identity does not match w=%d val=%d " % (c["line"], i, v))
                break
    if points is not None:                              # calculateExpAtPoint: the first stage-1 constraint's value at the rows asked for
        c0 = [c for c in g["expressionsInfo"]["constraints"] if c["stage"] == 1][0]
        return errors, [str(run(c0["code"], i, False)) for i in points]
    return errors


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_constraint_check_debug_mode_from_node(tmp_path):
    """calculateExps / callCalculateExps with debug = true and calculateExpAtPoint on the device (prover_helpers.js:46-80): a valid witness
    leaves no error; one altered cell leaves the reference's message -- same constraint line, same first failing row, same value -- for
    every boundary kind (everyRow on selector constants, everyFrame / firstRow / lastRow of pil2, an intermediate polynomial's identity)"""
    import json
    cases = {
        # (row, column, delta): transitions, the public inputs' rows, the output's row, a cell that breaks two constraints at different rows
        "fib_flow.json": [[], [[5, 0, 1]], [[0, 1, 7]], [[63, 0, 3]], [[0, 0, 1], [40, 3, 2]]],
        "fib_flow_boundaries_only.json": [[], [[17, 1, 1]], [[0, 0, 5]], [[63, 0, 1]], [[62, 2, 9]], [[63, 3, 4]]],
        "fib_flow_impols.json": [[], [[9, 2, 1]], [[30, 0, 1]]],
        "fib_flow_prevrow.json": [[], [[1, 1, 1]]],
    }
    for golden, variants in cases.items():
        g = json.load(open(os.path.join(ROOT, "tests", "golden", golden)))
        job = tmp_path / (golden + ".job.json")
        job.write_text(json.dumps({"golden": golden, "variants": [{"flips": f} for f in variants], "points": [0, 1, 63]}))
        out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "debug_flow.js"), str(job)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "debug flow OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
        res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        for f, r in zip(variants, res["results"]):
            want = _constraint_errors(g, f)
            assert (not f) == (not want), (golden, f)                 # every altered cell above is caught by some constraint
            assert r["host"] == want and r["dev"] == want, (golden, f, r, want)
        assert res["at"] == _constraint_errors(g, [], [0, 1, 63])[1], (golden, res["at"])     # calculateExpAtPoint (an everyFrame identity need not vanish on the last row)


def _canon(v):          # canonical text of a proof: decimal strings, no whitespace, keys in insertion order (= tests/js/prove_c3.js)
    if isinstance(v, dict):
        return "{" + ",".join('"%s":%s' % (k, _canon(x)) for k, x in v.items()) + "}"
    if isinstance(v, (list, tuple)):
        return "[" + ",".join(_canon(x) for x in v) + "]"
    return '"%d"' % int(v)


def _node_vs_python_proof(tmp_path, n_bits, n_cols, steps, n_queries):
    """the same witness proved twice on the device -- stage loop in Python (pil2gl.stark.stark_gen) and stage loop in Node
    (tests/js/prove_flow.js over the JS drop-ins, device-resident) -> (python seconds, node seconds, digests equal)"""
    import hashlib
    import json
    import time
    import numpy as np
    import torch
    import bench
    from pil2gl import stark
    ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": n_queries, "verificationHashType": "GL", "splitLinearHash": False,
          "steps": [{"nBits": b} for b in steps]}
    info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
    gpu = stark.GpuBackend(0, False)
    dev = torch.device("cuda", 0)
    cm, consts, publics = bench.fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
    start = [int(v) for v in cm[:n_cols].cpu().numpy().view(np.uint64)]
    setup = stark.build_const_tree(gpu, consts, info)
    best, res = 1e9, None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = stark.stark_gen(gpu, cm, setup, info, exprs, publics)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    digest = hashlib.sha256(_canon(res["proof"]).encode()).hexdigest()
    job = {"pilInfo": info, "expressionsInfo": exprs, "start": [str(v) for v in start], "publics": [str(v) for v in publics],
           "constRoot": [str(v) for v in setup["constRoot"]], "queries": res["queries"]}
    f = tmp_path / "job.json"
    f.write_text(json.dumps(job))
    del cm, setup, res
    torch.cuda.empty_cache()
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "prove_c3.js"), str(f), "3"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "prove c3 OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    return best, line["proof_seconds"], line["proofSha256"] == digest


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_driven_proof_equals_python_driven_proof_2e18(tmp_path):
    """2^18 x 20: same digest (the reference's JS orchestration and the Python mirror sequence the same C-ABI calls)"""
    _, _, same = _node_vs_python_proof(tmp_path, 18, 20, (21, 16, 11, 6), 32)
    assert same


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_config3_proof_driven_from_node(tmp_path):
    """BASELINE config 3 (2^24 x 100, FRI 27/22/17/12/7, 64 queries) with Node driving: the north_star boundary at the size
    that matters.  Same proof as the Python-driven one; the wall time is reported, and must stay within 10 % of it"""
    import torch
    if torch.cuda.mem_get_info()[0] < 230e9:
        pytest.skip("needs ~200 GB of free device memory")
    t_py, t_node, same = _node_vs_python_proof(tmp_path, 24, 100, (27, 22, 17, 12, 7), 64)
    print("config 3 proof: python-driven %.3f s, node-driven %.3f s" % (t_py, t_node))
    assert same
    assert t_node < 1.10 * t_py, (t_py, t_node)


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_verifier_drop_in_on_checker_proofs_from_node():
    """js/stark_verify.js (starkVerify with the reference's argument list; batched openings, the query program on the device evaluator,
    FRI.verify): accepts the six golden proofs of the CPU checker with their own verifier programs and rejects every alteration"""
    out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "verify_flow.js")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "verify flow OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _strs(v):
    if isinstance(v, dict):
        return {k: _strs(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_strs(x) for x in v]
    if isinstance(v, int) and not isinstance(v, bool):
        return str(v)
    return v


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_reference_written_proofs_verified_from_node(tmp_path):
    """BOTH proofs the reference prover wrote -- test/compressor (Goldilocks Poseidon, transcript replayed) and test/final (BN128 Poseidon,
    arity 4, the older transcript layout: challenges handed over as starkVerify's fourth argument) -- accepted whole by the JS verifier
    drop-in with the programs of their own verifier circuits, and rejected after every alteration"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_reference_proof as R
    from conftest import golden
    from pil2gl import io, stark
    info, vinfo, root_c = R._programs()
    z = golden("ref_compressor_verifier.proof.zkin.json")
    case = {"proof": io.zkin2proof(z, info), "publics": [int(v) for v in z["publics"]], "constRoot": root_c, "starkInfo": info, "verifierInfo": vinfo}
    big = lambda c: {k: (_strs(v) if k in ("proof", "publics", "constRoot", "challenges") else v) for k, v in c.items()}     # field elements as decimal strings; the infos keep their numbers
    f1 = tmp_path / "compressor.json"; f1.write_text(json.dumps(big(case)))
    info2, vinfo2, root_c2, z2 = R._final()
    ss = info2["starkStruct"]
    be = stark.GpuBackend(0, hash_type=ss["verificationHashType"], arity=ss["merkleTreeArity"], custom=ss["merkleTreeCustom"])
    tr = R._final_transcript(be.new_transcript(), z2, info2)
    case2 = {"proof": io.zkin2proof(z2, info2), "publics": [int(v) for v in z2["publics"]], "constRoot": root_c2, "starkInfo": info2, "verifierInfo": vinfo2, "challenges": tr}
    f2 = tmp_path / "final.json"; f2.write_text(json.dumps(big(case2)))
    for f in (f1, f2):
        out = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "verify_flow.js"), str(f)], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "verify case OK 1" in out.stdout, f.name + ": " + out.stdout[-2000:] + out.stderr[-4000:]
