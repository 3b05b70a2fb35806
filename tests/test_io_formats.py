"""Reference on-disk formats: pols files, and the proof -> zkin mapping checked for shape against a zkin
file the reference itself wrote."""
import json
import os

import numpy as np

from conftest import GOLDEN, rand_field


def _shape(x):
    s = []
    while isinstance(x, list):
        s.append(len(x)); x = x[0] if x else None
    return s


def test_pols_file_round_trip(tmp_path):
    from pil2gl import io
    rng = np.random.default_rng(1)
    a = rand_field(rng, (1000, 7))
    f = str(tmp_path / "x.commit")
    io.save_pols(a, f)
    assert os.path.getsize(f) == a.size * 8
    assert open(f, "rb").read(8) == int(a[0, 0]).to_bytes(8, "little")
    assert np.array_equal(io.load_pols(f, 1000, 7), a.reshape(-1))


def test_proof2zkin_has_the_reference_layout(oracle):
    """a proof of the synthetic AIR (1 stage + Q), converted, has the fields and nesting of the reference-written
    test/compressor/verifier.proof.zkin.json (3 stages + Q) restricted to its own stages"""
    import stark_ref
    from pil2gl import stark, io
    ss = {"nBits": 6, "nBitsExt": 9, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 9}, {"nBits": 5}, {"nBits": 2}]}
    info, exprs, vinfo = stark.fibonacci_air(1, ss)
    cm, consts, publics = stark.fibonacci_trace(6, 1)
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    z = io.proof2zkin(res["proof"], info)
    ref = json.load(open(os.path.join(GOLDEN, "ref_compressor_verifier.proof.zkin.json")))
    assert set(z) == {"root1", "root2", "evals", "s1_root", "s1_vals", "s1_siblings", "s2_root", "s2_vals", "s2_siblings",
                      "s0_valsC", "s0_vals1", "s0_vals2", "s0_siblingsC", "s0_siblings1", "s0_siblings2", "finalPol"}
    assert set(z) <= set(ref) | {"root2"}
    for k in ("root1", "s1_root"):
        assert _shape(z[k]) == _shape(ref[k]) == [4]
    assert _shape(z["evals"])[1] == _shape(ref["evals"])[1] == 3
    assert _shape(z["finalPol"]) == [4, 3] and _shape(ref["finalPol"])[1] == 3
    assert _shape(z["s0_vals1"]) == [8, info["mapSectionsN"]["cm1"]] and _shape(z["s0_siblings1"]) == [8, 9, 4]
    assert _shape(z["s0_valsC"]) == [8, 2] and _shape(z["s1_vals"]) == [8, 3 * (1 << (9 - 5))] and _shape(z["s1_siblings"]) == [8, 5, 4]
    assert _shape(ref["s0_siblings1"])[2] == 4 and _shape(ref["s1_siblings"])[2] == 4
    txt = io.zkin_json(z, res["publics"])
    back = json.loads(txt)
    assert back["publics"] == [str(v) for v in publics] and isinstance(back["root1"][0], str)
    assert int(back["s0_vals1"][3][1]) == z["s0_vals1"][3][1]


def test_proof2zkin_two_witness_stages(oracle):
    """the mapping for nStages = 2 (roots 1..3, stage-0 openings per stage, proof2zkin.js:12-66)"""
    import stark_ref
    from pil2gl import stark, io
    ss = {"nBits": 6, "nBitsExt": 9, "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": 9}, {"nBits": 5}, {"nBits": 2}]}
    info, exprs, _ = stark.permutation_air(ss)
    cm, consts, publics = stark.permutation_trace(6)
    be = stark_ref.OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    z = io.proof2zkin(res["proof"], info)
    assert {"root1", "root2", "root3", "s0_vals1", "s0_vals2", "s0_vals3", "s0_valsC", "s0_siblings2", "finalPol"} <= set(z)
    assert _shape(z["s0_vals2"]) == [8, 9] and _shape(z["s0_vals3"]) == [8, 6] and _shape(z["s0_vals1"]) == [8, 2]
    assert z["s0_vals2"][0] == res["proof"]["fri"][0]["polQueries"][0][1][0]
