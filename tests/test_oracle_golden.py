"""The CPU oracle against vectors produced by the reference's own modules
(tests/golden/*.json, generator: oracle/gen_golden.js) and the reference's KATs."""
import numpy as np
from conftest import golden, H, U, P, rand_field


def test_field_ops(oracle):
    g = golden("field.json")
    assert int(g["p"], 16) == P
    for k, v in enumerate(g["w"]):
        assert oracle.root(k) == int(v, 16)
    for k, v in enumerate(g["wi"]):
        assert oracle.root_inv(k) == int(v, 16)
    assert oracle.inv(7) == int(g["shiftInv"], 16) == 0x249249246db6db6e
    for a, b, m, s, d in H(g["mul"]):
        assert oracle.mul(a, b) == m == (a * b) % P
        assert oracle.add(a, b) == s
        assert oracle.sub(a, b) == d
    for a, ai in H(g["inv"]):
        assert oracle.inv(a) == ai
    for a, b, m, ai in H(g["ext"]):
        assert list(oracle.mul3(a, b)) == m
        assert list(oracle.inv3(a)) == ai
    a, r = H(g["batchInverse"])
    assert list(oracle.batch_inverse(a)) == r
    a, r = H(g["batchInverse3"])
    assert oracle.batch_inverse3(a).tolist() == r


def test_f3g_kat(oracle):
    # test/f3g.test.js:33-38 and SURVEY 8(c)(6)
    assert list(oracle.mul3([1, 2, 3], [4, 5, P - 1])) == [17, 23, 18]
    assert list(oracle.inv3([1, 2, 3])) == [0xba2e8ba22e8ba2ea, 0x2e8ba2e88ba2e8bb, 0x5d1745d11745d174]
    assert oracle.root(24) == 0x259b60f3625bae63 and oracle.root(32) == 0x64fdd1a46201e246


def test_scalar_ntt(oracle):
    g = golden("ntt.json")
    for c in g["cases"]:
        p = U(c["p"])
        assert oracle.fft(p).tolist() == H(c["fft"]), c["name"]
        assert oracle.ifft(p).tolist() == H(c["ifft"]), c["name"]
        for eb, v in c["ext"].items():
            assert oracle.extend_pol(p, int(eb)).tolist() == H(v), (c["name"], eb)
    e = g["ext3"]
    assert oracle.fft3(U(e["p"])).tolist() == H(e["fft"])
    assert oracle.fft3(U(e["p"]), inverse=True).tolist() == H(e["ifft"])


def test_extend_pol_kat(oracle):
    # extendPol([0..7], 1)
    exp = [0xeb97598f66614b7c, 0x4b60bddb32c4dde5, 0xc386f9d2f9d1b4db, 0xff30c18f876b47f1, 0x1426f9dff9d3772f,
           0x9c42431fa54a26f1, 0x4908a66d9991e092, 0x352f5b656daa8606, 0x4908a66d999cdf6b, 0xa972e3ef2f068223,
           0x07d9062e062c88ca, 0x08647aa3b8607a17, 0xb7390621062e4b38, 0x5c281b13f8ea7917, 0xeb97598f666ff49b,
           0xd5fd68655289b802]
    assert oracle.extend_pol(np.arange(8, dtype=np.uint64), 1).tolist() == exp


def test_cols_match_scalar(oracle):
    # test/fft_p.test.js semantics: multi-column == per-column scalar
    rng = np.random.default_rng(1)
    from conftest import rand_field
    for nb, npols, eb in [(3, 1, 1), (5, 2, 1), (8, 5, 3), (0, 3, 2)]:
        a = rand_field(rng, ((1 << nb), npols))
        f = oracle.fft_cols(a, nb); i = oracle.ifft_cols(a, nb); e = oracle.interpolate(a, nb, nb + eb)
        for c in range(npols):
            col = np.ascontiguousarray(a[:, c])
            assert (f[:, c] == oracle.fft(col)).all()
            assert (i[:, c] == oracle.ifft(col)).all()
            if nb > 0:
                assert (e[:, c] == oracle.extend_pol(col, eb)).all()


def test_poseidon(oracle):
    # first three entries are test/poseidon.test.js:14,26,38
    g = H(golden("poseidon.json"))
    assert g[0][2][:4] == [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca]
    assert g[1][2][:4] == [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2]
    assert g[2][2][:4] == [0xbe0085cfc57a8357, 0xd95af71847d05c09, 0xcf55a13d33c1c953, 0x95803a74f4530e82]
    for inp, cap, out in g:
        assert oracle.poseidon(inp, cap, 12).tolist() == out
        assert oracle.poseidon(inp, cap, 4).tolist() == out[:4]


def test_linear_hash(oracle):
    g = golden("linearhash.json")
    for w, plain, split in H(g["index"]):
        v = np.arange(w, dtype=np.uint64)
        assert oracle.linear_hash(v, False).tolist() == plain, w
        assert oracle.linear_hash(v, True).tolist() == split, w
    for v, plain, split in H(g["random"]):
        assert oracle.linear_hash(v, False).tolist() == plain
        assert oracle.linear_hash(v, True).tolist() == split


def test_merkle_roots(oracle):
    for N, w, split, root, leaf0, leaf_last in H(golden("merkle.json")):
        elems = (np.arange(N, dtype=np.uint64)[:, None] + np.uint64(1000) * np.arange(w, dtype=np.uint64)[None, :])
        nodes = oracle.merkelize(np.ascontiguousarray(elems), bool(split))
        assert nodes.size == oracle.merkle_num_nodes(N)
        assert nodes[-4:].tolist() == root, (N, w, split)
        assert nodes[:4].tolist() == leaf0 and nodes[4 * (N - 1):4 * N].tolist() == leaf_last
        # getGroupProof / verifyGroupProof round trip (test/merklehash_p.test.js:19-100)
        for idx in {3 % N, N - 1, 0}:
            sib = oracle.group_proof(nodes, N, idx)
            assert oracle.root_from_proof(elems[idx], idx, sib, bool(split)).tolist() == root


def test_merkle_num_nodes(oracle):
    # merklehash_p.js:28-42: 4*(2h-1) for powers of two >= 2
    for k in range(1, 20):
        assert oracle.merkle_num_nodes(1 << k) == 4 * (2 * (1 << k) - 1)
    assert oracle.merkle_num_nodes(33) == 4 * (34 + 18 + 10 + 6 + 4 + 2 + 1)
    assert oracle.merkle_num_nodes(1) == 8   # reference quirk: no level loop, root = zero words


def test_transcript(oracle):
    for c in golden("transcript.json"):
        t = oracle.Transcript()
        puts = H(c["put"])
        t.put(puts[0])
        fields = H(c.get("fields", []))
        if fields:
            assert t.get_field().tolist() == fields[0]
        if len(puts) > 1:
            t.put(puts[1])
            assert t.get_field().tolist() == fields[1]
            assert t.get_field().tolist() == fields[2]
        if "perms" in c:
            n, nb, res = c["perms"]
            assert t.get_permutations(n, nb).tolist() == res
        if "state" in c:
            assert t.get_state().tolist() == H(c["state"])


def test_fri_fold(oracle):
    for pol_bits, out_bits, bits0, bits_prev, ch, pol, res in H(golden("fri_fold.json")):
        sinv = oracle.fri_shift_inv(bits0, bits_prev)
        out = oracle.fri_fold(np.array(pol, dtype=np.uint64), out_bits, sinv, ch)
        assert out.tolist() == res, (pol_bits, out_bits)


def test_zerofiers(oracle):
    for nb, nbe, zh, first, last, frame in H(golden("zerofiers.json")):
        assert oracle.build_zhinv(nb, nbe).tolist() == zh
        assert oracle.build_one_row_zerofier_inv(nb, nbe, 0).tolist() == first
        assert oracle.build_one_row_zerofier_inv(nb, nbe, (1 << nb) - 1).tolist() == last
        assert oracle.build_frame_zerofier(nb, nbe, 2, 1).tolist() == frame


def test_hint_columns_against_python_integers(oracle):
    """calculateZ / calculateS (polutils.js:128-164) restated with Python integers on base-field columns"""
    rng = np.random.default_rng(3)
    n = 50
    num = [int(x) for x in rand_field(rng, n)]; den = [int(x) or 1 for x in rand_field(rng, n)]
    z = [1]
    for i in range(1, n):
        z.append(z[-1] * num[i - 1] * pow(den[i - 1], P - 2, P) % P)
    assert [int(x) for x in oracle.gprod(np.array(num, dtype=np.uint64), np.array(den, dtype=np.uint64), 1, 1)] == z
    s, acc = [], 0
    for i in range(n):
        acc = (acc + num[0] * pow(den[i], P - 2, P)) % P
        s.append(acc)
    assert [int(x) for x in oracle.gsum(np.array(num[:1], dtype=np.uint64), np.array(den, dtype=np.uint64), 1, 1)] == s
    # extension columns: dim-1 operands embed as (v, 0, 0)
    num3 = np.zeros((n, 3), np.uint64); num3[:, 0] = num
    den3 = np.zeros((n, 3), np.uint64); den3[:, 0] = den
    z3 = oracle.gprod(num3, den3, 3, 3).reshape(n, 3)
    assert [int(x) for x in z3[:, 0]] == z and not z3[:, 1:].any()


# ---- stage-2 hints and the zkin mapping against the reference's own functions (oracle/gen_golden.js runs
#      polutils.js calculateZ / calculateS / calculateH1H2 and src/proof2zkin.js as they are)
def _rows(v, dim):
    """reference values -> n x dim ints: an extension column may hold the scalar F.one in row 0 (polutils.js:135)"""
    out = []
    for r in v:
        r = H(r)
        out.append(list(r) if isinstance(r, list) else [r] + [0] * (dim - 1))
    return out


def test_hints_against_reference_vectors(oracle):
    g = golden("hints.json")
    for c in g["gprod"]:
        dn, dd = c["dimNum"], c["dimDen"]
        dim = max(dn, dd)
        num = np.array(_rows(c["num"], dn), dtype=np.uint64).reshape(-1); den = np.array(_rows(c["den"], dd), dtype=np.uint64).reshape(-1)
        assert oracle.gprod(num, den, dn, dd).reshape(-1, dim).tolist() == _rows(c["gprod"], dim), (c["n"], dn, dd)
    for c in g["gsum"]:
        dn, dd = c["dimNum"], c["dimDen"]
        dim = max(dn, dd)
        num = np.array(_rows([c["num"]], dn), dtype=np.uint64).reshape(-1); den = np.array(_rows(c["den"], dd), dtype=np.uint64).reshape(-1)
        assert oracle.gsum(num, den, dn, dd).reshape(-1, dim).tolist() == _rows(c["gsum"], dim), (c["n"], dn, dd)
    for c in g["h1h2"]:
        key = (lambda r: H(r)) if c["dim"] == 1 else (lambda r: tuple(H(r)))
        h1, h2 = oracle.h1h2([key(r) for r in c["f"]], [key(r) for r in c["t"]])
        assert h1 == [key(r) for r in c["h1"]] and h2 == [key(r) for r in c["h2"]], (c["n"], c["dim"])


def test_proof2zkin_against_reference_vectors():
    from pil2gl import io
    for c in golden("proof2zkin.json"):
        z = io.proof2zkin(H(c["proof"]), c["starkInfo"])
        want = H(c["zkin"])
        assert list(z) == list(want), "field order differs"            # JSON.stringify keeps insertion order: so does the file
        assert z == want
