"""GPU parity of the BN128 Merkle path through the C ABI against the CPU oracle, bit-exact, and the
reference-written final proof replayed through the device permutation."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, P, rand_field

pytestmark = pytest.mark.gpu

ROOT_C = 4996530440829051953383356903338638570209318620783152464040752714611525657817   # test/final/verifier.circom:3299


@pytest.fixture(scope="module")
def bn():
    import pil2gl
    pil2gl.init(0)
    from pil2gl import bn128
    return bn128


@pytest.fixture(scope="module")
def orc():
    import bn128_oracle
    return bn128_oracle


def test_poseidon_every_width(bn, orc):
    rng = np.random.default_rng(5)
    for n_in in range(1, 17):
        ins = [[int.from_bytes(rng.bytes(32), "little") % orc.R for _ in range(n_in)] for _ in range(3)]
        ins.append([0] * n_in); ins.append([orc.R - 1] * n_in)
        init = [int.from_bytes(rng.bytes(32), "little") % orc.R for _ in ins]
        init[-2] = 0
        n_out = 1 if n_in % 3 else n_in + 1
        got = bn.poseidon_batch(ins, init, n_out)
        for a, s, g in zip(ins, init, got):
            assert g == orc.poseidon(a, s, n_out), n_in
    # inputs above the modulus are reduced like F.e()
    assert bn.poseidon([orc.R + 5, 7], orc.R + 1, 2) == orc.poseidon([5, 7], 1, 2)
    # batches up to 2048 permutations get a wave each (the row-split kernel of the transcript chain), larger ones a lane
    # each: the same inputs through both, and a sample against the oracle
    for n_in, n_out in ((2, 1), (8, 9), (16, 3)):
        ins = [[int.from_bytes(rng.bytes(32), "little") % orc.R for _ in range(n_in)] for _ in range(2100)]
        init = [int.from_bytes(rng.bytes(32), "little") % orc.R for _ in ins]
        big = bn.poseidon_batch(ins, init, n_out)
        small = sum((bn.poseidon_batch(ins[k:k + 700], init[k:k + 700], n_out) for k in range(0, 2100, 700)), [])
        assert big == small, n_in
        for k in (0, 1234, 2099):
            assert big[k] == orc.poseidon(ins[k], init[k], n_out), (n_in, k)
    assert bn.poseidon_batch([[3, 4]] * 3, None, 1) == [[orc.poseidon([3, 4], 0, 1)[0]]] * 3          # no initial states given


def test_dense_statement_of_the_permutation_agrees():
    """the library's default runs the partial rounds in sparse form; PIL2GL_BN128_DENSE=1 runs poseidon.circom:22-44 as written"""
    import subprocess
    import sys
    code = ("import sys, os; sys.path[:0] = [%r, %r]; import numpy as np; import pil2gl; pil2gl.init(0); from pil2gl import bn128; import bn128_oracle as o\n"
            "for n in (1, 2, 4, 5, 8, 16):\n"
            "    a = [(7 ** (k + 3)) %% o.R for k in range(n)]\n"
            "    assert bn128.poseidon(a, 11, n + 1) == o.poseidon(a, 11, n + 1), n\n"
            "print('dense ok')\n") % (os.path.join(os.path.dirname(GOLDEN), "..", "pil2-stark-js_amd", "python"), os.path.join(os.path.dirname(GOLDEN), "..", "oracle"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PIL2GL_BN128_DENSE="1"))
    assert r.returncode == 0 and "dense ok" in r.stdout, r.stdout + r.stderr


def test_prefetching_instances_of_the_permutation_agree():
    """PIL2GL_BN128_WIDE=1 selects the kernel instances that request the next term's LDS operand and table entry before multiplying the
    current one (round 1's default for states of >= 10 elements; since round 3 the plain instances run everywhere and these are an A/B
    switch): permutations of every width class, the arity-16 leaf rule and a small tree must still be the oracle's"""
    import subprocess
    import sys
    code = ("import sys, os; sys.path[:0] = [%r, %r]; import numpy as np; import pil2gl; pil2gl.init(0); from pil2gl import bn128; import bn128_oracle as o\n"
            "for n in (1, 2, 4, 8, 9, 10, 12, 16):\n"
            "    a = [(5 ** (k + 7)) %% o.R for k in range(n)]\n"
            "    assert bn128.poseidon(a, 3, n + 1) == o.poseidon(a, 3, n + 1), n\n"
            "rows = [[(i * 131 + j * 7 + 1) %% 0xFFFFFFFF00000001 for j in range(100)] for i in range(70)]\n"
            "t = bn128.buildMerkleHash(16, False).merkelize(np.array(rows, dtype=np.uint64).reshape(-1), 100, 70)\n"
            "assert bn128.from_montgomery(t['nodes']) == o.merkelize(rows, 16, False)\n"
            "print('wide ok')\n") % (os.path.join(os.path.dirname(GOLDEN), "..", "pil2-stark-js_amd", "python"), os.path.join(os.path.dirname(GOLDEN), "..", "oracle"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, PIL2GL_BN128_WIDE="1"))
    assert r.returncode == 0 and "wide ok" in r.stdout, r.stdout + r.stderr


def test_montgomery_conversion(bn, orc):
    vals = [0, 1, orc.R - 1, 1 << 200, 0x123456789ABCDEF << 100]
    m = bn.to_montgomery(vals)
    for v, w in zip(vals, m):
        assert [int(x) for x in w] == orc.to_montgomery_words(v)
    assert bn.from_montgomery(m) == vals


@pytest.mark.parametrize("arity,custom", [(16, False), (4, True), (8, False), (4, False), (16, True), (2, False)])
def test_linear_hash_rows(bn, orc, arity, custom):
    from pil2gl import _lib
    import pil2gl
    rng = np.random.default_rng(arity * 2 + custom)
    for width in [1, 2, 3, 4, 5, 6, 7, 9, 12, 13, 21, 36, 47, 48, 49, 50, 97, 100]:
        if arity == 2 and width > 30:
            continue
        h = 5
        a = rand_field(rng, (h, width))
        a[0, :] = 0; a[1, :] = P - 1
        out = np.zeros((h, 4), np.uint64)
        _lib.call("pil2gl_bn128_linear_hash_rows", pil2gl._ptr(a), width, h, arity, int(custom), pil2gl._ptr(out))
        for i in range(h):
            want = orc.to_montgomery_words(orc.linear_hash_worker(a[i].tolist(), arity, custom))
            assert [int(x) for x in out[i]] == want, (width, i)


@pytest.mark.parametrize("arity,custom,N,nPols,idx", [(16, False, 256, 3, 3), (4, True, 256, 3, 3), (16, False, 256, 9, 3), (8, False, 33, 9, 32),
                                                      (4, True, 70, 21, 69), (16, False, 1, 5, 0), (16, False, 300, 100, 299), (2, False, 9, 7, 4)])
def test_merkle_tree_and_proofs(bn, orc, arity, custom, N, nPols, idx, tmp_path):
    """test/merklehash_bn128_p.test.js shapes (pols[i][j] = i + 1000 j) plus ragged heights"""
    import torch
    a = np.array([[i + j * 1000 for j in range(nPols)] for i in range(N)], dtype=np.uint64)
    MH = bn.buildMerkleHash(arity, custom)
    want = orc.merkelize(a.tolist(), arity, custom)
    for dev in (False, True):
        buf = torch.from_numpy(a.view(np.int64)).cuda().reshape(-1) if dev else a.reshape(-1).copy()
        tree = MH.merkelize(buf, nPols, N)
        nodes = tree["nodes"].cpu().numpy().view(np.uint64) if dev else tree["nodes"]
        assert len(nodes) == 4 * orc.merkle_num_nodes(N, arity)
        assert bn.from_montgomery(nodes) == want
        assert MH.root(tree) == want[-1]
        v, mp = MH.getGroupProof(tree, idx)
        assert v == a[idx].tolist() and mp == orc.group_proof(want, N, arity, idx)
        # the batch form (one launch per tree on the device): every row of a spread, repeats included, == the single openings
        rows = sorted({0, idx, N - 1, N // 2, (7 * idx + 3) % N}) + [idx]
        batch = MH.getGroupProofs(tree, rows)
        assert [b[0] for b in batch] == [a[i].tolist() for i in rows]
        assert [b[1] for b in batch] == [orc.group_proof(want, N, arity, i) for i in rows]
        if N > 1 and nPols != 4:
            assert MH.verifyGroupProof(MH.root(tree), mp, idx, v)
            bad = list(v); bad[0] ^= 1
            assert not MH.verifyGroupProof(MH.root(tree), mp, idx, bad)
    f = str(tmp_path / "t.bin")
    MH.writeToFile(tree, f)
    t2 = MH.readFromFile(f)
    assert (t2["nodes"] == nodes).all() and (t2["elements"] == a.reshape(-1)).all()
    with pytest.raises(bn.Pil2glError):
        MH.getGroupProof(tree, N)
    with pytest.raises(bn.Pil2glError):
        MH.getGroupProofs(tree, [0, N])


def test_reference_final_proof_through_gpu(bn):
    """every Merkle opening of test/final/verifier.proof.zkin.json (arity 4, t = 5 and 4) and its transcript, on the device"""
    p = json.load(open(os.path.join(GOLDEN, "ref_final_verifier.proof.zkin.json")))
    MH = bn.buildMerkleHash(4, False)
    T = bn.Transcript(16)
    T.put([int(x) for x in p["publics"]]); T.put(int(p["root1"])); T.getField(); T.getField()
    T.put(int(p["root2"])); T.getField(); T.getField()
    T.put(int(p["root3"])); T.getField()
    T.put(int(p["rootQ"])); T.getField()
    T.put([[int(x) for x in e] for e in p["evals"]]); T.getField(); T.getField()
    fri_ch = [T.getField()]
    for s in range(1, 5):
        T.put(int(p["s%d_root" % s])); fri_ch.append(T.getField())
    T.put([[int(x) for x in e] for e in p["finalPol"]])
    ys = T.getPermutations(32, 17)
    for q in range(32):
        for vk, sk, rt in (("s0_vals1", "s0_siblings1", int(p["root1"])), ("s0_vals3", "s0_siblings3", int(p["root3"])),
                           ("s0_valsQ", "s0_siblingsQ", int(p["rootQ"])), ("s0_valsC", "s0_siblingsC", ROOT_C)):
            assert MH.verifyGroupProof(rt, p[sk][q], ys[q], [int(x) for x in p[vk][q]]), (vk, q)
        for s, bits in ((1, 14), (2, 11), (3, 7), (4, 4)):
            assert MH.verifyGroupProof(int(p["s%d_root" % s]), p["s%d_siblings" % s][q], ys[q] % (1 << bits),
                                       [int(x) for x in p["s%d_vals" % s][q]]), (s, q)
    # the same 32 x 8 openings through the batch form: one batched permutation call per sponge chunk and per level
    for vk, sk, rt in (("s0_vals1", "s0_siblings1", int(p["root1"])), ("s0_vals3", "s0_siblings3", int(p["root3"])),
                       ("s0_valsQ", "s0_siblingsQ", int(p["rootQ"])), ("s0_valsC", "s0_siblingsC", ROOT_C)):
        proofs = [([int(x) for x in p[vk][q]], p[sk][q]) for q in range(32)]
        assert MH.calculateRootsFromGroupProofs(proofs, ys) == [rt] * 32, vk
    for s, bits in ((1, 14), (2, 11), (3, 7), (4, 4)):
        proofs = [([int(x) for x in p["s%d_vals" % s][q]], p["s%d_siblings" % s][q]) for q in range(32)]
        assert MH.verifyGroupProofs(int(p["s%d_root" % s]), proofs, [y % (1 << bits) for y in ys]), s
    bad = [(list(v), m) for v, m in proofs]; bad[5][0][0] += 1
    assert not MH.verifyGroupProofs(int(p["s4_root"]), bad, [y % 16 for y in ys])
    # FRI.verify (fri.js:107-174) on the reference prover's layers: BN128 trees, Goldilocks folds.  nBits 13 (verifier.circom:303),
    # steps 17/14/11/7/4; step 0 hands over the value layer 1 opened (the FRI polynomial needs the circuit's own program).
    import pil2gl
    ss = {"nBits": 13, "nBitsExt": 17, "nQueries": 32, "steps": [{"nBits": b} for b in (17, 14, 11, 7, 4)]}

    def ref_proof():
        layers = [{"root": int(p["s%d_root" % s]), "polQueries": [[[int(x) for x in p["s%d_vals" % s][q]], p["s%d_siblings" % s][q]] for q in range(32)]}
                  for s in range(1, 5)]
        return [{"polQueries": list(range(32))}] + layers + [[[int(x) for x in e] for e in p["finalPol"]]]

    def step0(q, idx):
        g = idx // (1 << 14)
        return [[int(x) for x in p["s1_vals"][q][3 * g:3 * g + 3]]]
    fri = pil2gl.FRI(ss, MH)
    assert fri.verify(fri_ch, list(ys), ref_proof(), step0)
    bad = ref_proof(); bad[3]["polQueries"][7][0][2] += 1
    assert not fri.verify(fri_ch, list(ys), bad, step0)
    bad = ref_proof(); bad[-1][5][1] += 1
    assert not fri.verify(fri_ch, list(ys), bad, step0)
    assert not fri.verify(fri_ch[:4] + [[1, 2, 3]], list(ys), ref_proof(), step0)


def test_transcript_chain_kernel(bn, orc):
    """lists go through the chained kernel (nIn+1 lanes share each permutation): the same state and the same challenges as
    the oracle's element-by-element transcript, for every sponge width and for lists that straddle block boundaries"""
    rng = np.random.default_rng(17)
    rnd = lambda n: [int.from_bytes(rng.bytes(32), "little") % orc.R for _ in range(n)]
    for n_in in (1, 2, 3, 4, 7, 8, 12, 15, 16):
        for nb in (1, 2, 5):
            blocks = [rnd(n_in) for _ in range(nb)]
            st = rnd(1)[0]
            want = None
            s = st
            for b in blocks:
                want = orc.poseidon(b, s, n_in + 1); s = want[0]
            assert bn.poseidon_chain(blocks, st) == want, (n_in, nb)
    assert bn.poseidon_chain([[orc.R - 1] * 16, [0] * 16], 0) == orc.poseidon([0] * 16, orc.poseidon([orc.R - 1] * 16, 0, 1)[0], 17)
    with pytest.raises(bn.Pil2glError):
        bn.poseidon_chain([[1] * 17], 0)
    for n_in in (4, 16):
        T, O = bn.Transcript(n_in), orc.TranscriptBN128(n_in)
        for step in ([3], rnd(2 * n_in), [rnd(3) for _ in range(n_in)], rnd(1)[0], rnd(5 * n_in - 1), [], rnd(n_in - 2), [[rnd(2), rnd(1)[0]]] * 7):
            T.put(step); O.put(step)
            assert (T.pending, T.out, T.out3, T.state) == (O.pending, O.out, O.out3, O.state)
            if not isinstance(step, int) and len(step) % 2:
                assert T.getField() == O.getField()
        assert T.getPermutations(20, 13) == O.getPermutations(20, 13) and T.getState() == O.getState()


@pytest.mark.parametrize("hbits", [24, 27])
def test_config4_shape_tree_opens(bn, hbits):
    """BASELINE config 4's shape (100 columns, BN128 linear hash, arity 16) at 2^24 rows and at the config's own 2^27 extended
    rows (107 GB of rows, a 15 s tree: skipped when the memory is not there): every opened path of the device-built tree
    recomputes the root through the ORACLE's rule (bn128_oracle.root_from_group_proof: the leaf's linear hash and the
    arity-16 levels in Python integers) and through the library's own verifier"""
    import bn128_oracle as orc
    import torch
    import gc
    gc.collect(); torch.cuda.empty_cache()
    h, w = 1 << hbits, 100
    if torch.cuda.mem_get_info()[0] < 8 * h * w * 1.25 + 8e9:
        pytest.skip("needs %.0f GB of free device memory" % ((8 * h * w * 1.25 + 8e9) / 1e9))
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    buf = torch.empty(h * w, dtype=torch.int64, device="cuda")
    for o in range(0, h * w, 1 << 28):
        m = min(1 << 28, h * w - o)
        buf[o:o + m] = torch.randint(0, 0x7FFFFFFFFFFFFFFF, (m,), dtype=torch.int64, device="cuda", generator=g) % 0xFFFFFFFF00000001
    MH = bn.buildMerkleHash(16, False)
    tree = MH.merkelize(buf, w, h)
    root = MH.root(tree)
    for idx in (0, 1, h - 1, 12345678, (h // 3) | 15):
        v, mp = MH.getGroupProof(tree, idx)
        assert len(mp) == (hbits + 3) // 4 and all(len(l) == 16 for l in mp)
        assert v == [int(x) for x in buf[idx * w:(idx + 1) * w].cpu().numpy().view(np.uint64)]
        assert MH.verifyGroupProof(root, mp, idx, v)
        assert orc.root_from_group_proof([[int(x) for x in l] for l in mp], idx, v, 16, False) == int(root)
    v[3] ^= 1
    assert not MH.verifyGroupProof(root, mp, idx, v)


def test_matrix_core_unit_kernels(tmp_path):
    """tools/bn_mfma_unit.hip on the device: the hand-scheduled carry / finish chains against a plain restatement (random, all-maximal and mixed byte
    positions), the reduction-free products and the radix-2^29 S-box against the canonical 32-bit-limb arithmetic -- 262 144 lanes each"""
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "bn_mfma_unit")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "pil2-stark-js_amd", "csrc"), os.path.join(ROOT, "tools", "bn_mfma_unit.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    lines = [l for l in out.stdout.splitlines() if "lanes differ" in l]
    assert len(lines) == 8 and all(": 0 of " in l for l in lines), out.stdout
