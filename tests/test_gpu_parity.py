"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, the golden vectors and the
reference's own test shapes.  Bit-exact (integer field arithmetic, no tolerance)."""
import numpy as np
import pytest

from conftest import golden, H, U, P, rand_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gl():
    import pil2gl
    pil2gl.init(0)
    return pil2gl


# ------------------------------------------------------------------ field
def test_field_ops_device(gl, oracle):
    from pil2gl import _lib
    g = golden("field.json")
    rows = H(g["mul"])
    rng = np.random.default_rng(7)
    a = np.concatenate([np.array([r[0] for r in rows], dtype=np.uint64), rand_field(rng, 4096)])
    b = np.concatenate([np.array([r[1] for r in rows], dtype=np.uint64), rand_field(rng, 4096)])
    m = np.zeros_like(a); s = np.zeros_like(a); d = np.zeros_like(a)
    _lib.call("pil2gl_selftest_field", gl._ptr(a), gl._ptr(b), a.size, gl._ptr(m), gl._ptr(s), gl._ptr(d))
    for i, r in enumerate(rows):
        assert (int(m[i]), int(s[i]), int(d[i])) == (r[2], r[3], r[4])
    for i in range(len(rows), a.size):
        x, y = int(a[i]), int(b[i])
        assert int(m[i]) == x * y % P and int(s[i]) == (x + y) % P and int(d[i]) == (x - y) % P


def test_handwritten_products_on_edge_operands(gl):
    """the carry-out products of the S-boxes (flagged form) and of the transform kernels (exact form) on operands chosen for their
    carries: every combination of 32-bit halves from {0, 1, 2^31, 2^32-2, 2^32-1}, values >= p, and pairs whose reduction borrows
    at its last step -- w1 > z with a zero low product, e.g. 2^63 squared -- where the exact form folds the borrow back and the
    flagged form must raise its flag (and is then not used); plus random operands, on which the flag stays down"""
    from pil2gl import _lib
    h = [0, 1, 1 << 31, (1 << 32) - 2, (1 << 32) - 1, 0x80000001, 0x7FFFFFFF]
    vals = [(hi << 32) | lo for hi in h for lo in h]
    pairs = [(x, y) for x in vals for y in vals]
    pairs += [(1 << 63, 1 << 63), ((1 << 63) + (1 << 31), 1 << 63), (0xFFFFFFFF00000000, 0xFFFFFFFF00000000), (P, P), (P - 1, P - 1), ((1 << 64) - 1, (1 << 64) - 1)]
    rng = np.random.default_rng(17)
    ra = rng.integers(0, 1 << 64, 1 << 16, dtype=np.uint64); rb = rng.integers(0, 1 << 64, 1 << 16, dtype=np.uint64)
    a = np.concatenate([np.array([p_[0] for p_ in pairs], dtype=np.uint64), ra]); b = np.concatenate([np.array([p_[1] for p_ in pairs], dtype=np.uint64), rb])
    x = np.zeros_like(a); pb = np.zeros_like(a); fl = np.zeros_like(a)
    _lib.call("pil2gl_selftest_products", gl._ptr(a), gl._ptr(b), a.size, gl._ptr(x), gl._ptr(pb), gl._ptr(fl))
    flagged = 0
    for i in range(a.size):
        want = int(a[i]) * int(b[i]) % P
        assert int(x[i]) == want, (hex(int(a[i])), hex(int(b[i])))
        if fl[i]:
            flagged += 1
        else:
            assert int(pb[i]) == want, (hex(int(a[i])), hex(int(b[i])))
    assert fl[len(pairs) - 6] == 1                       # 2^63 squared: low words zero, the last subtraction borrows with certainty
    assert flagged >= 1 and fl[len(pairs):].sum() == 0   # structured operands raise it, 2^16 random pairs do not (probability 2^-16)


def test_ext_ops_device(gl, oracle):
    from pil2gl import _lib
    rows = H(golden("field.json")["ext"])
    a = np.array([r[0] for r in rows], dtype=np.uint64); b = np.array([r[1] for r in rows], dtype=np.uint64)
    m = np.zeros_like(a); iv = np.zeros_like(a)
    _lib.call("pil2gl_selftest_ext", gl._ptr(a), gl._ptr(b), a.shape[0], gl._ptr(m), gl._ptr(iv))
    assert m.tolist() == [r[2] for r in rows] and iv.tolist() == [r[3] for r in rows]


# ------------------------------------------------------------------ NTT / LDE  (test/fft_p.test.js shapes + random)
def _index_matrix(nBits, nPols):
    return np.ascontiguousarray(np.repeat(np.arange(1 << nBits, dtype=np.uint64)[:, None], nPols, axis=1))


@pytest.mark.parametrize("nBits,nPols", [(5, 2), (3, 1), (1, 3), (0, 2), (10, 5), (11, 3), (12, 8), (13, 17), (18, 5)])
def test_fft_ifft(gl, oracle, nBits, nPols):
    rng = np.random.default_rng(nBits * 100 + nPols)
    for a in (_index_matrix(nBits, nPols), rand_field(rng, ((1 << nBits), nPols))):
        out = np.zeros_like(a)
        gl.fft(a, nPols, nBits, out)
        assert (out == oracle.fft_cols(a, nBits)).all()
        gl.ifft(a, nPols, nBits, out)
        assert (out == oracle.ifft_cols(a, nBits)).all()


@pytest.mark.parametrize("nBits,nPols,extBits", [(3, 1, 1), (18, 5, 1), (5, 2, 3), (1, 1, 1), (0, 3, 2), (4, 100, 3),
                                                 (10, 8, 3), (11, 8, 3), (12, 3, 2), (14, 9, 3), (16, 8, 3), (7, 33, 0)])
def test_interpolate(gl, oracle, nBits, nPols, extBits):
    rng = np.random.default_rng(nBits * 1000 + nPols * 10 + extBits)
    for a in (_index_matrix(nBits, nPols), rand_field(rng, ((1 << nBits), nPols))):
        out = np.zeros(((1 << (nBits + extBits)), nPols), np.uint64)
        gl.interpolate(a, nPols, nBits, out, nBits + extBits)
        assert (out == oracle.interpolate(a, nBits, nBits + extBits)).all()


@pytest.mark.parametrize("nBits,nPols,extBits", [(16, 30, 3), (16, 32, 1), (17, 100, 1)])
def test_ntt_fixed_geometry_kernels(gl, oracle, nBits, nPols, extBits, monkeypatch):
    """wide matrices with 8-stage passes run kernel instances whose strides are compile-time constants (ntt_pass_kernel<.., 8>,
    lde_mid_kernel<16, 15|16>): the oracle's result, and bit-for-bit what the any-geometry instances give"""
    rng = np.random.default_rng(nBits * 100 + nPols)
    a = rand_field(rng, ((1 << nBits), nPols))
    want = oracle.interpolate(a, nBits, nBits + extBits)
    for generic in ("0", "1"):
        monkeypatch.setenv("PIL2GL_NTT_GENERIC", generic)
        out = np.zeros(((1 << (nBits + extBits)), nPols), np.uint64)
        gl.interpolate(a, nPols, nBits, out, nBits + extBits)
        assert np.array_equal(out, want), generic
        f = np.zeros_like(a); gl.fft(a, nPols, nBits, f)
        g = np.zeros_like(a); gl.ifft(f, nPols, nBits, g)
        assert np.array_equal(g, a), generic
        if generic == "0":
            f0 = f
        else:
            assert np.array_equal(f, f0)
    assert np.array_equal(f0[:, :3], oracle.fft_cols(np.ascontiguousarray(a[:, :3]), nBits))


@pytest.mark.parametrize("nb,C", [(27, 3), (29, 3), (30, 1)])
def test_ntt_at_the_largest_domain(gl, oracle, nb, C):
    """2^27 rows (config 3's extended domain), 2^29 x 3 (config 5's: the quotient's transforms run there; 12.9 GB a buffer) and
    2^30 x 1, the largest domain the transforms take -- the reference's take any nBits <= 32 (f3g.js:40, fft_p.js:178): fft
    then ifft returns the input bit for bit; the transform of the coefficient vector (0, c, 0, ...) is c w^i (closed form,
    rows sampled up to the last one); the transform is linear on sampled rows; one bit more than 2^30 is refused"""
    import torch
    import gc
    n = 1 << nb
    gc.collect(); torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < 6.5 * 8 * n * C + 4e9:
        pytest.skip("needs %.0f GB of free device memory" % ((6.5 * 8 * n * C + 4e9) / 1e9))
    g = torch.Generator(device="cuda"); g.manual_seed(nb)
    a = torch.empty(n * C, dtype=torch.int64, device="cuda")
    for o in range(0, n * C, 1 << 28):
        m = min(1 << 28, n * C - o)
        a[o:o + m] = torch.randint(0, 1 << 62, (m,), dtype=torch.int64, device="cuda", generator=g)
    cvals = [5, P - 2, 0x123456789ABCDEF][:C]
    av = a.view(n, C)
    f = torch.empty_like(a); back = torch.empty_like(a)
    gl.fft(a, C, nb, f)
    gl.ifft(f, C, nb, back)
    assert torch.equal(back, a)
    del back
    e = torch.zeros(n * C, dtype=torch.int64, device="cuda")
    ev = e.view(n, C)
    for c_, v in enumerate(cvals):
        ev[1, c_] = np.array([v], dtype=np.uint64).view(np.int64)[0]
    fe = torch.empty_like(e)
    gl.fft(e, C, nb, fe)
    del e, ev
    w = int(oracle.root(nb))
    rows = [0, 1, 2, 12345, n // 2, n // 2 + 1, (n // 3) | 1, n - 2, n - 1]
    got = fe.view(n, C)[rows].cpu().numpy().view(np.uint64)
    del fe
    for k, i in enumerate(rows):
        assert [int(x) for x in got[k]] == [v * pow(w, i, P) % P for v in cvals], i
    # linearity on the same rows: fft(a + e) = fft(a) + fft(e)   (a + e differs from a in row 1 only)
    row1 = [(int(x) + v) % P for x, v in zip(av[1].cpu().numpy().view(np.uint64), cvals)]
    av[1] = torch.from_numpy(np.array(row1, dtype=np.uint64).view(np.int64)).cuda()
    fa = f.view(n, C)[rows].cpu().numpy().view(np.uint64)
    gl.fft(a, C, nb, f)
    fsv = f.view(n, C)[rows].cpu().numpy().view(np.uint64)
    for k in range(len(rows)):
        assert [int(x) for x in fsv[k]] == [(int(x) + int(y)) % P for x, y in zip(fa[k], got[k])]
    with pytest.raises(Exception):
        gl.fft(a, 1, 31, f)
    with pytest.raises(Exception):
        gl.interpolate(a, 1, 28, f, 31)


def test_interpolate_at_the_largest_domain(gl, oracle):
    """config 5's shape on one device, three columns: 2^26 rows extended to 2^29.  The trace c w_N^i is the polynomial c x, whose
    extension is c 7 w_E^i in natural row order (stark_gen_helpers.js:139-144); and the sum of two traces extends to the sum"""
    import torch
    nb, nbe, C = 26, 29, 3
    N, E = 1 << nb, 1 << nbe
    from pil2gl import stark
    be = stark.GpuBackend(0)
    cvals = [3, P - 5, 0xFEDCBA987654321]
    x = be.build_x(nb, 1)                                          # w_N^i
    src = be.empty(N * C)
    ops = [(stark.OPC["mul"], (stark.SEC, 1, 1, 0, k), (stark.SEC, 1, 0, 0, 0), (stark.SCALAR, 1, 0, 0, k)) for k in range(C)]
    be.eval_program(ops, 0, [(x, 1), (src, C)], np.array(cvals, dtype=np.uint64), nb, 0)
    dst = be.empty(E * C)
    gl.interpolate(src, C, nb, dst, nbe)
    wE = int(oracle.root(nbe))
    rows = [0, 1, 7, 8, 9, 123456789, E // 2, E - 9, E - 1]
    got = dst.view(E, C)[rows].cpu().numpy().view(np.uint64)
    for k, i in enumerate(rows):
        assert [int(v) for v in got[k]] == [c * 7 * pow(wE, i, P) % P for c in cvals], i
    # a second trace: random; extension of the sum = sum of the extensions on sampled rows
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    r = torch.randint(0, 1 << 62, (N * C,), dtype=torch.int64, device="cuda", generator=g)
    d2 = be.empty(E * C)
    gl.interpolate(r, C, nb, d2, nbe)
    er = d2.view(E, C)[rows].cpu().numpy().view(np.uint64)
    s_ = be.empty(N * C)
    ops = [(stark.OPC["add"], (stark.SEC, 1, 2, 0, k), (stark.SEC, 1, 0, 0, k), (stark.SEC, 1, 1, 0, k)) for k in range(C)]
    be.eval_program(ops, 0, [(src, C), (r, C), (s_, C)], np.zeros(1, np.uint64), nb, 0)
    gl.interpolate(s_, C, nb, d2, nbe)
    es = d2.view(E, C)[rows].cpu().numpy().view(np.uint64)
    for k in range(len(rows)):
        assert [int(v) for v in es[k]] == [(int(a_) + int(b_)) % P for a_, b_ in zip(got[k], er[k])]


def test_interpolate_golden_kat(gl):
    # SURVEY 8(c)(2) / tests/golden/ntt.json "index3" ext 1
    c = [x for x in golden("ntt.json")["cases"] if x["name"] == "index3"][0]
    out = np.zeros(16, np.uint64)
    gl.interpolate(U(c["p"]), 1, 3, out, 4)
    assert out.tolist() == H(c["ext"]["1"])


def test_ntt_three_passes(gl, oracle):
    # 2^21 rows forces three passes in every direction
    rng = np.random.default_rng(21)
    a = rand_field(rng, ((1 << 21), 1))
    out = np.zeros_like(a)
    gl.fft(a, 1, 21, out)
    assert (out == oracle.fft_cols(a, 21)).all()
    ext = np.zeros(((1 << 22), 1), np.uint64)
    gl.interpolate(a, 1, 21, ext, 22)
    assert (ext == oracle.interpolate(a, 21, 22)).all()


def test_fft_in_place_and_errors(gl, oracle):
    rng = np.random.default_rng(3)
    a = rand_field(rng, (1 << 12, 4)); ref = oracle.fft_cols(a, 12)
    gl.fft(a, 4, 12, a)
    assert (a == ref).all()
    with pytest.raises(gl.Pil2glError):
        gl.interpolate(a, 4, 12, np.zeros(4 << 11, np.uint64), 11)       # nBitsExt < nBits
    with pytest.raises(gl.Pil2glError):
        gl.fft(a, 4, 13, a)                                              # buffer too small


def test_bad_arguments_are_refused_before_any_launch(gl):
    """every entry point validates pointers and sizes on the host and returns PIL2GL_EINVAL (the addon: a thrown Error) -- a bad call
    must never reach a kernel.  Null pointers with non-zero sizes, domains beyond the limits, ranges outside their bounds."""
    import torch
    from pil2gl import _lib
    d = torch.zeros(4096, dtype=torch.int64, device="cuda")
    ok = d.data_ptr()
    ch = np.zeros(3, np.uint64)
    bad = [("pil2gl_fft_dev", (None, 1, 4, ok, None)), ("pil2gl_fft_dev", (ok, 1, 31, ok, None)), ("pil2gl_fft_dev", (ok, 1, 4, None, None)),
           ("pil2gl_interpolate_cosets_dev", (ok, 1, 4, ok, 6, 3, 2, None)),                 # cosets [3, 5) of 4
           ("pil2gl_interpolate_cosets_dev", (ok, 1, 4, ok, 3, 0, 1, None)),                 # nBitsExt < nBits
           ("pil2gl_poseidon_dev", (ok, None, 4, 13, ok, None)), ("pil2gl_poseidon_dev", (None, None, 4, 4, ok, None)),
           ("pil2gl_linear_hash_rows_dev", (None, 9, 16, 0, ok, None)), ("pil2gl_linear_hash_rows_dev", (ok, 1 << 31, 1, 0, ok, None)),
           ("pil2gl_merkelize_dev", (ok, 4, 0, 0, ok, None)), ("pil2gl_merkelize_dev", (ok, 4, 8, 0, None, None)),
           ("pil2gl_fri_fold_dev", (ok, 5, 6, 1, gl._ptr(ch), ok, None)), ("pil2gl_fri_fold_dev", (ok, 31, 6, 1, gl._ptr(ch), ok, None)),
           ("pil2gl_fri_fold_dev", (ok, 5, 3, 1, None, ok, None)),
           ("pil2gl_build_x_dev", (32, 1, ok, None)), ("pil2gl_build_x_dev", (4, 1, None, None)),
           ("pil2gl_geometric_dev", (1, 2, 8, None, None)), ("pil2gl_geometric_dev", (1, 2, 1 << 39, ok, None)), ("pil2gl_geometric_dev", (1, 2, 1 << 40, ok, None)),
           ("pil2gl_first_nonzero_row_dev", (ok, 2, 0, 8, gl._ptr(ch), gl._ptr(ch), None)), ("pil2gl_first_nonzero_row_dev", (ok, 1, 9, 8, gl._ptr(ch), gl._ptr(ch), None)),
           ("pil2gl_first_nonzero_row_dev", (None, 1, 0, 8, gl._ptr(ch), gl._ptr(ch), None)), ("pil2gl_first_nonzero_row_dev", (ok, 1, 0, 8, None, gl._ptr(ch), None)),
           ("pil2gl_build_zhinv_dev", (5, 4, ok, None)), ("pil2gl_build_zhinv_dev", (4, 6, None, None)),
           ("pil2gl_compute_q_split_dev", (ok, 4, 5, 3, 3, ok, None)),                        # qDeg * N > E
           ("pil2gl_compute_q_split_dev", (None, 4, 5, 3, 2, ok, None)),
           ("pil2gl_x_div_x_sub_xi_cosets_dev", (6, 2, gl._ptr(ch), 2, 2, 0, 1, ok, None)),   # iOpen >= nOpen
           ("pil2gl_x_div_x_sub_xi_cosets_dev", (6, 2, gl._ptr(ch), 2, 0, 3, 2, ok, None)),   # cosets [3, 5) of 4
           ("pil2gl_rows_dot_ext_dev", (None, 8, 8, gl._ptr(ch), 1, ok, 0, None)),
           ("pil2gl_gprod_dev", (ok, 2, ok, 1, 8, ok, None)), ("pil2gl_gprod_dev", (None, 1, ok, 1, 8, ok, None)),
           ("pil2gl_dev_upload", (None, gl._ptr(ch), 3))]
    for name, args in bad:
        with pytest.raises(gl.Pil2glError):
            _lib.call(name, *args)
    torch.cuda.synchronize()
    assert int(d.sum()) == 0                      # and nothing was written


def test_first_nonzero_row(gl):
    """pil2gl_first_nonzero_row_dev (the device half of calculateExps' debug mode, prover_helpers.js:46-70): the smallest row of
    [first, last) whose value is not zero, and that value; columns of dimension 1 and 3, ranges of every boundary kind"""
    import torch
    rng = np.random.default_rng(2)
    for dim in (1, 3):
        for n in (1, 7, 256, 1000, 70001):
            col = np.zeros((n, dim), np.uint64)
            assert gl.firstNonZeroRow(torch.from_numpy(col.view(np.int64)).cuda(), dim, 0, n) is None
            hits = sorted(set(int(x) for x in rng.integers(0, n, 5)))
            for r in hits:
                col[r, int(rng.integers(0, dim))] = int(rng.integers(1, 1 << 63))
            d = torch.from_numpy(col.view(np.int64)).cuda()
            for first, last in ((0, n), (0, 1), (n - 1, n), (min(1, n - 1), max(n - 2, 1)), (hits[0] + 1, n), (hits[-1], hits[-1]), (0, hits[0])):
                inside = [r for r in hits if first <= r < last]
                got = gl.firstNonZeroRow(d, dim, first, last)
                if not inside:
                    assert got is None, (dim, n, first, last, got)
                else:
                    assert got == (inside[0], [int(v) for v in col[inside[0]]]), (dim, n, first, last, got)


# ------------------------------------------------------------------ Poseidon / linear hash / Merkle
def test_poseidon_kats(gl):
    g = H(golden("poseidon.json"))
    assert gl.poseidon([0] * 8) == [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca]
    assert gl.poseidon(list(range(8)), [8, 9, 10, 11]) == [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2]
    assert gl.poseidon([-1] * 8, [-1] * 4) == [0xbe0085cfc57a8357, 0xd95af71847d05c09, 0xcf55a13d33c1c953, 0x95803a74f4530e82]
    inp = np.array([r[0] for r in g], dtype=np.uint64); cap = np.array([r[1] for r in g], dtype=np.uint64)
    assert gl.poseidon_batch(inp, cap, 12).tolist() == [r[2] for r in g]
    with pytest.raises(gl.Pil2glError):
        gl.poseidon([1, 2, 3])


def test_poseidon_sbox_borrow_path(gl, oracle):
    """the hand-written S-box product flags a lane whose last subtraction borrows (probability ~2^-32 per product) and the
    kernel recomputes it the slow way.  2^63 squared is 2^126 = 2^30 * 2^96: low words zero, so the borrow is certain; states
    whose round-0 S-box inputs (state + round constant, glwasm.js:377) are 2^63 exercise that path in every lane, in some
    lanes only (mixed with random states: the fallback must not disturb the other lanes), and in single elements"""
    import os, re
    from conftest import ROOT
    with open(os.path.join(ROOT, "oracle", "poseidon_gl_constants.h")) as f:
        rc = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", f.read())[:12]]
    trig = [((1 << 63) - c) % P for c in rc]
    rng = np.random.default_rng(5)
    states = []
    for k in range(600):
        s = [int(x) % P for x in rng.integers(0, 1 << 63, 12, dtype=np.uint64)]
        if k % 3 == 0:
            s = list(trig)
        elif k % 3 == 1:
            j = k % 12
            s[j] = trig[j]
        states.append(s)
    a = np.array(states, dtype=np.uint64)
    got = gl.poseidon_batch(np.ascontiguousarray(a[:, :8]), np.ascontiguousarray(a[:, 8:]), 12)
    for k, s in enumerate(states):
        assert [int(v) for v in got[k]] == [int(v) for v in oracle.poseidon(s[:8], s[8:], 12)], k
    # the same rows through the leaf kernel (sponge) and the tree-level kernel
    rows = np.ascontiguousarray(a[:, :8])
    assert (gl.linearHash(rows, 8).reshape(-1, 4) == np.array([oracle.linear_hash(r) for r in rows])).all()
    assert (gl.merkelizeLevel(rows).reshape(-1, 4) == np.array([oracle.poseidon(r, None, 4) for r in rows])).all()


def _poseidon_tables():
    import os, re
    from conftest import ROOT
    txt = open(os.path.join(ROOT, "pil2-stark-js_amd", "csrc", "poseidon_gl_constants.inc")).read()
    out = {}
    for name in ("POSEIDON_GL_RC", "POSEIDON_GL_PARTIAL_C0", "POSEIDON_GL_RC26F"):
        m = re.search(name + r"\[\d+\] = \{(.*?)\};", txt, re.S)
        out[name] = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", m.group(1))]
    return out


def test_poseidon_statements_agree(gl, oracle):
    """the permutation in the library's three statements of it -- matrix-core MDS with rounds 4..25 four to a linear layer (what
    the hash kernels run), matrix-core MDS with a layer per round, vector ALU only -- on random, non-canonical and structured
    states; the first against the oracle as well"""
    from pil2gl import _lib
    rng = np.random.default_rng(2024)
    states = [[0] * 12, [(1 << 64) - 1] * 12, [P - 1] * 12, [P] * 12, [1] * 12, list(range(12)), [0x8080808080808080] * 12,
              [0x7F7F7F7F7F7F7F7F] * 12, [0xFF00FF00FF00FF00] * 12, [0x00FF00FF00FF00FF] * 12, [1 << 63] * 12]
    states += [[int(x) for x in rng.integers(0, 1 << 64, 12, dtype=np.uint64)] for _ in range(5000)]
    a = np.array(states, dtype=np.uint64)
    outs = []
    for what in (0, 1, 2):
        o = np.zeros_like(a)
        _lib.call("pil2gl_selftest_poseidon", gl._ptr(a), a.shape[0], what, gl._ptr(o))
        outs.append(o)
    assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all()
    for k in list(range(11)) + list(range(11, 5011, 97)):
        s = [v % P for v in states[k]]
        assert [int(v) for v in outs[0][k]] == [int(v) for v in oracle.poseidon(s[:8], s[8:], 12)], k


def test_partial_rounds_blocked_recombination(gl):
    """rounds 4..25 alone, four to a linear layer (poseidon_blocks.cuh) against one layer per round and against plain integers,
    on arbitrary states: the blocks' recombination works on biased planes of four signed coefficient digits and ends in an
    addition that wraps with probability ~2^-15 per element (patched in a rarely taken branch) -- 2^18 random states put a few
    hundred elements through that branch; the structured states reach the extremes of every plane"""
    from pil2gl import _lib
    t = _poseidon_tables()
    PC0 = t["POSEIDON_GL_PARTIAL_C0"]
    MC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    M = [[MC[(j - i) % 12] + (8 if i == 0 and j == 0 else 0) for j in range(12)] for i in range(12)]

    def partial(st):
        st = [v % P for v in st]
        for r in range(22):
            st[0] = pow((st[0] + PC0[r]) % P, 7, P)
            st = [sum(M[i][j] * st[j] for j in range(12)) % P for i in range(12)]
        return st
    rng = np.random.default_rng(31)
    states = [[0] * 12, [(1 << 64) - 1] * 12, [P - 1] * 12, [P] * 12, [1] * 12, [0x8080808080808080] * 12, [0x7F7F7F7F7F7F7F7F] * 12,
              [0xFF00FF00FF00FF00] * 12, [0x00FF00FF00FF00FF] * 12, [0x0101010101010101 * k for k in range(12)]]
    for j in range(12):
        for v in (1, 0xFF, 1 << 63, (1 << 64) - 1, 0x80, 0xFFFFFFFF, 0xFFFFFFFF00000000):
            s = [0] * 12; s[j] = v; states.append(s)
            s = [(1 << 64) - 1] * 12; s[j] = v; states.append(s)
    for _ in range(400):                     # sparse bytes: planes far from their mean
        s = []
        for j in range(12):
            b = rng.integers(0, 256, 8) * (rng.random(8) < 0.3)
            s.append(int(sum(int(x) << (8 * k) for k, x in enumerate(b))))
        states.append(s)
    nstruct = len(states)
    a = np.concatenate([np.array(states, dtype=np.uint64), rng.integers(0, 1 << 64, ((1 << 18), 12), dtype=np.uint64)])
    blk = np.zeros_like(a); one = np.zeros_like(a)
    _lib.call("pil2gl_selftest_poseidon", gl._ptr(a), a.shape[0], 3, gl._ptr(blk))
    _lib.call("pil2gl_selftest_poseidon", gl._ptr(a), a.shape[0], 4, gl._ptr(one))
    assert (blk == one).all()
    for k in list(range(nstruct)) + list(range(nstruct, a.shape[0], 4001)):
        assert [int(v) for v in blk[k]] == partial([int(v) for v in a[k]]), k


@pytest.mark.parametrize("mfma", [1, 0])
def test_mds_layer_structured_inputs(gl, mfma):
    """the MDS layer alone (matrix-core and vector-ALU forms) on inputs the hash never produces by chance: zero and
    tiny bytes (row 0's signed accumulators go negative), all-ones bytes (largest sums), values >= p, one-hot states"""
    from pil2gl import _lib
    MC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    M = [[MC[(j - i) % 12] + (8 if i == 0 and j == 0 else 0) for j in range(12)] for i in range(12)]
    rng = np.random.default_rng(99)
    states = [[0] * 12, [(1 << 64) - 1] * 12, [P - 1] * 12, [P] * 12, [1] * 12, list(range(12)), [0xFF00FF00FF00FF00] * 12,
              [0x0101010101010101 * k for k in range(12)], [0x8080808080808080] * 12, [0x7F7F7F7F7F7F7F7F] * 12]
    for j in range(12):
        for v in (1, 0xFF, 1 << 63, (1 << 64) - 1, 0x80, 0x100):
            s = [0] * 12; s[j] = v; states.append(s)
    # states that make the LAST addition of an element's recombination carry out of 64 bits (probability ~2^-20 on random
    # data; the matrix-core form patches those lanes in a rarely taken branch): one non-zero high word v in column j with
    # M[i][j] * v = (k + 1) 2^32 - 1 - r and k - 1 > r, alone and together with random other elements
    hit = 0
    for j in range(12):
        for i in range(12):
            mij = M[i][j]
            for k in range(mij - 1, 0, -1):
                v = ((k + 1) * 2 ** 32 - 1) // mij
                if v < 2 ** 32 and ((mij * v >> 32) * (2 ** 32 - 1) >> 32) + (mij * v & 0xFFFFFFFF) >= 2 ** 32:
                    s = [0] * 12; s[j] = v << 32; states.append(s)
                    s2 = [int(x) for x in rng.integers(0, 1 << 20, 12, dtype=np.uint64)]; s2[j] = v << 32; states.append(s2)
                    hit += 1
                    break
    assert hit > 50
    for _ in range(300):                     # random sparse bytes
        s = []
        for j in range(12):
            b = rng.integers(0, 256, 8) * (rng.random(8) < 0.4)
            s.append(int(sum(int(x) << (8 * k) for k, x in enumerate(b))))
        states.append(s)
    states += [[int(x) for x in rng.integers(0, 1 << 64, 12, dtype=np.uint64)] for _ in range(300)]
    a = np.array(states, dtype=np.uint64)
    for layers in (1, 2, 5):
        out = np.zeros_like(a)
        _lib.call("pil2gl_selftest_mds", gl._ptr(a), a.shape[0], layers, mfma, gl._ptr(out))
        for k, s in enumerate(states):
            cur = list(s)
            for _ in range(layers):
                cur = [sum(M[i][j] * cur[j] for j in range(12)) % P for i in range(12)]
            assert [int(x) for x in out[k]] == cur, (layers, k, s)


def test_linear_hash_golden(gl):
    g = golden("linearhash.json")
    for w, plain, split in H(g["index"]):
        if w == 0:
            continue
        v = np.arange(w, dtype=np.uint64)
        assert gl.linearHash(v, w, False).tolist() == plain, w
        assert gl.linearHash(v, w, True).tolist() == split, w
    for v, plain, split in H(g["random"]):
        v = np.array(v, dtype=np.uint64)
        assert gl.linearHash(v, v.size, False).tolist() == plain
        assert gl.linearHash(v, v.size, True).tolist() == split


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("N,nPols,idx", [(256, 3, 3), (256, 9, 3), (33, 6, 32), (1 << 18, 10, 33), (2, 9, 1), (1, 5, 0), (1000, 100, 999)])
def test_merkle_tree(gl, oracle, N, nPols, idx, split):
    # test/merklehash_p.test.js:19-250: v = i + 1000 j ; merkelize -> getGroupProof -> verifyGroupProof
    pols = np.ascontiguousarray(np.arange(N, dtype=np.uint64)[:, None] + np.uint64(1000) * np.arange(nPols, dtype=np.uint64)[None, :])
    MH = gl.buildMerkleHash(split)
    tree = MH.merkelize(pols, nPols, N)
    assert (tree["nodes"] == oracle.merkelize(pols, split)).all()
    groupElements, mp = MH.getGroupProof(tree, idx)
    assert groupElements == pols[idx].tolist()
    assert np.array(mp, dtype=np.uint64).reshape(-1, 4).tolist() == oracle.group_proof(tree["nodes"], N, idx).tolist()
    root = MH.root(tree)
    if N == 1:      # reference quirk (merklehash_p.js:28-42,87): a 1-row tree has no level loop, root = zero words
        assert root == [0, 0, 0, 0] and mp == []
        return
    assert MH.verifyGroupProof(root, mp, idx, groupElements)
    bad = list(groupElements); bad[0] ^= 1
    assert not MH.verifyGroupProof(root, mp, idx, bad)
    with pytest.raises(gl.Pil2glError):
        MH.getGroupProof(tree, N)


def test_transcript_list_absorption(gl, oracle):
    """Transcript.put of a list goes through one chained device call (pil2gl_sponge_absorb: twelve lanes per permutation):
    same challenges as the oracle's element-by-element transcript, for every alignment of the list against the blocks"""
    from pil2gl import _lib
    rng = np.random.default_rng(5)
    for pre in (0, 1, 3, 7, 8, 9):
        for n in (0, 1, 7, 8, 15, 16, 17, 40, 96 * 8 + 5):
            vals = [int(v) for v in rand_field(rng, n)]
            head = [int(v) for v in rand_field(rng, pre)]
            t, o = gl.Transcript(), oracle.Transcript()
            for v in head:
                t.put(v)
            if head:
                o.put(np.array(head, dtype=np.uint64))
            t.put(vals if n % 2 else [vals[:n // 2], [vals[n // 2:]]])          # nested lists flatten in order
            if vals:
                o.put(np.array(vals, dtype=np.uint64))
            assert [int(x) for x in t.getField()] == [int(x) for x in o.get_field()], (pre, n)
            assert t.getPermutations(8, 11) == [int(x) for x in o.get_permutations(8, 11)]
    blocks = rand_field(rng, 5 * 8); cap = rand_field(rng, 4); out = np.zeros(12, np.uint64)
    _lib.call("pil2gl_sponge_absorb", gl._ptr(blocks), 5, gl._ptr(cap), gl._ptr(out))
    st = [int(v) for v in cap]
    for b in range(5):
        st12 = gl.poseidon([int(v) for v in blocks[8 * b:8 * b + 8]], st[:4], 12)
        st = st12
    assert [int(v) for v in out] == [int(v) for v in st]


def test_ntt_random_shapes_and_pass_splits(gl, oracle, monkeypatch):
    """interpolate / fft / ifft on random shapes with the pass planner forced to every split it can take (PIL2GL_NTT_KMAX:
    passes of up to 4, 6, 9, 10 stages besides the default 8 / 7), i.e. every register-step chunking of gl_fermat.cuh"""
    rng = np.random.default_rng(2024)
    n_cases = 0
    while n_cases < 40:
        nb = int(rng.integers(1, 14)); C = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 33, 64, 100])); eb = int(rng.integers(0, 4))
        if (C << (nb + eb)) > (1 << 22):
            continue
        kmax = rng.choice(["", "4", "6", "9", "10"])
        if kmax:
            monkeypatch.setenv("PIL2GL_NTT_KMAX", str(kmax))
        else:
            monkeypatch.delenv("PIL2GL_NTT_KMAX", raising=False)
        a = rand_field(rng, ((1 << nb), C))
        dst = np.zeros(((1 << (nb + eb)), C), np.uint64)
        gl.interpolate(a, C, nb, dst, nb + eb)
        assert np.array_equal(dst, oracle.interpolate(a, nb, nb + eb)), (nb, C, eb, kmax)
        f = np.zeros_like(a); gl.fft(a, C, nb, f)
        assert np.array_equal(f, oracle.fft_cols(a, nb)), (nb, C, kmax)
        g = np.zeros_like(a); gl.ifft(a, C, nb, g)
        assert np.array_equal(g, oracle.ifft_cols(a, nb)), (nb, C, kmax)
        n_cases += 1


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("N,nPols", [(256, 3), (256, 9), (33, 6), (1000, 100), (2, 9)])
def test_batch_root_from_group_proofs(gl, oracle, N, nPols, split):
    """verifier side (merklehash_p.js:169-215 for all queries of a tree in one call): the batch equals the one-by-one walk,
    reaches the root for every leaf asked, and a wrong value or a wrong sibling is caught"""
    rng = np.random.default_rng(N * 7 + nPols)
    pols = rand_field(rng, (N, nPols))
    MH = gl.buildMerkleHash(split)
    tree = MH.merkelize(pols, nPols, N)
    root = MH.root(tree)
    idxs = sorted({0, N - 1, N // 2, 1 % N, 5 % N, (N - 2) % N})
    proofs = MH.getGroupProofs(tree, idxs)
    roots = MH.calculateRootsFromGroupProofs(proofs, idxs)
    assert roots == [[int(v) for v in MH.calculateRootFromGroupProof(mp, i, vals)] for (vals, mp), i in zip(proofs, idxs)]
    assert all(r == [int(v) for v in root] for r in roots) and MH.verifyGroupProofs(root, proofs, idxs)
    bad = [(list(v), [list(s_) for s_ in mp]) for v, mp in proofs]
    bad[1][0][0] = (bad[1][0][0] + 1) % P
    assert not MH.verifyGroupProofs(root, bad, idxs)
    bad = [(list(v), [list(s_) for s_ in mp]) for v, mp in proofs]
    bad[-1][1][-1][3] ^= 1
    assert not MH.verifyGroupProofs(root, bad, idxs)
    assert MH.calculateRootsFromGroupProofs([], []) == []


def test_merkle_roots_golden(gl):
    for N, w, split, root, leaf0, leaf_last in H(golden("merkle.json")):
        elems = np.ascontiguousarray(np.arange(N, dtype=np.uint64)[:, None] + np.uint64(1000) * np.arange(w, dtype=np.uint64)[None, :])
        MH = gl.buildMerkleHash(bool(split))
        tree = MH.merkelize(elems, w, N)
        assert MH.root(tree) == root
        assert tree["nodes"][:4].tolist() == leaf0


def test_merkle_file_roundtrip(gl, tmp_path):
    pols = np.ascontiguousarray(np.arange(4096, dtype=np.uint64)[:, None] + np.uint64(1000) * np.arange(10, dtype=np.uint64)[None, :])
    MH = gl.buildMerkleHash(False)
    tree = MH.merkelize(pols, 10, 4096)
    fn = str(tmp_path / "t.consttree")
    MH.writeToFile(tree, fn)
    t2 = MH.readFromFile(fn)
    assert t2["width"] == 10 and t2["height"] == 4096
    assert (t2["elements"] == pols.reshape(-1)).all() and (t2["nodes"] == tree["nodes"]).all()


def test_merkelize_level_and_rows(gl, oracle):
    rng = np.random.default_rng(5)
    a = rand_field(rng, 8 * 1000)
    out = gl.merkelizeLevel(a)
    for i in (0, 1, 499, 999):
        assert out[4 * i:4 * i + 4].tolist() == oracle.poseidon(a[8 * i:8 * i + 8], None, 4).tolist()
    rows = rand_field(rng, (513, 21))
    d = gl.linearHash(rows, 21, False)
    for i in (0, 7, 512):
        assert d[4 * i:4 * i + 4].tolist() == oracle.linear_hash(rows[i], False).tolist()


@pytest.mark.parametrize("width", [5, 8, 9, 12, 16, 17, 27, 32, 33, 36, 47, 100, 131])
def test_split_linear_hash_rows_against_oracle(gl, oracle, width):
    """splitLinearHash (linearhash_gpu.js:30-66) on whole row blocks: two, three and four batches, a last batch of <= 4 columns
    (its own digest, unhashed), one batch only (the plain sponge), heights that leave the 512-row workgroups ragged"""
    rng = np.random.default_rng(width)
    for h in (1, 511, 513, 1500):
        rows = rand_field(rng, (h, width))
        d = gl.linearHash(rows, width, True).reshape(h, 4)
        for i in sorted({0, min(1, h - 1), h // 2, max(0, h - 2), h - 1}):
            assert d[i].tolist() == oracle.linear_hash(rows[i], True).tolist(), (width, h, i)
    rows = rand_field(rng, (4096, width))
    d = gl.linearHash(rows, width, True).reshape(-1, 4)
    want = np.array([oracle.linear_hash(r, True) for r in rows[1000:1064]], dtype=np.uint64)
    assert (d[1000:1064] == want).all()


def test_split_leaf_kernel_at_config3_size_is_the_composition_of_plain_hashes(gl):
    """BASELINE config 3's extended matrix (2^27 rows x 100 columns, 107 GB: skipped when the memory is not there): the split leaf
    kernel (one launch, digests parked in LDS) against the rule it implements spelled out with the PLAIN kernel -- linearhash_gpu.js:30-66:
    the plain hash of each batch of max(8, ceil(w / 4)) columns, then the plain hash of the four digests side by side -- on every row"""
    import torch
    import gc
    gc.collect(); torch.cuda.empty_cache()
    h, w = 1 << 27, 100
    if torch.cuda.mem_get_info()[0] < 8 * h * (w + 25 + 16 + 8) + 8e9:
        pytest.skip("needs %.0f GB of free device memory" % ((8 * h * (w + 49) + 8e9) / 1e9))
    g = torch.Generator(device="cuda"); g.manual_seed(100)
    buf = torch.empty(h * w, dtype=torch.int64, device="cuda")
    for o in range(0, h * w, 1 << 28):
        m = min(1 << 28, h * w - o)
        buf[o:o + m] = torch.randint(0, 0x7FFFFFFFFFFFFFFF, (m,), dtype=torch.int64, device="cuda", generator=g) % 0xFFFFFFFF00000001
    split = torch.empty(h * 4, dtype=torch.int64, device="cuda")
    gl.linearHash(buf, w, True, split)
    batch = max(8, (w + 3) // 4)
    second = torch.empty((h, 16), dtype=torch.int64, device="cuda")
    piece = torch.empty(h * batch, dtype=torch.int64, device="cuda")
    d = torch.empty(h * 4, dtype=torch.int64, device="cuda")
    bv = buf.view(h, w)
    for b in range(4):
        pv = piece.view(h, batch)
        pv.copy_(bv[:, b * batch:(b + 1) * batch])
        gl.linearHash(piece, batch, False, d)
        second[:, 4 * b:4 * b + 4] = d.view(h, 4)
    gl.linearHash(second.view(-1), 16, False, d)
    assert torch.equal(d, split)
    # and the plain kernel itself on three sampled rows against the oracle-independent golden rule is covered elsewhere; here: not all equal
    assert not torch.equal(split.view(h, 4)[0], split.view(h, 4)[1])


# ------------------------------------------------------------------ transcript + reference proof, through the product
def test_transcript_golden(gl):
    for c in golden("transcript.json"):
        t = gl.Transcript()
        puts = H(c["put"]); fields = H(c.get("fields", []))
        t.put(puts[0])
        if fields:
            assert t.getField() == fields[0]
        if len(puts) > 1:
            t.put(puts[1]); assert t.getField() == fields[1]; assert t.getField() == fields[2]
        if "perms" in c:
            n, nb, res = c["perms"]; assert t.getPermutations(n, nb) == res
        if "state" in c:
            assert t.getState() == H(c["state"])


def test_reference_proof_through_gpu(gl):
    import test_reference_proof as trp
    z = trp._ints(golden("ref_compressor_verifier.proof.zkin.json"))

    class _T:                                   # adapt the product Transcript to the replay helper's interface
        def __init__(self): self.t = gl.Transcript()
        def put(self, a): self.t.put(np.asarray(a, dtype=np.uint64).reshape(-1).tolist())
        def get_field(self): return np.array(self.t.getField(), dtype=np.uint64)
        def get_permutations(self, n, b): return np.array(self.t.getPermutations(n, b), dtype=np.uint64)

    class _O:
        Transcript = _T
    ch, fri_steps, queries = trp._replay(_O, z)
    assert queries == [891, 1628, 1228, 1991, 1856, 415, 833, 296]
    MH = gl.buildMerkleHash(False)
    for q, idx in enumerate(queries):
        for name, root in (("1", z["root1"]), ("2", z["root2"]), ("3", z["root3"]), ("4", z["root4"]), ("C", trp.ROOT_C)):
            assert MH.verifyGroupProof(root, z["s0_siblings" + name][q], idx, z["s0_vals" + name][q])
    # the same openings through the batch entry point: all eight queries of a tree in one call, for the five stage trees
    # and the two FRI trees the reference prover committed
    for name, root in (("1", z["root1"]), ("2", z["root2"]), ("3", z["root3"]), ("4", z["root4"]), ("C", trp.ROOT_C)):
        proofs = [(z["s0_vals" + name][q], z["s0_siblings" + name][q]) for q in range(len(queries))]
        assert MH.calculateRootsFromGroupProofs(proofs, queries) == [[int(v) for v in root]] * len(queries), name
    for s in (1, 2):
        idxs = [i % (1 << trp.STEPS[s]) for i in queries]
        proofs = [(z["s%d_vals" % s][q], z["s%d_siblings" % s][q]) for q in range(len(queries))]
        assert MH.verifyGroupProofs(z["s%d_root" % s], proofs, idxs), s
    # FRI fold of the opened groups (fri.js:107-150) on the GPU
    pol_bits = trp.STEPS[0]
    for s in (1, 2):
        out_bits = trp.STEPS[s]
        sinv0 = pow(pow(7, P - 2, P), 1 << (trp.STEPS[0] - trp.STEPS[s - 1]), P)
        wi = pow(int(gl_root(pol_bits)), P - 2, P)
        for q, idx0 in enumerate(queries):
            idx = idx0 % (1 << out_bits)
            group = np.array(z["s%d_vals" % s][q], dtype=np.uint64).reshape(-1, 3)
            out = np.zeros((1, 3), np.uint64)
            from pil2gl import _lib
            sinv_g = sinv0 * pow(wi, idx, P) % P
            ch_ = np.array([int(x) for x in fri_steps[s]], dtype=np.uint64)
            _lib.call("pil2gl_fri_fold", gl._ptr(group), group.shape[0].bit_length() - 1, 0, sinv_g, gl._ptr(ch_), gl._ptr(out))
            if s + 1 < len(trp.STEPS):
                nxt = np.array(z["s%d_vals" % (s + 1)][q], dtype=np.uint64).reshape(-1, 3)
                assert nxt[idx // (1 << trp.STEPS[s + 1])].tolist() == out[0].tolist()
            else:
                assert z["finalPol"][idx] == out[0].tolist()
        pol_bits = out_bits
    # ... and through FRI.verify itself (fri.js:107-174): the layer trees, the folds and the last polynomial are the reference
    # prover's.  The FRI polynomial's value at a query needs the circuit's verifier program, which is not part of the
    # fixture, so step 0 hands over the value layer 1 opened; every later layer is checked for real.
    nq = len(queries)
    ss = {"nBits": 10, "nBitsExt": trp.STEPS[0], "nQueries": nq, "steps": [{"nBits": b} for b in trp.STEPS]}

    def ref_proof():
        layers = [{"root": z["s%d_root" % s_], "polQueries": [[list(z["s%d_vals" % s_][q]), z["s%d_siblings" % s_][q]] for q in range(nq)]}
                  for s_ in range(1, len(trp.STEPS))]
        return [{"polQueries": list(range(nq))}] + layers + [[list(e) for e in z["finalPol"]]]

    def step0(q, idx):
        return [[int(v) for v in np.array(z["s1_vals"][q], dtype=np.uint64).reshape(-1, 3)[idx // (1 << trp.STEPS[1])]]]
    fri = gl.FRI(ss, MH)
    assert fri.verify(fri_steps, list(queries), ref_proof(), step0)
    bad = ref_proof(); bad[2]["polQueries"][3][0][5] = (int(bad[2]["polQueries"][3][0][5]) + 1) % P
    assert not fri.verify(fri_steps, list(queries), bad, step0)
    bad = ref_proof(); bad[-1][2][0] = (int(bad[-1][2][0]) + 1) % P
    assert not fri.verify(fri_steps, list(queries), bad, step0)
    bad = ref_proof(); bad[1]["root"] = [int(v) ^ 1 for v in bad[1]["root"]]
    assert not fri.verify(fri_steps, list(queries), bad, step0)
    assert not fri.verify(fri_steps, list(queries), ref_proof(), lambda q, idx: [[1, 2, 3]])


def gl_root(bits):
    w = 7277203076849721926
    for _ in range(32 - bits):
        w = w * w % P
    return w


# ------------------------------------------------------------------ FRI
def test_fri_fold_golden(gl, oracle):
    from pil2gl import _lib
    for pol_bits, out_bits, bits0, bits_prev, ch, pol, res in H(golden("fri_fold.json")):
        pol = np.array(pol, dtype=np.uint64); out = np.zeros((1 << out_bits, 3), np.uint64)
        sinv = oracle.fri_shift_inv(bits0, bits_prev)
        chn = np.array(ch, dtype=np.uint64)
        _lib.call("pil2gl_fri_fold", gl._ptr(pol), pol_bits, out_bits, sinv, gl._ptr(chn), gl._ptr(out))
        assert out.tolist() == res


def test_fri_class_fold_and_queries(gl, oracle):
    # fri.js fold/proofQueries over steps 11/7/3 with trees, against the oracle
    struct = {"nBits": 10, "nBitsExt": 11, "nQueries": 8, "steps": [{"nBits": 11}, {"nBits": 7}, {"nBits": 3}]}
    MH = gl.buildMerkleHash(False)
    fri = gl.FRI(struct, MH)
    rng = np.random.default_rng(11)
    pol = rand_field(rng, (1 << 11, 3))
    cur = pol; prev_bits = None
    for step in range(3):
        ch = rand_field(rng, 3)
        r = fri.fold(step, cur, ch.tolist())
        out_bits = struct["steps"][step]["nBits"]
        if step == 0:
            exp = cur
        else:
            exp = oracle.fri_fold(cur, out_bits, oracle.fri_shift_inv(11, prev_bits), ch)
        assert (np.asarray(r["pol"]) == exp).all()
        if step < 2:
            nb = struct["steps"][step + 1]["nBits"]
            tb = oracle.fri_transpose(exp, nb)
            nodes = oracle.merkelize(tb.reshape(1 << nb, -1), False)
            assert (r["tree"]["nodes"] == nodes).all() and r["proof"]["root"] == nodes[-4:].tolist()
            gp = MH.getGroupProof(r["tree"], 5)
            assert MH.verifyGroupProof(r["proof"]["root"], gp[1], 5, gp[0])
        else:
            assert r["proof"] == exp.tolist()
        cur = exp; prev_bits = out_bits


# ------------------------------------------------------------------ STARK step helpers
def _dev(gl, n):
    import torch
    return torch.zeros(n, dtype=torch.int64, device="cuda")


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def test_step_helpers(gl, oracle):
    from pil2gl import _lib
    import torch
    st = None
    for nb, nbe, zh, first, last, frame in H(golden("zerofiers.json")):
        o = _dev(gl, 1 << nbe)
        _lib.call("pil2gl_build_zhinv_dev", nb, nbe, gl._ptr(o), st); assert _host(o).tolist() == zh
        _lib.call("pil2gl_build_one_row_zerofier_inv_dev", nb, nbe, 0, gl._ptr(o), st); assert _host(o).tolist() == first
        _lib.call("pil2gl_build_one_row_zerofier_inv_dev", nb, nbe, (1 << nb) - 1, gl._ptr(o), st); assert _host(o).tolist() == last
        _lib.call("pil2gl_build_frame_zerofier_dev", nb, nbe, 2, 1, gl._ptr(o), st); assert _host(o).tolist() == frame
    nb, nbe = 9, 12
    o = _dev(gl, 1 << nbe)
    _lib.call("pil2gl_build_x_dev", nbe, 7, gl._ptr(o), st); assert (_host(o) == oracle.build_x(nbe, 7)).all()
    _lib.call("pil2gl_build_one_row_zerofier_inv_dev", nb, nbe, 3, gl._ptr(o), st)
    assert (_host(o) == oracle.build_one_row_zerofier_inv(nb, nbe, 3)).all()
    rng = np.random.default_rng(9)
    # q split (stark_gen_helpers.js:179-190)
    qdim, qdeg = 3, 2
    qq1 = rand_field(rng, (1 << nbe, qdim))
    d1 = torch.from_numpy(qq1.view(np.int64)).cuda(); d2 = _dev(gl, (1 << nbe) * qdim * qdeg)
    _lib.call("pil2gl_compute_q_split_dev", gl._ptr(d1), nb, nbe, qdim, qdeg, gl._ptr(d2), st)
    assert (_host(d2).reshape(1 << nbe, -1) == oracle.compute_q_split(qq1, nb, nbe, qdim, qdeg)).all()
    # xDivXSubXi (stark_gen_helpers.js:293-322), two openings
    xis = rand_field(rng, (2, 3))
    d = _dev(gl, (1 << nbe) * 6)
    for i in range(2):
        xi = np.ascontiguousarray(xis[i])
        _lib.call("pil2gl_x_div_x_sub_xi_dev", nbe, gl._ptr(xi), 2, i, gl._ptr(d), st)
    full = oracle.x_div_x_sub_xi(nbe, xis)
    assert (_host(d).reshape(1 << nbe, 6) == full).all()
    # the same table for a coset slice only (a rank of a sharded proof): row pos * cc + jl = extended row (pos << eb) + cb + jl
    eb_ = nbe - nb
    last = (1 << eb_) - 1
    for cb, cc in {(0, 1), (last, 1), (0, 1 << eb_), (2 if last >= 3 else 0, 2 if last >= 3 else 1)}:
        ds = _dev(gl, ((1 << nb) * cc) * 6)
        for i in range(2):
            xi = np.ascontiguousarray(xis[i])
            _lib.call("pil2gl_x_div_x_sub_xi_cosets_dev", nbe, eb_, gl._ptr(xi), 2, i, cb, cc, gl._ptr(ds), st)
        want = full.reshape(1 << nb, 1 << eb_, 6)[:, cb:cb + cc].reshape(-1, 6)
        assert (_host(ds).reshape(-1, 6) == want).all(), (cb, cc)
    with pytest.raises(Exception):
        _lib.call("pil2gl_x_div_x_sub_xi_cosets_dev", nbe, eb_, gl._ptr(xi), 2, 0, 0, 3, gl._ptr(d), st)
    # LEv + evals (stark_gen_helpers.js:216-264)
    xi = rand_field(rng, 3)
    lev = _dev(gl, (1 << nb) * 3)
    _lib.call("pil2gl_build_lev_dev", nb, gl._ptr(xi), gl._ptr(lev), st)
    lev_o = oracle.lev(nb, xi)
    assert (_host(lev).reshape(-1, 3) == lev_o).all()
    buf = rand_field(rng, (1 << nbe, 7)); dbuf = torch.from_numpy(buf.view(np.int64)).cuda()
    descs = (_lib.EvalDesc * 3)()
    for e, (off, dim) in enumerate([(0, 1), (2, 3), (6, 1)]):
        descs[e].buf = dbuf.data_ptr(); descs[e].width = 7; descs[e].offset = off; descs[e].dim = dim; descs[e].levIndex = 0
    import ctypes as C
    levs = (C.c_void_p * 1)(lev.data_ptr()); res = np.zeros((3, 3), np.uint64)
    _lib.call("pil2gl_compute_evals_dev", descs, 3, nb, nbe - nb, levs, 1, gl._ptr(res), st)
    for e, (off, dim) in enumerate([(0, 1), (2, 3), (6, 1)]):
        assert res[e].tolist() == oracle.eval_pol_at(buf, off, dim, nb, nbe - nb, lev_o).tolist()


# ------------------------------------------------------------------ expression evaluator
def _random_program(rng, n_ops, sections_w, n_scalars, dest_sec):
    """random straight-line program over (tmp, section, scalar) operands with mixed dims"""
    from gl_oracle import TMP, SEC, SCALAR
    ops = []; tmps = []      # (id, dim)
    def src():
        k = rng.integers(0, 3 if tmps else 2)
        if k == 2:
            i, d = tmps[rng.integers(0, len(tmps))]; return (TMP, d, 0, 0, i)
        if k == 1:
            d = 1 if rng.integers(0, 2) else 3
            return (SCALAR, d, 0, 0, int(rng.integers(0, n_scalars - 3)))
        s = int(rng.integers(0, len(sections_w) - 1)); d = 3 if (sections_w[s] >= 3 and rng.integers(0, 3) == 0) else 1
        return (SEC, d, s, int(rng.integers(-2, 3)), int(rng.integers(0, sections_w[s] - d + 1)))
    for k in range(n_ops):
        op = ["add", "sub", "mul", "copy"][rng.integers(0, 4)]
        a = src(); b = src() if op != "copy" else None
        dim = max(a[1], b[1] if b else 1)
        tid = len(tmps); tmps.append((tid, dim))
        ops.append((op, (TMP, dim, 0, 0, tid), a, b))
    # final: write an ext result into the destination section
    last3 = [t for t in tmps if t[1] == 3]
    t = last3[-1] if last3 else tmps[-1]
    ops.append(("copy", (SEC, t[1], dest_sec, 0, 0), (TMP, t[1], 0, 0, t[0]), None))
    return ops, len(tmps)


@pytest.mark.parametrize("n_ops,prime_shift", [(20, 0), (300, 3)])
def test_expression_evaluator(gl, oracle, n_ops, prime_shift):
    import torch
    import ctypes as C
    from pil2gl import _lib
    rng = np.random.default_rng(n_ops)
    n_bits = 9
    widths = [5, 9, 1, 3]           # last = destination (f_ext-like, width 3)
    secs = [rand_field(rng, (1 << n_bits, w)) for w in widths]
    secs[-1][:] = 0
    scalars = rand_field(rng, 40)
    ops, n_tmp = _random_program(rng, n_ops, widths, scalars.size, len(widths) - 1)
    ref_secs = [s.copy() for s in secs]
    oracle.eval_program(ops, n_tmp, ref_secs, scalars, n_bits, prime_shift)
    dsecs = [torch.from_numpy(s.view(np.int64)).cuda() for s in secs]
    prog = oracle.make_program(ops, n_tmp, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
    csecs = (_lib.GlxSection * len(dsecs))()
    for i, s in enumerate(dsecs):
        csecs[i].ptr = s.data_ptr(); csecs[i].width = widths[i]
    ctx = _lib.GlxCtx(n_bits, prime_shift, len(dsecs), scalars.size, csecs, scalars.ctypes.data_as(_lib.u64p))
    _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None)
    torch.cuda.synchronize()
    for d, r in zip(dsecs, ref_secs):
        assert (d.cpu().numpy().view(np.uint64).reshape(r.shape) == r).all()


@pytest.mark.parametrize("jit", ["0", "1"])
def test_expression_evaluator_minimal_programs_of_the_first_kernels_miscompile(gl, oracle, jit, monkeypatch):
    """The first evaluator kernel (commit 8435cfa) returned components 0 and 1 of ext values wrong whenever a program
    contained a COPY: hipcc -O1..-O3 dropped r[0] = a[0], r[1] = a[1] of the `default:` arm of its switch over the op code
    (reproduced and bisected with the old source in round 2).  These are the smallest programs that
    showed it -- a lone copy of an ext scalar, and sub -> tmp -> copy -- plus every op with every operand-dimension pair,
    through the interpreter and the run-time compiled kernel."""
    import torch
    import ctypes as C
    from pil2gl import _lib
    from gl_oracle import TMP, SEC, SCALAR
    monkeypatch.setenv("PIL2GL_EXPR_JIT", jit)
    rng = np.random.default_rng(4)
    n_bits, widths = 4, [5, 9, 1, 3]
    dest = (SEC, 3, 3, 0, 0)
    programs = [
        [("copy", dest, (SCALAR, 3, 0, 0, 18), None)],
        [("sub", (TMP, 3, 0, 0, 0), (SCALAR, 1, 0, 0, 6), (SCALAR, 3, 0, 0, 18)), ("copy", dest, (TMP, 3, 0, 0, 0), None)],
        [("sub", (TMP, 3, 0, 0, 0), (SCALAR, 1, 0, 0, 6), (SCALAR, 3, 0, 0, 18)), ("add", (TMP, 3, 0, 0, 1), (TMP, 3, 0, 0, 0), (TMP, 3, 0, 0, 0)),
         ("add", (TMP, 1, 0, 0, 2), (SEC, 1, 1, -1, 4), (SEC, 1, 2, -2, 0)), ("copy", dest, (TMP, 3, 0, 0, 1), None)],
        [("copy", (TMP, 3, 0, 0, 0), (SEC, 3, 1, 1, 2), None), ("copy", (TMP, 3, 0, 0, 1), (TMP, 3, 0, 0, 0), None), ("copy", dest, (TMP, 3, 0, 0, 1), None)],
    ]
    for op in ("add", "sub", "mul"):
        for da in (1, 3):
            for db in (1, 3):
                programs.append([(op, (TMP, max(da, db), 0, 0, 0), (SCALAR, da, 0, 0, 3), (SEC, db, 1, 0, 1)), ("copy", (SEC, max(da, db), 3, 0, 0), (TMP, max(da, db), 0, 0, 0), None)])
    for ops in programs:
        secs = [rand_field(rng, (1 << n_bits, w)) for w in widths]; secs[-1][:] = 0
        scalars = rand_field(rng, 40)
        n_tmp = 1 + max([o[1][4] for o in ops if o[1][0] == TMP] + [0])
        ref = [s_.copy() for s_ in secs]
        oracle.eval_program(ops, n_tmp, ref, scalars, n_bits, 0)
        dsecs = [torch.from_numpy(s_.view(np.int64)).cuda() for s_ in secs]
        prog = oracle.make_program(ops, n_tmp, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
        csecs = (_lib.GlxSection * 4)()
        for i, t in enumerate(dsecs):
            csecs[i].ptr = t.data_ptr(); csecs[i].width = widths[i]
        ctx = _lib.GlxCtx(n_bits, 0, 4, scalars.size, csecs, scalars.ctypes.data_as(_lib.u64p))
        _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None)
        torch.cuda.synchronize()
        assert (dsecs[3].cpu().numpy().view(np.uint64).reshape(ref[3].shape) == ref[3]).all(), ops


def test_expression_evaluator_many_live_tmps(gl, oracle):
    """more live temporaries than fit LDS: exercises the global-memory spill path of the evaluator"""
    import torch
    import ctypes as C
    from pil2gl import _lib
    from gl_oracle import TMP, SEC, SCALAR
    rng = np.random.default_rng(99)
    n_bits, n_live = 11, 120
    widths = [7, 3]
    secs = [rand_field(rng, (1 << n_bits, w)) for w in widths]; secs[1][:] = 0
    scalars = rand_field(rng, 16)
    ops = []
    for t in range(n_live):                    # t_i = sec0[col] * scalar   (all stay live)
        ops.append(("mul", (TMP, 3, 0, 0, t), (SEC, 1, 0, int(rng.integers(-3, 4)), int(rng.integers(0, 7))), (SCALAR, 3, 0, 0, int(rng.integers(0, 13)))))
    acc = n_live
    ops.append(("add", (TMP, 3, 0, 0, acc), (TMP, 3, 0, 0, 0), (TMP, 3, 0, 0, 1)))
    for t in range(2, n_live):                 # fold them all
        ops.append((["add", "sub", "mul"][t % 3], (TMP, 3, 0, 0, acc), (TMP, 3, 0, 0, acc), (TMP, 3, 0, 0, t)))
    ops.append(("copy", (SEC, 3, 1, 0, 0), (TMP, 3, 0, 0, acc), None))
    ref = [s.copy() for s in secs]
    oracle.eval_program(ops, acc + 1, ref, scalars, n_bits, 0)
    dsecs = [torch.from_numpy(s.view(np.int64)).cuda() for s in secs]
    prog = oracle.make_program(ops, acc + 1, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
    csecs = (_lib.GlxSection * 2)()
    for i, t in enumerate(dsecs):
        csecs[i].ptr = t.data_ptr(); csecs[i].width = widths[i]
    ctx = _lib.GlxCtx(n_bits, 0, 2, scalars.size, csecs, scalars.ctypes.data_as(_lib.u64p))
    _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None)
    torch.cuda.synchronize()
    assert (dsecs[1].cpu().numpy().view(np.uint64).reshape(ref[1].shape) == ref[1]).all()


# ------------------------------------------------------------------ device-resident path (buffers stay in HBM)
def test_device_resident_extend_and_merkelize(gl, oracle):
    import torch
    rng = np.random.default_rng(17)
    nBits, nPols, ext = 12, 8, 3
    a = rand_field(rng, (1 << nBits, nPols))
    src = torch.from_numpy(a.view(np.int64)).cuda()
    dst = torch.empty((1 << (nBits + ext)) * nPols, dtype=torch.int64, device="cuda")
    gl.interpolate(src, nPols, nBits, dst, nBits + ext)
    MH = gl.buildMerkleHash(False)
    tree = MH.merkelize(dst, nPols, 1 << (nBits + ext))
    torch.cuda.synchronize()
    e = oracle.interpolate(a, nBits, nBits + ext)
    assert (dst.cpu().numpy().view(np.uint64).reshape(e.shape) == e).all()
    nodes = oracle.merkelize(e, False)
    assert (tree["nodes"].cpu().numpy().view(np.uint64) == nodes).all()
    assert MH.root(tree) == nodes[-4:].tolist()
    v, mp = MH.getGroupProof(tree, 4097)
    assert v == e[4097].tolist() and MH.verifyGroupProof(MH.root(tree), mp, 4097, v)


@pytest.mark.parametrize("nBits,nPols,ext,cb,cc", [(12, 8, 3, 0, 8), (12, 8, 3, 2, 2), (10, 1, 3, 7, 1), (14, 33, 3, 4, 4),
                                                   (0, 3, 2, 1, 2), (3, 2, 1, 1, 1), (16, 100, 3, 5, 1), (21, 2, 2, 3, 1)])
def test_interpolate_coset_slice(gl, oracle, nBits, nPols, ext, cb, cc):
    """pil2gl_interpolate_cosets_dev = rows (pos << ext) + j, j in [cb, cb+cc), of the full interpolate"""
    import torch
    rng = np.random.default_rng(nBits * 31 + cb)
    a = rand_field(rng, (1 << nBits, nPols))
    src = torch.from_numpy(a.view(np.int64)).cuda()
    dst = torch.empty((nPols * cc) << nBits, dtype=torch.int64, device="cuda")
    gl.interpolateCosets(src, nPols, nBits, dst, nBits + ext, cb, cc)
    e = oracle.interpolate(a, nBits, nBits + ext).reshape(1 << nBits, 1 << ext, nPols)
    assert np.array_equal(dst.cpu().numpy().view(np.uint64), e[:, cb:cb + cc, :].reshape(-1))
    with pytest.raises(gl.Pil2glError):
        gl.interpolateCosets(src, nPols, nBits, dst, nBits + ext, (1 << ext) - cc + 1, cc)
    # caller-provided workspace, and the trace itself as workspace (overwritten, no library scratch)
    want = dst.clone()
    ws = torch.empty_like(src)
    dst.zero_(); gl.interpolateCosets(src, nPols, nBits, dst, nBits + ext, cb, cc, workspace=ws)
    assert torch.equal(dst, want) and np.array_equal(src.cpu().numpy().view(np.uint64).reshape(a.shape), a)
    dst.zero_(); gl.interpolateCosets(src, nPols, nBits, dst, nBits + ext, cb, cc, workspace=src)
    assert torch.equal(dst, want)


def test_merkelize_from_digests(gl, oracle):
    import torch
    rng = np.random.default_rng(23)
    for h in (1, 2, 33, 1000, 1 << 14):
        e = rand_field(rng, (h, 7))
        nodes = oracle.merkelize(e, False)
        MH = gl.buildMerkleHash(False)
        leaves = torch.from_numpy(nodes[:4 * h].copy().view(np.int64)).cuda()
        got = MH.merkelizeDigests(leaves, h).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, nodes)


# ------------------------------------------------------------------ stage-2 hints (csrc/hints.hip)
@pytest.mark.parametrize("n,dimNum,dimDen", [(1, 3, 3), (5, 1, 1), (2048, 3, 3), (2049, 1, 3), (70000, 3, 1), (1 << 18, 3, 3), (300001, 1, 1)])
def test_gprod_gsum_columns(gl, oracle, n, dimNum, dimDen):
    """calculateZ / calculateS (polutils.js:128-164) against the serial oracle; the product telescopes to 1 when the
    numerators are a rotation of the denominators (how a permutation argument uses it)"""
    import torch
    rng = np.random.default_rng(n + dimNum)
    num = rand_field(rng, n * dimNum); den = rand_field(rng, n * dimDen)
    den[den == 0] = 1
    dn, dd = torch.from_numpy(num.view(np.int64)).cuda(), torch.from_numpy(den.view(np.int64)).cuda()
    z = gl.calculateZ(dn, dd, dimNum, dimDen).cpu().numpy().view(np.uint64)
    assert np.array_equal(z, oracle.gprod(num, den, dimNum, dimDen))
    s = gl.calculateS(dn[:dimNum].contiguous(), dd, dimNum, dimDen).cpu().numpy().view(np.uint64)
    assert np.array_equal(s, oracle.gsum(num[:dimNum], den, dimNum, dimDen))
    if n >= 2048:
        # zero denominators: a lane inverts its eight denominators together, and a zero among them (0^(p-2) = 0 when inverted alone, as
        # the oracle does) must not reach that product -- one in a batch, a whole batch, the column's last row
        dz = den.reshape(n, dimDen).copy()
        dz[7] = 0; dz[1024:1032] = 0; dz[1033] = 0; dz[n - 1] = 0
        dz = dz.reshape(-1)
        ddz = torch.from_numpy(dz.view(np.int64)).cuda()
        assert np.array_equal(gl.calculateZ(dn, ddz, dimNum, dimDen).cpu().numpy().view(np.uint64), oracle.gprod(num, dz, dimNum, dimDen))
        assert np.array_equal(gl.calculateS(dn[:dimNum].contiguous(), ddz, dimNum, dimDen).cpu().numpy().view(np.uint64), oracle.gsum(num[:dimNum], dz, dimNum, dimDen))
    if dimNum == dimDen and n > 1:
        rot = np.roll(den.reshape(n, dimDen), 1, axis=0).reshape(-1).copy()
        z2 = gl.calculateZ(torch.from_numpy(rot.view(np.int64)).cuda(), dd, dimNum, dimDen).cpu().numpy().view(np.uint64).reshape(n, -1)
        last = oracle.gprod(rot, den, dimNum, dimDen).reshape(n, -1)[-1]
        assert np.array_equal(z2[-1], last)
        # z[n-1] * num[n-1]/den[n-1] closes the cycle: equals 1
        d = den.reshape(n, dimDen)
        if dimDen == 1:
            assert int(z2[-1][0]) * int(rot.reshape(n, 1)[-1][0]) % P == int(d[-1][0]) % P


@pytest.mark.parametrize("n,dim,distinct", [(1, 1, 1), (8, 1, 3), (1000, 1, 1000), (5000, 3, 70), (1 << 16, 1, 1 << 12), (70001, 3, 70001), (4096, 1, 1)])
def test_h1h2_columns(gl, oracle, n, dim, distinct):
    """calculateH1H2 (polutils.js:105-126) against the literal restatement: tables with repeated values (the last index
    wins), lookups hitting few or many entries, base and extension columns; a value missing from t is the reference's error"""
    import torch
    rng = np.random.default_rng(n * 7 + dim)
    vals = rand_field(rng, (distinct, dim))
    t = vals[rng.integers(0, distinct, n)] if distinct < n else vals[rng.permutation(n)]
    f = t[rng.integers(0, n, n)]
    if n >= 8:
        f[: n // 4] = t[0]                       # one heavily used entry
    dt, df = torch.from_numpy(t.reshape(-1).copy().view(np.int64)).cuda(), torch.from_numpy(f.reshape(-1).copy().view(np.int64)).cuda()
    h1, h2 = gl.calculateH1H2(df, dt, dim)
    key = (lambda r: int(r[0])) if dim == 1 else (lambda r: tuple(int(x) for x in r))
    w1, w2 = oracle.h1h2([key(r) for r in f], [key(r) for r in t])
    g1 = h1.cpu().numpy().view(np.uint64).reshape(n, dim); g2 = h2.cpu().numpy().view(np.uint64).reshape(n, dim)
    assert [key(r) for r in g1] == w1 and [key(r) for r in g2] == w2
    if n > 1 and distinct > 1:
        bad = f.copy(); bad[n // 2] = (bad[n // 2] + np.uint64(12345)) % np.uint64(P)
        if key(bad[n // 2]) not in {key(r) for r in t}:
            with pytest.raises(gl.Pil2glError, match="Number not included"):
                gl.calculateH1H2(torch.from_numpy(bad.reshape(-1).copy().view(np.int64)).cuda(), dt, dim)


def test_hints_against_reference_vectors(gl):
    """the device hint kernels against vectors written by the reference's own calculateZ / calculateS / calculateH1H2
    (polutils.js:105-164, run by oracle/gen_golden.js)"""
    import torch
    g = golden("hints.json")

    def rows(v, dim):
        out = []
        for r in v:
            r = H(r)
            out.append(list(r) if isinstance(r, list) else [r] + [0] * (dim - 1))       # F.one in row 0 of an extension column
        return out

    def dev(v, dim):
        return torch.from_numpy(np.array(rows(v, dim), dtype=np.uint64).reshape(-1).view(np.int64)).cuda()
    for c in g["gprod"]:
        dn, dd = c["dimNum"], c["dimDen"]; dim = max(dn, dd)
        z = gl.calculateZ(dev(c["num"], dn), dev(c["den"], dd), dn, dd).cpu().numpy().view(np.uint64).reshape(-1, dim)
        assert z.tolist() == rows(c["gprod"], dim), (c["n"], dn, dd)
    for c in g["gsum"]:
        dn, dd = c["dimNum"], c["dimDen"]; dim = max(dn, dd)
        sres = gl.calculateS(dev([c["num"]], dn), dev(c["den"], dd), dn, dd).cpu().numpy().view(np.uint64).reshape(-1, dim)
        assert sres.tolist() == rows(c["gsum"], dim), (c["n"], dn, dd)
    for c in g["h1h2"]:
        d = c["dim"]
        h1, h2 = gl.calculateH1H2(dev(c["f"], d), dev(c["t"], d), d)
        assert h1.cpu().numpy().view(np.uint64).reshape(-1, d).tolist() == rows(c["h1"], d), (c["n"], d)
        assert h2.cpu().numpy().view(np.uint64).reshape(-1, d).tolist() == rows(c["h2"], d), (c["n"], d)


# ------------------------------------------------------------------ extension-weighted sums (csrc/dot.hip)
def test_rows_dot_kernels_agree_on_many_tiles(gl, monkeypatch):
    """more row tiles than the persistent grid of the matrix-core kernel has workgroups (2 per CU), and extreme byte patterns
    (all bytes 0x00 / 0x7f / 0x80 / 0xff in values and weights): the three kernels give the same sums"""
    import torch
    from pil2gl import _lib
    rng = np.random.default_rng(77)
    n_rows, width, n_out = 100003, 100, 2
    m = rand_field(rng, (n_rows, width))
    pats = np.array([0, 0x7F7F7F7F7F7F7F7F, 0x8080808080808080, 0xFFFFFFFF00000000, P - 1, 0x00FF00FF00FF00FF], dtype=np.uint64)
    m[::7] = pats[rng.integers(0, len(pats), m[::7].shape)]
    coef = rand_field(rng, (n_out, width, 3)); coef[:, ::5] = pats[rng.integers(0, len(pats), coef[:, ::5].shape)] % np.uint64(P)
    dm = torch.from_numpy(m.view(np.int64).reshape(-1)).cuda()
    res = {}
    for mode in ("mfma", "stream", "tile"):
        monkeypatch.setenv("PIL2GL_ROWS_DOT_MFMA", "1" if mode == "mfma" else "0")
        monkeypatch.setenv("PIL2GL_ROWS_DOT_STREAM", "0" if mode == "tile" else "1")
        acc = torch.zeros(n_rows * n_out * 3, dtype=torch.int64, device="cuda")
        _lib.call("pil2gl_rows_dot_ext_dev", gl._ptr(dm), width, n_rows, gl._ptr(coef), n_out, gl._ptr(acc), 0, None)
        res[mode] = acc.cpu().numpy().view(np.uint64)
    assert np.array_equal(res["mfma"], res["tile"]) and np.array_equal(res["stream"], res["tile"])
    rows = [0, 1, 63, 64, 7 * 1000, n_rows - 1]
    want = (m[rows].astype(object) @ coef.astype(object).transpose(1, 0, 2).reshape(width, n_out * 3)) % P
    assert (res["mfma"].reshape(n_rows, n_out * 3)[rows].astype(object) == want).all()


@pytest.mark.parametrize("widths,n_out,n_rows", [((100, 6, 2), 2, 1000), ((100, 7, 2), 2, 333), ((40, 40, 30), 1, 129), ((34,), 2, 64),
                                                 ((2, 6, 100), 2, 4099), ((100, 6, 2, 2, 2), 2, 70), ((60, 60), 2, 200), ((100, 6, 2), 3, 100),
                                                 ((2, 18, 81, 6), 2, 1000), ((81,), 1, 130), ((35, 35, 35), 2, 257), ((111,), 2, 65), ((1, 31), 2, 100), ((113,), 2, 64),
                                                 ((200, 6, 2), 2, 777), ((2, 255, 6), 1, 300), ((120, 120, 120, 4, 4, 4), 2, 129), ((500, 3), 2, 70), ((30, 1), 2, 64),
                                                 ((100, 6, 2), 4, 300), ((2, 18, 81, 6), 3, 257), ((200, 7), 3, 130), ((16, 8), 4, 64)])
def test_rows_dot_over_several_matrices(gl, widths, n_out, n_rows):
    """pil2gl_rows_dot_ext_multi_dev: the stage matrices of the FRI polynomial side by side in one pass of the matrix-core kernel
    (<= 112 columns together with every odd width counted as the next even one -- an odd matrix is staged with a zero word after each row:
    the two-stage permutation AIR's 2 + 18 + 81 + 6 --, <= 4 matrices, <= 2 outputs); wider inputs go in column windows packed into several
    accumulating launches (config 5's 200 + 6 + 2: two), three or four outputs in two sweeps of those launches, what is left under 32 columns
    matrix by matrix on the vector kernels: sum over all columns"""
    import ctypes as C
    import torch
    from pil2gl import _lib
    rng = np.random.default_rng(sum(widths) + n_rows)
    ms = [rand_field(rng, (n_rows, w)) for w in widths]
    cs = [rand_field(rng, (n_out, w, 3)) for w in widths]
    ms[0][0, 0] = P - 1; cs[0][0, 0] = [P - 1, P - 1, P - 1]
    dms = [torch.from_numpy(m.view(np.int64).reshape(-1)).cuda() for m in ms]
    ptrs = (C.c_void_p * len(widths))(*[t.data_ptr() for t in dms])
    ws = np.array(widths, dtype=np.uint64)
    cps = (C.c_void_p * len(widths))(*[c.ctypes.data for c in cs])
    acc = torch.zeros(n_rows * n_out * 3, dtype=torch.int64, device="cuda")
    for accumulate in (0, 1):
        _lib.call("pil2gl_rows_dot_ext_multi_dev", ptrs, C.c_void_p(ws.ctypes.data), len(widths), n_rows, cps, n_out, gl._ptr(acc), accumulate, None)
    want = sum(m.astype(object) @ c.astype(object).transpose(1, 0, 2).reshape(w, n_out * 3) for m, c, w in zip(ms, cs, widths))
    got = acc.cpu().numpy().view(np.uint64).reshape(n_rows, n_out * 3)
    assert (got.astype(object) == (2 * want) % P).all()


@pytest.mark.parametrize("mode", ["mfma", "stream", "tile"])
def test_rows_and_cols_dot_ext(gl, oracle, monkeypatch, mode):
    """the three kernels behind pil2gl_rows_dot_ext_dev.  mfma (default): even rows of 32..112 columns with one or two outputs go
    to the matrix cores (rows_dot_mfma_kernel); stream: rows of 32 columns and more through the whole-row streaming kernel on
    the vector ALU (what the other shapes take anyway); tile: every shape through the column-tile kernel"""
    import ctypes as C
    monkeypatch.setenv("PIL2GL_ROWS_DOT_MFMA", "1" if mode == "mfma" else "0")
    monkeypatch.setenv("PIL2GL_ROWS_DOT_STREAM", "0" if mode == "tile" else "1")
    import torch
    from pil2gl import _lib
    rng = np.random.default_rng(31)
    # every row is checked; widths around the 16-column tile, row counts around the 256-row block, and a buffer that does not
    # start on a 128-byte line
    for n_rows, width, n_out, skew in [(1000, 37, 2, 0), (64, 1, 1, 0), (4097, 100, 3, 0), (300, 16, 4, 0), (515, 12, 1, 0), (1030, 20, 4, 0),
                                       (256, 4, 2, 0), (777, 36, 2, 0), (3, 100, 1, 0), (700, 100, 2, 1), (259, 44, 3, 0),
                                       (200, 128, 2, 0), (65, 129, 1, 0), (1000, 255, 4, 1), (63, 33, 3, 0), (129, 32, 2, 0),
                                       (4097, 100, 2, 0), (64, 112, 2, 0), (200, 110, 1, 0), (1, 34, 2, 0)]:
        m = rand_field(rng, (n_rows, width)); m[0, 0] = P - 1; m[-1, -1] = 0
        coef = rand_field(rng, (n_out, width, 3)); coef[0, 0] = [P - 1, P - 1, P - 1]
        store = torch.zeros(n_rows * width + skew, dtype=torch.int64, device="cuda")
        dm = store[skew:]; dm.copy_(torch.from_numpy(m.view(np.int64).reshape(-1)))
        assert dm.data_ptr() % 128 == 8 * skew
        acc = torch.zeros(n_rows * n_out * 3, dtype=torch.int64, device="cuda")
        _lib.call("pil2gl_rows_dot_ext_dev", gl._ptr(dm), width, n_rows, gl._ptr(coef), n_out, gl._ptr(acc), 0, None)
        _lib.call("pil2gl_rows_dot_ext_dev", gl._ptr(dm), width, n_rows, gl._ptr(coef), n_out, gl._ptr(acc), 1, None)   # accumulate: 2x
        got = acc.cpu().numpy().view(np.uint64).reshape(n_rows, n_out, 3)
        want = (2 * (m.astype(object) @ coef.astype(object).transpose(1, 0, 2).reshape(width, n_out * 3))) % P
        assert (got.reshape(n_rows, n_out * 3).astype(object) == want).all(), (n_rows, width, n_out, skew)
    # stages wider than the 1024 terms a lane's unreduced sums can hold (the library cuts them into windows): values and
    # weights whose every limb is at its maximum, so that a window one column too long would wrap the 64-bit partial sums
    for n_rows, width, n_out in [(130, 1500, 2), (70, 5000, 4), (64, 1024, 1), (64, 1025, 1)]:
        m = np.full((n_rows, width), P - 1, dtype=np.uint64); m[1::2] = rand_field(rng, (len(m[1::2]), width))
        coef = np.full((n_out, width, 3), P - 1, dtype=np.uint64); coef[:, ::3] = rand_field(rng, coef[:, ::3].shape)
        dm = torch.from_numpy(m.view(np.int64).reshape(-1)).cuda()
        acc = torch.zeros(n_rows * n_out * 3, dtype=torch.int64, device="cuda")
        _lib.call("pil2gl_rows_dot_ext_dev", gl._ptr(dm), width, n_rows, gl._ptr(coef), n_out, gl._ptr(acc), 0, None)
        got = acc.cpu().numpy().view(np.uint64).reshape(n_rows, n_out * 3)
        want = (m.astype(object) @ coef.astype(object).transpose(1, 0, 2).reshape(width, n_out * 3)) % P
        assert (got.astype(object) == want).all(), (n_rows, width, n_out)
    # several matrices in one sweep of the weights (pil2gl_cols_dot_ext_multi_dev) == matrix by matrix
    nb2, eb2 = 10, 3
    mats = [rand_field(rng, (1 << (nb2 + eb2), w)) for w in (100, 6, 2, 65)]
    dmats = [torch.from_numpy(m.view(np.int64)).cuda() for m in mats]
    levs2 = [rand_field(rng, (1 << nb2, 3)) for _ in range(3)]
    dlevs2 = [torch.from_numpy(l.view(np.int64)).cuda() for l in levs2]
    lv2 = (C.c_void_p * 3)(*[t.data_ptr() for t in dlevs2])
    single = []
    for m, dmat in zip(mats, dmats):
        o = np.zeros((3, m.shape[1], 3), np.uint64)
        _lib.call("pil2gl_cols_dot_ext_dev", gl._ptr(dmat), m.shape[1], 1 << nb2, 1 << eb2, lv2, 3, gl._ptr(o), None)
        single.append(o)
    multi = [np.zeros_like(o) for o in single]
    ptrs = (C.c_void_p * len(mats))(*[t.data_ptr() for t in dmats])
    ws = np.array([m.shape[1] for m in mats], dtype=np.uint64)
    outs = (C.c_void_p * len(mats))(*[o.ctypes.data for o in multi])
    _lib.call("pil2gl_cols_dot_ext_multi_dev", ptrs, C.c_void_p(ws.ctypes.data), len(mats), 1 << nb2, 1 << eb2, lv2, 3, outs, None)
    for a_, b_ in zip(single, multi):
        assert np.array_equal(a_, b_)
    assert single[1][2, 5].tolist() == oracle.eval_pol_at(mats[1], 5, 1, nb2, eb2, levs2[2]).tolist()
    # six opening points in one call (a sweep weighs four: two sweeps inside the library) == the points three at a time
    levs6 = dlevs2 + [torch.from_numpy(rand_field(rng, (1 << nb2, 3)).view(np.int64)).cuda() for _ in range(3)]
    lv6 = (C.c_void_p * 6)(*[t.data_ptr() for t in levs6])
    lvb = (C.c_void_p * 3)(*[t.data_ptr() for t in levs6[3:]])
    six = [np.zeros((6, m.shape[1], 3), np.uint64) for m in mats]
    outs6 = (C.c_void_p * len(mats))(*[o.ctypes.data for o in six])
    _lib.call("pil2gl_cols_dot_ext_multi_dev", ptrs, C.c_void_p(ws.ctypes.data), len(mats), 1 << nb2, 1 << eb2, lv6, 6, outs6, None)
    second = [np.zeros_like(o) for o in single]
    outs3 = (C.c_void_p * len(mats))(*[o.ctypes.data for o in second])
    _lib.call("pil2gl_cols_dot_ext_multi_dev", ptrs, C.c_void_p(ws.ctypes.data), len(mats), 1 << nb2, 1 << eb2, lvb, 3, outs3, None)
    for a_, b_, c_ in zip(six, multi, second):
        assert np.array_equal(a_[:3], b_) and np.array_equal(a_[3:], c_)
    # column sums against the oracle's per-column evaluation (stark_gen_helpers.js:250-264)
    nb, eb, width = 11, 3, 9
    buf = rand_field(rng, (1 << (nb + eb), width)); dbuf = torch.from_numpy(buf.view(np.int64)).cuda()
    levs = [rand_field(rng, (1 << nb, 3)) for _ in range(2)]
    dlevs = [torch.from_numpy(l.view(np.int64)).cuda() for l in levs]
    lv = (C.c_void_p * 2)(*[t.data_ptr() for t in dlevs]); out = np.zeros((2, width, 3), np.uint64)
    _lib.call("pil2gl_cols_dot_ext_dev", gl._ptr(dbuf), width, 1 << nb, 1 << eb, lv, 2, gl._ptr(out), None)
    for l in range(2):
        for c in range(width):
            assert out[l, c].tolist() == oracle.eval_pol_at(buf, c, 1, nb, eb, levs[l]).tolist()


@pytest.mark.parametrize("jit", ["0", "1", "wide0", "wide"])
def test_expression_evaluator_interpreter_and_jit_agree(gl, oracle, jit, monkeypatch):
    """the same random programs through the interpreter (PIL2GL_EXPR_JIT=0) and the hiprtc-compiled kernel (=1), on narrow sections and
    on WIDE ones (20 and 34 columns, row offsets -2..2 and -8..8: interpreter "wide0", compiled "wide")"""
    import torch
    import ctypes as C
    from pil2gl import _lib
    monkeypatch.setenv("PIL2GL_EXPR_JIT", "0" if jit in ("0", "wide0") else "1")
    for n_ops, prime_shift in [(40, 0), (300, 2)]:
        rng = np.random.default_rng(1000 + n_ops)
        n_bits = 10
        widths = [5, 9, 1, 3] if jit in ("0", "1") else [20, 34, 1, 3]
        secs = [rand_field(rng, (1 << n_bits, w)) for w in widths]; secs[-1][:] = 0
        scalars = rand_field(rng, 40)
        ops, n_tmp = _random_program(rng, n_ops, widths, scalars.size, len(widths) - 1)
        ref_secs = [s.copy() for s in secs]
        oracle.eval_program(ops, n_tmp, ref_secs, scalars, n_bits, prime_shift)
        dsecs = [torch.from_numpy(s.view(np.int64)).cuda() for s in secs]
        prog = oracle.make_program(ops, n_tmp, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
        csecs = (_lib.GlxSection * len(dsecs))()
        for i, s in enumerate(dsecs):
            csecs[i].ptr = s.data_ptr(); csecs[i].width = widths[i]
        ctx = _lib.GlxCtx(n_bits, prime_shift, len(dsecs), scalars.size, csecs, scalars.ctypes.data_as(_lib.u64p))
        _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None)
        torch.cuda.synchronize()
        assert (dsecs[-1].cpu().numpy().view(np.uint64).reshape(ref_secs[-1].shape) == ref_secs[-1]).all()


@pytest.mark.parametrize("mulcall", ["0", "1"])
def test_compiled_evaluator_lazy_products_reach_their_readers_non_canonical(gl, oracle, mulcall, monkeypatch):
    """the run-time compiled kernel keeps a product lazy when only products and fused multiply-accumulates read it.  Random
    operands give a representative >= p once in 2^32; here (2^32+1)(2^32-1) = 2^64-1 does in every row where the columns hold
    those two, so the readers -- a product, a dim-3 scaling, a Horner chain fused into lazy multiply-accumulates -- all see it"""
    import torch
    import ctypes as C
    from pil2gl import _lib
    from gl_oracle import TMP, SEC, SCALAR
    monkeypatch.setenv("PIL2GL_EXPR_JIT", "1")
    monkeypatch.setenv("PIL2GL_EXPR_MULCALL", mulcall)
    rng = np.random.default_rng(77)
    n_bits = 8
    widths = [6, 3, 3]
    secs = [rand_field(rng, (1 << n_bits, w)) for w in widths]; secs[-1][:] = 0
    secs[0][::2, 0] = (1 << 32) + 1; secs[0][::2, 1] = (1 << 32) - 1          # product 2^64 - 1 in the even rows
    secs[0][1::4, 0] = P - 1; secs[0][1::4, 1] = P - 1                          # and (-1)(-1) = 1 via the largest operands
    scalars = rand_field(rng, 12)
    T = lambda i, d=1: (TMP, d, 0, 0, i)
    S = lambda c, d=1, sec=0: (SEC, d, sec, 0, c)
    ops = [("mul", T(0), S(0), S(1)),                  # lazy: read by products and multiply-accumulates only
           ("mul", T(1), T(0), S(2)),                  # product of a lazy value (itself read by an addition: canonical)
           ("mul", T(2, 3), S(0, 3, 1), T(0)),         # dim 3 x lazy dim 1
           ("add", T(3), T(1), S(3))]
    # Horner chain over the extension scalar X = scalars[0..2] with the lazy product as every c_i:  t <- X t + c
    acc = 2
    for i in range(6):
        ops.append(("mul", T(4 + 2 * i, 3), (SCALAR, 3, 0, 0, 0), T(acc, 3)))
        ops.append(("add", T(5 + 2 * i, 3), T(4 + 2 * i, 3), T(0) if i % 2 == 0 else T(1)))
        acc = 5 + 2 * i
    ops.append(("mul", T(16, 3), T(acc, 3), T(3)))
    ops.append(("copy", (SEC, 3, 2, 0, 0), T(16, 3), None))
    n_tmp = 17
    ref = [x.copy() for x in secs]
    oracle.eval_program(ops, n_tmp, ref, scalars, n_bits, 0)
    dsecs = [torch.from_numpy(x.view(np.int64)).cuda() for x in secs]
    prog = oracle.make_program(ops, n_tmp, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
    csecs = (_lib.GlxSection * len(dsecs))()
    for i, x in enumerate(dsecs):
        csecs[i].ptr = x.data_ptr(); csecs[i].width = widths[i]
    ctx = _lib.GlxCtx(n_bits, 0, len(dsecs), scalars.size, csecs, scalars.ctypes.data_as(_lib.u64p))
    _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None)
    torch.cuda.synchronize()
    got = dsecs[-1].cpu().numpy().view(np.uint64).reshape(ref[-1].shape)
    assert (got < np.uint64(P)).all()
    assert (got == ref[-1]).all()


def test_clock_probe_reports_a_plausible_shader_clock(gl):
    """pil2gl_selftest_clock (what bench.py prices issue cycles with): a clock between the idle and the nominal one, spread sane"""
    import ctypes as C
    from pil2gl import _lib
    mhz = (C.c_double * 3)()
    _lib.call("pil2gl_selftest_clock", C.c_uint32(8), mhz)
    assert 500 < mhz[1] <= mhz[0] <= mhz[2] < 3000


def test_pols_file_to_device_and_back(gl, tmp_path):
    """`.commit`-style file (witnessCalculator.js:145-196) streamed into HBM, committed, written back"""
    import torch
    from pil2gl import io
    rng = np.random.default_rng(8)
    a = rand_field(rng, (1 << 12, 5))
    f = str(tmp_path / "t.commit")
    io.save_pols(a, f)
    d = io.load_pols(f, 1 << 12, 5, torch.device("cuda", 0))
    assert d.is_cuda and np.array_equal(d.cpu().numpy().view(np.uint64), a.reshape(-1))
    g = str(tmp_path / "u.commit")
    io.save_pols(d, g)
    assert open(f, "rb").read() == open(g, "rb").read()


def test_randomised_differential_run():
    """tests/fuzz/fuzz_parity.py for 25 s with a fixed seed: every operator and whole proofs on shapes drawn at random, each against the oracle
    (the long runs of the round are recorded in profiles/r04_fuzz_parity.txt)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_parity.py"), "25", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "failures: 0" in r.stdout and "'proof'" in r.stdout and "'eval_program'" in r.stdout


@pytest.mark.parametrize("nb,eb,qDim,qDeg", [(3, 1, 3, 2), (8, 3, 3, 2), (9, 2, 3, 3), (12, 3, 3, 2), (16, 3, 3, 2), (17, 1, 1, 2), (10, 4, 3, 5), (20, 3, 3, 2)])
def test_quotient_pieces_extended_from_their_coefficients(gl, oracle, nb, eb, qDim, qDeg):
    """pil2gl_compute_q_split_brev_dev + pil2gl_extend_coefs_brev_dev == computeQStark's split and plain transform of the zero-padded
    matrix (stark_gen_helpers.js:179-192), against the oracle's q_split + fft; sizes with one, two and three forward sweeps"""
    import torch
    from pil2gl import _lib
    nbe = nb + eb
    rng = np.random.default_rng(nb * 10 + eb)
    qq1 = rand_field(rng, (1 << nbe, qDim))
    want = oracle.fft_cols(oracle.compute_q_split(qq1, nb, nbe, qDim, qDeg), nbe)
    d1 = torch.from_numpy(qq1.view(np.int64).reshape(-1)).cuda()
    W = qDim * qDeg
    c = torch.zeros(W << nb, dtype=torch.int64, device="cuda"); out = torch.zeros(W << nbe, dtype=torch.int64, device="cuda")
    _lib.call("pil2gl_compute_q_split_brev_dev", gl._ptr(d1), nb, nbe, qDim, qDeg, gl._ptr(c), None)
    _lib.call("pil2gl_extend_coefs_brev_dev", gl._ptr(c), W, nb, gl._ptr(out), nbe, None)
    assert (out.cpu().numpy().view(np.uint64).reshape(1 << nbe, W) == want).all()
    # the coefficient matrix itself: row bitrev(i) = coefficient i of every piece
    full = oracle.compute_q_split(qq1, nb, nbe, qDim, qDeg)[:1 << nb]
    br = np.array([int(format(i, "0%db" % nb)[::-1], 2) if nb else 0 for i in range(1 << nb)])
    assert (c.cpu().numpy().view(np.uint64).reshape(1 << nb, W)[br] == full).all()
    with pytest.raises(gl.Pil2glError):
        _lib.call("pil2gl_extend_coefs_brev_dev", gl._ptr(c), W, nb, gl._ptr(out), 40, None)
    # coset slices of the same extension (a rank's part of the split quotient, pil2gl_extend_coefs_brev_cosets_dev): row (pos, j - cb) = row (pos << eb) + j
    for cb, cc in {(0, 1), ((1 << eb) - 1, 1), (0, 1 << eb), ((1 << eb) // 2, (1 << eb) // 2)}:
        sl = torch.zeros((W * cc) << nb, dtype=torch.int64, device="cuda")
        _lib.call("pil2gl_extend_coefs_brev_cosets_dev", gl._ptr(c), W, nb, gl._ptr(sl), nbe, cb, cc, None)
        assert (sl.cpu().numpy().view(np.uint64).reshape(1 << nb, cc, W) == want.reshape(1 << nb, 1 << eb, W)[:, cb:cb + cc]).all(), (cb, cc)
    with pytest.raises(gl.Pil2glError):
        _lib.call("pil2gl_extend_coefs_brev_cosets_dev", gl._ptr(c), W, nb, gl._ptr(out), nbe, 1 << eb, 1, None)


@pytest.mark.parametrize("wide", ["1", "0"])
def test_narrow_interpolate_with_wide_forward_passes(gl, oracle, wide, monkeypatch):
    """a matrix whose rows are under 128 bytes takes 7-stage passes, but its extension viewed as N x (C * cosets) is wide: the forward passes
    then take 8 stages (PIL2GL_LDE_WIDEFWD, default on; 17-26 % on narrow interpolates at 2^24 rows) -- same values either way"""
    monkeypatch.setenv("PIL2GL_LDE_WIDEFWD", wide)
    rng = np.random.default_rng(77)
    for nb, C, eb in [(15, 2, 3), (17, 6, 3), (16, 8, 2), (18, 3, 1), (14, 12, 4), (16, 1, 3)]:
        a = rand_field(rng, (1 << nb, C))
        out = np.zeros((1 << (nb + eb), C), np.uint64)
        gl.interpolate(a, C, nb, out, nb + eb)
        assert (out == oracle.interpolate(a, nb, nb + eb)).all(), (nb, C, eb)
