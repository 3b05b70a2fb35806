"""The reference's worker-level transform operators, fft_block and interpolatePrepareBlock (src/helpers/fft/fft_worker.js:6-67).

fft_worker.js cannot be loaded here (it needs `workerpool`), so no vector comes from it directly.  What the reference's own tests
check (test/fft_p.test.js) is the transform these operators compose under fft_p.js's block loop, so that is the pin: the CPU
restatement (oracle/fft_worker_ref.py), driven by a restatement of that loop (fft_p.js:114-176, 187-297), must reproduce the
reference's scalar fft / ifft / extendPol outputs (tests/golden/ntt.json) for every block size -- and the device operators must be
bit-identical to the restatement on arbitrary arguments, and reproduce the same goldens under the same loop."""
import numpy as np
import pytest

from conftest import golden, U, P, rand_field
import fft_worker_ref as R


def _cases():
    return [c for c in golden("ntt.json")["cases"] if c["nBits"] >= 1]


def _cols(p, nPols):
    """p in column 0, shifted copies in the others (row-major)"""
    n = len(p)
    return [p[(i + 3 * k) % n] for i in range(n) for k in range(nPols)]


# ------------------------------------------------------------------ the restatement against the reference's vectors (CPU)
def test_block_loop_reproduces_reference_fft_for_every_block_size():
    for c in _cases():
        p = [int(x) for x in U(c["p"])]
        f, fi = [int(x) for x in U(c["fft"])], [int(x) for x in U(c["ifft"])]
        for bb in (1, 2, 3, 4, 5, 7, 12):
            if bb > c["nBits"] and bb != 12:
                continue
            assert R.fft_p(p, 1, c["nBits"], False, bb) == f, (c["name"], bb)
            assert R.fft_p(p, 1, c["nBits"], True, bb) == fi, (c["name"], bb)


def test_block_loop_reproduces_reference_extendpol():
    for c in _cases():
        p = [int(x) for x in U(c["p"])]
        for eb, out in c["ext"].items():
            e = [int(x) for x in U(out)]
            for bb, bbe, per in ((12, 12, 1 << 12), (2, 3, 3), (1, 1, 1), (3, 2, 5)):
                assert R.interpolate_p(p, 1, c["nBits"], c["nBits"] + int(eb), bb, bbe, per) == e, (c["name"], eb, bb, bbe, per)


def test_block_loop_is_columnwise():
    """rows are nPols wide and the operators never mix columns (fft_worker.js:12-14, :53-58)"""
    c = [x for x in _cases() if x["name"] == "rand4"][0]
    p = [int(x) for x in U(c["p"])]
    a = R.fft_p(_cols(p, 3), 3, 4, False, 2)
    for k in range(3):
        col = [p[(i + 3 * k) % 16] for i in range(16)]
        assert a[k::3] == R.fft_p(col, 1, 4, False, 4)


# ------------------------------------------------------------------ the device operators (through the C ABI)
@pytest.fixture(scope="module")
def gl():
    import pil2gl
    pil2gl.init(0)
    return pil2gl


def _dev_block(gl):
    def op(bb, start_pos, nPols, nBits, s, blockBits, layers):
        a = np.array(bb, dtype=np.uint64)
        gl.fft_block(a, start_pos, nPols, nBits, s, blockBits, layers)
        return [int(x) for x in a]
    return op


def _dev_prepare(gl):
    def op(bb, width, start, inc):
        a = np.array(bb, dtype=np.uint64)
        gl.interpolatePrepareBlock(a, width, start, inc)
        return [int(x) for x in a]
    return op


@pytest.mark.gpu
def test_fft_block_device_equals_restatement_on_arbitrary_arguments(gl):
    rng = np.random.default_rng(2024)
    seen = set()
    for trial in range(160):
        nBits = int(rng.integers(1, 12))
        blockBits = int(rng.integers(0, min(nBits, 7) + 1))
        layers = int(rng.integers(0, blockBits + 1))
        s = int(rng.integers(layers, nBits + 1)) if trial % 5 else int(rng.integers(0, nBits + 1))      # stages below `layers` too (w0 = 1 there)
        if s > layers and s - layers > nBits:
            continue
        nPols = int(rng.integers(1, 6))
        start_pos = int(rng.integers(0, 1 << (nBits - blockBits))) << blockBits
        if trial % 7 == 0:
            start_pos = int(rng.integers(0, (1 << nBits) - (1 << blockBits) + 1))                        # unaligned blocks are legal arguments
        buf = rand_field(rng, nPols << blockBits)
        if trial % 11 == 0:
            buf[::3] = np.uint64(P) + (buf[::3] & np.uint64(0xFFFF))                                     # non-canonical words: F.mul / F.add reduce them
        want = R.fft_block([int(x) for x in buf], start_pos, nPols, nBits, s, blockBits, layers)
        got = buf.copy()
        assert gl.fft_block(got, start_pos, nPols, nBits, s, blockBits, layers) is got
        if layers == 0:
            assert (got == buf).all()                     # nothing is paired: the words stay as they are
            continue
        assert [int(x) for x in got] == want, (nBits, s, blockBits, layers, nPols, start_pos)
        seen.add((s > layers, layers < blockBits))
    assert len(seen) == 4
    # a device block, in place; the largest stage index; one wide block
    import torch
    buf = rand_field(rng, 5 << 6)
    d = torch.from_numpy(buf.view(np.int64)).cuda()
    assert gl.fft_block(d, 3 << 6, 5, 32, 32, 6, 6) is d
    assert [int(x) for x in d.cpu().numpy().view(np.uint64)] == R.fft_block([int(x) for x in buf], 3 << 6, 5, 32, 32, 6, 6)
    buf = rand_field(rng, 100 << 10)
    got = buf.copy(); gl.fft_block(got, 1 << 10, 100, 14, 12, 10, 8)
    assert [int(x) for x in got] == R.fft_block([int(x) for x in buf], 1 << 10, 100, 14, 12, 10, 8)


@pytest.mark.gpu
def test_fft_block_argument_errors(gl):
    from pil2gl import Pil2glError
    a = np.zeros(8, np.uint64)
    with pytest.raises(Pil2glError, match="layers"):
        gl.fft_block(a, 0, 1, 3, 3, 2, 3)                 # fft_worker.js:30-38 never end for layers > blockBits
    with pytest.raises(Pil2glError, match="nBits"):
        gl.fft_block(a, 0, 1, 33, 3, 3, 3)                # F.w has 33 entries
    with pytest.raises(Pil2glError, match="stage"):
        gl.fft_block(a, 0, 1, 3, 6, 2, 2)                 # n / width would not be an integer
    with pytest.raises(Pil2glError, match="needs"):
        gl.fft_block(a, 0, 1, 4, 4, 4, 4)                 # 16 rows in an 8-word buffer
    assert gl.fft_block(np.zeros(0, np.uint64), 0, 0, 3, 3, 3, 3).size == 0


@pytest.mark.gpu
def test_interpolate_prepare_block_device(gl):
    rng = np.random.default_rng(5)
    for width, height in ((1, 1), (3, 7), (100, 64), (8, 1000), (5, 0)):
        buf = rand_field(rng, width * height)
        start, inc = int(rand_field(rng, 1)[0]), int(rand_field(rng, 1)[0])
        if height == 7:
            start, inc = P + 5, (1 << 64) - 1             # F.mul reduces whatever BigInt it is given
        want = R.interpolatePrepareBlock([int(x) for x in buf], width, start, inc)
        got = buf.copy()
        assert gl.interpolatePrepareBlock(got, width, start, inc, 0, 1) is got
        assert [int(x) for x in got] == want


@pytest.mark.gpu
def test_reference_block_loop_over_device_operators_reproduces_reference_vectors(gl):
    """fft_p.js's own loop (host bit reversal and transposes, as there) with the device operators in the workers' place"""
    for c in _cases():
        if c["nBits"] > 7:
            continue
        p = [int(x) for x in U(c["p"])]
        f, fi = [int(x) for x in U(c["fft"])], [int(x) for x in U(c["ifft"])]
        for bb in (1, 3, 12):
            assert R.fft_p(p, 1, c["nBits"], False, bb, block_op=_dev_block(gl)) == f
            assert R.fft_p(p, 1, c["nBits"], True, bb, block_op=_dev_block(gl)) == fi
        for eb, out in c["ext"].items():
            e = [int(x) for x in U(out)]
            assert R.interpolate_p(p, 1, c["nBits"], c["nBits"] + int(eb), 2, 3, 3, block_op=_dev_block(gl), prepare_op=_dev_prepare(gl)) == e


@pytest.mark.gpu
def test_block_loop_over_device_operators_equals_the_one_call_transforms(gl):
    """the product path (pil2gl.fft / ifft / interpolate: one call) and the reference's loop over the worker-level operators agree
    on a wide random matrix"""
    rng = np.random.default_rng(77)
    nBits, nPols, eb = 9, 13, 2
    src = rand_field(rng, nPols << nBits)
    out = np.zeros_like(src)
    gl.fft(src, nPols, nBits, out)
    assert R.fft_p([int(x) for x in src], nPols, nBits, False, 4, block_op=_dev_block(gl)) == [int(x) for x in out]
    gl.ifft(src, nPols, nBits, out)
    assert R.fft_p([int(x) for x in src], nPols, nBits, True, 5, block_op=_dev_block(gl)) == [int(x) for x in out]
    ext = np.zeros(nPols << (nBits + eb), np.uint64)
    gl.interpolate(src, nPols, nBits, ext, nBits + eb)
    got = R.interpolate_p([int(x) for x in src], nPols, nBits, nBits + eb, 4, 6, 100, block_op=_dev_block(gl), prepare_op=_dev_prepare(gl))
    assert got == [int(x) for x in ext]
