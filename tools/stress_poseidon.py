"""the matrix-core permutation (blocked partial rounds, carry-out products with their rare fallbacks) against the plain vector-ALU
permutation of the same library, on the GPU, over many random states: the rare paths -- a product's final borrow (2^-32 per product),
a recombination's last carry (2^-15 per element) -- are taken a few dozen / many thousand times in 2^28 permutations, which no CPU
oracle run reaches.   usage: python tools/stress_poseidon.py [log2 of the number of permutations, default 26]"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl
from pil2gl import _lib
pil2gl.init(0)
P = 0xFFFFFFFF00000001
total_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 26
chunk_bits = min(total_bits, 22)
n = 1 << chunk_bits
rng = np.random.default_rng(20261004)
bad = 0; t0 = time.time()
for c in range(1 << (total_bits - chunk_bits)):
    st = rng.integers(0, P, size=(n, 12), dtype=np.uint64)
    if c % 4 == 1:                                      # structured extremes in a part of the chunks
        st[::3, ::2] = P - 1 - rng.integers(0, 4, size=st[::3, ::2].shape, dtype=np.uint64)
        st[1::3, 1::2] = rng.integers(0, 4, size=st[1::3, 1::2].shape, dtype=np.uint64)
    outs = []
    for what in (0, 2):
        o = np.empty_like(st)
        _lib.call("pil2gl_selftest_poseidon", st.ctypes.data_as(_lib.u64p), C.c_uint64(n), C.c_int(what), o.ctypes.data_as(_lib.u64p))
        outs.append(o)
    d = int((outs[0] != outs[1]).any(axis=1).sum())
    bad += d
    if d or c % 8 == 7:
        print("chunk %d: %d differing permutations so far (%.0f s)" % (c, bad, time.time() - t0), flush=True)
print("2^%d permutations, matrix-core form vs vector-ALU form: %d differ" % (total_bits, bad))
sys.exit(1 if bad else 0)
