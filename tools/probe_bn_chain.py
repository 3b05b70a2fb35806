"""Time of a BN128 transcript put of a long list: chained kernel against one permutation call per block."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pil2-stark-js_amd", "python"))
import pil2gl
pil2gl.init(0)
from pil2gl import bn128 as bn

vals = [(7 ** (k + 5)) % bn.R for k in range(16 * 40)]
blocks = [vals[k * 16:(k + 1) * 16] for k in range(40)]
bn.poseidon_chain(blocks[:2], 0); bn.poseidon(blocks[0], 0, 17)
t0 = time.time(); a = bn.poseidon_chain(blocks, 0); t1 = time.time()
s = 0
for b in blocks:
    o = bn.poseidon(b, s, 17); s = o[0]
t2 = time.time()
assert a == o
print("40 blocks of 16: chain %.2f ms (%.3f ms/perm), sequential calls %.2f ms (%.3f ms/perm)" % ((t1 - t0) * 1e3, (t1 - t0) * 25, (t2 - t1) * 1e3, (t2 - t1) * 25))
