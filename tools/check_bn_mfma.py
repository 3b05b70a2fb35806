#!/usr/bin/env python3
"""Every state width of the BN254 permutation through the one-lane-per-permutation kernels (the matrix-core layers) against the
Python-integer oracle, then an A/B timing of the 2^20 x 100 arity-16 commit with the layers on the vector ALU / matrix cores.
  PIL2GL_BN128_WAVE_PER_PERM_MAX=0 python tools/check_bn_mfma.py"""
import os
import subprocess
import sys

os.environ.setdefault("PIL2GL_BN128_WAVE_PER_PERM_MAX", "0")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "pil2-stark-js_amd", "python"), os.path.join(ROOT, "oracle")]
import numpy as np
import pil2gl
from pil2gl import bn128
import bn128_oracle as orc

pil2gl.init(0)
rng = np.random.default_rng(11)
bad = 0
for n_in in range(1, 17):
    ins = [[int.from_bytes(rng.bytes(32), "little") % orc.R for _ in range(n_in)] for _ in range(130)]
    ins[0] = [0] * n_in
    ins[1] = [orc.R - 1] * n_in
    ins[64] = [orc.R - 1 - k for k in range(n_in)]
    init = [int.from_bytes(rng.bytes(32), "little") % orc.R for _ in ins]
    init[0] = 0
    n_out = n_in + 1
    got = bn128.poseidon_batch(ins, init, n_out)
    ok = [k for k, (a, s, g) in enumerate(zip(ins, init, got)) if g == orc.poseidon(a, s, n_out)]
    nb = len(ins) - len(ok)
    if nb and os.environ.get("CHECK_VERBOSE"):
        print("   agree:", ok[:20], " first outputs of 5:", [hex(v) for v in got[5][:2]], "want", [hex(v) for v in orc.poseidon(ins[5], init[5], n_out)[:2]])
    print("t = %2d: %d of %d permutations differ from the oracle" % (n_in + 1, nb, len(ins)), flush=True)
    bad += nb
print("MISMATCH" if bad else "all widths agree")
if len(sys.argv) > 1:
    for mfma in ("0", "1", "0", "1"):
        env = dict(os.environ, PIL2GL_BN128_MFMA=mfma)
        env.pop("PIL2GL_BN128_WAVE_PER_PERM_MAX")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_bn128.py"), sys.argv[1], "100", "16"], env=env, capture_output=True, text=True)
        print("PIL2GL_BN128_MFMA=" + mfma, out.stdout.strip(), out.stderr.strip()[-300:], flush=True)
sys.exit(1 if bad else 0)
