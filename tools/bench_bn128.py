#!/usr/bin/env python3
"""Timing of the BN128 Merkle commitment (config 4 shape: 100 columns, arity 16) at a reduced row count.
  python tools/bench_bn128.py [log2 rows] [cols] [arity] [custom]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pil2-stark-js_amd", "python"))
import torch
import pil2gl
from pil2gl import bn128

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 18
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 100
arity = int(sys.argv[3]) if len(sys.argv) > 3 else 16
custom = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
pil2gl.init(0)
h = 1 << nb
g = torch.Generator(device="cuda"); g.manual_seed(1)
buf = torch.randint(0, 0x7FFFFFFFFFFFFFFF, (h * cols,), dtype=torch.int64, device="cuda", generator=g) % 0xFFFFFFFF00000001
MH = bn128.buildMerkleHash(arity, custom)
MH.merkelize(buf[:cols * 64], cols, 64); torch.cuda.synchronize()      # parameter generation + warm-up
t0 = time.perf_counter()
tree = MH.merkelize(buf, cols, h)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n_el = (cols + 2) // 3
leaf_perms = h * ((n_el + arity - 1) // arity) if cols > 4 else 0
tree_perms = 0
n = h
while n > 1:
    n = (n - 1) // arity + 1
    tree_perms += n
print({"rows": h, "cols": cols, "arity": arity, "custom": custom, "seconds": round(dt, 4), "leaf_perms": leaf_perms, "tree_perms": tree_perms,
       "Mperm_per_s": round((leaf_perms + tree_perms) / dt / 1e6, 3), "root": hex(MH.root(tree))})
