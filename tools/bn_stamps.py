#!/usr/bin/env python3
"""Where a wave of the matrix-core BN254 leaf kernel spends its cycles: a build with -DBN_STAMPS (lib_ab/libpil2gl_stamps.so, s_memtime around
every phase) runs the 2^20 x 100 arity-16 commit; the per-phase sums over all waves are printed as shares of the kernel's wave time.
  PIL2GL_LIB=pil2-stark-js_amd/lib_ab/libpil2gl_stamps.so python tools/bn_stamps.py [log2 rows] [columns]      (6 columns: one width-3 permutation per row)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pil2-stark-js_amd", "python"))
import torch
import pil2gl
from pil2gl import bn128, _lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pil2gl.init(0)
h = 1 << nb
g = torch.Generator(device="cuda"); g.manual_seed(1)
buf = torch.randint(0, 0x7FFFFFFFFFFFFFFF, (h * cols,), dtype=torch.int64, device="cuda", generator=g) % 0xFFFFFFFF00000001
MH = bn128.buildMerkleHash(16, False)
MH.merkelize(buf[:64 * cols], cols, 64); torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
out = (C.c_uint64 * 16)()
lib.pil2gl_bn128_debug_stamps(out, 1)
digests = torch.empty(h * 4, dtype=torch.int64, device="cuda")
_lib.call("pil2gl_bn128_linear_hash_rows_dev", buf.data_ptr(), cols, h, 16, 0, digests.data_ptr(), None)
torch.cuda.synchronize()
lib.pil2gl_bn128_debug_stamps(out, 0)
names = ["S-box layers (full rounds)", "dense layers", "partial rounds (whole)", "  rows on y (P)", "  rounds (S-box + cross terms + finish)", "  column updates (U)", "", "kernel", "waves"]
tot = out[7]
for i, nm in enumerate(names):
    if nm and i != 8:
        print("%-42s %6.1f %%   %9.0f cycles per wave" % (nm, 100.0 * out[i] / tot, out[i] / out[8]))
print("waves", out[8])
if out[12]:
    print("shader clock inside the kernel: %.2f GHz (s_memtime against the 100 MHz s_memrealtime)" % (out[7] / out[12] * 0.1))
if out[6]:
    print("dense rows (%d per wave): tiles + products issued %.0f cycles per row, carry of both accumulators (incl. the wait for the last product) %.0f, finish + store %.0f"
          % (out[6] / out[8], out[13] / out[6], out[14] / out[6], out[15] / out[6]))
if out[11]:
    print("P loop per column: load+prep %.0f cycles, tiles+products issued %.0f cycles (%d columns)" % (out[9] / out[11], out[10] / out[11], out[11]))
