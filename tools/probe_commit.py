"""one config-3 extend + merkelize (interpolate, linear hash, tree) for rocprofv3 counter passes"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl
pil2gl.init(0)
nBits = int(os.environ.get("NBITS", 24)); C = int(os.environ.get("NCOLS", 100)); eb = 3
N, E = 1 << nBits, 1 << (nBits + eb)
src = torch.randint(0, 2**62, (N * C,), dtype=torch.int64, device="cuda")
dst = torch.empty(E * C, dtype=torch.int64, device="cuda")
MH = pil2gl.buildMerkleHash(False)
for _ in range(int(os.environ.get("REPS", 1))):
    pil2gl.interpolate(src, C, nBits, dst, nBits + eb)
    tree = MH.merkelize(dst, C, E)
torch.cuda.synchronize()
print("root", [hex(int(v)) for v in MH.root(tree)])
