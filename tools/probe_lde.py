"""one config-3 interpolate (+ an E x 3 fft/ifft pair) for per-launch kernel timing under rocprofv3 --kernel-trace"""
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
for _ in range(2):
    pil2gl.interpolate(src, C, nBits, dst, nBits + eb)
q = dst[:E * 3]
q2 = torch.empty_like(q)
for _ in range(2):
    pil2gl.ifft(q, 3, nBits + eb, q2)
    pil2gl.fft(q2, 3, nBits + eb, q)
torch.cuda.synchronize()
