"""Quick GPU timing probe (not the benchmark): LDE / merkelize / poseidon at config-2 shapes."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl

pil2gl.init(0)
print(pil2gl.device_info())
nBits = int(os.environ.get("NBITS", 20)); C = int(os.environ.get("NCOLS", 8)); eb = 3
N, E = 1 << nBits, 1 << (nBits + eb)
g = torch.Generator(device="cuda"); g.manual_seed(1)
src = torch.randint(0, 2**62, (N * C,), dtype=torch.int64, device="cuda", generator=g)
dst = torch.empty(E * C, dtype=torch.int64, device="cuda")
MH = pil2gl.buildMerkleHash(False)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

t = timeit(lambda: pil2gl.interpolate(src, C, nBits, dst, nBits + eb))
print("interpolate 2^%d x %d -> x8: %.3f ms  (%.1f GB/s algorithmic)" % (nBits, C, t, 72 * N * C / t / 1e6))
t = timeit(lambda: MH.merkelize(dst, C, E))
perms = E * ((C + 7) // 8 if C > 4 else 0) + E - 1
print("merkelize %d x %d: %.3f ms  (%.2f Gperm/s)" % (E, C, t, perms / t / 1e6))
lv = torch.empty(E * 4, dtype=torch.int64, device="cuda")
t = timeit(lambda: pil2gl.merkelizeLevel(dst[:E * 8], lv))
print("merkelizeLevel %d ops: %.3f ms (%.2f Gperm/s)" % (E, t, E / t / 1e6))
t = timeit(lambda: pil2gl.fft(dst, C, nBits + eb, dst))
print("fft 2^%d x %d: %.3f ms (%.1f GB/s algorithmic)" % (nBits + eb, C, t, 16 * E * C / t / 1e6))
