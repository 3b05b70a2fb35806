"""leaf hashing and one tree level at config-3 size, for A/B runs of two builds of the library in one gpurun call (PIL2GL_LIB)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pil2-stark-js_amd", "python"))
import torch, pil2gl
pil2gl.init(0)
rows, C = 1 << int(os.environ.get("NBITS", 26)), 100
src = torch.randint(0, 2**62, (rows * C,), dtype=torch.int64, device="cuda")
dig = torch.empty(rows * 4, dtype=torch.int64, device="cuda"); lvl = torch.empty(rows * 2, dtype=torch.int64, device="cuda")
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for rnd in range(2):
    a = t(lambda: pil2gl.linearHash(src, C, False, dig)); b = t(lambda: pil2gl.merkelizeLevel(dig, lvl))
    print("%s: leaves %.2f ms (%.3f G perm/s), level %.2f ms (%.3f G perm/s)" % (os.environ.get("PIL2GL_LIB", "in-tree"), a, rows * 13 / a / 1e6, b, rows / 2 / b / 1e6), flush=True)
