"""config-3 interpolate timed for one build of the library (PIL2GL_LIB): same-call A/B of kernel variants"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl
pil2gl.init(0)
nBits = int(os.environ.get("NBITS", 24)); C = int(os.environ.get("NCOLS", 100)); eb = 3
N, E = 1 << nBits, 1 << (nBits + eb)
torch.manual_seed(0)
src = torch.randint(0, 2**62, (N * C,), dtype=torch.int64, device="cuda")
dst = torch.empty(E * C, dtype=torch.int64, device="cuda")
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
torch.manual_seed(1)
for rnd in range(2):
    print("%s %s: 2^%d x %d interpolate %.2f ms" % (os.path.basename(os.environ.get("PIL2GL_LIB", "in-tree")), os.environ.get("TAG", ""), nBits, C, timeit(lambda: pil2gl.interpolate(src, C, nBits, dst, nBits + eb))), flush=True)
# a checksum of the result, so that builds and geometries timed in one call can be seen to agree (wrap-around sums of the words and of word * index)
s0 = s1 = 0
for o in range(0, dst.numel(), 1 << 28):          # in pieces: the weighted sum's temporaries are as large as what they are taken of
    d = dst[o:o + (1 << 28)]
    w = torch.arange(o, o + d.numel(), dtype=torch.int64, device="cuda")
    s0 += int(d.sum()); s1 += int((d * (2 * w + 1)).sum())
print("   checksum %016x %016x" % (s0 & (2**64 - 1), s1 & (2**64 - 1)), flush=True)
