"""interpolateCosets (a rank's share of an extension) timed under the planner's environment switches (PIL2GL_NTT_KMAX, PIL2GL_NTT_TILE):
NBITS, NCOLS, COSETS (count, from coset 0), EXT (3)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl
pil2gl.init(0)
nBits = int(os.environ.get("NBITS", 26)); C = int(os.environ.get("NCOLS", 32)); eb = int(os.environ.get("EXT", 3)); cc = int(os.environ.get("COSETS", 1))
N = 1 << nBits
torch.manual_seed(0)
src = torch.randint(0, 2**62, (N * C,), dtype=torch.int64, device="cuda")
dst = torch.empty(N * C * cc, dtype=torch.int64, device="cuda")
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
t = min(timeit(lambda: pil2gl.interpolateCosets(src, C, nBits, dst, nBits + eb, 0, cc, None)) for _ in range(2))
s0 = s1 = 0
for o in range(0, dst.numel(), 1 << 28):
    d = dst[o:o + (1 << 28)]
    w = torch.arange(o, o + d.numel(), dtype=torch.int64, device="cuda")
    s0 += int(d.sum()); s1 += int((d * (2 * w + 1)).sum())
print("KMAX %s TILE %s: 2^%d x %d, %d of %d cosets: %.2f ms   checksum %016x %016x" % (os.environ.get("PIL2GL_NTT_KMAX", "-"), os.environ.get("PIL2GL_NTT_TILE", "-"), nBits, C, cc, 1 << eb, t,
                                                                                       s0 & (2**64 - 1), s1 & (2**64 - 1)), flush=True)
