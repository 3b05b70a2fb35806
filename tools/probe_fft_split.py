"""fft / ifft of a wide matrix under forced pass splits (PIL2GL_NTT_SPLIT, read per call): NBITS, NCOLS, SPLITS="7,7,6;8,6,6;..." """
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl
pil2gl.init(0)
nBits = int(os.environ.get("NBITS", 20)); C = int(os.environ.get("NCOLS", 100))
N = 1 << nBits
torch.manual_seed(0)
src = torch.randint(0, 2**62, (N * C,), dtype=torch.int64, device="cuda")
dst = torch.empty(N * C, dtype=torch.int64, device="cuda")
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for sp in os.environ.get("SPLITS", "-").split(";"):
    if sp == "-":
        os.environ.pop("PIL2GL_NTT_SPLIT", None)
    else:
        os.environ["PIL2GL_NTT_SPLIT"] = sp
    a = min(timeit(lambda: pil2gl.fft(src, C, nBits, dst)) for _ in range(2)); c0 = int(dst.sum()) & (2**64 - 1)
    b = min(timeit(lambda: pil2gl.ifft(src, C, nBits, dst)) for _ in range(2)); c1 = int(dst.sum()) & (2**64 - 1)
    print("2^%d x %d split %-8s fft %.2f ms  ifft %.2f ms   checksums %016x %016x" % (nBits, C, sp, a, b, c0, c1), flush=True)
