"""interpolate timing sweep over planner knobs (env read per call by the library)"""
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
def timeit(fn, n=2):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for tile, kmax, ldet, ldeth in [(8192, 10, 8192, 1024), (4096, 10, 8192, 1024), (2048, 10, 8192, 1024), (8192, 8, 8192, 1024), (8192, 10, 4096, 1024), (8192, 10, 4096, 512), (4096, 10, 4096, 512), (4096, 10, 2048, 256), (8192, 9, 8192, 1024), (8192, 6, 8192, 1024)]:
    os.environ["PIL2GL_NTT_TILE"] = str(tile); os.environ["PIL2GL_NTT_KMAX"] = str(kmax)
    os.environ["PIL2GL_LDE_TILE"] = str(ldet); os.environ["PIL2GL_LDE_THREADS"] = str(ldeth)
    try:
        t = timeit(lambda: pil2gl.interpolate(src, C, nBits, dst, nBits + eb))
        print("tile=%d kmax=%d lde_tile=%d lde_thr=%d: %.2f ms (%.0f GB/s alg)" % (tile, kmax, ldet, ldeth, t, 72 * N * C / t / 1e6), flush=True)
    except Exception as ex:
        print("tile=%d kmax=%d lde_tile=%d lde_thr=%d: FAIL %s" % (tile, kmax, ldet, ldeth, ex), flush=True)
