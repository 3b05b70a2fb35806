import os, sys, torch
sys.path.insert(0, "pil2-stark-js_amd/python")
import pil2gl
pil2gl.init(0)
E, C = 1 << 27, 100
dst = torch.empty(E * C, dtype=torch.int64, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
for o in range(0, E * C, 1 << 28):
    m = min(1 << 28, E * C - o)
    dst[o:o + m] = torch.randint(0, 2**62, (m,), dtype=torch.int64, device="cuda", generator=g)
dig = torch.empty(E * 4, dtype=torch.int64, device="cuda")
pil2gl.linearHash(dst, C, False, dig); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
for rnd in range(2):
    s.record(); pil2gl.linearHash(dst, C, False, dig); e.record(); torch.cuda.synchronize()
    print("%s: leaf %.1f ms  checksum %016x" % (os.path.basename(os.environ.get("PIL2GL_LIB", "in-tree")), s.elapsed_time(e), int(dig.sum()) & (2**64-1)), flush=True)
