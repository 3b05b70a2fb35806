"""config-3 interpolate timed with the library's planner switches flipped inside ONE process (boxes differ by several percent,
so only same-call A/B figures are comparable).  usage: python tools/probe_lde_ab.py VAR=a,b [VAR2=c,d ...]"""
import os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl
pil2gl.init(0)
nBits = int(os.environ.get("NBITS", 24)); C = int(os.environ.get("NCOLS", 100)); eb = 3
N, E = 1 << nBits, 1 << (nBits + eb)
src = torch.randint(0, 2**62, (N * C,), dtype=torch.int64, device="cuda")
dst = torch.empty(E * C, dtype=torch.int64, device="cuda")
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
ref = None
for rnd in range(2):
    for combo in itertools.product(*[v for _, v in axes]):
        for (name, _), val in zip(axes, combo):
            os.environ[name] = val
        t = timeit(lambda: pil2gl.interpolate(src, C, nBits, dst, nBits + eb))
        chk = int(dst[::1000003].sum().item())
        if ref is None: ref = chk
        print("round %d %s: %.2f ms  %s" % (rnd, " ".join("%s=%s" % (n, v) for (n, _), v in zip(axes, combo)), t, "same result" if chk == ref else "RESULT DIFFERS"), flush=True)
