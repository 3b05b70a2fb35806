import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pil2-stark-js_amd/python")
import torch, bench
from pil2gl import stark
dev = torch.device("cuda", 0)
n_bits, n_cols = 24, 100
ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": False, "steps": [{"nBits": b} for b in (27, 22, 17, 12, 7)]}
info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
be = stark.GpuBackend(0, False)
src, consts, publics = bench.fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
setup = stark.build_const_tree(be, consts, info)
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tm = {} if i % 2 else None
    stark.stark_gen(be, src, setup, info, exprs, publics, timings=tm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(i, round(dt, 4), round(sum(tm.values()), 4) if tm else None, torch.cuda.memory_reserved() / 1e9)
