"""ONE config-3 proof (the bench's default step, nothing else) for rocprofv3 counter passes"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
sys.path.insert(0, ROOT)
import pil2gl
import bench
from pil2gl import stark
pil2gl.init(0)
dev = torch.device("cuda", 0)
n_bits, n_cols = int(os.environ.get("NBITS", 24)), int(os.environ.get("NCOLS", 100))
ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": False,
      "steps": [{"nBits": b} for b in bench.fri_steps_for(n_bits + 3)]}
info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
be = stark.GpuBackend(0, False)
src, consts, publics = bench.fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
setup = stark.build_const_tree(be, consts, info)
for _ in range(int(os.environ.get("REPS", 1))):
    res = stark.stark_gen(be, src, setup, info, exprs, publics)
torch.cuda.synchronize()
print("root1", res["proof"]["root1"])
