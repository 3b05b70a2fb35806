"""constraint evaluation (q_expr) of the config-3 proof with the run-time compiled kernel's switches flipped inside ONE process:
   PIL2GL_EXPR_LAZYMUL x PIL2GL_EXPR_MULCALL x PIL2GL_EXPR_STAGE (COMBOS="l,c,s;...") (the kernels are cached by source, so every combination compiles its own).
   PIL2GL_EXPR_STAGE was removed in round 4 (the staged variants lost: LAB_NOTES 9.5, 10); the third field is ignored by today's library."""
import os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python")); sys.path.insert(0, ROOT)
import pil2gl, bench
from pil2gl import stark
pil2gl.init(0)
dev = torch.device("cuda", 0)
n_bits, n_cols = int(os.environ.get("NBITS", 24)), int(os.environ.get("NCOLS", 100))
air = os.environ.get("AIR", "fib")
ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": False,
      "steps": [{"nBits": b} for b in bench.fri_steps_for(n_bits + 3)]}
be = stark.GpuBackend(0, False)
if air == "perm":                      # the two-stage permutation-check AIR of bench.py --air perm
    copies = max(1, n_cols // 11)
    info, exprs, _ = stark.permutation_air(ss, copies)
    src, consts, publics = bench.permutation_trace_gpu(dev, n_bits, copies)
else:
    info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
    src, consts, publics = bench.fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
setup = stark.build_const_tree(be, consts, info)
roots = set()
for rnd in range(2):
    for lazy, call, stg in (tuple(x.split(",")) for x in os.environ.get("COMBOS", "0,1,0;1,1,0;1,1,1").split(";")):
        os.environ["PIL2GL_EXPR_LAZYMUL"] = lazy; os.environ["PIL2GL_EXPR_MULCALL"] = call; os.environ["PIL2GL_EXPR_STAGE"] = stg
        t = {}
        res = stark.stark_gen(be, src, setup, info, exprs, publics, timings=t)
        roots.add(str(res["proof"]["root2"] if "root2" in res["proof"] else res["proof"].get("rootQ", "")))
        print("round %d lazy=%s call=%s stage=%s: q_expr %.2f ms, fri_expr %.2f ms, stage2_witness %.2f ms" % (rnd, lazy, call, stg, t["q_expr"] * 1e3, t["fri_expr"] * 1e3, t.get("stage2_witness", 0) * 1e3), flush=True)
print("distinct Q roots:", len(roots))
