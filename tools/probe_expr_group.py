"""constraint evaluation (q_expr) of the config-3 proof with the run-time compiled kernel's section loads issued G columns at a time
(PIL2GL_EXPR_GROUP = 0: at first use, as shipped; 4, 8, 16, 32), all inside ONE process; the quotient root must not move.
HISTORICAL: the switch was removed after this measurement (profiles/r04_jit_eval_grouped_loads.txt); build the library of commit a1d7ce2
(PIL2GL_LIB) to repeat it -- with today's library every setting runs the shipped kernel."""
import os, sys
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
if air == "perm":
    copies = max(1, n_cols // 11)
    info, exprs, _ = stark.permutation_air(ss, copies)
    src, consts, publics = bench.permutation_trace_gpu(dev, n_bits, copies)
else:
    info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
    src, consts, publics = bench.fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
setup = stark.build_const_tree(be, consts, info)
os.environ["PIL2GL_JIT_INFO"] = "1"
roots = set()
for rnd in range(2):
    for g in os.environ.get("GROUPS", "0,4,8,16,32").split(","):
        os.environ["PIL2GL_EXPR_GROUP"] = g
        t = {}
        res = stark.stark_gen(be, src, setup, info, exprs, publics, timings=t)
        roots.add(str(res["proof"]["root%d" % (info["nStages"] + 1)]))
        print("round %d group=%s: q_expr %.2f ms, stage2_witness %.2f ms, proof %.1f ms" % (rnd, g, t["q_expr"] * 1e3, t.get("stage2_witness", 0) * 1e3, sum(t.values()) * 1e3), flush=True)
    os.environ["PIL2GL_JIT_INFO"] = "0"
print("distinct quotient roots:", len(roots))
