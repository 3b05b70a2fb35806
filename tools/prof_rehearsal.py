import sys, os, cProfile, pstats, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pil2-stark-js_amd/python")
import torch, bench
from pil2gl import stark, parallel
n_bits, n_cols, K = 24, 100, 8
ss = {"nBits": n_bits, "nBitsExt": n_bits + 3, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": False, "steps": [{"nBits": b} for b in bench.fri_steps_for(n_bits + 3)]}
info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
be = stark.GpuBackend(0, False)
src, consts, publics = bench.fibonacci_trace_gpu(torch.device("cuda", 0), n_bits, n_cols // 2, 0)
setup = stark.build_const_tree(be, consts, info)
parallel.stark_gen_sharded(be, src, setup, info, exprs, publics, rehearse_world=K)
for _ in range(3):
    st = {}
    t0 = time.perf_counter()
    parallel.stark_gen_sharded(be, src, setup, info, exprs, publics, rehearse_world=K, timings=st)
    torch.cuda.synchronize()
    print("instrumented %.1f ms" % (1e3 * (time.perf_counter() - t0)), {k: round(v * 1e3, 1) for k, v in st.items()}, flush=True)
pr = cProfile.Profile(); pr.enable()
parallel.stark_gen_sharded(be, src, setup, info, exprs, publics, rehearse_world=K)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
