"""One rank of the sharded-proof test (launched by tests/test_parallel.py under torch.distributed.run): every rank runs
pil2gl.parallel.stark_gen_sharded and compares the proof it receives with the proof of the ordinary single-process prove
loop on the same backend -- they must be identical, field by field.
--backend oracle : CPU checker backend over gloo (host logic of the partition: slices, gathers, evaluation owner, openings)
--backend gpu    : the HIP library on cuda:0 for every rank, exchange over gloo"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))

import torch.distributed as dist
import datetime
TIMEOUT = datetime.timedelta(seconds=120)      # a lost peer fails the test within two minutes instead of ten


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="oracle")
    ap.add_argument("--nbits", type=int, default=6)
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--steps", default="9,5,2")
    ap.add_argument("--air", default="fib", help="fib: one witness stage; perm: two (stage 2 = grand-product hint)")
    ap.add_argument("--hashcommits", type=int, default=0, help="starkStruct.hashCommits")
    ap.add_argument("--pg", default="gloo", help="process-group backend; nccl (= RCCL) needs one GPU per rank")
    ap.add_argument("--impols", type=int, default=0, help="fib: intermediate polynomials computed by the prover (fibonacci_air im_pols)")
    ap.add_argument("--boundaries", type=int, default=0, help="fib: constraints on pil2 boundaries (everyFrame / firstRow / lastRow) instead of selector constants")
    ap.add_argument("--shardsetup", type=int, default=0, help="1: the constant tree is split over the ranks too (parallel.build_const_tree_sharded)")
    ap.add_argument("--twice", type=int, default=0, help="1: prove a second time with the same setup (cached tables)")
    a = ap.parse_args()
    if a.pg == "nccl":
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=TIMEOUT)
    else:
        dist.init_process_group("gloo", timeout=TIMEOUT)
    rank = dist.get_rank()
    import gl_oracle as orc
    orc.build(); orc.set_threads(2)
    from pil2gl import stark, parallel
    steps = [int(x) for x in a.steps.split(",")]
    ss = {"nBits": a.nbits, "nBitsExt": steps[0], "nQueries": 8, "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
    if a.hashcommits:
        ss["hashCommits"] = True
    if a.air in ("permref", "permres"):                     # hints in the reference's shape: numerator / denominator are expressions; permres: + subproof values
        info, exprs, _ = stark.permutation_air(ss, max(1, a.pairs // 2), ref_hints="result" if a.air == "permres" else True)
        cm, consts, publics = stark.permutation_trace(a.nbits, copies=max(1, a.pairs // 2))
    elif a.air == "perm":
        info, exprs, _ = stark.permutation_air(ss)
        cm, consts, publics = stark.permutation_trace(a.nbits)
    else:
        info, exprs, _ = stark.fibonacci_air(a.pairs, ss, im_pols=bool(a.impols), boundaries=bool(a.boundaries))
        cm, consts, publics = stark.fibonacci_trace(a.nbits, a.pairs, im_pols=bool(a.impols))
    if a.backend == "gpu":
        be = stark.GpuBackend(0)
    else:
        from stark_backend import OracleBackend
        be = OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    setup_sh = setup
    if a.shardsetup:
        setup_sh = parallel.build_const_tree_sharded(be, consts, info)
        assert list(setup_sh["constRoot"]) == list(setup["constRoot"]), "rank %d: sharded constant tree has another root" % rank
    got = parallel.stark_gen_sharded(be, be.from_host(cm), setup_sh, info, exprs, publics)
    if a.twice:                                             # the rank's tables are kept with the setup (parallel.shard_tables): a second proof reuses them
        assert "_shardTables" in setup_sh
        again = parallel.stark_gen_sharded(be, be.from_host(cm), setup_sh, info, exprs, publics)
        assert again["proof"] == got["proof"], "rank %d: the second proof with the same setup differs" % rank
    want = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    for k in ("challenges", "challengesFRISteps", "queries", "publics"):
        assert got[k] == want[k], "rank %d: %s differ" % (rank, k)
    assert list(got["proof"]) == list(want["proof"])
    for k in [k for k in want["proof"] if k != "fri"]:
        assert got["proof"][k] == want["proof"][k], "rank %d: proof.%s differs" % (rank, k)
    assert got["proof"]["fri"] == want["proof"]["fri"], "rank %d: FRI part of the proof differs" % rank
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
