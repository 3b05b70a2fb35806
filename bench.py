#!/usr/bin/env python3
"""bench.py -- STARK prove time and trace-cells/s on MI355X.

One "step" (default --mode prove) = one full proof of a synthetic AIR whose witness is already resident in HBM:
the stage loop of src/prover/prover.js:7-127 -- extendAndMerkelize (LDE blow-up 8 + Poseidon Merkle tree),
constraint polynomial Q (expression evaluation on the extended domain, iNTT/split/NTT, tree), evaluations,
FRI polynomial, FRI folding with trees, query openings -- driven by pil2gl.stark.stark_gen through the C ABI
(libpil2gl.so).  The AIR is K = cols/2 copies of the reference's Fibonacci machine
(test/state_machines/sm_fibonacci/fibonacci.pil), so the trace satisfies its constraints and the proof is a
valid one (tests/test_stark_prove.py verifies such proofs and their bit-identity with the CPU oracle's).
--mode commit times only extendAndMerkelize (stark_gen_helpers.js:388-412) on a uniformly random trace.
Inputs, intermediates, trees and outputs stay on the device (no PCIe in the timed region).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5|NBITSxCOLS]
                  [--mode prove|commit|commit-sharded|prove-sharded] [--shard-of K]

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank runs the same step on its
own trace (weak scaling, no data-path collective); value = all ranks' cells / max-over-ranks time.
--mode commit-sharded instead splits ONE trace's extendAndMerkelize by cosets over the ranks (pil2gl.parallel:
all-gather of leaf digests over RCCL; strong scaling, value = that trace's cells / time); --mode prove-sharded does the
same for ONE whole proof (q, evaluations and the FRI polynomial exchanged as well).  With --shard-of K either sharded mode
runs rank 0's share of a K-GPU job alone on one GPU (per-GPU time and memory; not a contract line).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
WORKLOADS = {"c2": (20, 8), "c3": (24, 100), "c5": (26, 200)}      # BASELINE.json configs[1], configs[2], configs[4] (c5: sharded only)
EXT_BITS = 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("PIL2GL_BENCH_WORKLOAD", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default=os.environ.get("PIL2GL_BENCH_MODE", "prove"), choices=["prove", "commit", "commit-sharded", "prove-sharded"])
    ap.add_argument("--split", action="store_true", help="splitLinearHash leaves (linearhash_gpu.js)")
    ap.add_argument("--full-tree", action="store_true", help="commit-sharded: every rank builds the whole tree above the gathered leaves (default: the tree is split by leaf blocks, pil2gl.parallel.ShardedTree)")
    ap.add_argument("--shard-of", type=int, default=0, help="commit-sharded on ONE GPU: run rank 0's share of a K-GPU job (per-GPU time/memory rehearsal, e.g. --workload c5 --shard-of 8)")
    return ap.parse_args()


def make_trace(n_rows, n_cols, seed, device):
    """uniform canonical Goldilocks elements: hi in [0, 2^32-1), lo in [0, 2^32)  (value < p)"""
    g = torch.Generator(device=device); g.manual_seed(seed)
    n = n_rows * n_cols
    out = torch.empty(n, dtype=torch.int64, device=device)
    chunk = 1 << 26
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        hi = torch.randint(0, 0xFFFFFFFF, (m,), dtype=torch.int64, device=device, generator=g)
        lo = torch.randint(0, 1 << 32, (m,), dtype=torch.int64, device=device, generator=g)
        out[o:o + m] = (hi << 32) | lo
    return out


def fibonacci_trace_gpu(torch, dev, n_bits, n_pairs, rank):
    """witness of K Fibonacci machines (sm_fibonacci.js:12-23), generated on the device (one lane per machine,
    the recurrence is sequential in the row index), plus the constant columns L1/LLAST and the publics"""
    import ctypes as C
    import pil2gl
    N = 1 << n_bits
    rng = np.random.default_rng(0x5EED0000 + rank)
    init = rng.integers(0, 0xFFFFFFFF00000001, size=2 * n_pairs, dtype=np.uint64)      # (l1_k(0), l2_k(0)) pairs
    cm = torch.empty(N * 2 * n_pairs, dtype=torch.int64, device=dev)
    pil2gl.call("pil2gl_synth_fibonacci_dev", n_bits, n_pairs, C.c_void_p(init.ctypes.data), C.c_void_p(cm.data_ptr()),
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    consts = np.zeros((N, 2), dtype=np.uint64); consts[0, 0] = 1; consts[N - 1, 1] = 1
    first = cm[:2].cpu().numpy().view(np.uint64); last = cm[(N - 1) * 2 * n_pairs:(N - 1) * 2 * n_pairs + 1].cpu().numpy().view(np.uint64)
    publics = [int(first[1]), int(first[0]), int(last[0])]
    return cm, consts, publics


def ev_time(fn, iters):
    """average ms of fn() on torch's current stream (the stream the library launches on), HIP events"""
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def cpu_baseline_prove(n_cols, split, fri_delta=5):
    """the same full proof by the prove loop over the CPU oracle backend (C/OpenMP port) on a bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gl_oracle
    from stark_backend import OracleBackend
    from pil2gl import stark
    gl_oracle.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("PIL2GL_CPU_THREADS", "16")))
    gl_oracle.set_threads(cores)
    n_bits = 11 if n_cols > 16 else 14

    def run(nb):
        steps = [nb + EXT_BITS]
        while steps[-1] > 10:
            steps.append(max(steps[-1] - fri_delta, 6))
        ss = {"nBits": nb, "nBitsExt": nb + EXT_BITS, "nQueries": 64, "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
        info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
        cm, consts, publics = stark.fibonacci_trace(nb, n_cols // 2)
        be = OracleBackend(split)
        setup = stark.build_const_tree(be, consts, info)
        t0 = time.perf_counter()
        stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
        return time.perf_counter() - t0
    t = run(n_bits)
    while t < 8.0 and n_bits < 19:             # the reported sample is the last run: 8-16 s of CPU work
        n_bits += 1
        t = run(n_bits)
    cells = (1 << n_bits) * n_cols
    return {"value": cells / t, "unit": "trace-cells/s", "cores": cores, "kind": "port",
            "sample": "full proof of 2^%d x %d Fibonacci AIR, blow-up 8, prove loop over the OpenMP C oracle backend, %.1f s" % (n_bits, n_cols, t)}


def cpu_baseline(n_cols, split):
    """extend+merkelize by the CPU oracle (a C/OpenMP port of the reference algorithms) on a bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gl_oracle
    gl_oracle.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("PIL2GL_CPU_THREADS", "16")))      # the GPU box grants a 16-CPU share per GPU
    gl_oracle.set_threads(cores)
    n_bits = 14 if n_cols > 16 else 17
    rng = np.random.default_rng(1)

    def run(nb):
        a = rng.integers(0, 0xFFFFFFFF00000001, size=(1 << nb, n_cols), dtype=np.uint64)
        t0 = time.perf_counter()
        e = gl_oracle.interpolate(a, nb, nb + EXT_BITS)
        gl_oracle.merkelize(e, split)
        return time.perf_counter() - t0
    t = run(n_bits)
    while t < 4.0 and n_bits < 20:          # grow the sample until it is a few seconds of CPU work
        n_bits += 1
        t = run(n_bits)
    cells = (1 << n_bits) * n_cols
    return {"value": cells / t, "unit": "trace-cells/s", "cores": cores, "kind": "port",
            "sample": "extend+merkelize of 2^%d x %d random trace, blow-up 8, OpenMP C oracle, %.1f s" % (n_bits, n_cols, t)}


def load_pmc_lde_traffic():
    """HBM bytes of ONE interpolate from the committed PMC summary: all ntt_pass_kernel + lde_mid_kernel launches of the
    profiled run divided by the number of LDEs in it (one lde_mid launch each)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            sb = json.load(f)["_sum_bytes"]
        n = sb["lde_mid_kernel"]["launches"]
        return int((sb["lde_mid_kernel"]["bytes"] + sb["ntt_pass_kernel"]["bytes"]) / n)
    except Exception:
        return None


def load_pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (profiles/pmc_traffic.json), if any"""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(kernel)
    except Exception:
        return None


def rehearse_shard(args):
    """--mode commit-sharded --shard-of K on one GPU: rank 0's share of a K-GPU sharded commit (its cosets of the LDE in the
    trace's own memory, its leaves, a tree over stand-in digests), to show the per-GPU time and memory of e.g. config 5."""
    import pil2gl
    from pil2gl import stark, parallel
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n_bits, n_cols = WORKLOADS[args.workload] if args.workload in WORKLOADS else (int(v) for v in args.workload.lower().split("x"))
    K = args.shard_of
    be = stark.GpuBackend(0, args.split)
    N = 1 << n_bits
    src = make_trace(N, n_cols, 0x5EED0000, dev)
    times = []
    for i in range(args.warmup + args.steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = parallel.extend_and_merkelize_sharded(be, src, n_cols, n_bits, n_bits + EXT_BITS, overwrite_src=True, rehearse_world=K, split_tree=not args.full_tree)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if i >= args.warmup:
            times.append(dt)
        del st
    dt = sum(times) / len(times)
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({"metric": "per-GPU time of a %d-GPU coset-sharded commit (rank 0's share run alone; digests of the other ranks stood in)" % K,
                      "value": N * n_cols / dt, "unit": "trace-cells/s (whole trace / per-GPU time: the job rate if the %d ranks run in parallel)" % K,
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
                      "dtype": "u64", "data": "synthetic",
                      "config": {"workload": "extendAndMerkelize 2^%d rows x %d cols -> 2^%d rows, %d of %d cosets on this GPU" % (n_bits, n_cols, n_bits + EXT_BITS, (1 << EXT_BITS) // K, 1 << EXT_BITS),
                                 "mode": "commit-sharded rehearsal", "shard_of": K, "tree": "full on every rank" if args.full_tree else "split by leaf blocks"},
                      "peak_torch_GB": torch.cuda.max_memory_allocated() / 1e9, "device_GB_in_use_at_end": (total - free) / 1e9}), flush=True)


def rehearse_prove(args):
    """--mode prove-sharded --shard-of K on one GPU: rank 0's share of a K-GPU sharded proof (own slices standing in for the
    gathered q / FRI polynomial / digests; the evaluations are computed here anyway because rank 0 owns coset 0)"""
    import pil2gl
    from pil2gl import stark, parallel
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n_bits, n_cols = WORKLOADS[args.workload] if args.workload in WORKLOADS else (int(v) for v in args.workload.lower().split("x"))
    n_cols -= n_cols & 1
    K = args.shard_of
    fri_steps = [n_bits + EXT_BITS]
    while fri_steps[-1] > 10:
        fri_steps.append(max(fri_steps[-1] - 5, 6))
    ss = {"nBits": n_bits, "nBitsExt": n_bits + EXT_BITS, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": bool(args.split),
          "steps": [{"nBits": b} for b in fri_steps]}
    info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
    be = stark.GpuBackend(0, args.split)
    src, consts, publics = fibonacci_trace_gpu(torch, dev, n_bits, n_cols // 2, 0)
    setup = stark.build_const_tree(be, consts, info)
    times = []
    for i in range(args.warmup + args.steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        parallel.stark_gen_sharded(be, src, setup, info, exprs, publics, rehearse_world=K)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if i >= args.warmup:
            times.append(dt)
    dt = sum(times) / len(times)
    stages = {}
    parallel.stark_gen_sharded(be, src, setup, info, exprs, publics, rehearse_world=K, timings=stages)      # one more, instrumented
    print(json.dumps({"metric": "per-GPU time of ONE proof split over %d GPUs (rank 0's share run alone, exchanges stood in)" % K,
                      "value": (1 << n_bits) * n_cols / dt, "unit": "trace-cells/s (the job rate if the %d ranks run in parallel and the exchanges are free)" % K,
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "u64", "data": "synthetic",
                      "config": {"workload": "full proof of 2^%d rows x %d cols, blow-up 8, %d of %d cosets on this GPU" % (n_bits, n_cols, (1 << EXT_BITS) // K, 1 << EXT_BITS),
                                 "mode": "prove-sharded rehearsal", "shard_of": K}, "stages_s": {k: round(v, 4) for k, v in stages.items()},
                      "peak_torch_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)


def main():
    args = parse()
    if args.mode == "commit-sharded" and args.shard_of:
        return rehearse_shard(args)
    if args.mode == "prove-sharded" and args.shard_of:
        return rehearse_prove(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank %= max(1, torch.cuda.device_count())            # rehearsing N ranks on fewer GPUs: ranks share devices
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or (args.mode in ("commit-sharded", "prove-sharded") and not args.shard_of):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("PIL2GL_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path with ranks sharing a GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import pil2gl
    pil2gl.init(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = args.workload
    if wl == "auto":
        free, _ = torch.cuda.mem_get_info()
        wl = "c3" if free > 170e9 else "c2"
    if wl in WORKLOADS:
        n_bits, n_cols = WORKLOADS[wl]
    else:
        n_bits, n_cols = (int(v) for v in wl.lower().split("x"))
    N, E = 1 << n_bits, 1 << (n_bits + EXT_BITS)

    prove_ctx = None
    prove_sharded = args.mode == "prove-sharded"
    if args.mode in ("prove", "prove-sharded"):
        from pil2gl import stark
        n_cols -= n_cols & 1                                   # pairs of columns
        fri_steps = [n_bits + EXT_BITS]
        while fri_steps[-1] > 10:                              # decreasing by <= 5 bits, as zkevm.starkstruct.json does
            fri_steps.append(max(fri_steps[-1] - 5, 6))
        ss = {"nBits": n_bits, "nBitsExt": n_bits + EXT_BITS, "nQueries": 64, "verificationHashType": "GL",
              "splitLinearHash": bool(args.split), "steps": [{"nBits": b} for b in fri_steps]}
        info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
        be = stark.GpuBackend(local_rank, args.split)
        src, consts, publics = fibonacci_trace_gpu(torch, dev, n_bits, n_cols // 2, 0 if prove_sharded else rank)   # sharded: ONE trace
        setup = stark.build_const_tree(be, consts, info)
        prove_ctx = (stark, be, setup, info, exprs, publics)
    elif args.mode == "commit-sharded":                        # ONE trace, replicated; the cosets of its extension are split
        from pil2gl import stark, parallel
        if (1 << EXT_BITS) % (args.shard_of or world):
            raise SystemExit("commit-sharded needs a world size dividing %d" % (1 << EXT_BITS))
        shard_be = stark.GpuBackend(local_rank, args.split)
        src = make_trace(N, n_cols, 0x5EED0000, dev)
    else:
        src = make_trace(N, n_cols, 0x5EED0000 + rank, dev)
    sharded = args.mode == "commit-sharded"
    dst = torch.empty(E * n_cols, dtype=torch.int64, device=dev)
    MH = pil2gl.buildMerkleHash(args.split)
    nodes = torch.empty(MH._getNNodes(E * 4), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    import ctypes as C

    stage_times = {}

    def step():
        if prove_ctx is not None:
            stark_, be_, setup_, info_, exprs_, publics_ = prove_ctx
            if prove_sharded:
                from pil2gl import parallel
                parallel.stark_gen_sharded(be_, src, setup_, info_, exprs_, publics_)
                return
            stark_.stark_gen(be_, src, setup_, info_, exprs_, publics_, timings=stage_times if collect[0] else None)
            return
        if sharded:
            parallel.extend_and_merkelize_sharded(shard_be, src, n_cols, n_bits, n_bits + EXT_BITS,
                                                  overwrite_src=bool(args.shard_of), rehearse_world=args.shard_of or None, split_tree=not args.full_tree)
            return
        pil2gl.interpolate(src, n_cols, n_bits, dst, n_bits + EXT_BITS)
        pil2gl.call("pil2gl_merkelize_dev", pil2gl._ptr(dst), n_cols, E, int(args.split), pil2gl._ptr(nodes), C.c_void_p(stream))
    collect = [False]
    if prove_ctx is not None or sharded:
        del dst, nodes                                         # the prove loop allocates its own buffers
        dst = nodes = None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not a timed or counted step: on a box that has just been handed out the first pass after process start has
    # been seen ~180 ms (13 %) slower than the following ones (clock ramp, first touch of 150 GB); one untimed pass ahead of the
    # W warm-up steps keeps that out of a short run's mean.  PIL2GL_BENCH_PREWARM=0 skips it.
    if os.environ.get("PIL2GL_BENCH_PREWARM", "1") != "0":
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        step()
        step_marks.append(time.perf_counter())               # host clock only: no extra synchronisation inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    step_ms = [round((b - a) * 1e3, 2) for a, b in zip([t0] + step_marks[:-1], step_marks)]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = (1 if (sharded or prove_sharded) else world) * N * n_cols / (dt / args.steps)
    if (sharded or prove_sharded) and rank == 0:
        dst = torch.empty(E * n_cols, dtype=torch.int64, device=dev)
    if prove_ctx is not None and rank == 0 and not prove_sharded:   # one more, untimed-for-value, proof with a per-stage breakdown
        collect[0] = True
        step(); torch.cuda.synchronize()
        collect[0] = False
        dst = torch.empty(E * n_cols, dtype=torch.int64, device=dev)

    out = None
    if rank == 0:
        # ---- per-kernel timing with HIP events on the launch stream (each call below is exactly one kernel,
        #      or one kernel family named in DESIGN.md) ----
        iters = max(1, min(3, args.steps))
        digests = torch.empty(E * 4, dtype=torch.int64, device=dev)
        t_leaf = ev_time(lambda: pil2gl.linearHash(dst, n_cols, args.split, digests), iters)
        t_lde = ev_time(lambda: pil2gl.interpolate(src, n_cols, n_bits, dst, n_bits + EXT_BITS), iters)
        lvl = torch.empty(E * 2, dtype=torch.int64, device=dev)
        t_lvl = ev_time(lambda: pil2gl.merkelizeLevel(digests, lvl), iters)
        leaf_perms = E * ((n_cols + 7) // 8) if n_cols > 4 else 0
        if args.split and n_cols > 4:
            batch = max(8, (n_cols + 3) // 4); nb = (n_cols + batch - 1) // batch
            leaf_perms = E * (sum((min(batch, n_cols - b * batch) + 7) // 8 if min(batch, n_cols - b * batch) > 4 else 0 for b in range(nb)) + (((4 * nb) + 7) // 8 if nb > 1 else 0))
        kernels = [
            {"kernel": "linear_hash_kernel", "ms": t_leaf, "alg_bytes": 8 * E * n_cols + 32 * E, "perms": leaf_perms},
            {"kernel": "interpolate (ntt_pass_kernel x / lde_mid_kernel)", "ms": t_lde, "alg_bytes": 8 * N * n_cols * (1 + (1 << EXT_BITS))},
            {"kernel": "merkle_level_kernel (first level)", "ms": t_lvl, "alg_bytes": 32 * E + 16 * E, "perms": E // 2},
        ]
        if wl == "c3":                                          # the committed PMC passes were taken at config 3
            kernels[0]["traffic"] = load_pmc_traffic("linear_hash_kernel")
            kernels[1]["traffic"] = load_pmc_lde_traffic()
            kernels[2]["traffic"] = load_pmc_traffic("merkle_level_kernel")
        for k in kernels:
            k["GBps"] = k["alg_bytes"] / k["ms"] / 1e6
            k["hbm_frac"] = k["GBps"] / HBM_PEAK_GBS
            if "perms" in k:
                k["Gperm_s"] = k["perms"] / k["ms"] / 1e6
        dom = max(kernels, key=lambda k: k["ms"])
        roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": dom["hbm_frac"], "traffic": load_pmc_traffic(dom["kernel"].split(" ")[0]),
                    "note": "Poseidon hashing is integer-ALU bound (no 64-bit multiplier on gfx950); its HBM fraction is small by nature, see kernels[]"}
        # what does bind it: the committed SQ counters of the same kernel (profiles/r01_valu_utilisation.json, tools/pmc_valu.py):
        # share of a wave's cycles spent issuing vector-ALU instructions x waves per SIMD ~ share of the SIMD's issue slots
        try:
            with open(os.path.join(ROOT, "profiles", "r01_valu_utilisation.json")) as f:
                pm = json.load(f).get(dom["kernel"].split(" ")[0])
            if pm:
                roofline["valu"] = {"active_frac_of_wave_cycles": round(pm["valu_frac_of_wave_cycles"], 3), "wait_frac": round(pm["wait_any_frac"], 3),
                                    "source": "profiles/r01_valu_utilisation.json (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, rocprofv3 --pmc, config 3)"}
        except Exception:
            pass
        if prove_ctx is not None:
            metric = "STARK prove time (ms_per_step) and trace-cells/s, synthetic Fibonacci AIR, GL Poseidon Merkle + FRI, blow-up 8"
            workload = "full proof (commit, Q, evals, FRI %s, %d queries) of 2^%d rows x %d cols -> 2^%d rows, %s linear hash, %s" % (
                "/".join(str(x["nBits"]) for x in prove_ctx[3]["starkStruct"]["steps"]), prove_ctx[3]["starkStruct"]["nQueries"], n_bits, n_cols, n_bits + EXT_BITS, "split" if args.split else "plain",
                "ONE proof split by cosets over the GPUs (q, evaluations, FRI polynomial exchanged)" if prove_sharded else "per GPU")
        elif sharded:
            metric = "trace-cells/s, STARK commit step (extend+merkelize) of ONE trace split by cosets over the GPUs, GL Poseidon Merkle, blow-up 8"
            workload = "extendAndMerkelize 2^%d rows x %d cols -> 2^%d rows, %s linear hash, %d of 8 cosets per GPU + all-gather of leaf digests" % (
                n_bits, n_cols, n_bits + EXT_BITS, "split" if args.split else "plain", (1 << EXT_BITS) // world)
        else:
            metric = "trace-cells/s, STARK commit step (extend+merkelize), GL Poseidon Merkle, blow-up 8"
            workload = "extendAndMerkelize 2^%d rows x %d cols -> 2^%d rows, %s linear hash, per GPU" % (n_bits, n_cols, n_bits + EXT_BITS, "split" if args.split else "plain")
        out = {
            "metric": metric,
            "value": value, "unit": "trace-cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if (sharded or prove_sharded) else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload, "mode": args.mode,
                       "nBits": n_bits, "nCols": n_cols, "nBitsExt": n_bits + EXT_BITS, "hash": "GL-Poseidon-12",
                       "parallelism": ("coset-sharded x%d" % world) if (sharded or prove_sharded) else ("replicas x%d" % world if world > 1 else "single GPU")},
            "roofline": roofline, "kernels": kernels,
        }
        if prove_ctx is not None:
            out["prove"] = {"seconds": ms_per_step / 1e3, "stages_s": {k: round(v, 4) for k, v in stage_times.items()}, "host_ms_of_each_step": step_ms}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_prove(n_cols, args.split) if prove_ctx is not None else cpu_baseline(n_cols, args.split)
            out["speedup_vs_cpu_port"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
