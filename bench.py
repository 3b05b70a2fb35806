#!/usr/bin/env python3
"""bench.py -- STARK prove time and trace-cells/s on MI355X.

One "step" = one pass of the hot path over one synthetic input that is already resident in HBM.

  N = 1 (default)   --mode prove: one full proof -- the stage loop of src/prover/prover.js:7-127: extendAndMerkelize (LDE
                    blow-up 8 + Poseidon Merkle tree), constraint polynomial Q (expression evaluation on the extended
                    domain, iNTT/split/NTT, tree), evaluations, FRI polynomial, FRI folding with trees, query openings --
                    driven by pil2gl.stark.stark_gen through the C ABI (libpil2gl.so), at config 3 of BASELINE.json
                    (2^24 rows x 100 cols, GL Poseidon + FRI).  The AIR is K = cols/2 copies of the reference's Fibonacci
                    machine (test/state_machines/sm_fibonacci/fibonacci.pil): a valid trace, so the proof verifies.
  N > 1 (default)   --mode prove-sharded: ONE proof of the same config-3 trace, split by cosets over the N ranks
                    (pil2gl.parallel): strong scaling, value = that trace's cells / max-over-ranks time.
                    Exchanges (leaf digests, q, FRI polynomial, a few sums) go over RCCL when every rank has its own GPU
                    and through HIP IPC windows when ranks share one (one-GPU rehearsal).
  --mode commit / commit-sharded: only extendAndMerkelize (stark_gen_helpers.js:388-412) on a uniformly random trace;
                    commit-sharded on 8 ranks takes config 5 (2^26 x 200), which only fits sharded.
  --replicas        N > 1: every rank proves its own trace (weak scaling, no data-path exchange) -- the proving-farm mode.
  --workload c4     BN128 Poseidon (arity 16) commit of the config-4 trace; --workload c2: config 2.
  --shard-of K      (with a sharded mode, one GPU) rank 0's share of a K-rank job run alone; not a contract line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|c2|c3|c4|c5|NBITSxCOLS] [--mode ...]

`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run, 127.0.0.1) and relays
rank 0's JSON line; under torchrun WORLD_SIZE must equal --gpus.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_HZ = 2.4e9             # MI355X_MICROARCH.md: max clock (the clock held under an all-integer load is lower: conservative)
N_SIMD = 256 * 4
WORKLOADS = {"c2": (20, 8), "c3": (24, 100), "c4": (24, 100), "c5": (26, 200)}      # BASELINE.json configs[1..4] (c5: sharded only)
EXT_BITS = 3
# measured issue cost of the instructions the integer floor is priced in (tools/microbench.hip, tools/mfma_mds.hip, MI355X)
CYC_MAD_U64_U32 = 4.5
CYC_MFMA_ISSUE = 8.0
CYC_CARRY_OP = 4.4           # v_addc_co_u32 / v_subb_co_u32 (measured, twice a plain 32-bit operation's share)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("PIL2GL_BENCH_WORKLOAD", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default=os.environ.get("PIL2GL_BENCH_MODE", "auto"),
                    choices=["auto", "prove", "commit", "commit-sharded", "prove-sharded"])
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent proofs per rank (weak scaling) instead of ONE sharded proof")
    ap.add_argument("--split", action="store_true", help="splitLinearHash leaves (linearhash_gpu.js)")
    ap.add_argument("--full-tree", action="store_true", help="commit-sharded: every rank builds the whole tree above the gathered leaves (default: the tree is split by leaf blocks, pil2gl.parallel.ShardedTree)")
    ap.add_argument("--air", default="fib", choices=["fib", "perm"], help="prove modes: fib = C/2 Fibonacci machines, one witness stage (the headline line); "
                    "perm = C/11 permutation checks, TWO witness stages with grand-product hints (2 + 9 base columns each: polutils.js:105-164, hints_helpers.js:81-123)")
    ap.add_argument("--shard-of", type=int, default=0, help="a sharded mode on ONE GPU: run rank 0's share of a K-GPU job (per-GPU time/memory rehearsal, e.g. --workload c5 --shard-of 8)")
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


def fibonacci_trace_gpu(dev, n_bits, n_pairs, rank):
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


def permutation_trace_gpu(dev, n_bits, copies, seed=0x5EED0000):
    """stage-1 witness of `copies` permutation checks on the device: a_k uniform, b_k[i] = a_k[(s_k i + o_k) mod N] (s_k odd),
    row-major N x 2K like pil2gl.stark.permutation_trace; plus the constants L1 / LLAST"""
    N = 1 << n_bits
    cm = torch.empty((N, 2 * copies), dtype=torch.int64, device=dev)
    i = torch.arange(N, dtype=torch.int64, device=dev)
    for k in range(copies):
        a = make_trace(N, 1, seed + k, dev)
        cm[:, 2 * k] = a
        cm[:, 2 * k + 1] = a[(i * (5 + 2 * k) + 3 + k) & (N - 1)]
    consts = np.zeros((N, 2), dtype=np.uint64); consts[0, 0] = 1; consts[N - 1, 1] = 1
    return cm.reshape(-1), consts, []


def ev_time(fn, iters):
    """average ms of fn() on torch's current stream (the stream the library launches on), HIP events"""
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def host_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(cores, int(os.environ.get("PIL2GL_CPU_THREADS", "16")))      # the GPU box grants a 16-CPU share per GPU


def port_calibration():
    """how the C/OpenMP port relates to the reference's own JavaScript (measured in the build container, one thread, on the
    reference's dependency-free twins of the path: oracle/calibrate_ref.js, oracle/calibrate_port.py)"""
    for name in ("r04_cpu_port_vs_reference_js.json", "r01_cpu_port_vs_reference_js.json"):       # the latest measurement that is there
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                c = json.load(f)
            return {"source": "profiles/%s (build container, 1 thread)" % name, "data": c}
        except Exception:
            continue
    return None


def fri_steps_for(n_bits_ext):
    steps = [n_bits_ext]
    while steps[-1] > 10:                              # decreasing by <= 5 bits, as zkevm.starkstruct.json does
        steps.append(max(steps[-1] - 5, 6))
    return steps


def cpu_baseline_prove(n_cols, split, air="fib"):
    """the same full proof by the prove loop over the CPU oracle backend (C/OpenMP port) on a bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gl_oracle
    from stark_backend import OracleBackend
    from pil2gl import stark
    gl_oracle.build()
    cores = host_cores()
    gl_oracle.set_threads(cores)
    n_bits = 11 if n_cols > 16 else 14

    def run(nb):
        ss = {"nBits": nb, "nBitsExt": nb + EXT_BITS, "nQueries": 64, "verificationHashType": "GL", "steps": [{"nBits": b} for b in fri_steps_for(nb + EXT_BITS)]}
        if air == "perm":
            info, exprs, _ = stark.permutation_air(ss, max(1, n_cols // 11))
            cm, consts, publics = stark.permutation_trace(nb, copies=max(1, n_cols // 11))
        else:
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
            "sample": "full proof of 2^%d x %d %s AIR, blow-up 8, prove loop over the OpenMP C oracle backend, %.1f s" % (
                n_bits, n_cols, "two-stage permutation-check" if air == "perm" else "Fibonacci", t),
            "port_vs_reference_js": port_calibration()}


def cpu_baseline_commit(n_cols, split):
    """extend+merkelize by the CPU oracle (a C/OpenMP port of the reference algorithms) on a bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gl_oracle
    gl_oracle.build()
    cores = host_cores()
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
            "sample": "extend+merkelize of 2^%d x %d random trace, blow-up 8, OpenMP C oracle, %.1f s" % (n_bits, n_cols, t),
            "port_vs_reference_js": port_calibration()}


def cpu_baseline_bn128(n_cols, arity):
    """config 4's commit on the CPU: LDE by the C port of the reference's algorithm, BN128 tree by the C port of the BN254
    permutation (oracle/bn128_oracle.c: 4 x 64-bit Montgomery arithmetic, OpenMP over rows / nodes; checked against the
    Python-integer oracle, which the reference's constants and its `test/final` proof pin) on all host threads, bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gl_oracle
    import bn128_oracle
    gl_oracle.build(); gl_oracle.set_threads(host_cores())
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))
    rng = np.random.default_rng(1)
    nb = 9

    def run(nb):
        a = rng.integers(0, 0xFFFFFFFF00000001, size=(1 << nb, n_cols), dtype=np.uint64)
        t0 = time.perf_counter()
        e = gl_oracle.interpolate(a, nb, nb + EXT_BITS)
        bn128_oracle.c_merkelize_words(e, arity, False)
        return time.perf_counter() - t0
    run(6)                                                     # builds the library, generates the constants (Grain LFSR, Python)
    t = run(nb)
    while t < 8.0 and nb < 16:
        nb += 1
        t = run(nb)
    return {"value": (1 << nb) * n_cols / t, "unit": "trace-cells/s", "cores": host_cores(), "kind": "port",
            "sample": "extend (C port) + BN128 arity-%d merkelize (C port of the BN254 permutation, OpenMP) of 2^%d x %d, blow-up 8, %.1f s" % (arity, nb, n_cols, t)}


def load_pmc(name, kernel):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary under profiles/, if any"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f).get(kernel)
    except Exception:
        return None


def load_pmc_lde_traffic(name):
    """HBM bytes of ONE interpolate from a committed PMC summary: all NTT-family launches of the profiled run divided by
    the number of LDEs in it"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        if "_interpolate_bytes" in d:
            return int(d["_interpolate_bytes"])
        sb = d["_sum_bytes"]
        n = sb["lde_mid_kernel"]["launches"]
        return int((sb["lde_mid_kernel"]["bytes"] + sb["ntt_pass_kernel"]["bytes"]) / n)
    except Exception:
        return None


def clock_under_hash_load():
    """shader clock (MHz: median, 5th, 95th percentile over the workgroups) while the chip runs Poseidon permutations
    (pil2gl_selftest_clock: shader-clock counter against the 100 MHz counter inside the kernel).  sysfs reports the nominal DPM level
    (2.4 GHz) throughout; under this load the MI355X runs near 1.9 GHz, and SIMD cycles priced at the nominal clock overstate what a
    kernel spent by a quarter."""
    try:
        import ctypes as C
        from pil2gl import _lib
        mhz = (C.c_double * 3)()
        _lib.call("pil2gl_selftest_clock", C.c_uint32(40), mhz)
        return [round(x, 1) for x in mhz]
    except Exception:
        return None


class BoardPower:
    """board power (W) while the timed steps run: a thread reads the cards' hwmon power1_input every 50 ms (tools/power_probe.py, tools/energy_probe.hip;
    DESIGN.md 6 and 10: the Poseidon kernels run at the power cap, and their time follows joules, not stalls).  The sensor with the largest mean is the
    card under load.  Reported beside the line, never part of `value`; None when no sensor is readable."""

    def __init__(self):
        import glob
        self.files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")) or sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
        self.samples = [[] for _ in self.files]
        self.stop = False
        self.thread = None

    def _run(self):
        while not self.stop:
            for k, f in enumerate(self.files):
                try:
                    with open(f) as fh:
                        self.samples[k].append(int(fh.read().strip()) / 1e6)
                except Exception:
                    pass
            time.sleep(0.05)

    def start(self):
        if self.files:
            import threading
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()
        return self

    def finish(self):
        if self.thread is None:
            return None
        self.stop = True
        self.thread.join(timeout=2)
        best = None
        for k, v in enumerate(self.samples):
            if len(v) >= 4 and (best is None or sum(v) / len(v) > sum(self.samples[best]) / len(self.samples[best])):
                best = k
        if best is None:
            return None
        v = sorted(self.samples[best])
        cap = None
        try:
            with open(os.path.join(os.path.dirname(self.files[best]), "power1_cap")) as fh:
                cap = int(fh.read().strip()) / 1e6
        except Exception:
            pass
        return {"W_median": round(v[len(v) // 2], 1), "W_p5": round(v[len(v) // 20], 1), "W_p95": round(v[(len(v) * 19) // 20], 1), "cap_W": cap, "samples": len(v),
                "what": "hwmon power1_input of the card under load, sampled every 50 ms over the timed steps"}


def pmc_file():
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"):
        if os.path.exists(os.path.join(ROOT, "profiles", name)):
            return name
    return "pmc_traffic.json"


def poseidon_int_roofline(perms, ms, clock_mhz=None):
    """integer-issue roofline of the Goldilocks Poseidon kernels (118 S-boxes x 4 = 472 modular
    multiplications + 30 MDS layers per permutation).  achieved = SIMD cycles per wave of 64 permutations; floor = what the
    multiplications and the matrix-core MDS cost if nothing but their irreducible instructions were issued:
    472 x 5 v_mad_u64_u32 (four 32x32 partial products + the multiply by 2^32-1 that folds the high half; gfx950 has no
    64-bit multiplier) + the 450 v_mfma_i32_32x32x32_i8 of the round-3 layer schedule, priced at their measured issue costs."""
    achieved = N_SIMD * CLOCK_HZ * (ms * 1e-3) / (perms / 64.0)
    n_mfma = 10 * 18 + 5 * 54                                # 8 full-round layers + 2 single partial layers, 5 blocks of four partial rounds (poseidon_blocks.cuh)
    floor = 472 * 5 * CYC_MAD_U64_U32 + n_mfma * CYC_MFMA_ISSUE
    out = {"bound": "int-issue", "kernel": "linear_hash_kernel", "achieved": achieved, "floor": floor, "unit": "SIMD issue cycles per wave of 64 permutations",
           "frac": floor / achieved, "clock_GHz_assumed": CLOCK_HZ / 1e9,
           "floor_terms": {"v_mad_u64_u32": 472 * 5, "cycles_each": CYC_MAD_U64_U32, "v_mfma_i32_32x32x32_i8": n_mfma, "issue_cycles_each": CYC_MFMA_ISSUE},
           "note": "reductions, carries, byte-plane recombination and round constants are overhead by this definition"}
    if clock_mhz:                                              # the same at the clock such a kernel actually runs at (measured inside a kernel of the same load)
        out["clock_MHz_measured"] = {"median": clock_mhz[0], "p5": clock_mhz[1], "p95": clock_mhz[2]}
        out["achieved_at_measured_clock"] = achieved * clock_mhz[0] * 1e6 / CLOCK_HZ
        out["frac_at_measured_clock"] = floor / out["achieved_at_measured_clock"]
    return out


def lde_int_roofline(n_bits, n_cols, cosets, ms):
    """integer-issue roofline of the LDE (iNTT of 2^n rows, then n forward stages on each of `cosets` cosets).
    achieved = SIMD issue cycles per wave-wide element-stage (64 elements through one butterfly stage); floor = what a radix-2
    butterfly costs per element if nothing but its irreducible instructions were issued: half a modular multiplication
    (5 v_mad_u64_u32: four partial products + the fold of the high half) and one 64-bit add or sub (2 carry-chain
    instructions).  Twiddle generation, reductions' carry handling, LDS traffic and addressing are overhead by this
    definition; so is the 9th transform a row-major LDE cannot avoid (the inverse)."""
    stages = n_cols * (1 << n_bits) * n_bits * (1 + cosets)
    achieved = N_SIMD * CLOCK_HZ * (ms * 1e-3) / (stages / 64.0)
    floor = 2.5 * CYC_MAD_U64_U32 + 2 * CYC_CARRY_OP
    return {"bound": "int-issue", "kernel": "interpolate (ntt_pass_kernel x / lde_mid_kernel)", "achieved": achieved, "floor": floor,
            "unit": "SIMD issue cycles per 64 element-stages", "frac": floor / achieved, "clock_GHz_assumed": CLOCK_HZ / 1e9,
            "floor_terms": {"v_mad_u64_u32": 2.5, "cycles_each": CYC_MAD_U64_U32, "carry_chain_ops": 2, "cycles_each_carry": CYC_CARRY_OP},
            "note": "the passes run at 94-100 % of their vector-ALU issue slots (profiles/r03_valu_utilisation.json): the distance to the floor is instruction count, not memory"}


def bn128_work(t):
    """(v_mad_u64_u32 steps, matrix instructions per 64 permutations) of one BN254 Poseidon permutation of width t as bn128.hip
    computes it since round 5 (csrc/bn_mfma.cuh): every linear layer -- the dense MDS products, and the partial rounds four to a
    block: rows, cross terms, column updates -- is a constant-tile product on v_mfma_i32_32x32x32_i8; the vector ALU keeps the
    S-boxes (three Montgomery products of 128 steps each) and, per finished row, the 2 x 12 steps that carry the byte positions
    plus the 8 of the one 32-bit Montgomery step.  RF = 8, RP from poseidon.circom:7-9; RP % 4 rounds run in the old vector form."""
    rp = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68][t - 2]
    n, nb, tail = t - 1, rp // 4, rp % 4
    sbox = (8 * t + rp) * 3 * 128
    nsb = (nb + 1) // 2                                                   # blocks two to a super-block: one column update per eight rounds
    rows = 8 * t + n + 4 * nb + nsb * n                                   # dense rows, closing layer, round rows, column updates
    finish = rows * 32 + 4 * nb * 24                                      # + the block part of a round's row is carried on its own
    valu_tail = tail * ((t * 64 + 64) + n * 128)
    mfma = 2 * (8 * t * t + n * n) + 2 * (nb * (4 * n + 16) + (nb // 2) * 16 + nsb * n + 4 * nb * n)      # pairs: two 32-permutation tiles per wave
    return sbox + finish + valu_tail, mfma


def bn128_mads(t):
    return bn128_work(t)[0]


def start_watchdog(Progress):
    """A rank that has made no progress (stage boundary or collective, pil2gl.parallel.Progress) for PIL2GL_STALL_S seconds
    dumps the stack of every thread and exits non-zero: faulthandler.dump_traceback_later(limit, exit=True), whose timer
    is a C thread that needs no interpreter lock: a main thread stuck inside a driver call (where round 2's 2-rank run on one
    GPU sat, silently, until it was killed) cannot keep it from firing.  torch.distributed.run tears the other ranks down."""
    import faulthandler
    limit = float(os.environ.get("PIL2GL_STALL_S", "240"))
    if limit <= 0:
        return

    last = [-1e9]

    def rearm(what, now):
        # at most once every two seconds: a re-arm cancels and restarts faulthandler's timer thread, and a sharded proof passes ~30
        # marks (every stage and collective) inside the timed region -- the single-GPU line passes none
        if now - last[0] < 2.0:
            return
        last[0] = now
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(limit, exit=True)
    Progress.on_mark = rearm
    rearm(None, 0)


def sharded_memory_estimate(mode, n_bits, n_cols, world, overwrite_trace=False):
    """device bytes one rank of a coset-sharded run holds at its peak (8-byte words), so that a configuration that cannot fit is
    refused with a message instead of found out by the allocator half-way (or, with ranks sharing a GPU, by a peer's collective
    timing out).  Counted: the replicated trace, the coefficient scratch of the LDE, the rank's slice of the extension, its share
    of the node arrays, the quotient stage (slice + coefficients), FRI buffers, exchange windows, library scratch."""
    N = 1 << n_bits
    cc = (1 << EXT_BITS) // world
    w = 8
    trace = w * N * n_cols
    slice_ = trace * cc
    scratch = 0 if overwrite_trace else trace                  # coefficient matrix of the LDE (the witness buffer itself when it may be destroyed)
    nodes = 2 * w * 4 * N * cc * 2                             # leaf digests + subtree, two committed stages
    exchange = 2 * w * 4 * N * cc * 2                          # digests sent + received (all-to-all), windows when ranks share a GPU
    if mode == "prove-sharded":
        q = w * 3 * N * cc * 4 + w * 6 * N * cc * 2            # q slice, its coefficients and blocks; split quotient slice (qDim*qDeg = 6)
        fri = w * 3 * N * cc * 4 + w * 6 * N * cc              # xDivXSubXi (2 openings), f, accumulators, transposed copy
        const_ = w * 2 * N * (1 + cc) + w * 8 * N * cc         # constants (trace domain + own cosets), their leaves and subtree, x and ZhInv slices
    else:
        q = fri = const_ = 0
    total = trace + slice_ + scratch + nodes + exchange + q + fri + const_ + (2 << 30)
    # (rank 0's per-kernel timing pass after the timed steps takes one more slice, but from the blocks the proof has released
    # to torch's caching allocator: not additive)
    return int(total * 1.08)                                   # allocator granularity / fragmentation


def _digest_words(obj):
    """sha256 of the canonical text of a proof (or any nest of lists / dicts / integers) -> four 62-bit words"""
    import hashlib

    def canon(v):
        if isinstance(v, dict):
            return "{" + ",".join('"%s":%s' % (k, canon(v[k])) for k in sorted(v)) + "}"
        if isinstance(v, (list, tuple)):
            return "[" + ",".join(canon(x) for x in v) + "]"
        return str(int(v))
    h = hashlib.sha256(canon(obj).encode()).digest()
    return [int.from_bytes(h[8 * i:8 * i + 8], "little") >> 2 for i in range(4)]


def sharded_identity_check(be, comm, dev, rank, world, mode, args):
    """What a multi-rank run does FIRST, whatever the exchange backend: the sharded path must produce the single-process result before
    anything is timed.  prove-sharded: ONE proof of a config-2-size trace (2^20 x 8) split over the ranks -- every rank ends with a
    complete proof -- and, on rank 0, the ordinary single-process proof of the same trace (its own, unsharded constant tree); the
    digests of all world + 1 proofs must agree.  commit-sharded: the root of the sharded tree against the root of the single-device
    extendAndMerkelize.  -> a dict for the JSON line; on disagreement every rank exits non-zero (they all see all the digests)."""
    from pil2gl import stark, parallel
    import pil2gl
    n_bits, n_cols = WORKLOADS["c2"]
    t0 = time.perf_counter()
    if mode == "prove-sharded":
        ss = {"nBits": n_bits, "nBitsExt": n_bits + EXT_BITS, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": bool(args.split),
              "steps": [{"nBits": b} for b in fri_steps_for(n_bits + EXT_BITS)]}
        info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
        src, consts, publics = fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
        setup_sh = parallel.build_const_tree_sharded(be, consts, info, comm=comm)
        mine = _digest_words(parallel.stark_gen_sharded(be, src, setup_sh, info, exprs, publics, comm=comm)["proof"])
        want = None
        if rank == 0:
            setup = stark.build_const_tree(be, consts, info)
            want = _digest_words(stark.stark_gen(be, src, setup, info, exprs, publics)["proof"])
        what = "proof of 2^%d x %d (FRI %s, 64 queries)" % (n_bits, n_cols, "/".join(str(x["nBits"]) for x in ss["steps"]))
    else:
        src = make_trace(1 << n_bits, n_cols, 0x5EED0000, dev)
        st = parallel.extend_and_merkelize_sharded(be, src, n_cols, n_bits, n_bits + EXT_BITS, split_tree=not args.full_tree, comm=comm)
        mine = _digest_words(st["tree"].root if not args.full_tree else be.root({"nodes": st["nodes"]}))
        want = None
        if rank == 0:
            E = 1 << (n_bits + EXT_BITS)
            dst = torch.empty(E * n_cols, dtype=torch.int64, device=dev)
            pil2gl.interpolate(src, n_cols, n_bits, dst, n_bits + EXT_BITS)
            MH = pil2gl.buildMerkleHash(args.split)
            want = _digest_words(MH.root(MH.merkelize(dst, n_cols, E)))
            del dst
        what = "root of extendAndMerkelize 2^%d x %d" % (n_bits, n_cols)
    t = torch.zeros((world + 1) * 4, dtype=torch.int64)
    t[4 * rank:4 * rank + 4] = torch.tensor(mine, dtype=torch.int64)
    if rank == 0:
        t[4 * world:] = torch.tensor(want, dtype=torch.int64)
    t = comm.all_reduce_sum(t).reshape(world + 1, 4)
    ref = [int(v) for v in t[world]]
    bad = [r for r in range(world) if [int(v) for v in t[r]] != ref]
    if bad:
        sys.stderr.write("bench.py rank %d: SHARDED RESULT DIFFERS from the single-process one on rank(s) %s (%s): nothing timed\n" % (rank, bad, what)); sys.stderr.flush()
        sys.exit(3)
    return {"result": "ok", "checked": what + ", sharded over %d ranks on every rank == single process on rank 0" % world,
            "digest": "%016x" % ref[0], "seconds": round(time.perf_counter() - t0, 2)}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as children (never exec after touching the GPU) and
    relay their output; rank 0 prints the JSON line"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def rehearse_shard(args):
    """--mode commit-sharded --shard-of K on one GPU: rank 0's share of a K-GPU sharded commit (its cosets of the LDE in the
    trace's own memory, its leaves, a tree over stand-in digests), to show the per-GPU time and memory of e.g. config 5."""
    from pil2gl import stark, parallel
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n_bits, n_cols = WORKLOADS[args.workload] if args.workload in WORKLOADS else (int(v) for v in args.workload.lower().split("x"))
    K = args.shard_of
    be = stark.GpuBackend(0, args.split)
    N = 1 << n_bits
    src = make_trace(N, n_cols, 0x5EED0000, dev)
    times = []
    gc.collect(); gc.disable()                                 # (the cycle collector stays out of the measured passes, as in the main line)
    for i in range(args.warmup + args.steps):
        gc.collect()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = parallel.extend_and_merkelize_sharded(be, src, n_cols, n_bits, n_bits + EXT_BITS, overwrite_src=True, rehearse_world=K, split_tree=not args.full_tree)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if i >= args.warmup:
            times.append(dt)
        del st
    gc.enable()
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
    gathered q / FRI polynomial / digests; the evaluations are computed here anyway because rank 0 owns coset 0).  Config 5
    (2^26 x 200 -> 2^29 rows) runs this way too: constant tree split like the others, per-coset domain tables, and the witness
    buffer doubling as the LDE's workspace (107 GB trace + 107 GB slice; the witness is generated again before every step,
    outside the timing)."""
    from pil2gl import stark, parallel
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n_bits, n_cols = WORKLOADS[args.workload] if args.workload in WORKLOADS else (int(v) for v in args.workload.lower().split("x"))
    n_cols -= n_cols & 1
    K = args.shard_of
    ss = {"nBits": n_bits, "nBitsExt": n_bits + EXT_BITS, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": bool(args.split),
          "steps": [{"nBits": b} for b in fri_steps_for(n_bits + EXT_BITS)]}
    info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
    be = stark.GpuBackend(0, args.split)
    need = sharded_memory_estimate("prove-sharded", n_bits, n_cols, K)
    overwrite = need > 0.97 * torch.cuda.mem_get_info()[1]          # no room for trace + coefficient scratch + slice: the trace is the scratch
    if overwrite:
        need = sharded_memory_estimate("prove-sharded", n_bits, n_cols, K, overwrite_trace=True)
    src, consts, publics = fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
    comm = parallel.Comm(rehearse_world=K)
    setup = parallel.build_const_tree_sharded(be, consts, info, comm=comm)
    times = []
    # the interpreter's cycle collector stays out of the measured passes, as in the main line (round 5's record had a 33 ms collection inside
    # the instrumented pass's `queries` stage -- the stage that builds the proof's lists -- and none in the timed passes: its stages summed to
    # 199 ms against a 165 ms step); collected between the passes instead
    gc.collect(); gc.disable()
    for i in range(args.warmup + args.steps):
        if overwrite and i:
            del src
            src, _, _ = fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
        gc.collect()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        parallel.stark_gen_sharded(be, src, setup, info, exprs, publics, comm=comm, overwrite_trace=overwrite)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if i == 0:
            t_first = dt                                       # the rank's tables (x, zerofiers: shard_tables) and run-time compiled evaluators are built in this pass
        if i >= args.warmup:
            times.append(dt)
    dt = sum(times) / len(times)
    stages = {}
    if overwrite:
        del src
        src, _, _ = fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
    comm.reset_stats()
    gc.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = parallel.stark_gen_sharded(be, src, setup, info, exprs, publics, comm=comm, timings=stages, overwrite_trace=overwrite)      # one more, instrumented
    torch.cuda.synchronize(); t_instr = time.perf_counter() - t0
    gc.enable()
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({"metric": "per-GPU time of ONE proof split over %d GPUs (rank 0's share run alone, exchanges stood in)" % K,
                      "value": (1 << n_bits) * n_cols / dt, "unit": "trace-cells/s (the job rate if the %d ranks run in parallel and the exchanges are free)" % K,
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "u64", "data": "synthetic",
                      "config": {"workload": "full proof of 2^%d rows x %d cols, blow-up 8, %d of %d cosets on this GPU" % (n_bits, n_cols, (1 << EXT_BITS) // K, 1 << EXT_BITS),
                                 "mode": "prove-sharded rehearsal", "shard_of": K, "witness_buffer_is_lde_workspace": bool(overwrite)},
                      "stages_s": {k: round(v, 4) for k, v in stages.items()}, "stages_sum_ms": round(sum(stages.values()) * 1e3, 1),
                      "instrumented_pass_ms": round(t_instr * 1e3, 1),
                      "first_proof_ms_with_cold_tables": round(t_first * 1e3, 1), "kept_tables_GB": round(parallel.shard_tables_bytes(be, setup) / 1e9, 3),
                      "note": "ms_per_step is the steady state of a prover that keeps one setup: the rank's x / zerofier tables stay with the setup between proofs (the reference rebuilds them per proof, stark_gen_helpers.js:139-160, as this repository's single-GPU path does); first_proof_ms_with_cold_tables pays for them (and for the evaluators' run-time compilation) once",
                      "exchange_per_proof": r["exchange"], "peak_torch_GB": torch.cuda.max_memory_allocated() / 1e9, "device_GB_in_use_at_end": (total - free) / 1e9,
                      "estimated_peak_GB": need / 1e9}), flush=True)


def h2d_sample(dev, total_bytes):
    """what a caller that holds the witness in host memory pays before the timed step: pinned host -> HBM copy rate on a
    bounded sample (1 GiB), and the time that rate gives for the whole witness.  Reported beside the step, never inside it."""
    try:
        n = 1 << 27
        h = torch.empty(n, dtype=torch.int64).pin_memory()
        d = torch.empty(n, dtype=torch.int64, device=dev)
        d.copy_(h, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        gbs = 3 * n * 8 / (time.perf_counter() - t0) / 1e9
        return {"pinned_h2d_GBps": round(gbs, 1), "sample_GB": round(n * 8 / 1e9, 2), "witness_GB": round(total_bytes / 1e9, 2),
                "witness_upload_ms_at_that_rate": round(total_bytes / gbs / 1e6, 1)}
    except Exception as e:  # pragma: no cover
        return {"error": str(e)}


def prove_from_host(dev, src, step_with, n_proofs=4):
    """Proofs whose witnesses start in (pinned) HOST memory, as the reference's API hands them over (fft_p.js:187 takes a host
    BigBuffer; witnessCalculator.js:145-196 fills it): the upload of witness k+1 runs on a copy stream into a second device
    buffer while proof k computes.  Inside ONE proof nothing can start before the whole witness has landed -- the inverse
    transform's first butterflies pair rows N/2 apart, and column pieces of a row-major witness cross PCIe at half rate
    (tools/h2d_2d.hip: 25 GB/s for 128-byte pieces against 57 GB/s) -- so what is hidden is the NEXT witness's upload.
    -> first proof's latency (upload + proof), steady-state ms per proof, and the upload alone."""
    try:
        n = src.numel()
        host = torch.empty(n, dtype=torch.int64).pin_memory()
        host.copy_(src)                                        # the same valid witness for every proof of the pipeline
        bufs = [torch.empty_like(src), torch.empty_like(src)]
        # one untimed proof after the two witness buffers exist: when they do not fit beside the allocator's cached blocks it
        # releases those, and the next proof pays a second for carving its 100 GB blocks again (seen as a 2.5 s "first proof")
        bufs[0].copy_(src); step_with(bufs[0]); torch.cuda.synchronize()
        copy_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream()
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        freed = [torch.cuda.Event(), torch.cuda.Event()]

        def upload(k):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[k % 2])           # the proof that read this buffer has finished
                bufs[k % 2].copy_(host, non_blocking=True)
                ready[k % 2].record(copy_stream)
        torch.cuda.synchronize()
        t_up0 = time.perf_counter(); upload(0); copy_stream.synchronize(); t_upload = time.perf_counter() - t_up0
        freed[0].record(main); freed[1].record(main)
        torch.cuda.synchronize()
        marks = [time.perf_counter()]
        upload(0)
        for k in range(n_proofs):
            if k + 1 < n_proofs:
                upload(k + 1)
            main.wait_event(ready[k % 2])
            step_with(bufs[k % 2])
            freed[k % 2].record(main)
            torch.cuda.synchronize(); marks.append(time.perf_counter())
        per = [round((b - a) * 1e3, 1) for a, b in zip(marks[:-1], marks[1:])]
        return {"proofs": n_proofs, "upload_alone_ms": round(t_upload * 1e3, 1), "first_proof_ms": per[0], "ms_per_proof": per,
                "steady_state_ms_per_proof": round(sum(per[1:]) / max(1, len(per) - 1), 1),
                "note": "witness k+1 uploads from pinned host memory on a copy stream while proof k computes; the first proof pays its own upload"}
    except Exception as e:  # pragma: no cover
        return {"error": str(e)[:300]}


def proof_sha256(proof):
    """sha256 over the proof in the canonical text of tests/js/prove_c3.js canon(): decimal strings, keys in insertion order"""
    import hashlib

    def canon(v):
        if isinstance(v, dict):
            return "{" + ",".join('"%s":%s' % (k, canon(x)) for k, x in v.items()) + "}"
        if isinstance(v, (list, tuple)):
            return "[" + ",".join(canon(x) for x in v) + "]"
        return '"%d"' % int(v)
    return hashlib.sha256(canon(proof).encode()).hexdigest()


def node_driven_proof(info, exprs, setup, publics, start_row, res):
    """The same proof once more with the JS ORCHESTRATION driving (north_star: "the JS orchestration keeps driving"): a fresh `node`
    child process runs tests/js/prove_c3.js -- prover.js's stage order (src/prover/prover.js:7-127) over the JS drop-in modules, every
    large buffer resident in HBM -- on the same AIR and witness, and reports its best proof time and the digest of its proof, which
    must equal the digest of the proof this process made.  Outside `value`; PIL2GL_BENCH_NODE=0 skips it."""
    import shutil
    import subprocess
    import tempfile
    node = shutil.which("node")
    if node is None:
        return {"skipped": "node is not installed on this box"}
    want = proof_sha256(res["proof"])
    job = {"pilInfo": info, "expressionsInfo": exprs, "start": [str(v) for v in start_row], "publics": [str(v) for v in publics],
           "constRoot": [str(v) for v in setup["constRoot"]], "queries": res["queries"]}
    path = None
    try:
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            path = f.name
            json.dump(job, f)
        t0 = time.perf_counter()
        out = subprocess.run([node, os.path.join(ROOT, "tests", "js", "prove_c3.js"), path, "3"], capture_output=True, text=True, timeout=900)
        wall = time.perf_counter() - t0
        if out.returncode != 0 or "prove c3 OK" not in out.stdout:
            return {"error": (out.stdout[-300:] + out.stderr[-600:])}
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        return {"proof_seconds": line["proof_seconds"], "cells_per_s": line["cells_per_s"], "proofSha256_equal": line["proofSha256"] == want,
                "proofSha256": line["proofSha256"], "child_wall_s": round(wall, 1),
                "what": "node tests/js/prove_c3.js (best of 3): src/prover/prover.js:7-127's stage order over pil2-stark-js_amd/js, buffers in HBM, started as a fresh child process after this one's timed region"}
    except Exception as e:  # pragma: no cover
        return {"error": str(e)[:300]}
    finally:
        if path is not None and os.path.exists(path):          # (the job file holds pilInfo / expressionsInfo: never left behind, whatever the child did)
            os.unlink(path)


def bn128_commit_line(dev, n_bits, n_cols, steps, warmup):
    """config 4: extendAndMerkelize with the BN128 MerkleHash (merklehash_bn128_p.js:47-129, arity 16, non-custom): the GL
    LDE of the 2^24 x 100 trace and the BN254-Poseidon linear hash + 16-ary tree over all 2^27 extended rows"""
    import pil2gl
    from pil2gl import bn128
    arity = 16
    N, E = 1 << n_bits, 1 << (n_bits + EXT_BITS)
    src = make_trace(N, n_cols, 0x5EED0000, dev)
    dst = torch.empty(E * n_cols, dtype=torch.int64, device=dev)
    MH = bn128.buildMerkleHash(arity, False)
    MH.merkelize(src[:n_cols * 64], n_cols, 64); torch.cuda.synchronize()          # parameter generation

    def step():
        pil2gl.interpolate(src, n_cols, n_bits, dst, n_bits + EXT_BITS)
        return MH.merkelize(dst, n_cols, E)
    for _ in range(warmup):
        step()
    power = BoardPower().start() if os.environ.get("PIL2GL_BENCH_POWER", "1") != "0" else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        tree = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    board_power = power.finish() if power is not None else None
    t_lde = ev_time(lambda: pil2gl.interpolate(src, n_cols, n_bits, dst, n_bits + EXT_BITS), 1)
    t_tree = ev_time(lambda: MH.merkelize(dst, n_cols, E), 1)
    n_el = (n_cols + 2) // 3
    chunks = [min(arity, n_el - o) for o in range(0, n_el, arity)]
    leaf_mads = sum(bn128_mads(c + 1) for c in chunks)                              # last chunk: t = nLast + 1 (non-custom)
    leaf_perms = E * len(chunks)
    tree_perms, n = 0, E
    while n > 1:
        n = (n - 1) // arity + 1; tree_perms += n
    mads = E * leaf_mads + tree_perms * bn128_mads(arity + 1)
    mfmas = (E * sum(bn128_work(c + 1)[1] for c in chunks) + tree_perms * bn128_work(arity + 1)[1]) / 64.0    # per wave of 64 permutations
    floor_cyc = mads * CYC_MAD_U64_U32 / 64.0 / N_SIMD                              # chip-wide: a v_mad_u64_u32 serves 64 lanes of one of 1024 SIMDs
    floor_mfma_cyc = mfmas * 32.0 / N_SIMD                                          # 32 cycles of a SIMD's matrix pipe per 32x32x32 i8 instruction
    floor_ms = max(floor_cyc, floor_mfma_cyc) / CLOCK_HZ * 1e3                      # the two pipes run side by side (two waves per SIMD)
    alg = 8 * E * n_cols + 32 * (E + tree_perms)
    out = {"metric": "trace-cells/s, STARK commit step (extend + BN128 Poseidon Merkle tree, arity 16), blow-up 8",
           "value": N * n_cols / dt, "unit": "trace-cells/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (BN254 Fr, Montgomery) + u64 (Goldilocks LDE)", "data": "synthetic",
           "config": {"workload": "config 4: extendAndMerkelize 2^%d rows x %d cols -> 2^%d rows, BN128 Poseidon linear hash + %d-ary tree over all 2^%d extended rows" % (n_bits, n_cols, n_bits + EXT_BITS, arity, n_bits + EXT_BITS),
                      "mode": "commit", "nBits": n_bits, "nCols": n_cols, "nBitsExt": n_bits + EXT_BITS, "hash": "BN128-Poseidon (t = 17, 17, 3 per row; t = 17 in the tree)", "parallelism": "single GPU"},
           "roofline": {"bound": "hbm", "kernel": "bn_linear_hash_kernel + bn_merkle_level_kernel", "achieved": alg / t_tree / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / t_tree / 1e6 / HBM_PEAK_GBS, "traffic": None,
                        "note": "BN254 Poseidon is integer-issue bound by three orders of magnitude; see roofline_int_issue"},
           "roofline_int_issue": {"bound": "int-issue", "kernel": "bn_linear_hash_kernel", "achieved": t_tree, "floor": floor_ms, "unit": "ms per tree at %.1f GHz" % (CLOCK_HZ / 1e9),
                                  "frac": floor_ms / t_tree, "v_mad_u64_u32_per_row": leaf_mads, "cycles_each": CYC_MAD_U64_U32,
                                  "vector_floor_ms": floor_cyc / CLOCK_HZ * 1e3, "matrix_floor_ms": floor_mfma_cyc / CLOCK_HZ * 1e3,
                                  "v_mfma_i32_32x32x32_i8_per_row": sum(bn128_work(c + 1)[1] for c in chunks) / 64.0, "matrix_cycles_each": 32,
                                  "note": "floor = the larger of (a) every 32x32 product left on the vector ALU -- the S-boxes' Montgomery products and the rows' carry / reduction steps (bn128_work) -- issued back to back at the measured 4.5 cycles and (b) every matrix instruction of the linear layers at 32 cycles of its SIMD's matrix pipe; round 4's all-vector form needed 971 648 products per row (floor 3 847 ms)"},
           "kernels": [{"kernel": "BN128 merkelize (leaf hash + tree)", "ms": t_tree, "perms": leaf_perms + tree_perms, "Mperm_s": (leaf_perms + tree_perms) / t_tree / 1e3},
                       {"kernel": "interpolate", "ms": t_lde, "alg_bytes": 8 * N * n_cols * (1 + (1 << EXT_BITS)), "GBps": 8 * N * n_cols * 9 / t_lde / 1e6}],
           "root": hex(MH.root(tree))}
    if board_power is not None:
        out["board_power"] = board_power
    return out


def bench_bn128(args, dev, wl, n_bits, n_cols):
    out = bn128_commit_line(dev, n_bits, n_cols, args.steps, args.warmup)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_bn128(n_cols, 16)
        out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)


def bench_bn128_prove(args, dev, n_bits, n_cols):
    """`--workload c4 --mode prove`: a WHOLE proof with `verificationHashType: "BN128"` at config 4's size -- the stage loop of bench's default line
    (Fibonacci AIR, FRI 27/22/17/12/7, 64 queries) with every tree a BN254-Poseidon arity-16 tree and the BN128 transcript
    (stark_gen_helpers.js:93-101, 388-412; merklehash_bn128_p.js:47-129): GL LDE and evaluators as in config 3, commitments as in config 4."""
    import pil2gl
    from pil2gl import stark
    pil2gl.init(dev.index or 0)
    n_cols -= n_cols & 1
    ss = {"nBits": n_bits, "nBitsExt": n_bits + EXT_BITS, "nQueries": 64, "verificationHashType": "BN128", "merkleTreeArity": 16, "merkleTreeCustom": False,
          "steps": [{"nBits": b} for b in fri_steps_for(n_bits + EXT_BITS)]}
    info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
    be = stark.GpuBackend(dev.index or 0, False, "BN128", 16, False)
    src, consts, publics = fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
    t0 = time.perf_counter()
    setup = stark.build_const_tree(be, consts, info)
    torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
    for _ in range(max(1, args.warmup)):
        stark.stark_gen(be, src, setup, info, exprs, publics)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        res = stark.stark_gen(be, src, setup, info, exprs, publics)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    stages = {}
    stark.stark_gen(be, src, setup, info, exprs, publics, timings=stages); torch.cuda.synchronize()
    N = 1 << n_bits
    out = {"metric": "STARK prove time (ms_per_step) and trace-cells/s, synthetic Fibonacci AIR, BN128 Poseidon (arity 16) Merkle trees and transcript + FRI, blow-up 8",
           "value": N * n_cols / dt, "unit": "trace-cells/s", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": dt * 1e3,
           "higher_is_better": True, "scaling": None, "vs_baseline": None, "dtype": "u64 (Goldilocks) + u32 limbs (BN254 Fr)", "data": "synthetic",
           "config": {"workload": "full proof (commit, Q, evals, FRI %s, 64 queries) of 2^%d rows x %d cols -> 2^%d rows, verificationHashType BN128, 16-ary trees" % (
                          "/".join(str(b) for b in fri_steps_for(n_bits + EXT_BITS)), n_bits, n_cols, n_bits + EXT_BITS),
                      "mode": "prove", "config": "c4", "nBits": n_bits, "nCols": n_cols, "nBitsExt": n_bits + EXT_BITS, "hash": "BN128-Poseidon (arity 16)", "parallelism": "single GPU"},
           "prove": {"seconds": dt, "stages_s": {k: round(v, 4) for k, v in stages.items()}, "const_tree_s": round(t_setup, 3)},
           "root1": hex(res["proof"]["root1"]), "proofSha256": proof_sha256(res["proof"])}
    print(json.dumps(out), flush=True)


def other_configs(dev, be, main_wl, main_split):
    """The BASELINE configurations the headline line does not time, measured by the SAME run after its timed region and per-kernel pass,
    with every buffer of the headline workload freed first: config 2 (whole proof), config 3 with `splitLinearHash` leaves (whole proof),
    config 4 (BN128 commit).  Outside `value`; each entry carries its proof digest / root so that a reader can check what was computed.
    PIL2GL_BENCH_OTHER=0 skips it."""
    import pil2gl
    from pil2gl import stark
    res = {}

    def proof_line(n_bits, n_cols, split, steps):
        ss = {"nBits": n_bits, "nBitsExt": n_bits + EXT_BITS, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": bool(split),
              "steps": [{"nBits": b} for b in fri_steps_for(n_bits + EXT_BITS)]}
        info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
        src, consts, publics = fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0)
        be_ = stark.GpuBackend(dev.index or 0, bool(split))                         # (the leaf form is the backend's MerkleHash, not the starkStruct's flag)
        setup = stark.build_const_tree(be_, consts, info)
        stark.stark_gen(be_, src, setup, info, exprs, publics)                      # warm-up (run-time compiled evaluators, tables)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            r = stark.stark_gen(be_, src, setup, info, exprs, publics)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        return {"workload": "full proof of 2^%d rows x %d cols, blow-up 8, GL Poseidon %s linear hash, FRI %s, 64 queries" % (
                    n_bits, n_cols, "split" if split else "plain", "/".join(str(b) for b in fri_steps_for(n_bits + EXT_BITS))),
                "ms_per_proof": round(dt * 1e3, 2), "steps": steps, "cells_per_s": (1 << n_bits) * n_cols / dt, "proofSha256": proof_sha256(r["proof"])}

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            res[name] = fn()
        except Exception as e:  # pragma: no cover
            res[name] = {"error": repr(e)[:300]}
        res[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        gc.collect(); torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info(dev)[0]
    if main_wl != "c2":
        guarded("config2_proof", lambda: proof_line(20, 8, False, 5))
    if free >= 170e9:
        if not (main_wl == "c3" and main_split):
            guarded("config3_split_proof", lambda: proof_line(24, 100, True, 2))

        def c4():
            o = bn128_commit_line(dev, 24, 100, 1, 0)
            t = o["kernels"][0]
            return {"workload": o["config"]["workload"], "ms_per_commit": round(o["ms_per_step"], 1), "tree_ms": round(t["ms"], 1), "Mperm_s": round(t["Mperm_s"], 2),
                    "roofline_int_issue": round(o["roofline_int_issue"]["frac"], 4), "root": o["root"], "cells_per_s": o["value"],
                    "board_power_W_median": (o.get("board_power") or {}).get("W_median")}
        guarded("config4_bn128_commit", c4)
    else:
        res["skipped"] = "configs 3 (split) and 4 need 170 GB free on the GPU (%.0f GB are)" % (free / 1e9)
    return res


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1 and not args.shard_of:
        sys.exit(self_launch(args))                            # nothing has touched the GPU yet
    world = int(env_world or "1")
    if env_world is not None and args.gpus not in (1, world):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    mode = args.mode
    if mode == "auto":
        mode = "prove" if (world == 1 or args.replicas) else "prove-sharded"
        if args.shard_of:
            mode = "prove-sharded"
    elif mode == "commit" and world > 1 and not args.replicas:
        mode = "commit-sharded"
    if args.shard_of:
        if args.workload == "auto":
            args.workload = "c3"
        return rehearse_shard(args) if mode == "commit-sharded" else rehearse_prove(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    n_dev = max(1, torch.cuda.device_count())
    shared_gpu = world > n_dev                                 # rehearsing N ranks on fewer GPUs: ranks share devices
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    sharded_mode = mode in ("commit-sharded", "prove-sharded")
    backend = None
    if world > 1 or sharded_mode:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("PIL2GL_BENCH_BACKEND", "gloo" if shared_gpu else "nccl")   # RCCL wants one device per rank
        # a collective that has waited this long has lost its peer: fail (non-zero exit of every rank) instead of sitting out
        # the default ten minutes and leaving an empty record
        import datetime
        coll_timeout = datetime.timedelta(seconds=float(os.environ.get("PIL2GL_DIST_TIMEOUT", "120")))
        if backend == "nccl":
            # a collective whose peer is gone must END the job (non-zero exit of every rank), not hang it: the process group's watchdog aborts
            # the process when a collective has been pending for `coll_timeout` -- also while the host sits in the device synchronisation
            # behind it (all_reduce_sum's copy to the host); said explicitly, whatever this build's default is
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "0")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=coll_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=coll_timeout)
    import pil2gl
    pil2gl.init(local_rank)
    from pil2gl.parallel import Progress
    if dist is not None:
        start_watchdog(Progress)
    if dist is not None:
        Progress.mark("process group up: %d ranks, backend %s%s; a rank without progress for %s s dumps its stacks and exits" % (
            world, backend, ", ranks share GPUs" if shared_gpu else "", os.environ.get("PIL2GL_STALL_S", "240")), rank, key=True)

    wl = args.workload
    if wl == "auto":
        free, _ = torch.cuda.mem_get_info()
        if mode == "commit-sharded" and world == 8 and free > 235e9:
            wl = "c5"                                          # 2^26 x 200: 221 GB per GPU, fits only sharded
        else:
            wl = "c3" if free > (170e9 if not sharded_mode else 100e9) * (world if shared_gpu else 1) else "c2"
    if wl in WORKLOADS:
        n_bits, n_cols = WORKLOADS[wl]
    else:
        n_bits, n_cols = (int(v) for v in wl.lower().split("x"))
    N, E = 1 << n_bits, 1 << (n_bits + EXT_BITS)
    if wl == "c4":
        if world > 1:
            raise SystemExit("--workload c4 is a single-GPU line")
        if args.mode == "prove":
            return bench_bn128_prove(args, dev, n_bits, n_cols)
        return bench_bn128(args, dev, wl, n_bits, n_cols)
    if sharded_mode and (1 << EXT_BITS) % world:
        raise SystemExit("a sharded mode needs a world size dividing the %d cosets" % (1 << EXT_BITS))
    if sharded_mode:
        free, total_mem = torch.cuda.mem_get_info()
        sharers = (world + n_dev - 1) // n_dev if shared_gpu else 1
        need = sharded_memory_estimate(mode, n_bits, n_cols, world)
        need_all = need * sharers
        Progress.mark("set-up: %s %s, this rank needs about %.0f GB at its peak%s; device has %.0f GB" % (
            mode, wl, need / 1e9, (" (%.0f GB for the %d ranks sharing it)" % (need_all / 1e9, sharers)) if sharers > 1 else "", total_mem / 1e9), rank, key=True)
        if need_all > 0.97 * total_mem and os.environ.get("PIL2GL_SKIP_MEMCHECK", "0") in ("", "0"):
            raise SystemExit("bench.py: %s at %s needs about %.0f GB of device memory (%d rank%s on this GPU, %.0f GB each at the peak) but the device has %.0f GB: "
                             "refused (use --workload c2, or one GPU per rank)" % (mode, wl, need_all / 1e9, sharers, "s" if sharers > 1 else "", need / 1e9, total_mem / 1e9))

    from pil2gl import stark, parallel
    comm = parallel.Comm() if dist is not None else None
    prove_ctx = None
    be = stark.GpuBackend(local_rank, args.split)
    identity = None
    # (PIL2GL_BENCH_IDENTITY=0 skips the check; =force runs it at one rank too: the RCCL group of one a single-GPU box allows)
    if sharded_mode and (world > 1 or os.environ.get("PIL2GL_BENCH_IDENTITY") == "force") and os.environ.get("PIL2GL_BENCH_IDENTITY", "1") != "0":
        Progress.mark("identity check: sharded result against the single-process one", rank, key=True)
        identity = sharded_identity_check(be, comm, dev, rank, world, mode, args)
        Progress.mark("identity check %s (%s s)" % (identity["result"], identity["seconds"]), rank, key=True)
    if mode in ("prove", "prove-sharded"):
        n_cols -= n_cols & 1                                   # pairs of columns
        ss = {"nBits": n_bits, "nBitsExt": n_bits + EXT_BITS, "nQueries": 64, "verificationHashType": "GL",
              "splitLinearHash": bool(args.split), "steps": [{"nBits": b} for b in fri_steps_for(n_bits + EXT_BITS)]}
        if args.air == "perm":
            copies = max(1, n_cols // 11)
            n_cols = 11 * copies                               # cells of both witness stages together: 2 + 9 base columns per check
            info, exprs, _ = stark.permutation_air(ss, copies)
            src, consts, publics = permutation_trace_gpu(dev, n_bits, copies)
        else:
            info, exprs, _ = stark.fibonacci_air(n_cols // 2, ss)
            src, consts, publics = fibonacci_trace_gpu(dev, n_bits, n_cols // 2, 0 if mode == "prove-sharded" else rank)   # sharded: ONE trace
        # sharded: the constant tree is split over the ranks like the witness trees (no rank holds an extended column whole)
        setup = parallel.build_const_tree_sharded(be, consts, info, comm=comm) if mode == "prove-sharded" else stark.build_const_tree(be, consts, info)
        prove_ctx = (setup, info, exprs, publics)
    elif mode == "commit-sharded":                             # ONE trace, replicated; the cosets of its extension are split
        src = make_trace(N, n_cols, 0x5EED0000, dev)
    else:
        src = make_trace(N, n_cols, 0x5EED0000 + rank, dev)
    MH = pil2gl.buildMerkleHash(args.split)
    dst = nodes = None
    if mode == "commit":
        dst = torch.empty(E * n_cols, dtype=torch.int64, device=dev)
        nodes = torch.empty(MH._getNNodes(E * 4), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    import ctypes as C

    def step(timings=None):
        if mode == "prove":
            setup_, info_, exprs_, publics_ = prove_ctx
            return stark.stark_gen(be, src, setup_, info_, exprs_, publics_, timings=timings)
        if mode == "prove-sharded":
            setup_, info_, exprs_, publics_ = prove_ctx
            return parallel.stark_gen_sharded(be, src, setup_, info_, exprs_, publics_, comm=comm, timings=timings)
        if mode == "commit-sharded":
            return parallel.extend_and_merkelize_sharded(be, src, n_cols, n_bits, n_bits + EXT_BITS, overwrite_src=(wl == "c5"), split_tree=not args.full_tree, comm=comm)
        pil2gl.interpolate(src, n_cols, n_bits, dst, n_bits + EXT_BITS)
        pil2gl.call("pil2gl_merkelize_dev", pil2gl._ptr(dst), n_cols, E, int(args.split), pil2gl._ptr(nodes), C.c_void_p(stream))
        return None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not a timed or counted step: on a box that has just been handed out the first pass after process start has
    # been seen ~180 ms (13 %) slower than the following ones (clock ramp, first touch of 150 GB); one untimed pass ahead of the
    # W warm-up steps keeps that out of a short run's mean.  PIL2GL_BENCH_PREWARM=0 skips it.
    def mark(what):
        if dist is not None:
            Progress.mark(what, rank, key=(rank == 0))
    if os.environ.get("PIL2GL_BENCH_PREWARM", "1") != "0":
        mark("pre-warm pass"); step()
    for i in range(args.warmup):
        mark("warm-up step %d" % i); step()
    # the interpreter's cycle collector stays out of the timed region (round 1's line showed one step in eight or nine 43 ms
    # slower than its neighbours: a generation-2 collection walking the proof objects); collected once here instead
    gc.collect(); gc.disable()
    if comm is not None:
        comm.reset_stats()
    power = BoardPower().start() if rank == 0 and os.environ.get("PIL2GL_BENCH_POWER", "1") != "0" else None
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    for i in range(args.steps):
        mark("timed step %d" % i)
        step()
        step_marks.append(time.perf_counter())               # host clock only: no extra synchronisation inside the timed region
    barrier()
    mark("timed region done")
    dt = time.perf_counter() - t0
    gc.enable()
    board_power = power.finish() if power is not None else None
    step_ms = [round((b - a) * 1e3, 2) for a, b in zip([t0] + step_marks[:-1], step_marks)]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = (1 if sharded_mode else world) * N * n_cols / (dt / args.steps)
    exchange = None
    if comm is not None and sharded_mode:
        exchange = comm.stats()
        for k in ("collectives", "bytes_sent_per_rank", "bytes_received_per_rank"):
            exchange[k] = exchange[k] // args.steps            # per step
        exchange["backend"] = backend + (" + HIP IPC windows (ranks share a GPU)" if exchange["mode"] == "ipc" else (" (RCCL over xGMI)" if backend == "nccl" else ""))
    # one more pass, outside the timed region, with a synchronisation after every stage (collective in the sharded mode: all ranks)
    stage_times = {}
    last_proof = None
    if mode in ("prove", "prove-sharded") and (rank == 0 or mode == "prove-sharded"):
        last_proof = step(timings=stage_times); torch.cuda.synchronize()

    if rank == 0:
        # ---- per-kernel timing with HIP events on the launch stream (each call below is exactly one kernel, or one kernel
        #      family named in DESIGN.md), at the size this rank runs them: the whole extension, or its cosets ----
        cc = ((1 << EXT_BITS) // world) if sharded_mode else (1 << EXT_BITS)
        rows = N * cc
        if dst is None or dst.numel() != rows * n_cols:
            dst = None
            dst = torch.empty(rows * n_cols, dtype=torch.int64, device=dev)
        iters = max(1, min(3, args.steps))
        digests = torch.empty(rows * 4, dtype=torch.int64, device=dev)
        # (--air perm: the witness is two stages of 2K and 9K columns; the per-kernel figures are taken on one N x 11K matrix)
        ksrc = make_trace(N, n_cols, 77, dev) if (prove_ctx is not None and args.air == "perm") else src
        if sharded_mode:
            ws = src if wl == "c5" else None
            lde = lambda: be.interpolate_cosets(ksrc, n_cols, n_bits, dst, n_bits + EXT_BITS, 0, cc, ws)
        else:
            lde = lambda: pil2gl.interpolate(ksrc, n_cols, n_bits, dst, n_bits + EXT_BITS)
        t_lde = ev_time(lde, iters)
        t_leaf = ev_time(lambda: pil2gl.linearHash(dst, n_cols, args.split, digests), iters)
        leaf_clock_mhz = clock_under_hash_load()
        # the other leaf form (splitLinearHash, linearhash_gpu.js:30-66: the reference's own *.starkstruct.gpu.json files set it), same rows
        t_leaf_other = ev_time(lambda: pil2gl.linearHash(dst, n_cols, not args.split, digests), iters) if n_cols > 8 else None
        lvl = torch.empty(rows * 2, dtype=torch.int64, device=dev)
        t_lvl = ev_time(lambda: pil2gl.merkelizeLevel(digests, lvl), iters)
        leaf_perms = rows * ((n_cols + 7) // 8) if n_cols > 4 else 0
        if args.split and n_cols > 4:
            batch = max(8, (n_cols + 3) // 4); nb = (n_cols + batch - 1) // batch
            leaf_perms = rows * (sum((min(batch, n_cols - b * batch) + 7) // 8 if min(batch, n_cols - b * batch) > 4 else 0 for b in range(nb)) + (((4 * nb) + 7) // 8 if nb > 1 else 0))
        # the FRI polynomial's weighted row sums over the stage matrix (2 openings): the HBM-bound kernel of the proof
        t_dot = None
        if n_cols % 2 == 0 and 32 <= n_cols <= 112:
            import numpy as _np
            from pil2gl import _lib as _l
            coef = _np.random.default_rng(5).integers(0, 2 ** 63, (2, n_cols, 3), dtype=_np.uint64) % _np.uint64(0xFFFFFFFF00000001)
            dacc = torch.empty(rows * 6, dtype=torch.int64, device=dev)
            t_dot = ev_time(lambda: _l.call("pil2gl_rows_dot_ext_dev", pil2gl._ptr(dst), n_cols, rows, pil2gl._ptr(coef), 2, pil2gl._ptr(dacc), 0, None), iters)
            del dacc
        kernels = [
            {"kernel": "linear_hash_kernel", "ms": t_leaf, "alg_bytes": 8 * rows * n_cols + 32 * rows, "perms": leaf_perms},
            {"kernel": "interpolate (ntt_pass_kernel x / lde_mid_kernel)", "ms": t_lde, "alg_bytes": 8 * N * n_cols * (1 + cc)},
            {"kernel": "merkle_level_kernel (first level)", "ms": t_lvl, "alg_bytes": 32 * rows + 16 * rows, "perms": rows // 2},
        ]
        if t_dot is not None:
            kernels.append({"kernel": "rows_dot_mfma_kernel (FRI polynomial: weighted sums of every row, 2 openings)", "ms": t_dot, "alg_bytes": 8 * rows * n_cols + 48 * rows,
                            "bound": "hbm"})
        def split_perms(w):
            batch = max(8, (w + 3) // 4); nb_ = (w + batch - 1) // batch
            return sum((min(batch, w - b * batch) + 7) // 8 if min(batch, w - b * batch) > 4 else 0 for b in range(nb_)) + (((4 * nb_) + 7) // 8 if nb_ > 1 else 0)
        if t_leaf_other is not None:
            other_perms = rows * ((n_cols + 7) // 8 if args.split else split_perms(n_cols))
            kernels.append({"kernel": "linear_hash_kernel (plain form)" if args.split else "linear_hash_split_kernel (splitLinearHash form of the same leaves)",
                            "ms": t_leaf_other, "alg_bytes": 8 * rows * n_cols + 32 * rows, "perms": other_perms})
        if wl == "c3" and not sharded_mode:                    # the committed PMC passes were taken at config 3, one GPU
            pf = pmc_file()
            kernels[0]["traffic"] = load_pmc(pf, "linear_hash_kernel")
            kernels[1]["traffic"] = load_pmc_lde_traffic(pf)
            kernels[2]["traffic"] = load_pmc(pf, "merkle_level_kernel")
            for k in kernels[:3]:
                k["traffic_source"] = "committed rocprofv3 PMC passes of one config-3 proof (profiles/%s: bytes by request size; not measured in this run)" % pf
        for k in kernels:
            k["GBps"] = k["alg_bytes"] / k["ms"] / 1e6
            k["hbm_frac"] = k["GBps"] / HBM_PEAK_GBS
            if k.get("perms"):
                k["Gperm_s"] = k["perms"] / k["ms"] / 1e6
        dom = max(kernels[:3] + [k for k in kernels[3:] if k.get("bound") == "hbm"], key=lambda k: k["ms"])
        roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": dom["hbm_frac"], "traffic": dom.get("traffic"), "traffic_source": dom.get("traffic_source"),
                    "note": "Poseidon hashing is integer-issue bound (no 64-bit multiplier on gfx950): its HBM fraction is small by nature; roofline_int_issue prices it against its own roof, kernels[1] is the HBM-bound LDE"}
        out = {}
        if prove_ctx is not None:
            info = prove_ctx[1]
            air_name = "two-stage permutation-check AIR (%d checks: 2 + 9 base columns each, grand-product hints)" % (n_cols // 11) if args.air == "perm" else "Fibonacci AIR"
            metric = "STARK prove time (ms_per_step) and trace-cells/s, synthetic %s, GL Poseidon Merkle + FRI, blow-up 8" % air_name
            workload = "full proof (commit, Q, evals, FRI %s, %d queries) of 2^%d rows x %d cols -> 2^%d rows, %s linear hash, %s" % (
                "/".join(str(x["nBits"]) for x in info["starkStruct"]["steps"]), info["starkStruct"]["nQueries"], n_bits, n_cols, n_bits + EXT_BITS, "split" if args.split else "plain",
                "ONE proof split by cosets over the GPUs (leaf digests, q, evaluations, FRI polynomial exchanged)" if mode == "prove-sharded" else "per GPU")
        elif mode == "commit-sharded":
            metric = "trace-cells/s, STARK commit step (extend+merkelize) of ONE trace split by cosets over the GPUs, GL Poseidon Merkle, blow-up 8"
            workload = "extendAndMerkelize 2^%d rows x %d cols -> 2^%d rows, %s linear hash, %d of 8 cosets per GPU + all-gather of leaf digests" % (
                n_bits, n_cols, n_bits + EXT_BITS, "split" if args.split else "plain", (1 << EXT_BITS) // world)
        else:
            metric = "trace-cells/s, STARK commit step (extend+merkelize), GL Poseidon Merkle, blow-up 8"
            workload = "extendAndMerkelize 2^%d rows x %d cols -> 2^%d rows, %s linear hash, per GPU" % (n_bits, n_cols, n_bits + EXT_BITS, "split" if args.split else "plain")
        out = {
            "metric": metric,
            "value": value, "unit": "trace-cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": ("strong" if sharded_mode else "weak") if world > 1 else None, "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload, "mode": mode, "config": wl,
                       "nBits": n_bits, "nCols": n_cols, "nBitsExt": n_bits + EXT_BITS, "hash": "GL-Poseidon-12",
                       "parallelism": ("coset-sharded x%d" % world) if sharded_mode else ("replicas x%d" % world if world > 1 else "single GPU")},
            "roofline": roofline, "roofline_int_issue": poseidon_int_roofline(kernels[0]["perms"], kernels[0]["ms"], leaf_clock_mhz) if kernels[0]["perms"] else None,
            "roofline_int_issue_lde": lde_int_roofline(n_bits, n_cols, cc, kernels[1]["ms"]),
            "kernels": kernels,
        }
        if board_power is not None:
            out["board_power"] = board_power
        if t_leaf_other is not None:
            out["leaf_plain_ms" if args.split else "leaf_split_ms"] = t_leaf_other
        if dist is not None:
            out["n_ranks"] = dist.get_world_size()
            out["ranks_share_gpus"] = bool(shared_gpu)
        if exchange is not None:
            out["exchange_per_step"] = exchange
        if identity is not None:
            out["sharded_identity"] = identity["result"]
            out["sharded_identity_check"] = identity
        if prove_ctx is not None:
            cexp = prove_ctx[2]["expressionsCode"][prove_ctx[1]["cExpId"]]["code"]["code"]
            fexp = prove_ctx[2]["expressionsCode"][prove_ctx[1]["friExpId"]]["code"]["code"]
            out["prove"] = {"seconds": ms_per_step / 1e3, "stages_s": {k: round(v, 4) for k, v in stage_times.items()}, "host_ms_of_each_step": step_ms,
                            "air": {"name": args.air, "machines": (n_cols // 11) if args.air == "perm" else n_cols // 2, "witness_stages": prove_ctx[1]["nStages"],
                                    "stage_widths": {k: v for k, v in prove_ctx[1]["mapSectionsN"].items()}, "hints": len(prove_ctx[1].get("hints", [])),
                                    "constraint_ops_per_extended_row": len(cexp), "fri_ops_per_extended_row": len(fexp),
                                    "openings": len(prove_ctx[1]["openingPoints"]), "evaluations": len(prove_ctx[1]["evMap"])}}
        if world == 1:
            out["witness_upload"] = h2d_sample(dev, 8 * N * n_cols)
        if world == 1 and mode == "prove" and os.environ.get("PIL2GL_BENCH_FROM_HOST", "1") != "0":
            del dst, digests, lvl                               # (back to torch's allocator cache: the proofs below reuse the blocks)
            setup_, info_, exprs_, publics_ = prove_ctx
            out["prove_from_host"] = prove_from_host(dev, src, lambda buf: stark.stark_gen(be, buf, setup_, info_, exprs_, publics_))
        if world == 1 and mode == "prove" and args.air != "perm" and last_proof is not None and os.environ.get("PIL2GL_BENCH_NODE", "1") != "0":
            import numpy as _np2
            start_row = [int(v) for v in src[:n_cols].cpu().numpy().view(_np2.uint64)]
            dst = digests = lvl = None
            gc.collect(); torch.cuda.empty_cache()             # the child needs its own 143 GB at config 3; this process keeps the witness and the constant tree (~25 GB): 168 of 288 GB
            out["node_driven"] = node_driven_proof(prove_ctx[1], prove_ctx[2], prove_ctx[0], prove_ctx[3], start_row, last_proof)
            out["node_driven_ok"] = bool(out["node_driven"].get("proofSha256_equal")) or "skipped" in out["node_driven"]      # a failed child shows at the top level, not only inside its record
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_prove(n_cols, args.split, args.air) if prove_ctx is not None else cpu_baseline_commit(n_cols, args.split)
            out["speedup_vs_cpu_port"] = value / out["cpu_baseline"]["value"]
        if world == 1 and mode == "prove" and args.air != "perm" and os.environ.get("PIL2GL_BENCH_OTHER", "1") != "0":
            # the other BASELINE configurations, timed by this run too (outside `value`): everything of the headline workload goes first
            src = dst = digests = lvl = nodes = last_proof = prove_ctx = setup = consts = ksrc = lde = None
            gc.collect(); torch.cuda.empty_cache()
            out["other_configs"] = other_configs(dev, be, wl, args.split)
        print(json.dumps(out), flush=True)
    if dist is not None:
        mark("closing")
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
