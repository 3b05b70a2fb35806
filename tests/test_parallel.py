"""Multi-process extendAndMerkelize: the cosets of the extension are split across ranks and only leaf
digests are exchanged.  CPU: world_size 2, 4 and 8 (one coset per rank, the driver's largest launch) over gloo on the checker backend.  GPU: 2 ranks driving the HIP library on
the one card of the box (gloo exchange; the nccl path differs only in where the gathered tensor lives)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "sharded_commit_worker.py")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


PROVE_WORKER = os.path.join(ROOT, "tests", "workers", "sharded_prove_worker.py")


def _launch(world, *args, timeout=600, worker=None, env_extra=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker or WORKER, *args]
    env = dict(os.environ, OMP_NUM_THREADS="2", **(env_extra or {}))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok") == world


def test_coset_range_partition():
    sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
    from pil2gl import parallel
    for eb in (1, 3, 4):
        for world in (1, 2, 1 << eb):
            seen = []
            for r in range(world):
                b, c = parallel.coset_range(r, world, eb)
                seen += list(range(b, b + c))
            assert seen == list(range(1 << eb))
            for idx in range(4 << eb):
                r, lr = parallel.owner_of_row(idx, eb, world)
                b, c = parallel.coset_range(r, world, eb)
                j = idx & ((1 << eb) - 1)
                assert b <= j < b + c and lr == (idx >> eb) * c + (j - b)
    with pytest.raises(ValueError):
        parallel.coset_range(0, 3, 3)


@pytest.mark.parametrize("world,split", [(2, 0), (4, 1), (8, 0)])
def test_sharded_commit_gloo_cpu(oracle, world, split):
    _launch(world, "--backend", "oracle", "--nbits", "5", "--extbits", "3", "--npols", "5", "--split", str(split))


@pytest.mark.gpu
@pytest.mark.parametrize("world,nbits,npols", [(2, 10, 9), (4, 13, 33)])
def test_sharded_commit_gpu_ranks(oracle, world, nbits, npols):
    _launch(world, "--backend", "gpu", "--nbits", str(nbits), "--extbits", "3", "--npols", str(npols))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_proof_equals_single_process_proof_cpu(oracle, world):
    """one proof over `world` ranks (commit, constraint evaluation and FRI polynomial split by cosets; q, evaluations and
    the FRI polynomial exchanged): every rank ends with the proof of the ordinary prove loop, bit for bit"""
    _launch(world, "--backend", "oracle", worker=PROVE_WORKER)


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_proof_with_sharded_constant_tree_cpu(oracle, world):
    """the constant tree split over the ranks like the witness trees (what a domain beyond one device's memory needs): same root,
    same proof -- the constants' opened rows and lower siblings travel in the one batched sum with the others"""
    _launch(world, "--backend", "oracle", "--shardsetup", "1", worker=PROVE_WORKER)


@pytest.mark.gpu
def test_sharded_proof_with_sharded_constant_tree_gpu_ranks(oracle):
    """the same with the HIP library: the rank's rows of x_ext come from pil2gl_geometric_dev, coset by coset"""
    _launch(4, "--backend", "gpu", "--nbits", "10", "--pairs", "20", "--steps", "13,9,4", "--shardsetup", "1", worker=PROVE_WORKER)


@pytest.mark.gpu
@pytest.mark.parametrize("world,nbits,pairs,steps", [(2, 8, 3, "11,7,3"), (4, 10, 20, "13,9,4")])
def test_sharded_proof_gpu_ranks(oracle, world, nbits, pairs, steps):
    _launch(world, "--backend", "gpu", "--nbits", str(nbits), "--pairs", str(pairs), "--steps", steps, worker=PROVE_WORKER)


@pytest.mark.gpu
def test_sharded_proof_gpu_ranks_exchange_in_pieces(oracle):
    """ranks sharing a GPU exchange through HIP-IPC windows of at most 1 GiB (a 2 GiB window could not be mapped: round 2's
    2-rank config-3 run hung in hipIpcOpenMemHandle); larger exchanges go through them in pieces.  Windows of 2^16 words
    here, so that the digests (2^13 rows x 4 cosets x 4 words per rank) and the quotient rows take several pieces each: the
    proof must still be the single-process proof"""
    _launch(2, "--backend", "gpu", "--nbits", "13", "--pairs", "6", "--steps", "16,11,6", worker=PROVE_WORKER,
            env_extra={"PIL2GL_IPC_WINDOW_WORDS": str(1 << 16)})


@pytest.mark.gpu
def test_sharded_proof_over_rccl_one_rank(oracle):
    """the RCCL branch of the exchange layer (device tensors into the collectives, the digest all-gather started asynchronously
    per chunk of hashed leaves, sums on the device): all the box's one GPU allows is a group of ONE rank, which still runs every
    collective call of the N-rank schedule; 2^16 rows so that the chunked path is taken"""
    _launch(1, "--backend", "gpu", "--pg", "nccl", "--nbits", "16", "--pairs", "4", "--steps", "19,14,9,4", worker=PROVE_WORKER)
    # and with the constant tree split too (its all-to-all, root sum and opened rows through the same RCCL calls)
    _launch(1, "--backend", "gpu", "--pg", "nccl", "--nbits", "16", "--pairs", "4", "--steps", "19,14,9,4", "--shardsetup", "1", worker=PROVE_WORKER)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_proof_with_boundaries_gpu_ranks(oracle, world):
    """pil2 boundaries in the sharded proof: the rank's rows of the firstRow / lastRow / everyFrame zerofiers are built from its own
    cosets (parallel.zi_slice: batched inversion over the rank's rows, the evaluator over its x rows), never sliced out of a
    2^nBitsExt-row table; the tables are kept with the setup and a second proof reuses them"""
    _launch(world, "--backend", "gpu", "--nbits", "10", "--pairs", "5", "--steps", "13,9,4", "--boundaries", "1", "--twice", "1", worker=PROVE_WORKER)


def test_sharded_proof_with_boundaries_cpu(oracle):
    _launch(2, "--backend", "oracle", "--boundaries", "1", "--twice", "1", worker=PROVE_WORKER)


@pytest.mark.parametrize("steps", ["9,2", "9"])
def test_sharded_proof_with_fri_groups_across_cosets_cpu(oracle, steps):
    """the first FRI tree's groups stay inside one coset only while steps[1].nBits >= the extension bits; below that (or
    with a single FRI step, whose polynomial is absorbed whole) the FRI polynomial is all-gathered and folded everywhere"""
    _launch(2, "--backend", "oracle", "--steps", steps, worker=PROVE_WORKER)


def test_sharded_hash_commits_proof_cpu(oracle):
    """starkStruct.hashCommits in the sharded prove loop (publics, evaluations and last polynomial absorbed as hashes)"""
    _launch(2, "--backend", "oracle", "--hashcommits", "1", worker=PROVE_WORKER)


def test_sharded_two_stage_proof_cpu(oracle):
    """two witness stages (stage 2 = challenge + grand-product hint): stage 2 is committed by cosets like stage 1"""
    _launch(2, "--backend", "oracle", "--air", "perm", worker=PROVE_WORKER)


@pytest.mark.gpu
def test_sharded_two_stage_proof_gpu_ranks(oracle):
    _launch(4, "--backend", "gpu", "--air", "perm", "--nbits", "10", "--steps", "13,9,4", worker=PROVE_WORKER)


@pytest.mark.gpu
def test_config5_rank_share_on_one_gpu(oracle):
    """BASELINE config 5 (2^26 rows x 200 columns -> 2^29 extended rows: 859 GB, fits only sharded over 8 GPUs): rank 0's share
    (coset 0 of 8: the 107 GB trace doubling as the LDE's workspace, a 107 GB slice, its leaves, its block of the tree)
    run alone on one GPU, the other ranks' digests stood in.  Column c of the trace is the x table rotated by s_c rows, i.e.
    the polynomial w^(s_c) X, so every extended value has a closed form: slice row pos of coset 0 is 7 w^(s_c + pos).
    Checked against it: sampled rows up to the last one (word offsets beyond 2^33); against the oracle: the leaf digests of
    those rows and the sibling paths of rows in this rank's leaf block up to the root."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
    import pil2gl
    from pil2gl import stark, parallel
    import gl_oracle as orc
    import gc
    pil2gl.shutdown(); pil2gl.init(0)                               # the library's scratch slots of earlier tests
    gc.collect(); torch.cuda.empty_cache()                          # what they left in torch's allocator cache
    free, _ = torch.cuda.mem_get_info()
    if free < 245e9:
        pytest.skip("needs 245 GB of free device memory (%.0f GB free)" % (free / 1e9))
    P = 0xFFFFFFFF00000001
    nb, C, eb, world = 26, 200, 3, 8
    N = 1 << nb
    be = stark.GpuBackend(0, False)
    x = be.build_x(nb, 1)                                           # w^i
    trace = be.empty(N * C)
    shifts = [(c * 7919 + 1 + (c % 3) * (N // 3)) % N for c in range(C)]
    tv = trace.view(N, C)
    for c in range(C):
        tv[:, c] = torch.roll(x, -shifts[c])
    del x
    comm = parallel.Comm(rehearse_world=world)
    cb, cc = parallel.coset_range(0, world, eb)
    assert (cb, cc) == (0, 1)
    local = be.empty(N * cc * C)
    be.interpolate_cosets(trace, C, nb, local, nb + eb, cb, cc, trace)    # the trace is the workspace: nothing else fits
    del trace, tv
    digests = be.linear_hash_rows(local, C, N * cc)
    w = int(orc.root(nb))
    rows = [0, 1, 5, (N >> 3) - 1, (N >> 3), N // 2 + 12345, N - 2, N - 1]
    lv = local.view(N, C)
    dg = be.as_torch(digests).view(N, 4)
    for pos in rows:
        want = [7 * pow(w, shifts[c] + pos, P) % P for c in range(C)]
        got = [int(v) for v in lv[pos].cpu().numpy().view(np.uint64)]
        assert got == want, pos
        assert [int(v) for v in dg[pos].cpu().numpy().view(np.uint64)] == [int(v) for v in orc.linear_hash(np.array(want, dtype=np.uint64))], pos
    tree = parallel.ShardedTree(be, [be.as_torch(digests).reshape(-1)] * world, N, cc, comm)
    block = N * cc                                                  # leaves of this rank's block: positions < N/8, all cosets
    idxs = [0, 8, 8 * 5 + 3, block - 1, block // 2 + 77]
    sib = tree.siblings(idxs)
    for leaf, mp in zip(idxs, sib):
        assert len(mp) == nb + eb
        pos = leaf >> eb                                             # every coset's stand-in digest is coset 0's
        cur = [int(v) for v in dg[pos].cpu().numpy().view(np.uint64)]
        i = leaf
        for s in mp:                                                 # merklehash_p.js:169-189
            pair = (cur + list(s)) if i % 2 == 0 else (list(s) + cur)
            cur = [int(v) for v in orc.poseidon(np.array(pair, dtype=np.uint64), np.zeros(4, np.uint64), 4)]
            i >>= 1
        assert cur == [int(v) for v in tree.root], leaf
    del local, digests, tree
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_config5_proof_rank_share_on_one_gpu(oracle):
    """BASELINE config 5 PROVED sharded, in rehearsal: rank 0's share of an 8-rank proof of a 2^26 x 200 trace (2^29 extended rows: no
    single device holds one extended column set) run alone on one GPU -- constant tree split like the witness trees, x / ZhInv built per
    coset, the witness buffer doubling as the LDE's workspace, the quotient's coefficients from per-coset transforms.  The exchanges
    are stood in by the rank's own data, so the result is not a proof; what IS checked, on sampled slice rows up to the last one, is
    everything the rank computes from its own data, against closed forms on host integers: the witness is column c = w^(s_c) X, so
    its extension on coset 0 is 7 w^(s_c + pos); L1(x) = (x^N - 1) / (N (x - 1)), LLAST(x) = L1(w x); x = 7 w^pos; ZhInv = 1 / (7^N - 1);
    and the 2 600-op constraint expression of the AIR on those rows (big-integer interpreter, stark_verify.js:222-298 restated)
    equals the rank's q_ext there."""
    import numpy as np
    import torch
    import gc
    sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
    import pil2gl
    from pil2gl import stark, parallel
    pil2gl.shutdown(); pil2gl.init(0)
    gc.collect(); torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 250e9:
        pytest.skip("needs 250 GB of free device memory (%.0f GB free)" % (free / 1e9))
    P = 0xFFFFFFFF00000001
    nb, C, eb, world = 26, 200, 3, 8
    N = 1 << nb
    ss = {"nBits": nb, "nBitsExt": nb + eb, "nQueries": 16, "verificationHashType": "GL", "steps": [{"nBits": b} for b in (29, 24, 19, 14, 9, 4)]}
    info, exprs, _ = stark.fibonacci_air(C // 2, ss)
    be = stark.GpuBackend(0, False)
    x = be.build_x(nb, 1)
    trace = be.empty(N * C)
    shifts = [(c * 7919 + 1 + (c % 3) * (N // 3)) % N for c in range(C)]
    tv = trace.view(N, C)
    for c in range(C):
        tv[:, c] = torch.roll(x, -shifts[c])
    del x, tv
    consts = np.zeros((N, 2), dtype=np.uint64); consts[0, 0] = 1; consts[N - 1, 1] = 1
    publics = [5, 6, 7]
    comm = parallel.Comm(rehearse_world=world)
    setup = parallel.build_const_tree_sharded(be, consts, info, comm=comm)
    rows = [0, 1, 2, 77, N // 2 + 12345, N - 2, N - 1]
    samples = {"rows": rows}
    r = parallel.stark_gen_sharded(be, trace, setup, info, exprs, publics, comm=comm, overwrite_trace=True, samples=samples)
    assert torch.cuda.max_memory_allocated() < 262e9
    w = stark.root_of_unity(nb)
    inv = lambda a: pow(a % P, P - 2, P)
    s7N = pow(7, N, P)
    L1 = lambda xx: (s7N - 1) * inv(N * (xx - 1)) % P            # xx on the coset 7 <w>: xx^N = 7^N
    vc = [int(v) for v in r["challenges"][info["nStages"]][0]]
    code = exprs["expressionsCode"][info["cExpId"]]["code"]["code"]
    for k, pos in enumerate(rows):
        xx = 7 * pow(w, pos, P) % P
        assert [int(v) for v in samples["cm1_ext"][k]] == [7 * pow(w, shifts[c] + pos, P) % P for c in range(C)], pos
        assert [int(v) for v in samples["const_ext"][k]] == [L1(xx), L1(xx * w % P)], pos
        assert int(samples["x_ext"][k][0]) == xx and int(samples["Zi_ext#0"][k][0]) == inv(s7N - 1), pos

        def resolve(ref):
            ty = ref["type"]
            xp = xx * pow(w, ref.get("prime", 0), P) % P           # "next row" of a slice row is the same coset one step on
            if ty == "cm":
                pm = info["cmPolsMap"][ref["id"]]
                assert pm["stage"] == 1 and pm["dim"] == 1
                return pow(w, shifts[pm["stagePos"]], P) * xp % P
            if ty == "const": return L1(xp) if ref["id"] == 0 else L1(xp * w % P)
            if ty == "challenge": return [int(v) for v in r["challenges"][ref["stage"] - 1][ref["stageId"]]]
            if ty == "public": return publics[ref["id"]]
            if ty == "number": return int(ref["value"]) % P
            if ty == "Zi": return inv(s7N - 1)
            if ty == "x": return xx
            raise ValueError(ty)
        want = stark.execute_code(code, resolve)
        want = want if isinstance(want, list) else [want, 0, 0]
        assert [int(v) for v in samples["q_ext"][k]] == want, pos
    assert len(vc) == 3 and len(r["proof"]["evals"]) == len(info["evMap"])
