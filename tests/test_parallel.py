"""Multi-process extendAndMerkelize: the cosets of the extension are split across ranks and only leaf
digests are exchanged.  CPU: world_size 2 and 4 over gloo on the checker backend.  GPU: 2 ranks driving the HIP library on
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


def _launch(world, *args, timeout=600, worker=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker or WORKER, *args]
    env = dict(os.environ, OMP_NUM_THREADS="2")
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


@pytest.mark.parametrize("world,split", [(2, 0), (4, 1)])
def test_sharded_commit_gloo_cpu(oracle, world, split):
    _launch(world, "--backend", "oracle", "--nbits", "5", "--extbits", "3", "--npols", "5", "--split", str(split))


@pytest.mark.gpu
@pytest.mark.parametrize("world,nbits,npols", [(2, 10, 9), (4, 13, 33)])
def test_sharded_commit_gpu_ranks(oracle, world, nbits, npols):
    _launch(world, "--backend", "gpu", "--nbits", str(nbits), "--extbits", "3", "--npols", str(npols))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_proof_equals_single_process_proof_cpu(oracle, world):
    """one proof over `world` ranks (commit, constraint evaluation and FRI polynomial split by cosets; q, evaluations and
    the FRI polynomial exchanged): every rank ends with the proof of the ordinary prove loop, bit for bit"""
    _launch(world, "--backend", "oracle", worker=PROVE_WORKER)


@pytest.mark.gpu
@pytest.mark.parametrize("world,nbits,pairs,steps", [(2, 8, 3, "11,7,3"), (4, 10, 20, "13,9,4")])
def test_sharded_proof_gpu_ranks(oracle, world, nbits, pairs, steps):
    _launch(world, "--backend", "gpu", "--nbits", str(nbits), "--pairs", str(pairs), "--steps", steps, worker=PROVE_WORKER)


def test_sharded_hash_commits_proof_cpu(oracle):
    """starkStruct.hashCommits in the sharded prove loop (publics, evaluations and last polynomial absorbed as hashes)"""
    _launch(2, "--backend", "oracle", "--hashcommits", "1", worker=PROVE_WORKER)


def test_sharded_two_stage_proof_cpu(oracle):
    """two witness stages (stage 2 = challenge + grand-product hint): stage 2 is committed by cosets like stage 1"""
    _launch(2, "--backend", "oracle", "--air", "perm", worker=PROVE_WORKER)


@pytest.mark.gpu
def test_sharded_two_stage_proof_gpu_ranks(oracle):
    _launch(4, "--backend", "gpu", "--air", "perm", "--nbits", "10", "--steps", "13,9,4", worker=PROVE_WORKER)
