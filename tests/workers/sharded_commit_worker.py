"""One rank of the sharded extendAndMerkelize test (launched by tests/test_parallel.py under torch.distributed.run).

--backend oracle : CPU ranks over gloo; the partition logic (coset ranges, digest all-gather order, tree assembly,
                   row opening) is run on the CPU checker and compared with the unsharded oracle result.
--backend gpu    : every rank drives the HIP library on cuda:0 (a 1-GPU box), exchange over gloo; compared with the
                   unsharded library result and with the oracle.
Exits non-zero on any mismatch."""
import argparse
import os
import sys

import numpy as np

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
    ap.add_argument("--extbits", type=int, default=3)
    ap.add_argument("--npols", type=int, default=5)
    ap.add_argument("--split", type=int, default=0)
    a = ap.parse_args()
    dist.init_process_group("gloo", timeout=TIMEOUT)
    rank, world = dist.get_rank(), dist.get_world_size()
    import gl_oracle as orc
    from conftest import rand_field
    from pil2gl import parallel
    orc.build(); orc.set_threads(2)
    if a.backend == "gpu":
        from pil2gl.stark import GpuBackend
        be = GpuBackend(0, split=bool(a.split))
    else:
        from stark_backend import OracleBackend
        be = OracleBackend(split=bool(a.split))
    nb, nbe, C = a.nbits, a.nbits + a.extbits, a.npols
    trace = rand_field(np.random.default_rng(1234), ((1 << nb), C))     # same seed on every rank = replicated trace
    st = parallel.extend_and_merkelize_sharded(be, be.from_host(trace), C, nb, nbe)
    ext = orc.interpolate(trace, nb, nbe)
    want_nodes = orc.merkelize(ext, bool(a.split))
    got_nodes = be.to_host(st["nodes"])
    assert np.array_equal(got_nodes, want_nodes), "rank %d: tree nodes differ" % rank
    cb, cc = st["cosetBegin"], st["cosetCount"]
    assert (cb, cc) == parallel.coset_range(rank, world, a.extbits)
    want_local = ext.reshape(1 << nb, 1 << a.extbits, C)[:, cb:cb + cc, :].reshape(-1)
    assert np.array_equal(be.to_host(st["local"]), want_local), "rank %d: local cosets differ" % rank
    if a.backend == "gpu":   # and the unsharded library path gives the same tree
        full = be.empty(C << nbe)
        be.interpolate(be.from_host(trace), C, nb, full, nbe)
        t1 = be.merkelize(full, C, 1 << nbe)
        assert np.array_equal(be.to_host(t1["nodes"]), got_nodes)
    idxs = [0, 1, (1 << nbe) - 1, 37 % (1 << nbe), (5 << a.extbits) + 3]
    # the same commit with the tree split by leaf blocks: same root, and the sibling paths of the full tree
    st2 = parallel.extend_and_merkelize_sharded(be, be.from_host(trace), C, nb, nbe, split_tree=True)
    assert [int(v) for v in st2["tree"].root] == [int(v) for v in want_nodes[-4:]], "rank %d: split-tree root differs" % rank
    sib = st2["tree"].siblings(idxs)
    for k, i in enumerate(idxs):
        want_sib, off, n, j = [], 0, 1 << nbe, i
        while n > 1:
            want_sib.append([int(v) for v in want_nodes[off + 4 * (j ^ 1): off + 4 * (j ^ 1) + 4]])
            off += 4 * (n + (n & 1)); n = (n + 1) // 2; j //= 2
        assert [[int(v) for v in s_] for s_ in sib[k]] == want_sib, "rank %d: split-tree path of leaf %d differs" % (rank, i)
    # the commit with the leaf hashing cut into four pieces, each piece's digests gathered while the next is hashed (the RCCL branch's
    # overlap, commit_local_slice(chunks=4): here over gloo, where the pieces go one after the other): same node array
    comm = parallel.Comm(None)
    chunked = parallel.commit_local_slice(be, st["local"], C, nb, cc, comm, split_tree=False, chunks=4)
    assert np.array_equal(be.to_host(chunked), want_nodes), "rank %d: chunked commit differs" % rank
    chunked_split = parallel.commit_local_slice(be, st["local"], C, nb, cc, comm, split_tree=False, chunks=3)      # ragged pieces
    assert np.array_equal(be.to_host(chunked_split), want_nodes), "rank %d: commit in three ragged pieces differs" % rank
    rows = parallel.open_rows(be, st, idxs)
    assert np.array_equal(rows, ext[idxs]), "rank %d: opened rows differ" % rank
    for i in idxs:
        r, lr = parallel.owner_of_row(i, a.extbits, world)
        assert 0 <= r < world and 0 <= lr < (cc << nb)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
