"""Multi-GPU partition of extendAndMerkelize: one process per GPU, the 2^extendBits cosets of the
extended domain are split across the ranks.

Reference path: stark_gen_helpers.js:302-318 extendAndMerkelize = interpolate (fft_p.js:187) + merkelize
(merklehash_p.js:44).  Row (pos << b) + j of the extended matrix is the evaluation on coset j of the size-N subgroup,
and a coset needs nothing but the N trace coefficients, so rank r computes cosets [r*cc, (r+1)*cc) on its own trace
copy, hashes its own leaves, and the only exchange is an all-gather of the 32-byte leaf digests (RCCL over xGMI when
the process group is nccl, gloo on CPU): 4/C of the extended matrix in words.  Every rank then builds the (small) upper
tree redundantly, so each holds the full node array and can open any Merkle path; the row values of a query come from
the rank that owns its coset (open_rows).

`be` is a backend object (pil2gl.stark.GpuBackend in production; the tests pass their CPU checker) providing
interpolate_cosets / linear_hash_rows / merkelize_digest_parts / as_torch.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def coset_range(rank, world, ext_bits):
    """cosets [begin, begin+count) owned by `rank`; world must divide 2^ext_bits"""
    n = 1 << ext_bits
    if world < 1 or n % world:
        raise ValueError("world size %d does not divide the %d cosets of the extension" % (world, n))
    cc = n // world
    return rank * cc, cc


def _comm_tensor(be, t, group):
    """tensor handed to the collective: device tensor for nccl, host tensor for gloo"""
    x = be.as_torch(t)
    if dist.get_backend(group) != "nccl" and x.is_cuda:
        x = x.cpu()
    return x


def extend_and_merkelize_sharded(be, src, n_pols, n_bits, n_bits_ext, group=None, overwrite_src=False, rehearse_world=None):
    """Sharded extendAndMerkelize.  `src` = the full N x n_pols trace on every rank.
    Returns {"local": N x (cc*n_pols) slice (row pos, coset jl, col c), "nodes": full tree.nodes, "width", "height",
    "cosetBegin", "cosetCount", "extBits"}; tree root = last 4 words of nodes, identical on all ranks and to the
    single-GPU merkelize of the full extension.
    overwrite_src: let the LDE use the trace buffer for its coefficient matrix (config 5: 107 GB trace + 107 GB slice per
    GPU, no third buffer).  rehearse_world=K: run rank 0's share of a K-rank job alone, standing in copies of the own
    digests for the gathered ones (a one-GPU rehearsal of the per-GPU time and memory; the tree is not a real root)."""
    if rehearse_world:
        rank, world = 0, int(rehearse_world)
    else:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    eb = n_bits_ext - n_bits
    cb, cc = coset_range(rank, world, eb)
    N = 1 << n_bits
    local = be.empty(N * cc * n_pols)
    be.interpolate_cosets(src, n_pols, n_bits, local, n_bits_ext, cb, cc, src if overwrite_src else None)
    digests = be.linear_hash_rows(local, n_pols, N * cc)             # [N*cc][4], local row = pos*cc + jl
    if rehearse_world:
        mine = be.as_torch(digests).reshape(-1)
        gathered = [mine] * world
    else:
        mine = _comm_tensor(be, digests, group).reshape(-1)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
    # part r is [N][cc*4]; leaf index = pos*2^eb + r*cc + jl (natural row order of the extension), written straight into
    # the node array (no stacked / permuted copies: at config 5 the leaf level alone is 17 GB)
    height = N << eb
    nodes = be.merkelize_digest_parts(gathered, N, cc)
    return {"local": local, "nodes": nodes, "width": n_pols, "height": height,
            "cosetBegin": cb, "cosetCount": cc, "extBits": eb}


def owner_of_row(idx, ext_bits, world):
    """(rank, local row) holding extended row idx"""
    cc = (1 << ext_bits) // world
    j = idx & ((1 << ext_bits) - 1)
    return j // cc, (idx >> ext_bits) * cc + (j % cc)


def open_rows(be, stree, idxs, group=None):
    """values of the extended rows idxs (fri.js:83-105 opens every tree at the query rows): each rank fills the rows of
    its own cosets, one all-reduce (sum with zeros, exact) completes them everywhere.  -> numpy [len(idxs)][width]"""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    w = stree["width"]
    loc = be.as_torch(stree["local"]).reshape(-1, w)
    sel = [(q, owner_of_row(i, stree["extBits"], world)) for q, i in enumerate(idxs)]
    out = torch.zeros((len(idxs), w), dtype=torch.int64, device=loc.device)
    own = [(q, lr) for q, (r, lr) in sel if r == rank]
    if own:
        qi = torch.tensor([q for q, _ in own], device=loc.device)
        li = torch.tensor([lr for _, lr in own], device=loc.device)
        out[qi] = loc[li].to(torch.int64)
    x = out if dist.get_backend(group) == "nccl" or not out.is_cuda else out.cpu()
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x.cpu().numpy().view(np.uint64)
