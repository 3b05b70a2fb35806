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
import json
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


MAX_NTT_BITS = 30        # csrc/common.h PIL2GL_MAX_NTT_BITS: the largest single transform


def coset_range(rank, world, ext_bits):
    """cosets [begin, begin+count) owned by `rank`; world must divide 2^ext_bits"""
    n = 1 << ext_bits
    if world < 1 or n % world:
        raise ValueError("world size %d does not divide the %d cosets of the extension" % (world, n))
    cc = n // world
    return rank * cc, cc


class _Done:
    """handle of an exchange that has already completed"""
    def __init__(self, parts): self.parts = parts
    def wait(self): return self.parts


class _Pending:
    def __init__(self, work, parts): self.work, self.parts = work, parts
    def wait(self):
        if self.work is not None:
            self.work.wait()                                    # nccl: the current stream waits for the collective's stream
            self.work = None
        return self.parts


class Progress:
    """Where a sharded run is, for whoever has to explain a run that stopped: every stage boundary and every collective
    calls mark().  PIL2GL_TRACE=1 prints one flushed stderr line per mark (rank, seconds since start, free device memory),
    PIL2GL_TRACE=0 none, unset only the marks a caller flags as key (bench.py: set-up and one per step); on_mark lets the
    caller re-arm a stall timer at every mark."""
    t0 = None
    last = ("start", 0.0)
    count = 0
    on_mark = None              # bench.py re-arms its stall timer here

    @classmethod
    def mark(cls, what, rank=None, key=False):
        import os, sys, time as _t
        if cls.t0 is None:
            cls.t0 = _t.perf_counter()
        now = _t.perf_counter() - cls.t0
        cls.last = (what, now)
        cls.count += 1
        if cls.on_mark is not None:
            cls.on_mark(what, now)
        if os.environ.get("PIL2GL_TRACE", "1" if key else "0") not in ("", "0"):       # unset: only the caller's key marks
            mem = ""
            try:
                if torch.cuda.is_available() and torch.cuda.is_initialized():
                    free, total = torch.cuda.mem_get_info()
                    mem = " free %.1f/%.1f GB" % (free / 1e9, total / 1e9)
            except Exception:
                pass
            r = rank if rank is not None else os.environ.get("RANK", "0")
            sys.stderr.write("[pil2gl r%s %8.3fs]%s %s\n" % (r, now, mem, what)); sys.stderr.flush()


class Comm:
    """Exchange layer of the sharded prover: the few collectives SURVEY.md 8e names (all-gather of leaf digests, of q and of
    the FRI polynomial; sums of a few opened rows / evaluations), with the bytes they move counted.

    mode "nccl"  one rank per GPU: torch.distributed's nccl backend, which on ROCm is RCCL over xGMI.  Every tensor stays on
                 the device; an all-gather may be started asynchronously (it runs on RCCL's stream) so that the next
                 chunk of leaf hashing overlaps it.
    mode "gloo"  host tensors (the CPU tests on the checker backend).
    mode "ipc"   several ranks SHARING one GPU (the one-GPU rehearsal of an N-rank job: RCCL refuses two ranks on one device,
                 and staging through host memory and gloo costs hundreds of ms).  Each rank exposes two windows of device
                 memory to the others (HIP IPC, through torch's CUDA tensor sharing); an all-gather is then: copy into the
                 own window, stream synchronise, one host barrier, `world` device-to-device copies out of the peers'
                 windows.  The windows alternate, so a window is only rewritten after every rank has passed the barrier of
                 the following exchange, i.e. has finished reading it (each rank synchronises its stream before a barrier).
                 Small sums travel through gloo on the host.
    mode "rehearse"  rank 0 of a K-rank job run alone: its own data stands in for the other ranks' (timing / memory only).
    """

    def __init__(self, group=None, rehearse_world=None, mode=None):
        self.group = group
        self.bytes_sent = self.bytes_received = self.collectives = 0
        self._auto_ipc = False
        if rehearse_world:
            self.mode, self.rank, self.world = "rehearse", 0, int(rehearse_world)
            return
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        backend = dist.get_backend(group)
        import os
        self.mode = mode or ("nccl" if backend == "nccl" else "gloo")
        # device tensors under a gloo group = ranks sharing a GPU: exchange through IPC windows unless told to stage through the host
        self._auto_ipc = mode is None and backend != "nccl" and self.world > 1 and os.environ.get("PIL2GL_EXCHANGE", "ipc") == "ipc"
        if self.mode == "ipc" and self.world == 1:
            self.mode = "gloo"
        self._win, self._peer, self._turn = [None, None], [None, None], 0
        self.IPC_WINDOW_WORDS = max(1 << 16, int(os.environ.get("PIL2GL_IPC_WINDOW_WORDS", Comm.IPC_WINDOW_WORDS)))   # (small in the tests: pieces)

    # ---- helpers
    def _count(self, sent, received):
        self.collectives += 1; self.bytes_sent += int(sent); self.bytes_received += int(received)

    def stats(self):
        return {"mode": self.mode, "ranks": self.world, "collectives": self.collectives,
                "bytes_sent_per_rank": self.bytes_sent, "bytes_received_per_rank": self.bytes_received}

    def reset_stats(self):
        self.bytes_sent = self.bytes_received = self.collectives = 0

    def barrier(self):
        if self.mode != "rehearse":
            Progress.mark("barrier", self.rank)
            dist.barrier(group=self.group)

    IPC_WINDOW_WORDS = 1 << 27      # 1 GiB: hipIpcOpenMemHandle on a 2 GiB allocation never returned on this driver (dmabuf IPC);
                                    # larger exchanges go through the windows in pieces

    def _windows(self, n, device):
        """both windows hold min(n, IPC_WINDOW_WORDS) words on every rank (n is the same on every rank)"""
        n = min(n, self.IPC_WINDOW_WORDS)
        if self._win[0] is not None and self._win[0].numel() >= n:
            return
        from torch.multiprocessing.reductions import reduce_tensor
        cap = max(n, 1 << 16)
        Progress.mark("IPC windows: 2 x %.2f GB" % (8 * cap / 1e9), self.rank)
        torch.cuda.synchronize()
        dist.barrier(group=self.group)                          # nobody still reads the old windows
        self._peer = [None, None]
        self._win = [torch.empty(cap, dtype=torch.int64, device=device) for _ in range(2)]
        mine = [reduce_tensor(w) for w in self._win]
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        self._peer = [[(self._win[k] if r == self.rank else everyone[r][k][0](*everyone[r][k][1])) for r in range(self.world)] for k in range(2)]
        Progress.mark("IPC windows mapped", self.rank)

    def _ipc_pieces(self, mine, take):
        """ranks sharing a GPU: `mine` (1-D, same length everywhere) goes through the windows in pieces of at most
        IPC_WINDOW_WORDS; after a piece [c0, c1) of every rank stands in its window, take(r, view, c0, c1) copies what this
        rank wants of rank r's piece (view = that piece, the own tensor's slice for r = rank)"""
        n = mine.numel()
        self._windows(n, mine.device)
        step = self._win[0].numel()
        for c0 in range(0, n, step):
            c1 = min(n, c0 + step)
            k = self._turn; self._turn ^= 1
            self._win[k][:c1 - c0].copy_(mine[c0:c1])
            torch.cuda.synchronize()
            dist.barrier(group=self.group)                      # every rank's piece is in place (and the other window is free again)
            for r in range(self.world):
                take(r, mine[c0:c1] if r == self.rank else self._peer[k][r][:c1 - c0], c0, c1)

    # ---- collectives
    def all_gather_start(self, mine, outs=None):
        """mine: 1-D int64 tensor (same length on every rank).  -> handle whose wait() returns the list of every rank's
        tensor (rank order), written into `outs` when given (a list of `world` contiguous tensors)"""
        n = mine.numel()
        Progress.mark("all_gather of %d words per rank (%s)" % (n, self.mode), self.rank)
        if self.mode == "rehearse":
            self._count(8 * n, 8 * n * (self.world - 1))
            if outs is not None:
                for o in outs:
                    o.copy_(mine)
                return _Done(outs)
            return _Done([mine] * self.world)
        self._count(8 * n, 8 * n * (self.world - 1))
        if self.mode == "gloo" and self._auto_ipc and mine.is_cuda:
            self.mode = "ipc"                                    # every rank takes this branch at the same exchange
        if self.mode == "ipc" and mine.is_cuda:
            if outs is None:
                outs = [torch.empty_like(mine) for _ in range(self.world)]
            self._ipc_pieces(mine, lambda r, view, c0, c1: outs[r][c0:c1].copy_(view))
            return _Done(outs)
        x = mine
        if self.mode != "nccl" and x.is_cuda:                  # gloo with device data: through the host (slow; tests only)
            x = x.cpu()
        host_staged = x is not mine
        if self.mode == "nccl" and outs is None:
            # one output buffer, rank-major: RCCL writes every rank's part in place (a LIST of outputs is gathered into a buffer of the
            # library's own first and copied out part by part)
            flat = torch.empty(self.world * n, dtype=x.dtype, device=x.device)
            work = dist.all_gather_into_tensor(flat, x.contiguous(), group=self.group, async_op=True)
            return _Pending(work, list(flat.view(self.world, n)))
        parts = outs if (outs is not None and not host_staged) else [torch.empty_like(x) for _ in range(self.world)]
        if self.mode == "nccl":
            work = dist.all_gather(parts, x, group=self.group, async_op=True)
            return _Pending(work, parts)
        dist.all_gather(parts, x, group=self.group)
        if host_staged:
            if outs is not None:
                for o, p_ in zip(outs, parts):
                    o.copy_(p_)
                parts = outs
            else:
                parts = [p_.to(mine.device) for p_ in parts]
        return _Done(parts)

    def all_gather(self, mine, outs=None):
        return self.all_gather_start(mine, outs).wait()

    def all_to_all(self, mine):
        """mine: 1-D int64 tensor of `world` equal chunks, chunk s meant for rank s.  -> list of `world` tensors: the chunk every
        rank meant for THIS rank (rank order).  (xGMI is point-to-point: an all-to-all uses every link at once.)"""
        n = mine.numel()
        if n % self.world:
            raise ValueError("all_to_all needs %d equal chunks" % self.world)
        ch = n // self.world
        Progress.mark("all_to_all of %d words per rank (%s)" % (n, self.mode), self.rank)
        self._count(8 * (n - ch), 8 * (n - ch))
        if self.mode == "rehearse":
            return [mine[:ch]] * self.world
        if self.mode == "gloo" and self._auto_ipc and mine.is_cuda:
            self.mode = "ipc"
        if self.mode == "ipc" and mine.is_cuda:
            got = [torch.empty(ch, dtype=mine.dtype, device=mine.device) for _ in range(self.world)]
            lo, hi = self.rank * ch, (self.rank + 1) * ch        # the chunk of every rank's tensor meant for this rank

            def take(r, view, c0, c1):
                a, b = max(lo, c0), min(hi, c1)
                if a < b:
                    got[r][a - lo:b - lo].copy_(view[a - c0:b - c0])
            self._ipc_pieces(mine, take)
            return got
        x = mine
        if self.mode != "nccl" and x.is_cuda:
            x = x.cpu()
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x.contiguous(), group=self.group)
        if x is not mine:
            out = out.to(mine.device)
        return list(out.reshape(self.world, ch))

    def all_reduce_sum(self, t):
        """sum over the ranks of a SMALL int64 tensor (every entry is non-zero on one rank only, so the sum is exact);
        returns a host tensor"""
        if self.mode == "rehearse":
            return t.cpu()
        Progress.mark("all_reduce of %d words (%s)" % (t.numel(), self.mode), self.rank)
        self._count(8 * t.numel(), 8 * t.numel())
        if self.mode == "nccl":
            x = t if t.is_cuda else t.cuda()
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            return x.cpu()
        x = t.cpu() if t.is_cuda else t
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x


def _comm_of(comm, group, rehearse_world):
    return comm if comm is not None else Comm(group, rehearse_world)


def commit_local_slice(be, local, width, n_bits, cc, comm, split_tree=False, chunks=None):
    """leaf digests of a rank's slice (N*cc rows x width), all-gathered, then the levels above them -> the complete node
    array on every rank, or (split_tree) a ShardedTree.  Under RCCL the rows are hashed in `chunks` pieces and the
    all-gather of a piece runs (on RCCL's stream) while the next piece is being hashed."""
    N = 1 << n_bits
    rows = N * cc
    if split_tree and N % comm.world == 0:
        # every rank only builds the subtree over ITS block of leaves (positions [r N/w, (r+1) N/w), all cosets): it needs that
        # block's digests from every rank, a contiguous 1/w of each rank's digest array -- an all-to-all of (w-1)/w of the own
        # digests instead of an all-gather of everybody's (config 3, 8 ranks: 0.47 GB received per rank and stage, not 3.8 GB)
        digests = be.as_torch(be.linear_hash_rows(local, width, rows)).reshape(-1)
        return ShardedTree(be, None, N, cc, comm, block_parts=comm.all_to_all(digests))
    if chunks is None:
        chunks = 4 if (comm.mode == "nccl" and rows >= (1 << 16) and hasattr(be, "linear_hash_rows_into")) else 1
    if chunks == 1:
        digests = be.linear_hash_rows(local, width, rows)       # [N*cc][4], local row = pos*cc + jl
        gathered = comm.all_gather(be.as_torch(digests).reshape(-1))
    else:
        mine = be.empty(rows * 4)
        gathered = [be.empty(rows * 4) for _ in range(comm.world)]
        mt, gt = be.as_torch(mine), [be.as_torch(g) for g in gathered]      # (views of the same memory: the exchange layer speaks torch)
        pending = []
        for k in range(chunks):
            r0, r1 = rows * k // chunks, rows * (k + 1) // chunks
            be.linear_hash_rows_into(local[r0 * width:r1 * width], width, r1 - r0, mine[r0 * 4:r1 * 4])
            pending.append(comm.all_gather_start(mt[r0 * 4:r1 * 4], [g[r0 * 4:r1 * 4] for g in gt]))
        for h in pending:
            h.wait()
        gathered = gt
    # part r is [N][cc*4]; leaf index = pos*2^eb + r*cc + jl (natural row order of the extension), written straight into
    # the node array (no stacked / permuted copies: at config 5 the leaf level alone is 17 GB)
    if split_tree:
        return ShardedTree(be, gathered, N, cc, comm)
    return be.merkelize_digest_parts(gathered, N, cc)


class ShardedTree:
    """A Merkle tree (merklehash_p.js layout, power-of-two height) whose levels above the leaves are split by CONTIGUOUS
    leaf blocks: rank r keeps the subtree over leaves [r*E/w, (r+1)*E/w) (built from the gathered leaf digests), the w
    subtree roots are exchanged and the log2(w) levels above them are computed by everybody.  Same root and same paths as
    the single tree; no rank hashes more than 1/w of it (plus w-1 nodes)."""

    def __init__(self, be, parts, N, cc, comm, block_digests=None, block_parts=None):
        """parts: the all-gathered coset-ordered leaf digests of a committed stage (commit_local_slice); or block_parts: from
        every rank the digests of THIS rank's positions only ([N/w][cc*4] each: an all-to-all instead of an all-gather); or
        block_digests: the digests of this rank's own contiguous block of N*cc leaves (nothing to exchange)"""
        self.be, self.comm, self.rank, self.world = be, comm, comm.rank, comm.world
        self.block = N * cc                                     # leaves per rank block: E / world
        if block_digests is not None:
            self.sub = be.merkelize_digests(block_digests, self.block)
        elif block_parts is not None:
            self.sub = be.merkelize_digest_block(block_parts, N, cc, self.rank, sliced=True)
        else:
            if N % self.world:
                raise ValueError("world size must divide the number of rows")
            self.sub = be.merkelize_digest_block(parts, N, cc, self.rank)
        mine = torch.zeros(self.world * 4, dtype=torch.int64)
        mine[self.rank * 4:self.rank * 4 + 4] = torch.tensor(np.array(be.root({"nodes": self.sub}), dtype=np.uint64).view(np.int64))
        roots = comm.all_reduce_sum(mine) if comm.mode != "rehearse" else mine[:4].repeat(self.world)
        cur = roots.numpy().view(np.uint64).reshape(self.world, 4).tolist()
        self.top = [cur]
        if self.world > 1 and hasattr(be, "merkelize_digests"):  # the log2(w) levels above the subtree roots in ONE device call (merklehash_p.js:109-132)
            top = be.as_torch(be.merkelize_digests(be.from_torch(roots), self.world)).cpu().numpy().view(np.uint64).reshape(-1, 4)
            o, n = self.world, self.world // 2
            while n >= 1:
                cur = top[o:o + n].tolist(); self.top.append(cur); o += n; n //= 2
        else:
            while len(cur) > 1:
                cur = [[int(v) for v in be.poseidon(cur[2 * i] + cur[2 * i + 1], [0, 0, 0, 0], 4)] for i in range(len(cur) // 2)]
                self.top.append(cur)
        self.root = cur[0]

    def siblings_local(self, idxs):
        """this rank's share of the lower levels of the paths to leaves idxs (zeros for leaves of other blocks): host tensor
        [len(idxs)][log2(block)][4]"""
        low = _log2(self.block)
        t = torch.zeros((len(idxs), max(low, 1), 4), dtype=torch.int64)
        mine = [(q, i % self.block) for q, i in enumerate(idxs) if i // self.block == self.rank]
        if mine and low:
            sib = self.be.merkle_siblings(self.sub, self.block, [li for _, li in mine])
            t[torch.tensor([q for q, _ in mine]), :low] = torch.from_numpy(np.array(sib, dtype=np.uint64).view(np.int64).reshape(len(mine), low, 4))
        return t

    def siblings_finish(self, idxs, summed):
        """summed: the sum over the ranks of siblings_local (host tensor) -> the sibling digests of every path: the lower
        levels from the owners, the upper log2(w) levels from the replicated top"""
        low = _log2(self.block)
        a = summed.numpy().view(np.uint64).reshape(len(idxs), max(low, 1), 4)
        out = []
        for q, i in enumerate(idxs):
            mp = a[q, :low].tolist()
            b = i // self.block
            for level in self.top[:-1]:
                mp.append(list(level[b ^ 1]))
                b >>= 1
            out.append(mp)
        return out

    def siblings(self, idxs):
        """sibling digests of the paths to leaves idxs (one sum over the ranks)"""
        return self.siblings_finish(idxs, self.comm.all_reduce_sum(self.siblings_local(idxs)))


def extend_and_merkelize_sharded(be, src, n_pols, n_bits, n_bits_ext, group=None, overwrite_src=False, rehearse_world=None, split_tree=False, comm=None):
    """Sharded extendAndMerkelize.  `src` = the full N x n_pols trace on every rank.
    Returns {"local": N x (cc*n_pols) slice (row pos, coset jl, col c), "nodes": full tree.nodes, "width", "height",
    "cosetBegin", "cosetCount", "extBits"}; tree root = last 4 words of nodes, identical on all ranks and to the
    single-GPU merkelize of the full extension.
    overwrite_src: let the LDE use the trace buffer for its coefficient matrix (config 5: 107 GB trace + 107 GB slice per
    GPU, no third buffer).  rehearse_world=K: run rank 0's share of a K-rank job alone, standing in copies of the own
    digests for the gathered ones (a one-GPU rehearsal of the per-GPU time and memory; the tree is not a real root).
    split_tree: instead of the full node array on every rank ("nodes"), return "tree": a ShardedTree -- each rank builds the
    subtree over its block of leaves only and the top log2(world) levels are replicated (same root, same paths).
    comm: the exchange layer (Comm); built from group / rehearse_world when not given."""
    comm = _comm_of(comm, group, rehearse_world)
    rank, world = comm.rank, comm.world
    eb = n_bits_ext - n_bits
    cb, cc = coset_range(rank, world, eb)
    N = 1 << n_bits
    local = be.empty(N * cc * n_pols)
    be.interpolate_cosets(src, n_pols, n_bits, local, n_bits_ext, cb, cc, src if overwrite_src else None)
    tree = commit_local_slice(be, local, n_pols, n_bits, cc, comm, split_tree=split_tree)
    height = N << eb
    out = {"local": local, "width": n_pols, "height": height, "cosetBegin": cb, "cosetCount": cc, "extBits": eb}
    out["tree" if split_tree else "nodes"] = tree
    return out


def owner_of_row(idx, ext_bits, world):
    """(rank, local row) holding extended row idx"""
    cc = (1 << ext_bits) // world
    j = idx & ((1 << ext_bits) - 1)
    return j // cc, (idx >> ext_bits) * cc + (j % cc)


def open_rows_local(be, stree, idxs, rank, world):
    """this rank's rows among the extended rows idxs (zeros for rows of other ranks' cosets): int64 tensor [len(idxs)][width]
    on the slice's device"""
    w = stree["width"]
    loc = be.as_torch(stree["local"]).reshape(-1, w)
    sel = [(q, owner_of_row(i, stree["extBits"], world)) for q, i in enumerate(idxs)]
    out = torch.zeros((len(idxs), w), dtype=torch.int64, device=loc.device)
    own = [(q, lr) for q, (r, lr) in sel if r == rank]
    if own:
        qi = torch.tensor([q for q, _ in own], device=loc.device)
        li = torch.tensor([lr for _, lr in own], device=loc.device)
        out[qi] = loc[li].to(torch.int64)
    return out


def open_rows(be, stree, idxs, group=None, comm=None):
    """values of the extended rows idxs (fri.js:83-105 opens every tree at the query rows): each rank fills the rows of
    its own cosets, one sum (with zeros, exact) completes them everywhere.  -> numpy [len(idxs)][width]"""
    comm = _comm_of(comm, group, None)
    return comm.all_reduce_sum(open_rows_local(be, stree, idxs, comm.rank, comm.world)).numpy().view(np.uint64)


# ------------------------------------------------------------------------------------------------------------------------
# One proof over several GPUs: every rank runs the same transcript (all absorbed values are
# replicated), the big stage-1 objects stay split by cosets, and three small exchanges complete the proof:
#   all-gather of leaf digests (commit), all-gather of q on the extended domain (3 words per row), all-gather of the FRI
#   polynomial (3 words per row); the evaluations come from the rank that owns coset 0.
# A local slice holds rows (pos, jl) = full row pos*2^b + cosetBegin + jl, pos-major; "next row" (prime 1 = +2^b rows) is
# +cosetCount rows in a slice, so the evaluator runs on it with its prime shift set to log2(cosetCount).
def _log2(n):
    b = n.bit_length() - 1
    if 1 << b != n:
        raise ValueError("%d is not a power of two" % n)
    return b


def coset_slice(be, full, n_bits, ext_bits, cb, cc, width):
    """rows of cosets [cb, cb+cc) of a full (2^(n+b) x width) buffer, in local-slice order"""
    t = be.as_torch(full).reshape(1 << n_bits, 1 << ext_bits, width)
    return be.from_torch(t[:, cb:cb + cc, :].contiguous().reshape(-1))


def x_slice(be, n_bits, ext_bits, cb, cc, shift):
    """rows of cosets [cb, cb+cc) of x_ext (stark_gen_helpers.js:139-144: x[i] = shift w_E^i) in local-slice order, built coset by
    coset -- column jl is the geometric sequence (shift w_E^(cb+jl)) w_N^pos -- where the backend can (a rank then never holds a
    whole extended column: at config 5 that table alone is 4.3 GB); sliced out of the full table otherwise"""
    from . import stark as S
    if not hasattr(be, "geometric"):
        return coset_slice(be, be.build_x(n_bits + ext_bits, shift), n_bits, ext_bits, cb, cc, 1)
    wE, wN = S.root_of_unity(n_bits + ext_bits), S.root_of_unity(n_bits)
    cols = [be.geometric(shift * pow(wE, cb + jl, S.P) % S.P, wN, 1 << n_bits) for jl in range(cc)]
    if cc == 1:
        return cols[0]
    return be.from_torch(torch.stack([be.as_torch(c) for c in cols], dim=1).reshape(-1))


def zhinv_slice(be, n_bits, ext_bits, cb, cc):
    """the same rows of buildZhInv's table (polutils.js:39-55): 1 / ((7 w_E^row)^N - 1) depends on the coset of the row only"""
    from . import stark as S
    sN = pow(S.SHIFT, 1 << n_bits, S.P)
    wc = S.root_of_unity(ext_bits) if ext_bits else 1                      # w_E^N
    z = [S._inv((sN * pow(wc, cb + jl, S.P) - 1) % S.P) for jl in range(cc)]
    return be.from_torch(be.as_torch(be.from_host(np.array(z, dtype=np.uint64))).repeat(1 << n_bits))     # tiled where the backend's memory is


def zi_slice(be, boundary, n_bits, ext_bits, cb, cc, x_loc):
    """the rows of cosets [cb, cb+cc) of one boundary's column of Zi_ext (stark_gen_helpers.js:146-160, polutils.js:39-102) in local-slice
    order, built FROM THE RANK'S OWN ROWS: no table of 2^nBitsExt rows exists on the way.
      everyRow   1 / (x^N - 1): depends on the coset only (zhinv_slice);
      everyFrame prod (x - root_k): the evaluator over the rank's x rows;
      firstRow / lastRow  (x^N - 1) / (x - root): x / (x - root) by the batched-inversion kernel of the FRI table
                 (pil2gl_x_div_x_sub_xi_cosets_dev with the point (root, 0, 0)), times 1 / x -- a geometric sequence per coset -- times
                 the coset's x^N - 1.
    Backends without those operators (the CPU checker) slice the whole table."""
    from . import stark as S
    name = boundary["name"]
    if name == "everyRow":
        return zhinv_slice(be, n_bits, ext_bits, cb, cc)
    N, lb = 1 << n_bits, _log2(cc)
    if not (hasattr(be, "geometric") and hasattr(be, "x_div_x_sub_xi_cosets") and cc & (cc - 1) == 0):
        return coset_slice(be, S.build_zi_table(be, boundary, n_bits, n_bits + ext_bits), n_bits, ext_bits, cb, cc, 1)
    wN, wE = S.root_of_unity(n_bits), S.root_of_unity(n_bits + ext_bits)
    out = be.empty(N * cc)
    if name == "everyFrame":
        roots = [pow(wN, i, S.P) for i in range(boundary["offsetMin"])] + [pow(wN, N - i - 1, S.P) for i in range(boundary["offsetMax"])]
        if not roots:
            return be.from_torch(torch.ones(N * cc, dtype=torch.int64, device=be.as_torch(out).device))
        ops = []
        for k in range(len(roots)):
            ops.append((S.OPC["sub"], (S.TMP, 1, 0, 0, 1), (S.SEC, 1, 0, 0, 0), (S.SCALAR, 1, 0, 0, k)))
            if k == 0:
                ops.append((S.OPC["copy"], (S.TMP, 1, 0, 0, 0), (S.TMP, 1, 0, 0, 1), None))
            else:
                ops.append((S.OPC["mul"], (S.TMP, 1, 0, 0, 0), (S.TMP, 1, 0, 0, 0), (S.TMP, 1, 0, 0, 1)))
        ops.append((S.OPC["copy"], (S.SEC, 1, 1, 0, 0), (S.TMP, 1, 0, 0, 0), None))
        be.eval_program(ops, 2, [(x_loc, 1), (out, 1)], np.array(roots, dtype=np.uint64), n_bits + lb, 0)
        return out
    if name not in ("firstRow", "lastRow"):
        raise ValueError("Boundary " + str(name) + " not supported")
    root = 1 if name == "firstRow" else pow(wN, N - 1, S.P)
    xdiv = be.x_div_x_sub_xi_cosets(n_bits + ext_bits, ext_bits, [[root, 0, 0]], cb, cc)        # [row][3]: x / (x - root), components 1, 2 zero
    sN = pow(S.SHIFT, N, S.P)
    wc = S.root_of_unity(ext_bits) if ext_bits else 1
    cols = [be.geometric(S._inv(S.SHIFT * pow(wE, cb + jl, S.P) % S.P) * ((sN * pow(wc, cb + jl, S.P) - 1) % S.P) % S.P, S._inv(wN), N) for jl in range(cc)]
    k = cols[0] if cc == 1 else be.from_torch(torch.stack([be.as_torch(c) for c in cols], dim=1).reshape(-1))    # (x^N - 1) / x, row by row
    ops = [(S.OPC["mul"], (S.SEC, 1, 2, 0, 0), (S.SEC, 1, 0, 0, 0), (S.SEC, 1, 1, 0, 0))]
    be.eval_program(ops, 1, [(xdiv, 3), (k, 1), (out, 1)], np.zeros(1, dtype=np.uint64), n_bits + lb, 0)
    return out


def shard_tables(be, setup, info, cb, cc):
    """the rank's rows of the tables that depend on the setup and the coset range only -- the constants' extension, x, the zerofiers --,
    built once per (setup, coset range) and kept with the setup: a prover that proves again and again with one setup (the bench, a
    service) pays for them once (they were 8 % of a config-3 proof's rank share)"""
    from . import stark as S
    ss = info["starkStruct"]
    nb, eb = ss["nBits"], ss["nBitsExt"] - ss["nBits"]
    cache = setup.setdefault("_shardTables", {})
    # the key names everything the tables depend on: the coset range, the backend that holds them (a CPU checker run after a GPU run with the
    # same setup must not be handed device tensors) and the AIR's domain, constants and boundaries
    key = (cb, cc, getattr(be, "name", type(be).__name__), str(getattr(be, "dev", "")), nb, eb, info["nConstants"],
           json.dumps(info.get("boundaries", [{"name": "everyRow"}]), sort_keys=True))
    if key not in cache:
        constShard, constTree = setup.get("constShard"), setup.get("constTree")
        t = {"const_ext": constShard["local"] if constShard is not None else coset_slice(be, constTree["elements"], nb, eb, cb, cc, info["nConstants"]),
             "x_ext": x_slice(be, nb, eb, cb, cc, S.SHIFT)}
        for bi, boundary in enumerate(info.get("boundaries", [{"name": "everyRow"}])):
            t["Zi_ext#%d" % bi] = zi_slice(be, boundary, nb, eb, cb, cc, t["x_ext"])
        cache[key] = t
    return dict(cache[key])


def clear_shard_tables(setup):
    """drops the tables shard_tables keeps with a setup (several GB of HBM per rank at config 5: x, every zerofier column and, without a
    sharded constant tree, a copy of the constants' slice); the next proof builds them again"""
    setup.pop("_shardTables", None)


def shard_tables_bytes(be, setup):
    """bytes the kept tables hold on this rank (for the peak-memory figures)"""
    n = 0
    for t in setup.get("_shardTables", {}).values():
        for k, v in t.items():
            if k == "const_ext" and setup.get("constShard") is not None:
                continue                                      # the sharded constant tree's own slice, not a copy
            n += int(be.as_torch(v).numel()) * 8
    return n


def build_const_tree_sharded(be, consts, info, group=None, rehearse_world=None, comm=None):
    """buildConstTree (stark_buildConstTree.js:13-35) for a coset-sharded proof: every rank extends the constants on its own cosets
    and hashes its own leaves; the tree above them is split by leaf blocks (ShardedTree).  Same root as the single tree; no rank
    holds the extended constants or the node array whole (config 5, two constant columns: 8.6 + 34 GB otherwise).  Collective:
    every rank of the group calls it.  -> setup for stark_gen_sharded"""
    comm = _comm_of(comm, group, rehearse_world)
    ss = info["starkStruct"]
    nb, nbe, nC = ss["nBits"], ss["nBitsExt"], info["nConstants"]
    eb = nbe - nb
    cb, cc = coset_range(comm.rank, comm.world, eb)
    const_n = be.from_host(consts) if isinstance(consts, np.ndarray) else consts
    local = be.empty((nC << nb) * cc)
    be.interpolate_cosets(const_n, nC, nb, local, nbe, cb, cc, None)
    tree = commit_local_slice(be, local, nC, nb, cc, comm, split_tree=True)
    shard = {"local": local, "width": nC, "height": 1 << nbe, "cosetBegin": cb, "cosetCount": cc, "extBits": eb}
    return {"constTree": None, "constRoot": tree.root, "const_n": const_n, "constShard": shard, "constTreeSharded": tree}


def all_gather_rows(be, local, n_bits, cc, width, comm):
    """local slices (N*cc rows x width) of every rank -> the full buffer in natural row order, on every rank"""
    parts = comm.all_gather(be.as_torch(local).reshape(-1))
    N = 1 << n_bits
    full = torch.stack([p.reshape(N, cc * width) for p in parts], dim=1).reshape(-1)      # [N][world][cc*width]
    return be.from_torch(full)


def quotient_coefficients_sharded(be, q_loc, nb, eb, cb, cc, qDim, qDeg, comm, brev=False):
    """computeQStark's first half (stark_gen_helpers.js:168-190: qq1 = ifft over the extended domain of q, then the split into qDeg
    chunks of N coefficients scaled by shift^(-N p)) WITHOUT gathering q: -> [N][qDeg*qDim] on every rank, the first N rows of
    pil2gl_compute_q_split_dev's result.  With r = 2^eb pos + j the size-E inverse transform factors by cosets,
        c'_(pN+i) = 2^-eb  sum_j  w_2^eb^(-pj)  w_E^(-ij)  C_j[i],       C_j = ifft_N(q on coset j),
    so a rank transforms ITS cosets (N rows each), multiplies by w_E^(-ij), and the sum over the cosets is taken where row i
    lives: rows are dealt over the ranks in blocks (an all-to-all of 1/w of everybody's rows), the qDeg chunks that are
    kept are combined there, and the blocks are all-gathered.  Exchanged per rank at config 3 on 8 ranks: 0.35 + 0.7 GB
    instead of the 2.8 GB of q, and no rank runs the size-E transform."""
    from . import stark as S
    N, E, w = 1 << nb, 1 << (nb + eb), comm.world
    nblk = N // w
    qt = be.as_torch(q_loc).reshape(N, cc, qDim)
    geo = hasattr(be, "geometric")
    wE_inv = S._inv(S.root_of_unity(nb + eb))
    if not geo:
        x_e = be.as_torch(be.build_x(nb + eb, 1))                       # w_E^i
        ar = torch.arange(N, dtype=torch.int64, device=x_e.device)
    T = []
    for jl in range(cc):
        j = cb + jl
        qj = be.from_torch(qt[:, jl, :].contiguous().reshape(-1))
        Cj = be.empty(qDim << nb)
        be.ifft(qj, qDim, nb, Cj)
        tw = be.geometric(1, pow(wE_inv, j, S.P), N) if geo else be.from_torch(x_e[((E - j) * ar) % E].contiguous())   # w_E^(-i j)
        Tj = be.empty(qDim << nb)
        ops = [(S.OPC["mul"], (S.SEC, 1, 2, 0, k), (S.SEC, 1, 0, 0, k), (S.SEC, 1, 1, 0, 0)) for k in range(qDim)]
        be.eval_program(ops, 1, [(Cj, qDim), (tw, 1), (Tj, qDim)], np.zeros(1, np.uint64), nb, 0)
        T.append(be.as_torch(Tj).reshape(w, nblk, qDim))
        del tw, Cj, qj
    if not geo:
        del x_e, ar
    mine = torch.stack(T, dim=1).reshape(-1).contiguous()               # [w][cc][nblk][qDim]: chunk s = my cosets' rows of block s
    parts = comm.all_to_all(mine)                                       # parts[s] = [cc][nblk][qDim] of rank s's cosets
    G = torch.stack([p_.reshape(cc, nblk, qDim) for p_ in parts], dim=0).reshape(w * cc, nblk, qDim).permute(1, 0, 2).contiguous()   # [nblk][2^eb][qDim]
    nco = w * cc
    inv_n = S._inv(nco)
    w_c = pow(S.root_of_unity(nb + eb), E // nco, S.P)                   # w_(2^eb)
    s_in = pow(S._inv(S.SHIFT), N, S.P)
    scal = np.array([inv_n * pow(s_in, p_, S.P) % S.P * pow(w_c, (-p_ * j) % nco, S.P) % S.P for p_ in range(qDeg) for j in range(nco)], dtype=np.uint64)
    ops = []
    for p_ in range(qDeg):
        for k in range(qDim):
            ops.append((S.OPC["mul"], (S.TMP, 1, 0, 0, 0), (S.SEC, 1, 0, 0, k), (S.SCALAR, 1, 0, 0, p_ * nco)))
            for j in range(1, nco):
                ops.append((S.OPC["mul"], (S.TMP, 1, 0, 0, 1), (S.SEC, 1, 0, 0, j * qDim + k), (S.SCALAR, 1, 0, 0, p_ * nco + j)))
                dest = (S.SEC, 1, 1, 0, p_ * qDim + k) if j == nco - 1 else (S.TMP, 1, 0, 0, 0)
                ops.append((S.OPC["add"], dest, (S.TMP, 1, 0, 0, 0), (S.TMP, 1, 0, 0, 1)))
            if nco == 1:
                ops.append((S.OPC["copy"], (S.SEC, 1, 1, 0, p_ * qDim + k), (S.TMP, 1, 0, 0, 0), None))
    blk = be.empty(nblk * qDeg * qDim)
    be.eval_program(ops, 2, [(be.from_torch(G.reshape(-1)), nco * qDim), (blk, qDeg * qDim)], scal, _log2(nblk), 0)
    if not brev:
        return be.from_torch(torch.cat([p_.reshape(-1) for p_ in comm.all_gather(be.as_torch(blk).reshape(-1))]))
    # the same matrix with coefficient i at row bitrev(i) (what pil2gl_extend_coefs_brev_cosets_dev reads): row i = r nblk + l sits at
    # bitrev(l) w + bitrev(r) -- each rank reverses its own block's rows, the gathered blocks interleave in reversed rank order
    lb_, lw_ = _log2(nblk), _log2(w)
    bt = be.as_torch(blk).reshape(nblk, qDeg * qDim)
    parts = comm.all_gather(bt[_bitrev_index(lb_, bt.device)].reshape(-1).contiguous())
    order = [int(format(r, "0%db" % lw_)[::-1], 2) if lw_ else 0 for r in range(w)]
    return be.from_torch(torch.stack([parts[order[r]].reshape(nblk, qDeg * qDim) for r in range(w)], dim=1).reshape(-1).contiguous())


_BITREV = {}


def _bitrev_index(bits, device):
    """bitrev(i) for i < 2^bits as an index tensor on `device` (kept: one per size and device)"""
    key = (bits, str(device))
    if key not in _BITREV:
        i = torch.arange(1 << bits, dtype=torch.int64, device=device)
        r = torch.zeros_like(i)
        for b in range(bits):
            r |= ((i >> b) & 1) << (bits - 1 - b)
        _BITREV[key] = r
    return _BITREV[key]


def _evals_by_opening(be, S, info, loc, widths, xis, nb, nbe, lb, cb, rank, world, comm, nC):
    """the evaluations with the opening points dealt over the ranks (rank i mod world takes opening i on its own first coset, the
    others contribute zeros, one all-reduce): the path of backends without column-range sums (the CPU checker)"""
    n_ev = len(info["evMap"])
    mine = [i for i in range(len(xis)) if i % world == rank]
    ev_t = torch.zeros(n_ev * 3, dtype=torch.int64)
    if mine:
        g_inv = S._inv(S.SHIFT * pow(S.root_of_unity(nbe), cb, S.P) % S.P)          # 1 / (7 w_E^cb): this rank's first coset
        levs = [be.build_lev(nb, S.ext_scale(xis[i], g_inv)) for i in mine]
        sel = [k for k, ev in enumerate(info["evMap"]) if info["openingPoints"].index(ev["prime"]) in mine]
        sub = dict(info); sub["evMap"] = [info["evMap"][k] for k in sel]; sub["openingPoints"] = [info["openingPoints"][i] for i in mine]
        if hasattr(be, "evals_fast") and len(levs) <= 4:
            evals = be.evals_fast(sub, loc, widths, nb, lb, levs)          # row k of the coset is local row k << log2(cc)
        else:
            descs = []
            for ev in sub["evMap"]:
                li = sub["openingPoints"].index(ev["prime"])
                if ev["type"] == "const":
                    descs.append((loc["const_ext"], nC, ev["id"], 1, li))
                else:
                    p = info["cmPolsMap"][ev["id"]]
                    descs.append((loc["cm%d_ext" % p["stage"]], widths["cm%d_ext" % p["stage"]], p["stagePos"], p["dim"], li))
            evals = be.compute_evals(descs, nb, lb, levs)
        full = np.zeros((n_ev, 3), dtype=np.uint64)
        full[sel] = np.array(evals, dtype=np.uint64).reshape(len(sel), 3)
        ev_t = torch.from_numpy(full.reshape(-1).view(np.int64).copy())
        del levs
    ev_t = comm.all_reduce_sum(ev_t)                                      # every entry is non-zero on one rank only
    return [[int(v) for v in r] for r in ev_t.numpy().view(np.uint64).reshape(n_ev, 3)]


def stark_gen_sharded(be, cm1_n, setup, info, exprs, publics, group=None, rehearse_world=None, timings=None, comm=None, overwrite_trace=False, samples=None):
    """pil2gl.stark.stark_gen with every witness stage, the constraint evaluation and the FRI polynomial split by cosets over the
    ranks of `group`.  Every rank passes the same trace and setup and receives the same (complete) proof, identical to the
    single-process one.  Replicated: the N-row transform of the split quotient and the FRI steps after the first fold (the
    quotient's coefficients come from per-coset transforms combined by row blocks, the first FRI tree's leaves and the first
    fold are computed by cosets); the stage trees above the leaves are split by leaf blocks (ShardedTree).
    rehearse_world=K: rank 0's share of a K-rank proof run alone (own slices stand in for the gathered ones, so the
    result is not a valid proof): per-GPU time and memory on one GPU.  timings: dict that receives seconds per stage.
    setup: stark.build_const_tree's (the constant tree whole on every rank) or build_const_tree_sharded's (split like the witness
    trees: what a domain beyond one device's memory needs).  overwrite_trace: the last witness stage's trace buffer doubles as its
    LDE's coefficient workspace and is destroyed (config 5: 107 GB trace + 107 GB slice per GPU, no room for a third buffer).
    samples: {"rows": [local rows]} -> receives those rows of the rank's slices (cm1_ext, const_ext, x_ext, Zi_ext, q_ext): the
    full-size rehearsal test checks them against closed forms."""
    from . import stark as S
    import time
    t_last = [time.perf_counter()]

    def lap(name):
        Progress.mark("stage done: " + name, comm.rank)
        if timings is not None:
            be.sync(); now = time.perf_counter(); timings[name] = timings.get(name, 0.0) + now - t_last[0]; t_last[0] = now
    comm = _comm_of(comm, group, rehearse_world)
    rank, world = comm.rank, comm.world
    rehearse_world = world if comm.mode == "rehearse" else None
    ss = info["starkStruct"]
    nb, nbe = ss["nBits"], ss["nBitsExt"]
    eb, N, E = nbe - nb, 1 << ss["nBits"], 1 << ss["nBitsExt"]
    cb, cc = coset_range(rank, world, eb)
    lb = _log2(cc)
    nloc = nb + lb                                              # local slices have 2^nloc rows
    qDim, qDeg = info["qDim"], info["qDeg"]
    nStages = info["nStages"]
    qStage = nStages + 1
    nQ, nC = info["mapSectionsN"]["cm%d" % qStage], info["nConstants"]
    assert ss["steps"][0]["nBits"] == nbe
    # the quotient's coefficients come by cosets and row blocks (no buffer of 2^nBitsExt rows) when the partition allows it; otherwise q is
    # all-gathered and transformed whole on every rank -- which a domain beyond one transform, or beyond the memory for a replicated
    # 2^nBitsExt x qDim buffer, cannot do: say so before any stage has been committed
    import os
    split_q = os.environ.get("PIL2GL_Q_GATHER", "0") != "1" and N % world == 0 and (N // world) >= 2 and cc & (cc - 1) == 0
    if not split_q and nbe > MAX_NTT_BITS:
        raise ValueError("sharded proof: 2^%d rows over %d ranks would all-gather the quotient and transform 2^%d rows on every rank (limit 2^%d); "
                         "the coset-wise path needs world | N and a power-of-two coset count per rank" % (nb, world, nbe, MAX_NTT_BITS))
    ctx = {"pilInfo": info, "publics": list(publics), "challenges": [[] for _ in range(nStages + 3)], "evals": [], "subproofValues": [0] * info.get("nSubproofValues", 0)}
    constTree = setup.get("constTree")
    constShard, constSTree = setup.get("constShard"), setup.get("constTreeSharded")
    transcript = be.new_transcript()
    hash_commits = bool(ss.get("hashCommits", False))
    transcript.put(setup["constRoot"])

    sl = lambda full, w: coset_slice(be, full, nb, eb, cb, cc, w)
    # the rank's rows of the domain tables, built per coset and once per setup.  With the HIP backend and a power-of-two coset count per rank
    # no buffer of this function has 2^nBitsExt rows; the CPU checker (no per-coset operators) and odd partitions slice whole tables
    loc = shard_tables(be, setup, info, cb, cc)
    widths = {"const_n": nC, "const_ext": nC, "q_ext": qDim, "f_ext": 3, "x_ext": 1, "x_n": 1,
              "xDivXSubXi_ext": 3 * len(info["openingPoints"])}
    for bi in range(len(info.get("boundaries", [{"name": "everyRow"}]))):
        widths["Zi_ext#%d" % bi] = 1
    for s_ in range(1, qStage + 1):
        widths["cm%d_n" % s_] = widths["cm%d_ext" % s_] = info["mapSectionsN"]["cm%d" % s_]

    def run_local(code):
        ops, n_tmp, secs, scalars = S.encode_code(code["code"], "ext", ctx)
        be.eval_program(ops, n_tmp, [(loc[s], widths[s]) for s in secs], scalars, nloc, lb)

    lap("tables")
    # witness stages, split by cosets; each tree is split by leaf blocks (ShardedTree), the rows stay with their owners.
    # From stage 2 on (prover.js:49-77): challenges, stage code and hints on the trace domain, replicated - it is N rows
    # against the N * 2^b of the extension, and every rank needs the whole stage-s trace for its own cosets anyway
    trace = {"const_n": setup.get("const_n"), "cm1_n": cm1_n}
    strees, shards, roots = {}, {}, {}
    for s_ in range(1, nStages + 1):
        name, w = "cm%d" % s_, widths["cm%d_n" % s_]
        if s_ > 1:
            n_ch = sum(1 for c in info["challengesMap"] if c["stage"] == s_)
            ctx["challenges"][s_ - 1] = [transcript.getField() for _ in range(n_ch)]
            trace[name + "_n"] = be.zeros(w << nb)
            if "x_n" not in trace:
                trace["x_n"] = be.build_x(nb, 1)
            for sc in exprs.get("stageCode", {}).get(s_, []):
                ops, n_tmp, secs, scalars = S.encode_code(sc["code"], "n", ctx)
                be.eval_program(ops, n_tmp, [(trace[x], widths[x]) for x in secs], scalars, nb, 0)
            S.resolve_hints(be, info, s_, trace, widths, nb, ctx, exprs)
        else:                                              # stage 1: its hints (publics read off the witness), then the publics (prover.js:41-52)
            if exprs.get("hintsInfo"):
                S.resolve_hints(be, info, 1, trace, widths, nb, ctx, exprs)
            S.put_commit(be, transcript, list(ctx["publics"]), hash_commits)
        im = exprs.get("imPolsCode", [])
        if s_ == nStages and len(im) >= s_ and im[s_ - 1].get("code"):      # intermediate polynomials (prover.js:212-214), replicated like the stage code
            ops, n_tmp, secs, scalars = S.encode_code(im[s_ - 1]["code"], "n", ctx)
            be.eval_program(ops, n_tmp, [(trace[x], widths[x]) for x in secs], scalars, nb, 0)
        loc[name + "_ext"] = be.empty(w << nloc)
        be.interpolate_cosets(trace[name + "_n"], w, nb, loc[name + "_ext"], nbe, cb, cc, trace[name + "_n"] if (overwrite_trace and s_ == nStages) else None)
        lap("stage%d_lde" % s_)
        strees[s_] = commit_local_slice(be, loc[name + "_ext"], w, nb, cc, comm, split_tree=True)
        shards[s_] = {"local": loc[name + "_ext"], "width": w, "height": E, "cosetBegin": cb, "cosetCount": cc, "extBits": eb}
        roots[s_] = strees[s_].root; transcript.put(roots[s_])
        lap("stage%d_merkle" % s_)
    del trace

    # quotient: the constraint expression on the local rows, then one all-gather of q
    ctx["challenges"][qStage - 1] = [transcript.getField()]
    loc["q_ext"] = be.empty(qDim << nloc)
    run_local(S.expr_code(exprs, info["cExpId"]))
    if samples is not None:
        ri = torch.tensor(list(samples["rows"]))
        for k in ("cm1_ext", "const_ext", "x_ext", "Zi_ext#0", "q_ext"):
            t_ = be.as_torch(loc[k]).reshape(1 << nloc, -1)
            samples[k] = t_[ri.to(t_.device)].cpu().numpy().view(np.uint64)
    lap("q_expr")
    q_ext = None if split_q else all_gather_rows(be, loc["q_ext"], nb, cc, qDim, comm)
    # computeQStark (stark_gen_helpers.js:168-208): the coefficients of q by cosets and row blocks (quotient_coefficients_sharded;
    # PIL2GL_Q_GATHER=1: all-gather q and transform it everywhere); the split quotient has degree < N per column, so its
    # extension is again "one coset per rank": the own cosets of the unshifted extension straight from the coefficients every rank
    # holds (pil2gl_extend_coefs_brev_cosets_dev: no forward transform to the subgroup and back), own leaves, exchanged digests
    qname = "cm%d_ext" % qStage
    if split_q:                                                # the coefficients without gathering q (by cosets, then by row blocks),
        qq2 = quotient_coefficients_sharded(be, loc["q_ext"], nb, eb, cb, cc, qDim, qDeg, comm, brev=True)      # in the order the extension reads
        del loc["q_ext"]
        loc[qname] = be.empty(nQ << nloc)
        be.extend_coefs_brev_cosets(qq2, nQ, nb, loc[qname], nbe, cb, cc)           # own cosets straight from the coefficients every rank holds
        del qq2
    else:
        qq1 = be.empty(qDim << nbe)
        be.ifft(q_ext, qDim, nbe, qq1)
        qq2 = be.q_split(qq1, nb, nbe, qDim, qDeg)
        del qq1, q_ext
        del loc["q_ext"]
        q_sub = be.empty(nQ << nb)
        be.fft(qq2[:nQ << nb], nQ, nb, q_sub)                # rows >= N of qq2 are zero: these are all its coefficients
        del qq2
        loc[qname] = be.empty(nQ << nloc)
        be.extend_cosets_unshifted(q_sub, nQ, nb, loc[qname], nbe, cb, cc)
    lap("q_ntt")
    strees[qStage] = commit_local_slice(be, loc[qname], nQ, nb, cc, comm, split_tree=True)
    shards[qStage] = {"local": loc[qname], "width": nQ, "height": E, "cosetBegin": cb, "cosetCount": cc, "extBits": eb}
    roots[qStage] = strees[qStage].root; transcript.put(roots[qStage])
    lap("q_merkle")

    # evaluations (computeEvalsStark :210-273).  The reference reads rows k << b, i.e. coset 0 with LEv = iNTT of (xi w/7)^k;
    # any coset 7 w_E^c of the subgroup determines the same polynomial values with LEv_c = iNTT of (xi w / (7 w_E^c))^k, so
    # the opening points are dealt over the ranks: rank i mod world takes opening i on its own first coset, the others
    # contribute zeros, one all-reduce
    xi = transcript.getField()
    ctx["challenges"][qStage] = [xi]
    wN = S.root_of_unity(nb)
    xis, n_ev = [], len(info["evMap"])
    for opening in info["openingPoints"]:
        w = pow(wN, abs(opening), S.P)
        if opening < 0:
            w = S._inv(w)
        xis.append(S.ext_scale(xi, w))
    if hasattr(be, "col_sums"):
        # (opening, share of the columns) pairs dealt over the ranks: with w ranks and n openings each opening has w // n ranks, rank r takes
        # opening r % n and the (r // n)-th share of every matrix's columns.  Its weights are LEv on ITS first coset (a closed form: one
        # sweep, pil2gl_build_lev_dev), its cells the only ones it reads; the column sums of all ranks meet in one all-reduce
        # (n x sum(widths) x 3 words) and every rank derives the evaluations from them
        n_open = len(xis)
        names = be.eval_matrices(info)
        offs, tot = {}, 0
        for nm in names:
            offs[nm] = tot; tot += widths[nm]
        sums_t = torch.zeros(n_open * tot * 3, dtype=torch.int64)
        G = max(1, world // n_open)
        tasks = [(rank % n_open, rank // n_open)] if world >= n_open else [(i, 0) for i in range(n_open) if i % world == rank]
        g_inv = S._inv(S.SHIFT * pow(S.root_of_unity(nbe), cb, S.P) % S.P)          # 1 / (7 w_E^cb): this rank's first coset
        for o, c in tasks:
            if c >= G:
                continue
            lev = be.build_lev(nb, S.ext_scale(xis[o], g_inv))
            lap("evals_lev")
            ranges = {nm: (widths[nm] * c // G, widths[nm] * (c + 1) // G) for nm in names}
            part = be.col_sums(names, loc, widths, nb, lb, [lev], ranges)           # row k of the coset is local row k << log2(cc)
            v = sums_t.numpy().view(np.uint64).reshape(n_open, tot, 3)
            for nm in names:
                v[o, offs[nm]:offs[nm] + widths[nm]] = part[nm][0]
            del lev
            lap("evals_dot")
        sums_t = comm.all_reduce_sum(sums_t)                                          # every entry is non-zero on one rank only
        v = sums_t.numpy().view(np.uint64).reshape(n_open, tot, 3)
        ctx["evals"] = S.evals_from_col_sums(info, {nm: v[:, offs[nm]:offs[nm] + widths[nm]] for nm in names})
    else:
        ctx["evals"] = _evals_by_opening(be, S, info, loc, widths, xis, nb, nbe, lb, cb, rank, world, comm, nC)
    S.put_commit(be, transcript, ctx["evals"], hash_commits)

    lap("evals")
    # FRI polynomial on the local rows, then one all-gather
    vfs = [transcript.getField(), transcript.getField()]
    ctx["challenges"][qStage + 1] = vfs
    if hasattr(be, "x_div_x_sub_xi_cosets") and cc & (cc - 1) == 0:    # only this rank's rows of the table
        loc["xDivXSubXi_ext"] = be.x_div_x_sub_xi_cosets(nbe, eb, xis, cb, cc)
    else:
        loc["xDivXSubXi_ext"] = sl(be.x_div_x_sub_xi(nbe, xis), widths["xDivXSubXi_ext"])
    loc["f_ext"] = be.empty(3 << nloc)
    if not (hasattr(be, "fri_polynomial_fast") and be.fri_polynomial_fast(info, loc, widths, ctx["evals"], vfs[0], vfs[1], nloc, loc["f_ext"])):
        run_local(S.expr_code(exprs, info["friExpId"]))
    lap("fri_expr")
    # Folding.  The first FRI tree commits the UNFOLDED polynomial in groups {i 2^b1 + g : i < 2^(b0-b1)} (fri.js:62-74 at
    # step 0, where nothing is folded), and the first fold combines exactly those groups (fri.js:45-60 at step 1).  A group
    # lies in ONE coset (g mod 2^eb; b1 >= eb), and coset j's rows are the same matrix one size down: group g = 2^eb g' + j
    # is rows {i 2^(b1-eb) + g'} of the rank's slice, with sinv_g = shift^-1 w^-g = (shift^-1 w^-j) (w^(2^eb))^-g'.  So a rank
    # transposes, hashes and folds its own cosets with the ordinary kernels, and what travels is the 2^b1 leaf digests and the
    # 2^b1 folded values (0.23 GB at config 3) instead of the polynomial (3.2 GB); the tree above the leaves and every
    # later step (2^(b0-b1) times smaller) are replicated.
    steps = ss["steps"]
    fri_first = None
    if len(steps) >= 2 and steps[1]["nBits"] >= eb:
        b0, b1 = steps[0]["nBits"], steps[1]["nBits"]
        nX, gl_ = 1 << (b0 - b1), 1 << (b1 - eb)                         # group size; groups (= tree-1 leaves) per coset
        ch0 = transcript.getField()
        ft = be.as_torch(loc["f_ext"]).reshape(N, cc, 3)
        f_cos = [be.from_torch(ft[:, jl, :].contiguous().reshape(-1)) if cc > 1 else loc["f_ext"] for jl in range(cc)]
        tbs = [be.fri_transpose(fj, nb, b1 - eb) for fj in f_cos]          # [gl_][nX*3]: row g' = the values of group 2^eb g' + j
        dig = torch.stack([be.as_torch(be.linear_hash_rows(tb, 3 * nX, gl_)).reshape(gl_, 4) for tb in tbs], dim=1)   # [gl_][cc][4]
        lap("fri_first_leaves")
        stree1 = nodes1 = None
        if gl_ % world == 0:
            # the tree over the 2^b1 leaves split by leaf blocks like the stage trees: leaf g = g' 2^eb + coset is "position g', coset" of a
            # 2^(b1-eb)-row matrix, so each rank receives its block's digests (all-to-all) and builds 1/w of the tree
            stree1 = ShardedTree(be, None, gl_, cc, comm, block_parts=comm.all_to_all(dig.reshape(-1).contiguous()))
            root1 = stree1.root
        else:
            parts = comm.all_gather(dig.reshape(-1).contiguous())
            leaves = torch.stack([p_.reshape(gl_, cc * 4) for p_ in parts], dim=1).reshape(-1)      # leaf g = g' 2^eb + rank cc + jl
            nodes1 = be.merkelize_digests(be.from_torch(leaves), 1 << b1)
            root1 = be.root({"nodes": nodes1})
        lap("fri_first_tree")
        transcript.put(root1)
        ch1 = transcript.getField()
        w_inv = S._inv(S.root_of_unity(b0))
        sinv = S._inv(S.SHIFT)
        fold = torch.stack([be.as_torch(be.fri_fold(fj, nb, b1 - eb, sinv * pow(w_inv, cb + jl, S.P) % S.P, ch1)).reshape(gl_, 3)
                            for jl, fj in enumerate(f_cos)], dim=1)          # [gl_][cc][3]
        del f_cos, ft
        parts = comm.all_gather(fold.reshape(-1).contiguous())
        friPol = be.from_torch(torch.stack([p_.reshape(gl_, cc * 3) for p_ in parts], dim=1).reshape(-1))
        friTrees, friProof = [None] * len(steps), [{} for _ in range(len(steps) + 1)]
        friTrees[1] = {"nodes": nodes1, "sharded_leaves": True, "tree": stree1}
        friProof[1] = {"root": list(root1)}
        fri_first = (tbs, 3 * nX, nodes1, 1 << b1, stree1)
        lap("fri_first_fold")
        friTrees, friProof, challengesFRI = S.fri_commit_phase(be, ss, None, transcript, resume=(1, friPol, friTrees, friProof, [ch0, ch1]))
    else:
        f_ext = all_gather_rows(be, loc["f_ext"], nb, cc, 3, comm)
        friTrees, friProof, challengesFRI = S.fri_commit_phase(be, ss, f_ext, transcript)
    lap("fri_fold")
    chq = transcript.getField(); challengesFRI.append(chq)
    tq = be.new_transcript(); tq.put(chq)
    queries = tq.getPermutations(ss["nQueries"], ss["steps"][0]["nBits"])
    # every opened row and every lower sibling comes from the one rank that owns it: all of them (the witness stages, the
    # quotient, the split FRI tree) travel in ONE sum over the ranks
    q1 = [qi % (1 << ss["steps"][1]["nBits"]) for qi in queries] if len(ss["steps"]) > 1 else []
    pieces = []
    for s_ in range(1, qStage + 1):
        w = shards[s_]["width"]
        pieces.append(open_rows_local(be, shards[s_], queries, rank, world).cpu().reshape(-1) if not rehearse_world else torch.zeros(len(queries) * w, dtype=torch.int64))
        pieces.append(strees[s_].siblings_local(queries).reshape(-1))
    if constSTree is not None:                                         # the constant tree is split like the others: its rows and lower siblings too
        pieces.append(open_rows_local(be, constShard, queries, rank, world).cpu().reshape(-1) if not rehearse_world else torch.zeros(len(queries) * nC, dtype=torch.int64))
        pieces.append(constSTree.siblings_local(queries).reshape(-1))
    if fri_first is not None:                                          # the opened groups of the first FRI tree, from the cosets' owners
        tbs, w1, nodes1, h1, stree1 = fri_first
        if stree1 is not None:
            pieces.append(stree1.siblings_local(q1).reshape(-1))
        g1 = torch.zeros((len(q1), w1), dtype=torch.int64)
        own = [(k, (g >> eb), (g & ((1 << eb) - 1)) - cb) for k, g in enumerate(q1) if cb <= (g & ((1 << eb) - 1)) < cb + cc]
        for jl in range(cc):
            sel = [(k, gp) for k, gp, j_ in own if j_ == jl]
            if sel:
                tt = be.as_torch(tbs[jl]).reshape(-1, w1)
                g1[torch.tensor([k for k, _ in sel])] = tt[torch.tensor([gp for _, gp in sel], device=tt.device)].cpu()
        pieces.append(g1.reshape(-1))
    lap("queries_gather")
    summed = comm.all_reduce_sum(torch.cat(pieces))
    parts, o = [], 0
    for p_ in pieces:
        parts.append(summed[o:o + p_.numel()]); o += p_.numel()
    opened = []
    for k, s_ in enumerate(range(1, qStage + 1)):
        rows = parts[2 * k].numpy().view(np.uint64).reshape(len(queries), shards[s_]["width"])
        opened.append((rows, strees[s_].siblings_finish(queries, parts[2 * k + 1])))
    if constSTree is not None:
        pc = list(zip([[int(v) for v in r] for r in parts[2 * qStage].numpy().view(np.uint64).reshape(len(queries), nC)],
                      constSTree.siblings_finish(queries, parts[2 * qStage + 1])))
    else:
        pc = be.group_proofs(constTree, queries)
    opened = [(rows.tolist(), sib) for rows, sib in opened]
    friProof[0]["polQueries"] = [[[rows[i], sib[i]] for rows, sib in opened] + [list(pc[i])] for i in range(len(queries))]
    q = list(queries)
    for step in range(1, len(ss["steps"])):
        q = [qi % (1 << ss["steps"][step]["nBits"]) for qi in q]
        if step == 1 and fri_first is not None:
            sib = stree1.siblings_finish(q, parts[-2]) if stree1 is not None else be.merkle_siblings(nodes1, h1, q)
            vals = parts[-1].numpy().view(np.uint64).reshape(len(q), w1).tolist()
            friProof[step]["polQueries"] = [[vals[i], sib[i]] for i in range(len(q))]
        else:
            friProof[step]["polQueries"] = [list(p_) for p_ in be.group_proofs(friTrees[step], q)]
    lap("queries")
    proof = {"root%d" % s_: roots[s_] for s_ in range(1, qStage + 1)}
    proof["evals"] = ctx["evals"]; proof["fri"] = friProof
    if info.get("nSubproofValues"):
        proof["subproofValues"] = list(ctx.get("subproofValues", []))
    return {"proof": proof, "publics": list(ctx["publics"]), "challenges": ctx["challenges"], "challengesFRISteps": challengesFRI, "queries": queries,
            "exchange": comm.stats()}
