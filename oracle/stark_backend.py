"""CPU backend for the prove loop (pil2gl.stark.stark_gen) built on the oracle.

TEST INFRASTRUCTURE ONLY: used by tests/ to produce the expected proof and by bench.py's cpu_baseline leg;
the product package never imports it (it receives a backend object from its caller)."""
import numpy as np

import gl_oracle as orc

P = 0xFFFFFFFF00000001


class OracleBackend:
    """same interface as pil2gl.stark.GpuBackend, numpy arrays + oracle functions"""
    name = "oracle"

    def __init__(self, split=False, hash_type="GL", arity=16, custom=False):
        self.split, self.hash_type, self.arity, self.custom = split, hash_type, arity, custom

    def new_transcript(self):
        if self.hash_type == "BN128":
            import bn128_oracle
            return bn128_oracle.TranscriptBN128(self.arity if self.custom else 16)
        from pil2gl.stark import Transcript
        return Transcript(self)

    def empty(self, n): return np.zeros(int(n), np.uint64)
    def zeros(self, n): return np.zeros(int(n), np.uint64)
    def from_host(self, a): return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1).copy()
    def to_host(self, t): return t.reshape(-1)
    def sync(self): pass

    def interpolate(self, src, C, nb, dst, nbe): dst[:] = orc.interpolate(src.reshape(-1, C), nb, nbe).reshape(-1)
    def fft(self, src, C, nb, dst): dst[:] = orc.fft_cols(src.reshape(-1, C), nb).reshape(-1)
    def ifft(self, src, C, nb, dst): dst[:] = orc.ifft_cols(src.reshape(-1, C), nb).reshape(-1)

    # multi-process partition (pil2gl.parallel) restated on the full-domain oracle functions: the slice of the full
    # extension, one linear hash per row, and the level loop of merklehash_p.js:87-103 with one poseidon per node pair
    def interpolate_cosets(self, src, C, nb, dst, nbe, cb, cc, ws=None):
        full = orc.interpolate(src.reshape(-1, C), nb, nbe).reshape(1 << nb, 1 << (nbe - nb), C)
        dst[:] = full[:, cb:cb + cc, :].reshape(-1)

    def extend_cosets_unshifted(self, src, C, nb, dst, nbe, cb, cc):
        coef = orc.ifft_cols(src.reshape(-1, C), nb)
        pad = np.zeros((1 << nbe, C), np.uint64); pad[:1 << nb] = coef
        full = orc.fft_cols(pad, nbe).reshape(1 << nb, 1 << (nbe - nb), C)
        dst[:] = full[:, cb:cb + cc, :].reshape(-1)

    def extend_coefs_brev_cosets(self, coef_brev, C, nb, dst, nbe, cb, cc):
        n = 1 << nb
        rev = np.array([int(format(i, "0%db" % nb)[::-1], 2) if nb else 0 for i in range(n)])
        pad = np.zeros((1 << nbe, C), np.uint64); pad[:n] = coef_brev.reshape(n, C)[rev]      # row bitrev(m) holds coefficient m
        full = orc.fft_cols(pad, nbe).reshape(n, 1 << (nbe - nb), C)
        dst[:] = full[:, cb:cb + cc, :].reshape(-1)

    def linear_hash_rows(self, buf, w, h):
        return np.concatenate([orc.linear_hash(buf[i * w:(i + 1) * w], self.split) for i in range(h)])

    def linear_hash_rows_into(self, buf, w, h, out):
        out[:4 * h] = self.linear_hash_rows(buf, w, h)

    def merkelize_digests(self, leaves, h):
        nodes = np.zeros(orc.merkle_num_nodes(h), np.uint64)
        nodes[:4 * h] = leaves[:4 * h]
        p_in, n = 0, 4 * h
        nxt = ((n - 1) // 8 + 1) * 4
        p_out = p_in + nxt * 2
        while n > 4:
            for i in range(nxt // 4):
                nodes[p_out + 4 * i:p_out + 4 * i + 4] = orc.poseidon(nodes[p_in + 8 * i:p_in + 8 * i + 8], None, 4)
            n = nxt; nxt = ((n - 1) // 8 + 1) * 4; p_in = p_out; p_out = p_in + nxt * 2
        return nodes

    def merkelize_digest_parts(self, parts, N, cc):
        world = len(parts)
        leaves = np.stack([p.cpu().numpy().view(np.uint64).reshape(N, cc * 4) for p in parts], axis=1).reshape(-1)
        return self.merkelize_digests(leaves, N * cc * world)

    def get_column(self, buf, width, off, dim, n): return np.ascontiguousarray(buf.reshape(n, width)[:, off:off + dim]).reshape(-1)
    def set_column(self, buf, width, off, dim, n, col): buf.reshape(n, width)[:, off:off + dim] = col.reshape(n, dim)
    def gprod(self, num, dn, den, dd): return orc.gprod(num, den, dn, dd)
    def gsum(self, num, dn, den, dd): return orc.gsum(num, den, dn, dd)

    def h1h2(self, f, t, dim):
        key = (lambda r: int(r[0])) if dim == 1 else (lambda r: tuple(int(x) for x in r))
        w1, w2 = orc.h1h2([key(r) for r in f.reshape(-1, dim)], [key(r) for r in t.reshape(-1, dim)])
        arr = lambda w: np.array([[v] if dim == 1 else list(v) for v in w], dtype=np.uint64).reshape(-1)
        return arr(w1), arr(w2)

    def merkle_siblings(self, nodes, height, idxs):
        return [[[int(x) for x in s] for s in orc.group_proof(nodes, height, i)] for i in idxs]

    def merkelize_digest_block(self, parts, N, cc, r, sliced=False):
        world = len(parts)
        nb_ = N // world
        leaves = np.stack([p.cpu().numpy().view(np.uint64).reshape(nb_, cc * 4) if sliced else p.cpu().numpy().view(np.uint64).reshape(N, cc * 4)[r * nb_:(r + 1) * nb_]
                           for p in parts], axis=1).reshape(-1)
        return self.merkelize_digests(leaves, nb_ * cc * world)

    def as_torch(self, t):
        import torch
        return torch.from_numpy(np.ascontiguousarray(t).view(np.int64))

    def from_torch(self, t): return t.cpu().numpy().view(np.uint64)

    def merkelize(self, buf, w, h):
        if self.hash_type == "BN128":
            import bn128_oracle
            if h >= 512:      # large trees through the C restatement (tests/test_bn128_oracle.py: == the Python-integer statement below, node by node)
                return {"elements": buf, "nodes": bn128_oracle.c_merkelize(buf.reshape(h, w), self.arity, self.custom), "width": w, "height": h}
            rows = [[int(v) for v in r] for r in buf.reshape(h, w)]
            return {"elements": buf, "nodes": bn128_oracle.merkelize(rows, self.arity, self.custom), "width": w, "height": h}
        return {"elements": buf, "nodes": orc.merkelize(buf.reshape(h, w), self.split), "width": w, "height": h}

    def root(self, tree):
        if self.hash_type == "BN128":
            return int(tree["nodes"][-1])
        return [int(v) for v in tree["nodes"][-4:]]

    def group_proof(self, tree, idx):
        w = tree["width"]
        vals = [int(v) for v in tree["elements"][idx * w:(idx + 1) * w]]
        if self.hash_type == "BN128":
            import bn128_oracle
            return vals, bn128_oracle.group_proof(tree["nodes"], tree["height"], self.arity, idx)
        return vals, [[int(x) for x in s] for s in orc.group_proof(tree["nodes"], tree["height"], idx)]

    def group_proofs(self, tree, idxs): return [self.group_proof(tree, i) for i in idxs]

    def poseidon(self, inp, cap, n): return [int(v) for v in orc.poseidon([int(v) % P for v in inp], [int(v) % P for v in cap], n)]
    def build_x(self, nb, shift): return orc.build_x(nb, shift)
    def build_zhinv(self, nb, nbe): return orc.build_zhinv(nb, nbe)
    def build_one_row_zerofier_inv(self, nb, nbe, row): return orc.build_one_row_zerofier_inv(nb, nbe, row)
    def build_frame_zerofier(self, nb, nbe, off_min, off_max): return orc.build_frame_zerofier(nb, nbe, off_min, off_max)
    def q_split(self, qq1, nb, nbe, qDim, qDeg): return orc.compute_q_split(qq1, nb, nbe, qDim, qDeg).reshape(-1)
    def x_div_x_sub_xi(self, nbe, xis): return orc.x_div_x_sub_xi(nbe, np.array(xis, dtype=np.uint64)).reshape(-1)
    def build_lev(self, nb, xi): return orc.lev(nb, np.array(xi, dtype=np.uint64)).reshape(-1)

    def compute_evals(self, descs, nb, eb, levs):
        return [[int(v) for v in orc.eval_pol_at(buf.reshape(-1, width), off, dim, nb, eb, levs[li].reshape(-1, 3))]
                for (buf, width, off, dim, li) in descs]

    def eval_program(self, ops, n_tmp, sections, scalars, nb, prime_shift):
        secs = [t.reshape(-1, w) for (t, w) in sections]          # views: destinations are written in place
        orc.eval_program(ops, n_tmp, secs, scalars, nb, prime_shift)

    def fri_fold(self, pol, pol_bits, out_bits, shift_inv, challenge):
        return orc.fri_fold(pol.reshape(-1, 3), out_bits, shift_inv, np.array([int(c) % P for c in challenge], dtype=np.uint64)).reshape(-1)

    def fri_transpose(self, pol, pol_bits, t_bits): return orc.fri_transpose(pol.reshape(-1, 3), t_bits).reshape(-1)


