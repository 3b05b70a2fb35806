"""Host-side mirror of the reference's BN128 Merkle commitment modules over libpil2gl (no arithmetic of its own):
  buildMerkleHash(arity, custom)  <-> src/helpers/hash/merklehash/merklehash_bn128_p.js:10-285
  LinearHashBN                    <-> src/helpers/hash/linearhash/linearhash.bn128.js:4-62
  Transcript                      <-> src/helpers/transcript/transcript.bn128.js:1-106
  poseidon(inputs, initState, nOut) <-> circomlibjs buildPoseidon() as the reference calls it
Field elements cross this API as Python ints in normal form (the JS modules use BigInt / F.toObject)."""
import ctypes as C

import numpy as np

from ._lib import Pil2glError, load, call
from . import _is_dev, _ptr, _stream, _check_len, _to_host, torch

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _words(vals):
    a = np.zeros((len(vals), 4), np.uint64)
    for i, v in enumerate(vals):
        v = int(v) % R
        for k in range(4):
            a[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return a


def _ints(words):
    w = np.asarray(words, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(x) << (64 * k) for k, x in enumerate(r)) for r in w]


def poseidon(inputs, initState=0, nOut=1):
    if not 1 <= len(inputs) <= 16:
        raise Pil2glError("BN128 Poseidon takes 1..16 inputs")
    i = _words(inputs); s = _words([initState]); o = np.zeros((nOut, 4), np.uint64)
    call("pil2gl_bn128_poseidon", _ptr(i), _ptr(s), 1, len(inputs), nOut, _ptr(o))
    return _ints(o)


def poseidon_chain(blocks, initState=0):
    """blocks: [nBlocks][nIn] ints absorbed one after the other, each permutation's output 0 being the next one's state
    element 0 (transcript.bn128.js:56-66) -> the nIn+1 outputs of the last permutation; one launch, nIn+1 lanes"""
    nB, nIn = len(blocks), len(blocks[0])
    i = _words([v for row in blocks for v in row]); s = _words([initState]); o = np.zeros((nIn + 1, 4), np.uint64)
    call("pil2gl_bn128_sponge_absorb", _ptr(i), nB, nIn, _ptr(s), _ptr(o))
    return _ints(o)


def poseidon_batch(inputs, init=None, nOut=1):
    """inputs: [count][nIn] ints, init: [count] ints or None -> [count][nOut] ints"""
    count, nIn = len(inputs), len(inputs[0])
    i = _words([v for row in inputs for v in row]); s = _words(init) if init is not None else None
    o = np.zeros((count * nOut, 4), np.uint64)
    call("pil2gl_bn128_poseidon", _ptr(i), _ptr(s), count, nIn, nOut, _ptr(o))
    flat = _ints(o)
    return [flat[k * nOut:(k + 1) * nOut] for k in range(count)]


def from_montgomery(words):
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, 4); o = np.zeros_like(w)
    call("pil2gl_bn128_convert", _ptr(w), w.shape[0], 0, _ptr(o))
    return _ints(o)


def to_montgomery(vals):
    w = _words(vals); o = np.zeros_like(w)
    call("pil2gl_bn128_convert", _ptr(w), w.shape[0], 1, _ptr(o))
    return o


class LinearHashBN:
    """linearhash.bn128.js: hash(vals) -> Fr (normal form int)"""

    def __init__(self, arity, custom):
        self.arity, self.custom = int(arity), bool(custom)

    def hash(self, vals):
        flat = []
        for v in vals:
            if isinstance(v, (list, tuple, np.ndarray)):
                flat.extend(int(x) for x in v)
            else:
                flat.append(int(v))
        el = [sum(x << (64 * k) for k, x in enumerate(flat[i:i + 3])) % R for i in range(0, len(flat), 3)]
        if not el:
            return 0
        if len(el) == 1:
            return el[0]
        st = 0
        for i in range(0, len(el), self.arity):                 # linearhash.bn128.js:46-57
            chunk = el[i:i + self.arity]
            if len(chunk) < self.arity and self.custom:
                chunk = chunk + [0] * (self.arity - len(chunk))
            st = poseidon(chunk, st, 1)[0]
        return st


class MerkleHashBN128:
    """merklehash_bn128_p.js:16-285.  tree = {elements, nodes, width, height}; nodes = u64 words, Montgomery form."""

    def __init__(self, arity=16, custom=False):
        if arity not in (2, 4, 8, 16):
            raise Pil2glError("arity must be 2, 4, 8 or 16")
        self.arity, self.custom = int(arity), bool(custom)
        self.lh = LinearHashBN(arity, custom)
        load()

    def _getNNodes(self, n):
        return int(load().pil2gl_bn128_merkle_num_nodes(n, self.arity))

    def merkelize(self, buff, width, height):
        if height <= 0:
            raise Pil2glError("height must be > 0")
        _check_len(buff, width * height, "buff")
        n_words = self._getNNodes(height) * 4
        if _is_dev(buff):
            nodes = torch.empty(n_words, dtype=torch.int64, device=buff.device)
            call("pil2gl_bn128_merkelize_dev", _ptr(buff), width, height, self.arity, int(self.custom), _ptr(nodes), _stream())
        else:
            nodes = np.zeros(n_words, np.uint64)
            call("pil2gl_bn128_merkelize", _ptr(buff), width, height, self.arity, int(self.custom), _ptr(nodes))
        return {"elements": buff, "nodes": nodes, "width": width, "height": height}

    def root(self, tree):
        last = tree["nodes"][-4:]
        if _is_dev(last):
            last = last.cpu().numpy().view(np.uint64)
        return from_montgomery(last)[0]

    def getGroupProof(self, tree, idx):
        if idx < 0 or idx >= tree["height"]:
            raise Pil2glError("Out of range")
        width, height, a = tree["width"], tree["height"], self.arity
        if _is_dev(tree["nodes"]):
            vals = np.zeros(max(width, 1), np.uint64); sib = np.zeros((64, a, 4), np.uint64); nl = C.c_uint32()
            call("pil2gl_bn128_group_proof_dev", _ptr(tree["elements"]), _ptr(tree["nodes"]), width, height, a, idx,
                 _ptr(vals), _ptr(sib), C.byref(nl))
            return [int(v) for v in vals[:width]], [_ints(sib[l]) for l in range(nl.value)]
        el = tree["elements"].reshape(-1)
        v = [int(x) for x in el[idx * width:(idx + 1) * width]]
        nodes = tree["nodes"].reshape(-1, 4); nbits = (a - 1).bit_length()
        mp, offset, n = [], 0, height
        while n > 1:                                            # merklehash_bn128_p.js:155-181
            si = idx ^ (idx & (a - 1))
            grp = from_montgomery(nodes[offset + si:offset + si + a])
            mp.append([g if i < n else 0 for i, g in enumerate(grp)])
            nxt = (n - 1) // a + 1
            offset += nxt * a; n = nxt; idx >>= nbits
        return v, mp

    def getGroupProofs(self, tree, idxs):
        """getGroupProof for a batch of rows of a device-resident tree in one launch (pil2gl_bn128_group_proofs_dev); host trees: one by one"""
        idxs = [int(i) for i in idxs]
        if not idxs or not _is_dev(tree["nodes"]):
            return [self.getGroupProof(tree, i) for i in idxs]
        width, height, a, n = tree["width"], tree["height"], self.arity, len(idxs)
        if any(i < 0 or i >= height for i in idxs):
            raise Pil2glError("Out of range")
        ii = np.array(idxs, dtype=np.uint64)
        vals = np.zeros((n, max(width, 1)), np.uint64); nl = C.c_uint32()
        sib_flat = np.zeros(n * 40 * a * 4, np.uint64)
        call("pil2gl_bn128_group_proofs_dev", _ptr(tree["elements"]), _ptr(tree["nodes"]), width, height, a, _ptr(ii), n,
             _ptr(vals), _ptr(sib_flat), C.byref(nl))
        lv = nl.value
        v2 = vals.reshape(-1)[:n * width].reshape(n, width).tolist() if width else [[] for _ in range(n)]
        s2 = sib_flat[:n * lv * a * 4].reshape(n, lv, a, 4)
        return [(v2[q], [_ints(s2[q, l]) for l in range(lv)]) for q in range(n)]

    def calculateRootFromGroupProof(self, mp, idx, vals):
        value = self.lh.hash(vals)                              # merklehash_bn128_p.js:184-232
        nbits = (self.arity - 1).bit_length()
        for sibs in mp:
            cur = idx & (self.arity - 1)
            idx >>= nbits
            group = [int(s) % R for s in sibs]
            group[cur] = value
            value = poseidon(group, 0, 1)[0]
        return value

    def calculateRootsFromGroupProofs(self, proofs, idxs):
        """calculateRootFromGroupProof for a batch of openings [(vals, siblings), ...] of one tree: every sponge chunk and
        every tree level is ONE batched permutation call over all the openings (a single BN254 permutation has the
        latency of a whole wave of them)"""
        if not proofs:
            return []
        flat = []
        for vals, _ in proofs:
            f = []
            for v in vals:
                f.extend(int(x) for x in v) if isinstance(v, (list, tuple, np.ndarray)) else f.append(int(v))
            flat.append(f)
        if len({len(f) for f in flat}) != 1 or len({len(mp) for _, mp in proofs}) != 1:
            raise Pil2glError("openings of different shapes in one batch")
        els = [[sum(x << (64 * k) for k, x in enumerate(f[i:i + 3])) % R for i in range(0, len(f), 3)] for f in flat]
        n_el = len(els[0])
        if n_el == 0:
            value = [0] * len(proofs)
        elif n_el == 1:
            value = [e[0] for e in els]
        else:                                                   # linearhash.bn128.js:46-57, all openings per chunk
            value = [0] * len(proofs)
            for i in range(0, n_el, self.arity):
                chunks = [e[i:i + self.arity] for e in els]
                if len(chunks[0]) < self.arity and self.custom:
                    chunks = [c + [0] * (self.arity - len(c)) for c in chunks]
                value = [r[0] for r in poseidon_batch(chunks, value, 1)]
        nbits = (self.arity - 1).bit_length()
        pos = [int(i) for i in idxs]
        for level in range(len(proofs[0][1])):                  # merklehash_bn128_p.js:207-231
            groups = []
            for q, (_, mp) in enumerate(proofs):
                g = [int(s_) % R for s_ in mp[level]]
                g[pos[q] & (self.arity - 1)] = value[q]
                groups.append(g)
                pos[q] >>= nbits
            value = [r[0] for r in poseidon_batch(groups, None, 1)]
        return value

    def verifyGroupProofs(self, root, proofs, idxs):
        return all(self.eqRoot(r, root) for r in self.calculateRootsFromGroupProofs(proofs, idxs))

    def eqRoot(self, r1, r2):
        return int(r1) == int(r2)

    def verifyGroupProof(self, root, mp, idx, groupElements):
        return self.eqRoot(self.calculateRootFromGroupProof(mp, idx, groupElements), root)

    def writeToFile(self, tree, fileName):
        """merklehash_bn128_p.js:243-263: [width u64][height u64][elements][nodes]"""
        with open(fileName, "wb") as f:
            np.array([tree["width"], tree["height"]], dtype="<u8").tofile(f)
            _to_host(tree["elements"]).astype("<u8", copy=False).tofile(f)
            _to_host(tree["nodes"]).astype("<u8", copy=False).tofile(f)

    def readFromFile(self, fileName, device=None):
        with open(fileName, "rb") as f:
            width, height = (int(v) for v in np.fromfile(f, dtype="<u8", count=2))
            elements = np.fromfile(f, dtype="<u8", count=width * height).astype(np.uint64)
            nodes = np.fromfile(f, dtype="<u8", count=self._getNNodes(height) * 4).astype(np.uint64)
        if device is not None:
            elements = torch.from_numpy(elements.view(np.int64)).to(device)
            nodes = torch.from_numpy(nodes.view(np.int64)).to(device)
        return {"elements": elements, "nodes": nodes, "width": width, "height": height}


def buildMerkleHash(arity=16, custom=False):
    """merklehash_bn128_p.js:10"""
    return MerkleHashBN128(arity, custom)


class Transcript:
    """transcript.bn128.js:1-106 over the device permutation"""

    def __init__(self, nInputs=16):
        self.nInputs, self.state = nInputs, 0
        self.pending, self.out, self.out3 = [], [], []

    def getState(self):
        if self.pending:
            self.updateState()
        return self.state

    def getField(self):
        return [self.getFields1(), self.getFields1(), self.getFields1()]

    def getFields1(self):
        if self.out3:
            return self.out3.pop(0)
        if self.out:
            v = self.out.pop(0)
            self.out3 = [v & 0xFFFFFFFFFFFFFFFF, (v >> 64) & 0xFFFFFFFFFFFFFFFF, (v >> 128) & 0xFFFFFFFFFFFFFFFF]
            return self.getFields1()
        self.updateState()
        return self.getFields1()

    def getFields253(self):
        if self.out:
            return self.out.pop(0)
        self.updateState()
        return self.getFields253()

    def updateState(self):
        while len(self.pending) < self.nInputs:
            self.pending.append(0)
        self.out = poseidon(self.pending, self.state, self.nInputs + 1)
        self.out3, self.pending = [], []
        self.state = self.out[0]

    def put(self, a):
        flat = []

        def walk(v):
            if isinstance(v, (list, tuple)):
                for x in v:
                    walk(x)
            else:
                flat.append(int(v) % R)
        walk(a)
        if not flat:
            return
        n = self.nInputs
        if (len(self.pending) + len(flat)) // n >= 2:      # a chain of dependent permutations: one launch for all of them
            allv = self.pending + flat
            nb = len(allv) // n
            self.out = poseidon_chain([allv[k * n:(k + 1) * n] for k in range(nb)], self.state)
            self.state, self.out3, self.pending = self.out[0], [], allv[nb * n:]
            if self.pending:
                self.out = []
            return
        for v in flat:
            self.out = []
            self.pending.append(v)
            if len(self.pending) == n:
                self.updateState()

    def getPermutations(self, n, nBits):
        total = n * nBits
        fields = [self.getFields253() for _ in range((total - 1) // 253 + 1)]
        res, cf, cb = [], 0, 0
        for _ in range(n):
            a = 0
            for j in range(nBits):
                if (fields[cf] >> cb) & 1:
                    a += 1 << j
                cb += 1
                if cb == 253:
                    cb = 0; cf += 1
            res.append(a)
        return res
