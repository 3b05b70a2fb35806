"""Python host-side mirror of the reference's L2 operator interface over libpil2gl.so.

Same names, argument order and error behaviour as the JavaScript modules they stand for
(`interpolate/fft/ifft`: src/helpers/fft/fft_p.js:178-302; `buildMerkleHash`/`MerkleHash`:
src/helpers/hash/merklehash/merklehash_p.js:12-279; `FRI`: src/stark/fri.js:7-175; `Transcript`:
src/helpers/transcript/transcript.js), so the parity tests read like the reference's own tests.
The Node.js binding of the same C ABI lives in ../../js and ../../addon.

Buffers are numpy uint64 arrays (host: copy in / compute on the GPU / copy out) or torch CUDA
tensors of dtype int64/uint64 (device resident, enqueued on torch's current stream, no sync).
Every computation runs in the HIP library; nothing falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Pil2glError, load, call  # noqa: F401

P = 0xFFFFFFFF00000001
SHIFT = 7

try:  # torch is only plumbing (device memory + streams); the library itself does not need it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_dev(b):
    return torch is not None and isinstance(b, torch.Tensor)


def _ptr(b):
    if b is None:
        return None
    if _is_dev(b):
        if not b.is_cuda:
            raise Pil2glError("torch tensors must live on the GPU (use numpy arrays for host buffers)")
        if b.dtype not in (torch.int64, torch.uint64) or not b.is_contiguous():
            raise Pil2glError("device buffers must be contiguous int64/uint64 tensors")
        return C.c_void_p(b.data_ptr())
    if not isinstance(b, np.ndarray) or b.dtype != np.uint64 or not b.flags["C_CONTIGUOUS"]:
        raise Pil2glError("host buffers must be C-contiguous numpy uint64 arrays")
    return C.c_void_p(b.ctypes.data)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _same_side(*bufs):
    dev = [_is_dev(b) for b in bufs if b is not None]
    if any(dev) and not all(dev):
        raise Pil2glError("mixing host and device buffers in one call")
    return bool(dev and dev[0])


def init(device=0):
    call("pil2gl_init", int(device))


def shutdown():
    """pil2gl_shutdown: releases the library's device state (tables, scratch slots, compiled kernels); init() starts afresh"""
    _lib.load().pil2gl_shutdown()


def device_info():
    name = C.create_string_buffer(64); cus = C.c_uint32(); mem = C.c_uint64()
    call("pil2gl_device_info", name, 64, C.byref(cus), C.byref(mem))
    return name.value.decode(), cus.value, mem.value


# ----------------------------------------------------------------------------- fft_p.js
def _check_len(buf, n, what):
    if buf is not None and int(np.prod(buf.shape)) < n:
        raise Pil2glError("%s has %d elements, needs %d" % (what, int(np.prod(buf.shape)), n))


def interpolate(buffSrc, nPols, nBits, buffDst, nBitsExt):
    """fft_p.js:187 interpolate(buffSrc, nPols, nBits, buffDst, nBitsExt); dst is caller-allocated."""
    _check_len(buffSrc, nPols << nBits, "buffSrc"); _check_len(buffDst, nPols << nBitsExt, "buffDst")
    if _same_side(buffSrc, buffDst):
        call("pil2gl_interpolate_dev", _ptr(buffSrc), nPols, nBits, _ptr(buffDst), nBitsExt, _stream())
    else:
        call("pil2gl_interpolate", _ptr(buffSrc), nPols, nBits, _ptr(buffDst), nBitsExt)


def interpolateCosets(buffSrc, nPols, nBits, buffDst, nBitsExt, cosetBegin, cosetCount, workspace=None):
    """interpolate restricted to cosets [cosetBegin, cosetBegin+cosetCount): dst is N x (cosetCount*nPols), device only
    (the per-GPU slice of extendAndMerkelize, see pil2gl.parallel).  workspace: N*nPols words for the coefficients
    (default: library scratch); passing buffSrc itself overwrites the trace and saves that much memory."""
    _check_len(buffSrc, nPols << nBits, "buffSrc"); _check_len(buffDst, (nPols * cosetCount) << nBits, "buffDst")
    if not _same_side(buffSrc, buffDst):
        raise Pil2glError("interpolateCosets works on device buffers")
    if workspace is None:
        call("pil2gl_interpolate_cosets_dev", _ptr(buffSrc), nPols, nBits, _ptr(buffDst), nBitsExt, cosetBegin, cosetCount, _stream())
    else:
        _check_len(workspace, nPols << nBits, "workspace")
        call("pil2gl_interpolate_cosets_ws_dev", _ptr(buffSrc), nPols, nBits, _ptr(buffDst), nBitsExt, cosetBegin, cosetCount, _ptr(workspace), _stream())


def fft(buffSrc, nPols, nBits, buffDst):
    """fft_p.js:178"""
    _check_len(buffSrc, nPols << nBits, "buffSrc"); _check_len(buffDst, nPols << nBits, "buffDst")
    if _same_side(buffSrc, buffDst):
        call("pil2gl_fft_dev", _ptr(buffSrc), nPols, nBits, _ptr(buffDst), _stream())
    else:
        call("pil2gl_fft", _ptr(buffSrc), nPols, nBits, _ptr(buffDst))


def ifft(buffSrc, nPols, nBits, buffDst):
    """fft_p.js:182"""
    _check_len(buffSrc, nPols << nBits, "buffSrc"); _check_len(buffDst, nPols << nBits, "buffDst")
    if _same_side(buffSrc, buffDst):
        call("pil2gl_ifft_dev", _ptr(buffSrc), nPols, nBits, _ptr(buffDst), _stream())
    else:
        call("pil2gl_ifft", _ptr(buffSrc), nPols, nBits, _ptr(buffDst))


def _block_on_device(buff, fn):
    """worker-level operators work in place on a block: a device tensor as it stands, a host array through a staging tensor"""
    if _is_dev(buff):
        fn(buff)
        return buff
    if buff.size:
        d = torch.from_numpy(buff.view(np.int64)).cuda()
        fn(d)
        buff[...] = d.cpu().numpy().view(np.uint64).reshape(buff.shape)
    return buff


def fft_block(buff, start_pos, nPols, nBits, s, blockBits, layers):
    """fft_worker.js:62 fft_block: `layers` butterfly stages ending at stage s on the 2^blockBits x nPols block at row start_pos"""
    _check_len(buff, nPols << blockBits, "buff")
    return _block_on_device(buff, lambda d: call("pil2gl_fft_block_dev", _ptr(d), start_pos, nPols, nBits, s, blockBits, layers, _stream()))


def interpolatePrepareBlock(buff, width, start, inc, st_i=0, st_n=1):
    """fft_worker.js:6 interpolatePrepareBlock: row i of the block times start * inc^i"""
    height = int(np.prod(buff.shape)) // width
    return _block_on_device(buff, lambda d: call("pil2gl_interpolate_prepare_block_dev", _ptr(d), width, height, int(start), int(inc), _stream()))


# ----------------------------------------------------------------------------- stage-2 hints (polutils.js:128-164)
def _hint(fn, num, den, dimNum, dimDen, out):
    n = int(np.prod(den.shape)) // dimDen
    if not _same_side(num, den, out) :
        raise Pil2glError("hint columns must be device buffers")
    if out is None:
        out = torch.empty(n * (3 if 3 in (dimNum, dimDen) else 1), dtype=torch.int64, device=den.device)
    call(fn, _ptr(num), dimNum, _ptr(den), dimDen, n, _ptr(out), _stream())
    return out


def calculateZ(num, den, dimNum=3, dimDen=3, out=None):
    """polutils.js:128 calculateZ: z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1] (device columns)"""
    return _hint("pil2gl_gprod_dev", num, den, dimNum, dimDen, out)


def calculateS(num, den, dimNum=3, dimDen=3, out=None):
    """polutils.js:145 calculateS: s[i] = s[i-1] + num / den[i], num one element (device buffers)"""
    return _hint("pil2gl_gsum_dev", num, den, dimNum, dimDen, out)


def calculateH1H2(f, t, dim=1):
    """polutils.js:105 calculateH1H2(F, f, t) on device columns -> (h1, h2)"""
    n = int(np.prod(t.shape)) // dim
    if not _same_side(f, t):
        raise Pil2glError("hint columns must be device buffers")
    h1 = torch.empty(n * dim, dtype=torch.int64, device=t.device); h2 = torch.empty_like(h1)
    call("pil2gl_h1h2_dev", _ptr(f), _ptr(t), n, dim, _ptr(h1), _ptr(h2), _stream())
    return h1, h2


def firstNonZeroRow(col, dim, first, last):
    """the device half of calculateExps(ctx, code, dom, debug = true) (prover_helpers.js:46-70): the first row of [first, last) where a
    constraint's value column (dim 1 or 3 words per row, device buffer) is not zero -> (row, [value words]) or None"""
    row = np.zeros(1, np.uint64); val = np.zeros(3, np.uint64)
    call("pil2gl_first_nonzero_row_dev", _ptr(col), dim, first, last, _ptr(row), _ptr(val), _stream())
    return None if int(row[0]) == 0xFFFFFFFFFFFFFFFF else (int(row[0]), [int(v) for v in val[:dim]])


# ----------------------------------------------------------------------------- poseidon / linear hash
def poseidon(inputs, capacity=None, nOuts=4):
    """hash/poseidon/poseidon.js:57 poseidon(inputs[8], capacity[4]?, nOuts=4) -> list of ints"""
    if len(inputs) != 8:
        raise Pil2glError("Invalid Input size (must be 8)")
    if capacity is not None and len(capacity) != 4:
        raise Pil2glError("Invalid Capacity size (must be 4)")
    i = np.array([int(v) % P for v in inputs], dtype=np.uint64)
    c = np.array([int(v) % P for v in capacity], dtype=np.uint64) if capacity is not None else None
    o = np.zeros(nOuts, np.uint64)
    call("pil2gl_poseidon", _ptr(i), _ptr(c), 1, nOuts, _ptr(o))
    return [int(v) for v in o]


def poseidon_batch(inp, cap=None, nOuts=4):
    inp = np.ascontiguousarray(inp, dtype=np.uint64).reshape(-1, 8)
    if cap is not None:
        cap = np.ascontiguousarray(cap, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((inp.shape[0], nOuts), np.uint64)
    call("pil2gl_poseidon", _ptr(inp), _ptr(cap), inp.shape[0], nOuts, _ptr(out))
    return out


def linearHash(buffIn, width, splitLinearHash=False, out=None):
    """merklehash_worker.js:37 linearHash(buffIn, width, st_i, st_n, splitLinearHash) -> height x 4 digests"""
    n = int(np.prod(buffIn.shape))
    height = n // width if width else 0
    if _is_dev(buffIn):
        if out is None:
            out = torch.empty(height * 4, dtype=torch.int64, device=buffIn.device)
        call("pil2gl_linear_hash_rows_dev", _ptr(buffIn), width, height, int(bool(splitLinearHash)), _ptr(out), _stream())
    else:
        if out is None:
            out = np.zeros(height * 4, np.uint64)
        call("pil2gl_linear_hash_rows", _ptr(buffIn), width, height, int(bool(splitLinearHash)), _ptr(out))
    return out


def merkelizeLevel(buffIn, out=None):
    """merklehash_worker.js:86 merkelizeLevel(buffIn, st_i, st_n): 8 words in -> 4 words out per op"""
    nOps = int(np.prod(buffIn.shape)) // 8
    if _is_dev(buffIn):
        if out is None:
            out = torch.empty(nOps * 4, dtype=torch.int64, device=buffIn.device)
        call("pil2gl_merkelize_level_dev", _ptr(buffIn), nOps, _ptr(out), _stream())
    else:
        if out is None:
            out = np.zeros(nOps * 4, np.uint64)
        call("pil2gl_merkelize_level", _ptr(buffIn), nOps, _ptr(out))
    return out


class _LinearHash:
    """linearhash.js / linearhash_gpu.js: hash(vals) -> [4 ints]"""

    def __init__(self, split):
        self.split = bool(split)

    def hash(self, vals):
        flat = []
        for v in vals:
            if isinstance(v, (list, tuple, np.ndarray)):
                flat.extend(int(x) for x in v)
            else:
                flat.append(int(v))
        if not flat:
            return [0, 0, 0, 0]
        a = np.array(flat, dtype=np.uint64)
        return [int(x) for x in linearHash(a, len(flat), self.split)]


# ----------------------------------------------------------------------------- merklehash_p.js
class MerkleHash:
    """merklehash_p.js:19-279.  `tree` is a dict {elements, nodes, width, height}; elements aliases the
    caller's buffer (merklehash_p.js:47), nodes is newly allocated on the same side as elements."""

    def __init__(self, splitLinearHash=False):
        self.splitLinearHash = bool(splitLinearHash)
        self.lh = _LinearHash(splitLinearHash)
        load()

    def _getNNodes(self, n):
        """merklehash_p.js:28 (n = number of u64 words of the leaf level = 4*height)"""
        return int(load().pil2gl_merkle_num_nodes(n // 4))

    def merkelize(self, buff, width, height):
        if height <= 0:
            raise Pil2glError("height must be > 0")
        _check_len(buff, width * height, "buff")
        n_nodes = self._getNNodes(height * 4)
        if _is_dev(buff):
            nodes = torch.empty(n_nodes, dtype=torch.int64, device=buff.device)
            call("pil2gl_merkelize_dev", _ptr(buff), width, height, int(self.splitLinearHash), _ptr(nodes), _stream())
        else:
            nodes = np.zeros(n_nodes, np.uint64)
            call("pil2gl_merkelize", _ptr(buff), width, height, int(self.splitLinearHash), _ptr(nodes))
        return {"elements": buff, "nodes": nodes, "width": width, "height": height}

    def merkelizeDigests(self, leaves, height):
        """upper levels only (merklehash_p.js:87-103) from `height` leaf digests already computed (device buffer)"""
        n_nodes = self._getNNodes(height * 4)
        _check_len(leaves, height * 4, "leaves")
        nodes = torch.empty(n_nodes, dtype=torch.int64, device=leaves.device)
        nodes[:height * 4] = leaves.reshape(-1)[:height * 4]
        call("pil2gl_merkelize_digests_dev", _ptr(nodes), height, _stream())
        return nodes

    def getElement(self, tree, idx, subIdx):
        return _word(tree["elements"], tree["width"] * idx + subIdx)

    def root(self, tree):
        n = tree["nodes"]
        last = n[-4:]
        if _is_dev(last):
            last = last.cpu().numpy().view(np.uint64)
        return [int(v) for v in last]

    def getGroupProof(self, tree, idx):
        if idx < 0 or idx >= tree["height"]:
            raise Pil2glError("Out of range")
        width, height = tree["width"], tree["height"]
        if _is_dev(tree["nodes"]):
            vals = np.zeros(max(width, 1), np.uint64); sib = np.zeros((64, 4), np.uint64); nl = C.c_uint32()
            call("pil2gl_group_proof_dev", _ptr(tree["elements"]), _ptr(tree["nodes"]), width, height, idx,
                 _ptr(vals), _ptr(sib), C.byref(nl))
            return [int(v) for v in vals[:width]], [[int(x) for x in s] for s in sib[:nl.value]]
        el = tree["elements"].reshape(-1)
        v = [int(x) for x in el[idx * width:(idx + 1) * width]]
        nodes = tree["nodes"]; mp = []; offset = 0; n = height * 4
        while n > 4:                                    # merklehash_p.js:154-167
            si = (idx ^ 1) * 4
            mp.append([int(x) for x in nodes[offset + si:offset + si + 4]])
            nextN = ((n - 1) // 8 + 1) * 4
            offset += nextN * 2; n = nextN; idx >>= 1
        return v, mp

    def getGroupProofs(self, tree, idxs):
        """getGroupProof for a list of indices (fri.js:83-105 opens each tree at every query): one gather on the device"""
        if not _is_dev(tree["nodes"]):
            return [self.getGroupProof(tree, i) for i in idxs]
        width, height = tree["width"], tree["height"]
        for i in idxs:
            if i < 0 or i >= height:
                raise Pil2glError("Out of range")
        nl = 0; n = height * 4
        while n > 4:
            n = ((n - 1) // 8 + 1) * 4; nl += 1
        ii = np.array(idxs, dtype=np.uint64); out = np.zeros((len(idxs), width + 4 * nl), np.uint64); lv = C.c_uint32()
        call("pil2gl_group_proofs_dev", _ptr(tree["elements"]), _ptr(tree["nodes"]), width, height, _ptr(ii), len(idxs), _ptr(out), C.byref(lv))
        vals, sibs = out[:, :width].tolist(), out[:, width:].reshape(len(idxs), nl, 4).tolist()      # python ints, as getGroupProof gives
        return list(zip(vals, sibs))

    def calculateRootFromGroupProof(self, mp, idx, vals):
        value = self.lh.hash(vals)                      # merklehash_p.js:170-210
        for sib in mp:
            if idx & 1 == 0:
                value = poseidon(list(value) + list(sib))
            else:
                value = poseidon(list(sib) + list(value))
            idx >>= 1
        return value

    def calculateRootsFromGroupProofs(self, proofs, idxs):
        """calculateRootFromGroupProof for a batch of openings [(vals, siblings), ...] of one tree (the verifier checks every
        query of a tree: stark_verify.js:165-178, fri.js:140): leaf hashes and path walks on the device, one call"""
        if not proofs:
            return []
        width, nl = len(proofs[0][0]), len(proofs[0][1])
        packed = np.zeros((len(proofs), width + 4 * nl), np.uint64)
        for q, (vals, sib) in enumerate(proofs):
            if len(vals) != width or len(sib) != nl:
                raise Pil2glError("openings of different shapes in one batch")
            packed[q, :width] = [int(v) % P for v in vals]
            packed[q, width:] = [int(x) % P for s_ in sib for x in s_]
        ii = np.array([int(i) for i in idxs], dtype=np.uint64); roots = np.zeros((len(proofs), 4), np.uint64)
        call("pil2gl_roots_from_group_proofs", _ptr(packed), width, nl, _ptr(ii), len(proofs), int(self.splitLinearHash), _ptr(roots))
        return [[int(v) for v in r] for r in roots]

    def verifyGroupProofs(self, root, proofs, idxs):
        return all(self.eqRoot(r, root) for r in self.calculateRootsFromGroupProofs(proofs, idxs))

    def eqRoot(self, r1, r2):
        return all(int(a) % P == int(b) % P for a, b in zip(r1, r2))

    def verifyGroupProof(self, root, mp, idx, groupElements):
        return self.eqRoot(self.calculateRootFromGroupProof(mp, idx, groupElements), root)

    def writeToFile(self, tree, fileName):
        """merklehash_p.js:228-246: [width u64][height u64][elements][nodes], little-endian u64"""
        with open(fileName, "wb") as f:
            np.array([tree["width"], tree["height"]], dtype="<u8").tofile(f)
            _to_host(tree["elements"]).astype("<u8", copy=False).tofile(f)
            _to_host(tree["nodes"]).astype("<u8", copy=False).tofile(f)

    def readFromFile(self, fileName, device=None):
        """merklehash_p.js:248-278"""
        with open(fileName, "rb") as f:
            width, height = (int(v) for v in np.fromfile(f, dtype="<u8", count=2))
            elements = np.fromfile(f, dtype="<u8", count=width * height).astype(np.uint64)
            nodes = np.fromfile(f, dtype="<u8", count=self._getNNodes(height * 4)).astype(np.uint64)
        if device is not None:
            elements = torch.from_numpy(elements.view(np.int64)).to(device)
            nodes = torch.from_numpy(nodes.view(np.int64)).to(device)
        return {"elements": elements, "nodes": nodes, "width": width, "height": height}


def buildMerkleHash(splitLinearHash=False):
    """merklehash_p.js:12"""
    return MerkleHash(splitLinearHash)


def _to_host(b):
    if _is_dev(b):
        return b.reshape(-1).cpu().numpy().view(np.uint64)
    return b.reshape(-1)


def _word(b, i):
    if _is_dev(b):
        return int(b.reshape(-1)[i:i + 1].cpu().numpy().view(np.uint64)[0])
    return int(b.reshape(-1)[i])


# ----------------------------------------------------------------------------- transcript.js
class Transcript:
    """transcript.js:2-85 (host-side duplex sponge; every permutation runs through pil2gl_poseidon)"""

    def __init__(self):
        self.state = [0, 0, 0, 0]
        self.pending = []
        self.out = []

    def getState(self):
        if self.pending:
            self.updateState()
        return self.state

    def getField(self):
        return [self.getFields1(), self.getFields1(), self.getFields1()]

    def getFields1(self):
        if not self.out:
            self.updateState()
        return self.out.pop(0)

    def put(self, a):
        """transcript.js:49-66.  A list is absorbed block-wise: all the full blocks of 8 it completes go through ONE device
        call (pil2gl_sponge_absorb chains their permutations), which leaves exactly the state element-by-element `put`s leave"""
        flat = []

        def walk(v):
            if isinstance(v, (list, tuple, np.ndarray)):
                for x in v:
                    walk(x)
            else:
                flat.append(int(v) % P)
        walk(a)
        i = 0
        while i < len(flat):
            need = 8 - len(self.pending)
            if len(flat) - i < need:
                self.pending.extend(flat[i:]); self.out = []
                return
            n_full = 1 + (len(flat) - i - need) // 8
            take = need + 8 * (n_full - 1)
            blocks = np.array(self.pending + flat[i:i + take], dtype=np.uint64)
            if n_full == 1:
                self.out = poseidon(blocks.tolist(), self.state, 12)
            else:
                cap = np.array([int(x) % P for x in self.state], dtype=np.uint64); out = np.zeros(12, np.uint64)
                call("pil2gl_sponge_absorb", _ptr(blocks), n_full, _ptr(cap), _ptr(out))
                self.out = [int(v) for v in out]
            self.pending = []; self.state = self.out[:4]
            i += take

    def updateState(self):
        while len(self.pending) < 8:
            self.pending.append(0)
        self.out = poseidon(self.pending, self.state, 12)
        self.pending = []
        self.state = self.out[:4]

    def _add1(self, a):
        self.out = []
        self.pending.append(a)
        if len(self.pending) == 8:
            self.out = poseidon(self.pending, self.state, 12)
            self.pending = []
            self.state = self.out[:4]

    def getPermutations(self, n, nBits):
        totalBits = n * nBits
        NFields = (totalBits - 1) // 63 + 1
        fields = [self.getFields1() for _ in range(NFields)]
        res = []; curField = 0; curBit = 0
        for _ in range(n):
            a = 0
            for j in range(nBits):
                if (fields[curField] >> curBit) & 1:
                    a += 1 << j
                curBit += 1
                if curBit == 63:
                    curBit = 0; curField += 1
            res.append(a)
        return res


# ----------------------------------------------------------------------------- fri.js
def _inv(a):
    return pow(int(a), P - 2, P)


def _root_of_unity(bits):
    w = 7277203076849721926                              # f3g.js:40: F.w[32]; w[k] = w[k+1]^2 (fft.js:45-50)
    for _ in range(32 - bits):
        w = w * w % P
    return w


class FRI:
    """fri.js:7-175.  Polynomials are (n,3) uint64 arrays / int64 CUDA tensors of extension elements."""

    def __init__(self, starkStruct, MH):
        if not starkStruct:
            raise Pil2glError("stark struct not defined")
        self.inNBits = starkStruct["nBitsExt"]
        self.maxDegNBits = starkStruct["nBits"]
        self.nQueries = starkStruct["nQueries"]
        self.steps = starkStruct["steps"]
        self.MH = MH

    def fold(self, step, pol, challenge):
        n = int(np.prod(pol.shape)) // 3
        polBits = n.bit_length() - 1
        if step == 0:
            assert polBits == self.inNBits, "Invalid polynomial size"
        else:
            assert (1 << polBits) == n, "Invalid polynomial size"
        shiftInv = _inv(SHIFT)                          # fri.js:31-36
        if step > 0:
            for _ in range(self.steps[0]["nBits"] - self.steps[step - 1]["nBits"]):
                shiftInv = shiftInv * shiftInv % P
        outBits = self.steps[step]["nBits"]
        dev = _is_dev(pol)
        if step == 0:                                   # fri.js:48-49
            pol2_e = pol
        else:
            ch = np.array([int(c) % P for c in challenge], dtype=np.uint64)
            if dev:
                pol2_e = torch.empty((1 << outBits, 3), dtype=torch.int64, device=pol.device)
                call("pil2gl_fri_fold_dev", _ptr(pol), polBits, outBits, shiftInv, _ptr(ch), _ptr(pol2_e), _stream())
            else:
                pol2_e = np.zeros((1 << outBits, 3), np.uint64)
                call("pil2gl_fri_fold", _ptr(pol), polBits, outBits, shiftInv, _ptr(ch), _ptr(pol2_e))
        tree = None
        if step != len(self.steps) - 1:                 # fri.js:64-71
            nGroupsBits = self.steps[step + 1]["nBits"]
            nGroups = 1 << nGroupsBits
            groupSize = (1 << outBits) // nGroups
            if dev:
                tb = torch.empty((1 << outBits) * 3, dtype=torch.int64, device=pol.device)
                call("pil2gl_fri_transpose_dev", _ptr(pol2_e), outBits, nGroupsBits, _ptr(tb), _stream())
            else:
                tb = np.zeros((1 << outBits) * 3, np.uint64)
                call("pil2gl_fri_transpose", _ptr(np.ascontiguousarray(pol2_e)), outBits, nGroupsBits, _ptr(tb))
            tree = self.MH.merkelize(tb, 3 * groupSize, nGroups)
            proof = {"root": self.MH.root(tree)}
        else:                                           # fri.js:72-75
            proof = [[int(x) for x in row] for row in _to_host(pol2_e).reshape(-1, 3)]
        return {"pol": pol2_e, "tree": tree, "proof": proof}

    def verify(self, friChallenges, friQueries, proof, checkQuery):
        """fri.js:107-174.  checkQuery(polQuery, idx) -> the step-0 group of a query ([value], stark_verify.js:158-215) or a false
        value.  Per layer the openings of all queries are checked in one device call (MH.verifyGroupProofs) and their groups
        folded in one (pil2gl_fri_verify_fold); friQueries is reduced in place, as the reference does."""
        assert len(proof) == len(self.steps) + 1, "Invalid proof size"
        nQ = self.nQueries
        polBits, shift = self.inNBits, SHIFT
        for si in range(len(self.steps)):
            item = proof[si]
            reductionBits = polBits - self.steps[si]["nBits"]
            if si == 0:
                groups = []
                for i in range(nQ):
                    g = checkQuery(item["polQueries"][i], friQueries[i])
                    if g is None or g is False:
                        return False
                    groups.append(g)
            else:
                pq = item["polQueries"]
                if not self.MH.verifyGroupProofs(item["root"], [(q[0], q[1]) for q in pq[:nQ]], list(friQueries[:nQ])):
                    return False
                groups = [np.array([int(v) % P for v in q[0]], dtype=np.uint64).reshape(-1, 3) for q in pq[:nQ]]   # split3, fri.js:179-185
            G = np.array([[[int(c) % P for c in e] for e in g] for g in groups], dtype=np.uint64).reshape(nQ, -1, 3)
            nX = G.shape[1]
            foldBits = nX.bit_length() - 1
            assert (1 << foldBits) == nX, "Invalid group size"
            w = _root_of_unity(polBits)
            sinv = np.array([_inv(shift * pow(w, int(friQueries[i]), P) % P) for i in range(nQ)], dtype=np.uint64)   # fri.js:126
            ch = np.array([int(c) % P for c in friChallenges[si]], dtype=np.uint64)
            ev = np.zeros((nQ, 3), np.uint64)
            Gt = np.ascontiguousarray(G.transpose(1, 0, 2))      # row i = element i of every query's group
            call("pil2gl_fri_verify_fold", _ptr(Gt), foldBits, nQ, _ptr(sinv), _ptr(ch), _ptr(ev))
            for i in range(nQ):
                if si < len(self.steps) - 1:
                    groupIdx = int(friQueries[i]) // (1 << self.steps[si + 1]["nBits"])
                    query = proof[si + 1]["polQueries"][i][0]
                    nxt = [int(v) % P for v in query[3 * groupIdx:3 * groupIdx + 3]]
                else:
                    nxt = [int(v) % P for v in proof[si + 1][int(friQueries[i])]]
                if nxt != [int(v) for v in ev[i]]:
                    return False
            polBits = self.steps[si]["nBits"]
            for _ in range(reductionBits):
                shift = shift * shift % P
            if si < len(self.steps) - 1:
                for i in range(len(friQueries)):
                    friQueries[i] = friQueries[i] % (1 << self.steps[si + 1]["nBits"])
        last = np.array([[int(c) % P for c in e] for e in proof[-1]], dtype=np.uint64).reshape(-1, 3)
        maxDeg = 0 if polBits - (self.inNBits - self.maxDegNBits) < 0 else 1 << (polBits - (self.inNBits - self.maxDegNBits))
        if last.shape[0] > 1:                            # the extension iNTT is the base-field one on each of the three coordinates
            coef = np.zeros_like(last)
            ifft(last, 3, last.shape[0].bit_length() - 1, coef)
            if coef[maxDeg + 1:].any():                  # no division by the shift needed to look for zeros (fri.js:166)
                return False
        return True

    def proofQueries(self, proof, trees, friQueries):
        """fri.js:83-105"""
        for step in range(len(self.steps)):
            proof[step]["polQueries"] = []
            if step == 0:
                for q in friQueries:
                    proof[step]["polQueries"].append([self.MH.getGroupProof(t, q) for t in trees[step]])
            else:
                for i in range(len(friQueries)):
                    friQueries[i] = friQueries[i] % (1 << self.steps[step]["nBits"])
                for q in friQueries:
                    proof[step]["polQueries"].append(self.MH.getGroupProof(trees[step], q))
