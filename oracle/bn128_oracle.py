"""CPU restatement of the BN128 Merkle commitment path on Python integers.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke): the product never imports this file.

The field arithmetic and Poseidon of this path live in third-party packages that are NOT under /root/reference:
circomlibjs@0.1.7 (buildPoseidon / buildPoseidonWasm) and wasmcurves@0.1.5 (buildF1m) -- call sites
merklehash_bn128_p.js:4-5,12,290-292 and merklehash_bn128_worker.js:55-95,138.  What is restated here:
  * the permutation: round structure of circuits.bn128/custom/poseidon.circom:6-45 (the in-tree statement of the
    same function: add C[t*r+j], x^5 on all lanes in the 4+4 outer rounds / lane 0 otherwise, dense M);
  * its parameters: the published Poseidon parameter generation (Grain LFSR stream: field=1, sbox=0, n=254, t, RF=8,
    RP(t); round constants by rejection sampling, then 2t stream values x_i, y_j -> Cauchy matrix 1/(x_i+y_j)),
    PINNED against every constant set the reference tree holds: poseidon_constants_original.circom (t=3,5,7,9,17)
    and src/final/poseidon_constants.js (4,7,8,16 inputs) -- tests/golden/poseidon_bn128_constants.json;
  * leaf rule merklehash_bn128_worker.js:42-98, tree merklehash_bn128_p.js:28-129, openings :142-182, verification
    :184-232 with linearhash.bn128.js:13-59, transcript transcript.bn128.js:1-106.
Pinned end to end by a proof the reference itself wrote: test/final/verifier.proof.zkin.json (arity 4, t=5 and the
generated t=4), tests/test_bn128_oracle.py.
"""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617     # BN254 scalar field
N_ROUNDS_F = 8
N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]        # poseidon.circom:8, t = 2..17
MONT_R = 1 << 256                                                                    # wasmcurves F1m: 4 x 64-bit limbs
GL_P = 0xFFFFFFFF00000001

_CONST = {}


def poseidon_constants(t):
    """(C[(RF+RP)*t], M[t][t]) for state width t (2..17)"""
    if t in _CONST:
        return _CONST[t]
    if not 2 <= t <= 17:
        raise ValueError("Poseidon BN128 supports 1..16 inputs")
    rp = N_ROUNDS_P[t - 2]
    bits = []

    def put(v, w):
        bits.extend(int(c) for c in bin(v)[2:].zfill(w))
    put(1, 2); put(0, 4); put(254, 12); put(t, 12); put(N_ROUNDS_F, 10); put(rp, 10); bits.extend([1] * 30)

    def step():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0); bits.append(nb)
        return nb
    for _ in range(160):
        step()

    def nextbit():                       # self-shrinking: a 1 lets the following bit through, a 0 drops it
        nb = step()
        while nb == 0:
            step(); nb = step()
        return step()

    def rnd():
        v = 0
        for _ in range(254):
            v = (v << 1) | nextbit()
        return v
    C = []
    for _ in range((N_ROUNDS_F + rp) * t):
        v = rnd()
        while v >= R:
            v = rnd()
        C.append(v)
    xy = [rnd() % R for _ in range(2 * t)]
    assert len(set(xy)) == 2 * t
    M = [[pow((xy[i] + xy[t + j]) % R, R - 2, R) for j in range(t)] for i in range(t)]
    _CONST[t] = (C, M)
    return C, M


def poseidon(inputs, init_state=0, n_out=1):
    """circomlibjs poseidon(inputs, initState, nOut) = poseidon.circom:6-45 with in[0] = initState; -> list of n_out"""
    t = len(inputs) + 1
    C, M = poseidon_constants(t)
    rp = N_ROUNDS_P[t - 2]
    st = [init_state % R] + [int(x) % R for x in inputs]
    for r in range(N_ROUNDS_F + rp):
        st = [(a + C[t * r + j]) % R for j, a in enumerate(st)]
        if r < N_ROUNDS_F // 2 or r >= N_ROUNDS_F // 2 + rp:
            st = [pow(a, 5, R) for a in st]
        else:
            st[0] = pow(st[0], 5, R)
        st = [sum(M[i][j] * st[j] for j in range(t)) % R for i in range(t)]
    return st[:n_out]


def linear_hash_worker(vals, arity, custom):
    """leaf digest as the Merkle worker computes it (merklehash_bn128_worker.js:42-98) -> Fr value (normal form).
    width <= 4: the words are taken as ONE 256-bit little-endian integer (:45-50, reduced by toMontgomery), unlike
    LinearHashBN.hash which packs 3 per element -- the two differ at width 4 (reference quirk, kept)."""
    vals = [int(v) for v in vals]
    if len(vals) <= 4:
        return sum(v << (64 * i) for i, v in enumerate(vals)) % R
    st = 0
    elems = [sum(v << (64 * k) for k, v in enumerate(vals[i:i + 3])) for i in range(0, len(vals), 3)]
    for i in range(0, len(elems), arity):
        chunk = elems[i:i + arity]
        if len(chunk) < arity and custom:
            chunk = chunk + [0] * (arity - len(chunk))
        st = poseidon(chunk, st, 1)[0]
    return st


def linear_hash_class(vals, arity, custom):
    """LinearHashBN.hash (linearhash.bn128.js:13-59), used by calculateRootFromGroupProof"""
    flat = []
    for v in vals:
        if isinstance(v, (list, tuple)):
            flat.extend(int(x) for x in v)
        else:
            flat.append(int(v))
    elems = [sum(v << (64 * k) for k, v in enumerate(flat[i:i + 3])) % R for i in range(0, len(flat), 3)]
    if not elems:
        return 0
    if len(elems) == 1:
        return elems[0]
    st = 0
    for i in range(0, len(elems), arity):
        chunk = elems[i:i + arity]
        if len(chunk) < arity and custom:
            chunk = chunk + [0] * (arity - len(chunk))
        st = poseidon(chunk, st, 1)[0]
    return st


def merkle_num_nodes(height, arity):
    """_getNNodes (merklehash_bn128_p.js:31-45), in Fr nodes"""
    n = height
    nxt = (n - 1) // arity + 1
    acc = nxt * arity
    while n > 1:
        n = nxt
        nxt = (n - 1) // arity + 1
        acc += nxt * arity if n > 1 else 1
    return acc


def merkelize(rows, arity, custom):
    """rows: list of rows (lists of GL values) -> list of Fr node values (normal form), layout of merklehash_bn128_p.js:
    leaf level padded with zero nodes to a multiple of arity, then each level likewise, the root alone at the end"""
    h = len(rows)
    nodes = [0] * merkle_num_nodes(h, arity)
    for i, r in enumerate(rows):
        nodes[i] = linear_hash_worker(r, arity, custom)
    p_in, n = 0, h
    nxt = (n - 1) // arity + 1
    p_out = p_in + nxt * arity
    while n > 1:
        for i in range(nxt):
            nodes[p_out + i] = poseidon(nodes[p_in + i * arity:p_in + (i + 1) * arity], 0, 1)[0]
        n = nxt
        nxt = (n - 1) // arity + 1
        p_in = p_out
        p_out = p_in + nxt * arity
    return nodes


def root(nodes):
    return nodes[-1]


def group_proof(nodes, height, arity, idx):
    """siblings per level: all `arity` nodes of idx's group (merklehash_bn128_p.js:155-181)"""
    nbits = (arity - 1).bit_length()
    mp, offset, n = [], 0, height
    while n > 1:
        si = idx ^ (idx & (arity - 1))
        mp.append([nodes[offset + si + i] if i < n else 0 for i in range(arity)])
        nxt = (n - 1) // arity + 1
        offset += nxt * arity
        n = nxt
        idx >>= nbits
    return mp


def root_from_group_proof(mp, idx, vals, arity, custom):
    """calculateRootFromGroupProof (merklehash_bn128_p.js:184-232)"""
    value = linear_hash_class(vals, arity, custom)
    nbits = (arity - 1).bit_length()
    for sibs in mp:
        cur = idx & (arity - 1)
        idx >>= nbits
        group = [int(s) % R for s in sibs]
        group[cur] = value
        value = poseidon(group, 0, 1)[0]
    return value


def to_montgomery_words(x):
    """Fr value -> 4 little-endian u64 words of x*2^256 mod r, the form tree.nodes holds (frm_toMontgomery)"""
    m = (x % R) * MONT_R % R
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_montgomery_words(w):
    m = sum(int(v) << (64 * i) for i, v in enumerate(w))
    return m * pow(MONT_R, R - 2, R) % R


class TranscriptBN128:
    """transcript.bn128.js:1-106"""

    def __init__(self, n_inputs=16):
        self.nInputs = n_inputs
        self.state = 0
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
        for v in (a if isinstance(a, (list, tuple)) else [a]):
            if isinstance(v, (list, tuple)):
                self.put(v)
            else:
                self.out = []
                self.pending.append(int(v) % R)
                if len(self.pending) == self.nInputs:
                    self.updateState()

    def getPermutations(self, n, n_bits):
        total = n * n_bits
        fields = [self.getFields253() for _ in range((total - 1) // 253 + 1)]
        res, cur_field, cur_bit = [], 0, 0
        for _ in range(n):
            a = 0
            for j in range(n_bits):
                if (fields[cur_field] >> cur_bit) & 1:
                    a += 1 << j
                cur_bit += 1
                if cur_bit == 253:
                    cur_bit = 0
                    cur_field += 1
            res.append(a)
        return res


# ---- the same statements in C (oracle/bn128_oracle.c): the config-4 cpu_baseline of bench.py runs on these; checked against the
#      Python-integer functions above in tests/test_bn128_oracle.py
_CLIB = None


def c_lib():
    """builds (make) and loads oracle/libbn128_oracle.so; hands it the constants of the widths it is asked for on demand"""
    global _CLIB
    if _CLIB is None:
        import ctypes as C
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "libbn128_oracle.so")
        src = os.path.join(here, "bn128_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", here, "-s", "libbn128_oracle.so"])
        L = C.CDLL(so)
        L.bn_merkle_num_nodes.restype = C.c_uint64
        L.bn_merkle_num_nodes.argtypes = [C.c_uint64, C.c_int]
        _CLIB = L
    return _CLIB


def _words(vals):
    import numpy as np
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v)
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def c_need_params(t):
    import ctypes as C
    L = c_lib()
    if not L.bn_have_params(int(t)):
        Cs, M = poseidon_constants(t)
        cw, mw = _words(Cs), _words([M[i][j] for i in range(t) for j in range(t)])
        assert L.bn_set_params(int(t), C.c_void_p(cw.ctypes.data), C.c_void_p(mw.ctypes.data)) == 0


def c_poseidon(inputs, init_state=0, n_out=1):
    import ctypes as C
    import numpy as np
    t = len(inputs) + 1
    c_need_params(t)
    iw, sw = _words([int(x) % R for x in inputs]), _words([init_state % R])
    out = np.zeros((n_out, 4), dtype=np.uint64)
    assert c_lib().bn_poseidon(len(inputs), C.c_void_p(iw.ctypes.data), C.c_void_p(sw.ctypes.data), n_out, C.c_void_p(out.ctypes.data)) == 0
    return [sum(int(out[i, k]) << (64 * k) for k in range(4)) for i in range(n_out)]


def c_merkelize_words(rows, arity, custom):
    """rows: numpy uint64 [height][width] -> numpy uint64 [nNodes][4], MONTGOMERY words (the reference's tree.nodes)"""
    import ctypes as C
    import numpy as np
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    h, w = rows.shape
    n_el = (w + 2) // 3
    if w > 4:
        for e0 in range(0, n_el, arity):
            c_need_params((arity if custom else min(arity, n_el - e0)) + 1)
    if h > 1:
        c_need_params(arity + 1)
    L = c_lib()
    nodes = np.zeros((int(L.bn_merkle_num_nodes(h, arity)), 4), dtype=np.uint64)
    rc = L.bn_merkelize(C.c_void_p(rows.ctypes.data), C.c_uint64(w), C.c_uint64(h), int(arity), int(bool(custom)), C.c_void_p(nodes.ctypes.data))
    assert rc == 0
    return nodes


def c_merkelize(rows, arity, custom):
    """-> list of Fr node values in normal form, like merkelize()"""
    return [from_montgomery_words(w) for w in c_merkelize_words(rows, arity, custom)]
