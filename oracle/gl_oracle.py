"""ctypes binding of the CPU oracle (oracle/libgl_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
Arrays are numpy uint64, matrices row-major (rows, cols).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libgl_oracle.so")
P = 0xFFFFFFFF00000001

u64p = C.POINTER(C.c_uint64)


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


def _ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


class _GlxRef(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("dim", C.c_uint8), ("section", C.c_uint16),
                ("prime", C.c_int32), ("index", C.c_uint32), ("pad_", C.c_uint32)]


class _GlxOp(C.Structure):
    _fields_ = [("op", C.c_uint32), ("pad_", C.c_uint32), ("dest", _GlxRef), ("src", _GlxRef * 2)]


class _GlxSection(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_uint64)]


class _GlxCtx(C.Structure):
    _fields_ = [("nBits", C.c_uint32), ("primeShift", C.c_uint32), ("nSections", C.c_uint32),
                ("nScalars", C.c_uint32), ("sections", C.POINTER(_GlxSection)), ("scalars", u64p)]


class _GlxProgram(C.Structure):
    _fields_ = [("nOps", C.c_uint32), ("nTmp", C.c_uint32), ("ops", C.POINTER(_GlxOp))]


class _Transcript(C.Structure):
    _fields_ = [("state", C.c_uint64 * 4), ("pending", C.c_uint64 * 8), ("nPending", C.c_int),
                ("out", C.c_uint64 * 12), ("nOut", C.c_int), ("outPos", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        for n in ("or_add", "or_sub", "or_mul"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_uint64, C.c_uint64]
        L.or_inv.restype = C.c_uint64; L.or_inv.argtypes = [C.c_uint64]
        L.or_exp.restype = C.c_uint64; L.or_exp.argtypes = [C.c_uint64, C.c_uint64]
        L.or_root.restype = C.c_uint64; L.or_root.argtypes = [C.c_int]
        L.or_root_inv.restype = C.c_uint64; L.or_root_inv.argtypes = [C.c_int]
        L.or_merkle_num_nodes.restype = C.c_uint64; L.or_merkle_num_nodes.argtypes = [C.c_uint64]
        L.or_fri_shift_inv.restype = C.c_uint64; L.or_fri_shift_inv.argtypes = [C.c_int, C.c_int]
        L.or_transcript_get1.restype = C.c_uint64
        L.or_group_proof.restype = C.c_int
        _lib = L
    return _lib


def set_threads(n):
    lib().or_set_threads(C.c_int(int(n)))


def _u(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


# ---- field ----
def add(a, b): return lib().or_add(a, b)
def sub(a, b): return lib().or_sub(a, b)
def mul(a, b): return lib().or_mul(a, b)
def inv(a): return lib().or_inv(a)
def exp(a, e): return lib().or_exp(a, e)
def root(bits): return lib().or_root(bits)
def root_inv(bits): return lib().or_root_inv(bits)


def mul3(a, b):
    a, b = _u(a), _u(b); r = np.zeros(3, np.uint64)
    lib().or3_mul(_ptr(a), _ptr(b), _ptr(r)); return r


def inv3(a):
    a = _u(a); r = np.zeros(3, np.uint64)
    lib().or3_inv(_ptr(a), _ptr(r)); return r


def batch_inverse(a):
    a = _u(a); r = np.zeros_like(a)
    lib().or_batch_inverse(_ptr(a), C.c_uint64(a.size), _ptr(r)); return r


def batch_inverse3(a):
    a = _u(a).reshape(-1, 3); r = np.zeros_like(a)
    lib().or3_batch_inverse(_ptr(a), C.c_uint64(a.shape[0]), _ptr(r)); return r


# ---- stage-2 hints (polutils.js:128-164) ----
def _hint(fn, num, den, dim_num, dim_den):
    num, den = _u(num).reshape(-1), _u(den).reshape(-1)
    n = den.size // dim_den
    out = np.zeros(n * (3 if 3 in (dim_num, dim_den) else 1), np.uint64)
    getattr(lib(), fn)(_ptr(num), C.c_int(dim_num), _ptr(den), C.c_int(dim_den), C.c_uint64(n), _ptr(out)); return out


def gprod(num, den, dim_num, dim_den): return _hint("or_gprod", num, den, dim_num, dim_den)
def gsum(num, den, dim_num, dim_den): return _hint("or_gsum", num, den, dim_num, dim_den)


def h1h2(f, t):
    """calculateH1H2 (polutils.js:105-126), literally: f, t lists of ints or of 3-tuples (small cases; pure Python)"""
    idx_t, s = {}, []
    for i, v in enumerate(t):
        idx_t[v] = i
        s.append((v, i))
    for i, v in enumerate(f):
        if v not in idx_t:
            raise ValueError("Number not included: w:%d" % i)
        s.append((v, idx_t[v]))
    s.sort(key=lambda e: e[1])                       # stable, like Array.prototype.sort in Node >= 11
    return [s[2 * i][0] for i in range(len(f))], [s[2 * i + 1][0] for i in range(len(f))]


# ---- NTT ----
def fft(p):
    p = _u(p).copy(); lib().or_fft(_ptr(p), C.c_int(int(p.size).bit_length() - 1), C.c_uint64(1)); return p


def ifft(p):
    p = _u(p).copy(); lib().or_ifft(_ptr(p), C.c_int(int(p.size).bit_length() - 1), C.c_uint64(1)); return p


def fft3(p, inverse=False):
    """component-wise (i)fft of an (n,3) array of extension elements"""
    p = _u(p).reshape(-1, 3).copy(); nb = int(p.shape[0]).bit_length() - 1
    f = lib().or_ifft if inverse else lib().or_fft
    for c in range(3):
        f(C.cast(C.c_void_p(p.ctypes.data + 8 * c), u64p), C.c_int(nb), C.c_uint64(3))
    return p


def extend_pol(p, ext_bits):
    p = _u(p); nb = int(p.size).bit_length() - 1
    out = np.zeros(p.size << ext_bits, np.uint64)
    lib().or_extend_pol(_ptr(p), C.c_int(nb), C.c_int(ext_bits), _ptr(out)); return out


def fft_cols(src, n_bits):
    src = _u(src); dst = np.empty_like(src)
    lib().or_fft_cols(_ptr(src), C.c_uint64(src.shape[1]), C.c_int(n_bits), _ptr(dst)); return dst


def ifft_cols(src, n_bits):
    src = _u(src); dst = np.empty_like(src)
    lib().or_ifft_cols(_ptr(src), C.c_uint64(src.shape[1]), C.c_int(n_bits), _ptr(dst)); return dst


def interpolate(src, n_bits, n_bits_ext):
    src = _u(src); dst = np.empty((1 << n_bits_ext, src.shape[1]), np.uint64)
    lib().or_interpolate(_ptr(src), C.c_uint64(src.shape[1]), C.c_int(n_bits), _ptr(dst), C.c_int(n_bits_ext)); return dst


# ---- hashing ----
def poseidon(inp, cap=None, n_out=4):
    inp = _u(inp); out = np.zeros(n_out, np.uint64)
    capp = _ptr(_u(cap)) if cap is not None else None
    lib().or_poseidon(_ptr(inp), capp, _ptr(out), C.c_int(n_out)); return out


def linear_hash(vals, split=False):
    vals = _u(vals); out = np.zeros(4, np.uint64)
    lib().or_linear_hash(_ptr(vals) if vals.size else None, C.c_uint64(vals.size), C.c_int(int(split)), _ptr(out)); return out


def merkle_num_nodes(height): return lib().or_merkle_num_nodes(height)


def merkelize(elems, split=False):
    elems = _u(elems); h, w = elems.shape
    nodes = np.zeros(merkle_num_nodes(h), np.uint64)
    lib().or_merkelize(_ptr(elems), C.c_uint64(w), C.c_uint64(h), C.c_int(int(split)), _ptr(nodes)); return nodes


def group_proof(nodes, height, idx):
    sib = np.zeros((64, 4), np.uint64)
    n = lib().or_group_proof(_ptr(_u(nodes)), C.c_uint64(height), C.c_uint64(idx), _ptr(sib))
    return sib[:n].copy()


def root_from_proof(vals, idx, siblings, split=False):
    vals = _u(vals); sib = _u(siblings).reshape(-1, 4); r = np.zeros(4, np.uint64)
    lib().or_root_from_proof(_ptr(vals), C.c_uint64(vals.size), C.c_int(int(split)), C.c_uint64(idx),
                             _ptr(sib), C.c_int(sib.shape[0]), _ptr(r)); return r


# ---- transcript ----
class Transcript:
    def __init__(self):
        self.t = _Transcript(); lib().or_transcript_init(C.byref(self.t))

    def put(self, a):
        a = _u(a).reshape(-1); lib().or_transcript_put(C.byref(self.t), _ptr(a), C.c_uint64(a.size))

    def get_field(self):
        r = np.zeros(3, np.uint64); lib().or_transcript_get_field(C.byref(self.t), _ptr(r)); return r

    def get_state(self):
        r = np.zeros(4, np.uint64); lib().or_transcript_get_state(C.byref(self.t), _ptr(r)); return r

    def get_permutations(self, n, n_bits):
        r = np.zeros(n, np.uint64); lib().or_transcript_get_permutations(C.byref(self.t), C.c_int(n), C.c_int(n_bits), _ptr(r)); return r


# ---- FRI ----
def fri_shift_inv(bits0, bits_prev): return lib().or_fri_shift_inv(bits0, bits_prev)


def fri_fold(pol, out_bits, shift_inv, challenge):
    pol = _u(pol).reshape(-1, 3); pb = int(pol.shape[0]).bit_length() - 1
    out = np.zeros((1 << out_bits, 3), np.uint64); ch = _u(challenge)
    lib().or_fri_fold(_ptr(pol), C.c_int(pb), C.c_int(out_bits), C.c_uint64(shift_inv), _ptr(ch), _ptr(out)); return out


def fri_transpose(pol, transpose_bits):
    pol = _u(pol).reshape(-1, 3); pb = int(pol.shape[0]).bit_length() - 1
    out = np.zeros_like(pol)
    lib().or_fri_transpose(_ptr(pol), C.c_int(pb), C.c_int(transpose_bits), _ptr(out)); return out


# ---- STARK step helpers ----
def build_x(n_bits, shift=1):
    x = np.zeros(1 << n_bits, np.uint64); lib().or_build_x(C.c_int(n_bits), C.c_uint64(shift), _ptr(x)); return x


def build_zhinv(n_bits, n_bits_ext):
    o = np.zeros(1 << n_bits_ext, np.uint64); lib().or_build_zhinv(C.c_int(n_bits), C.c_int(n_bits_ext), _ptr(o)); return o


def build_one_row_zerofier_inv(n_bits, n_bits_ext, row):
    o = np.zeros(1 << n_bits_ext, np.uint64)
    lib().or_build_one_row_zerofier_inv(C.c_int(n_bits), C.c_int(n_bits_ext), C.c_uint64(row), _ptr(o)); return o


def build_frame_zerofier(n_bits, n_bits_ext, off_min, off_max):
    o = np.zeros(1 << n_bits_ext, np.uint64)
    lib().or_build_frame_zerofier(C.c_int(n_bits), C.c_int(n_bits_ext), C.c_uint64(off_min), C.c_uint64(off_max), _ptr(o)); return o


def compute_q_split(qq1, n_bits, n_bits_ext, q_dim, q_deg):
    qq1 = _u(qq1); o = np.zeros(((1 << n_bits_ext), q_dim * q_deg), np.uint64)
    lib().or_compute_q_split(_ptr(qq1), C.c_int(n_bits), C.c_int(n_bits_ext), C.c_int(q_dim), C.c_int(q_deg), _ptr(o)); return o


def x_div_x_sub_xi(n_bits_ext, xis):
    """xis: (nOpen,3) -> (extN, 3*nOpen) as ctx.xDivXSubXi_ext"""
    xis = _u(xis).reshape(-1, 3); n_open = xis.shape[0]
    o = np.zeros((1 << n_bits_ext, 3 * n_open), np.uint64)
    for i in range(n_open):
        xi = np.ascontiguousarray(xis[i])
        lib().or_x_div_x_sub_xi(C.c_int(n_bits_ext), _ptr(xi), C.c_uint64(n_open), C.c_uint64(i), _ptr(o))
    return o


def lev(n_bits, xi):
    o = np.zeros((1 << n_bits, 3), np.uint64); xi = _u(xi)
    lib().or_lev(C.c_int(n_bits), _ptr(xi), _ptr(o)); return o


def eval_pol_at(buf, offset, dim, n_bits, extend_bits, lev_arr):
    buf = _u(buf); r = np.zeros(3, np.uint64); lev_arr = _u(lev_arr)
    lib().or_eval_pol_at(_ptr(buf), C.c_uint64(buf.shape[1]), C.c_uint64(offset), C.c_int(dim), C.c_int(n_bits),
                         C.c_int(extend_bits), _ptr(lev_arr), _ptr(r)); return r


# ---- expression evaluator ----
OP = {"add": 0, "sub": 1, "mul": 2, "copy": 3}
TMP, SEC, SCALAR = 0, 1, 2


def make_program(ops, n_tmp, struct_op=_GlxOp, struct_prog=_GlxProgram):
    """ops: list of (op, dest, src0, src1|None); ref = (kind, dim, section, prime, index)"""
    arr = (struct_op * len(ops))()
    for k, (op, d, s0, s1) in enumerate(ops):
        arr[k].op = OP[op] if isinstance(op, str) else op
        for tgt, r in ((arr[k].dest, d), (arr[k].src[0], s0), (arr[k].src[1], s1)):
            if r is None:
                continue
            tgt.kind, tgt.dim, tgt.section, tgt.prime, tgt.index = r
    prog = struct_prog(len(ops), n_tmp, arr)
    prog._keep = arr
    return prog


def eval_program(ops, n_tmp, sections, scalars, n_bits, prime_shift, row_begin=0, row_end=None):
    """sections: list of 2-D uint64 numpy arrays (modified in place for destinations)"""
    prog = make_program(ops, n_tmp)
    secs = (_GlxSection * len(sections))()
    for i, s in enumerate(sections):
        assert s.dtype == np.uint64 and s.flags["C_CONTIGUOUS"]
        secs[i].ptr = s.ctypes.data; secs[i].width = s.shape[1]
    scalars = _u(scalars)
    ctx = _GlxCtx(n_bits, prime_shift, len(sections), scalars.size, secs, _ptr(scalars))
    if row_end is None:
        row_end = 1 << n_bits
    rc = lib().or_eval_program(C.byref(prog), C.byref(ctx), C.c_uint64(row_begin), C.c_uint64(row_end))
    if rc != 0:
        raise RuntimeError("or_eval_program failed")
