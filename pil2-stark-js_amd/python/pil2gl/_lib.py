"""ctypes loader for libpil2gl.so (the C ABI declared in include/pil2gl.h).

There is no fallback: if the library is missing, or a call fails (e.g. no HIP device), an
exception is raised -- nothing here computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))                 # pil2-stark-js_amd/
LIB_PATH = os.environ.get("PIL2GL_LIB") or os.path.join(PKG_ROOT, "lib", "libpil2gl.so")      # PIL2GL_LIB: another build of the same ABI (A/B measurements)

u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class Pil2glError(RuntimeError):
    pass


class GlxRef(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("dim", C.c_uint8), ("section", C.c_uint16),
                ("prime", C.c_int32), ("index", C.c_uint32), ("pad_", C.c_uint32)]


class GlxOp(C.Structure):
    _fields_ = [("op", C.c_uint32), ("pad_", C.c_uint32), ("dest", GlxRef), ("src", GlxRef * 2)]


class GlxSection(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_uint64)]


class GlxCtx(C.Structure):
    _fields_ = [("nBits", C.c_uint32), ("primeShift", C.c_uint32), ("nSections", C.c_uint32),
                ("nScalars", C.c_uint32), ("sections", C.POINTER(GlxSection)), ("scalars", u64p)]


class GlxProgram(C.Structure):
    _fields_ = [("nOps", C.c_uint32), ("nTmp", C.c_uint32), ("ops", C.POINTER(GlxOp))]


class EvalDesc(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("width", C.c_uint64), ("offset", C.c_uint64),
                ("dim", C.c_uint32), ("levIndex", C.c_uint32)]


# name -> (restype, argtypes); pointers that may be host or device are c_void_p
_U64, _U32, _I = C.c_uint64, C.c_uint32, C.c_int
SIGNATURES = {
    "pil2gl_init": (_I, [_I]),
    "pil2gl_shutdown": (None, []),
    "pil2gl_last_error": (C.c_char_p, []),
    "pil2gl_version": (_I, []),
    "pil2gl_device_info": (_I, [C.c_char_p, _U32, C.POINTER(_U32), C.POINTER(_U64)]),
    "pil2gl_dev_alloc": (_I, [_U64, C.POINTER(vp)]),
    "pil2gl_dev_free": (_I, [vp]),
    "pil2gl_dev_zero": (_I, [vp, _U64, vp]),
    "pil2gl_dev_upload": (_I, [vp, vp, _U64]),
    "pil2gl_add": (_U64, [_U64, _U64]),
    "pil2gl_mul": (_U64, [_U64, _U64]),
    "pil2gl_square": (_U64, [_U64]),
    "pil2gl_dev_download": (_I, [vp, vp, _U64]),
    "pil2gl_sync": (_I, [vp]),
    "pil2gl_interpolate": (_I, [vp, _U64, _U32, vp, _U32]),
    "pil2gl_interpolate_dev": (_I, [vp, _U64, _U32, vp, _U32, vp]),
    "pil2gl_interpolate_cosets_dev": (_I, [vp, _U64, _U32, vp, _U32, _U32, _U32, vp]),
    "pil2gl_extend_cosets_unshifted_dev": (_I, [vp, _U64, _U32, vp, _U32, _U32, _U32, vp]),
    "pil2gl_interpolate_cosets_ws_dev": (_I, [vp, _U64, _U32, vp, _U32, _U32, _U32, vp, vp]),
    "pil2gl_fft": (_I, [vp, _U64, _U32, vp]),
    "pil2gl_ifft": (_I, [vp, _U64, _U32, vp]),
    "pil2gl_fft_dev": (_I, [vp, _U64, _U32, vp, vp]),
    "pil2gl_ifft_dev": (_I, [vp, _U64, _U32, vp, vp]),
    "pil2gl_poseidon": (_I, [vp, vp, _U64, _U32, vp]),
    "pil2gl_poseidon_dev": (_I, [vp, vp, _U64, _U32, vp, vp]),
    "pil2gl_linear_hash_rows": (_I, [vp, _U64, _U64, _I, vp]),
    "pil2gl_linear_hash_rows_dev": (_I, [vp, _U64, _U64, _I, vp, vp]),
    "pil2gl_merkelize_level": (_I, [vp, _U64, vp]),
    "pil2gl_merkelize_level_dev": (_I, [vp, _U64, vp, vp]),
    "pil2gl_merkle_num_nodes": (_U64, [_U64]),
    "pil2gl_merkelize": (_I, [vp, _U64, _U64, _I, vp]),
    "pil2gl_merkelize_dev": (_I, [vp, _U64, _U64, _I, vp, vp]),
    "pil2gl_merkelize_digests_dev": (_I, [vp, _U64, vp]),
    "pil2gl_group_proof_dev": (_I, [vp, vp, _U64, _U64, _U64, vp, vp, C.POINTER(_U32)]),
    "pil2gl_group_proofs_dev": (_I, [vp, vp, _U64, _U64, vp, _U32, vp, C.POINTER(_U32)]),
    "pil2gl_roots_from_group_proofs": (_I, [vp, _U64, _U32, vp, _U32, C.c_int, vp]),
    "pil2gl_sponge_absorb": (_I, [vp, _U64, vp, vp]),
    "pil2gl_bn128_sponge_absorb": (_I, [vp, _U64, _U32, vp, vp]),
    "pil2gl_fri_fold": (_I, [vp, _U32, _U32, _U64, vp, vp]),
    "pil2gl_fri_fold_dev": (_I, [vp, _U32, _U32, _U64, vp, vp, vp]),
    "pil2gl_fri_verify_fold": (_I, [vp, _U32, _U32, vp, vp, vp]),
    "pil2gl_fri_verify_fold_dev": (_I, [vp, _U32, _U32, vp, vp, vp, vp]),
    "pil2gl_fri_transpose": (_I, [vp, _U32, _U32, vp]),
    "pil2gl_fri_transpose_dev": (_I, [vp, _U32, _U32, vp, vp]),
    "pil2gl_build_x_dev": (_I, [_U32, _U64, vp, vp]),
    "pil2gl_geometric_dev": (_I, [_U64, _U64, _U64, vp, vp]),
    "pil2gl_build_zhinv_dev": (_I, [_U32, _U32, vp, vp]),
    "pil2gl_build_one_row_zerofier_inv_dev": (_I, [_U32, _U32, _U64, vp, vp]),
    "pil2gl_build_frame_zerofier_dev": (_I, [_U32, _U32, _U64, _U64, vp, vp]),
    "pil2gl_compute_q_split_dev": (_I, [vp, _U32, _U32, _U32, _U32, vp, vp]),
    "pil2gl_compute_q_split_brev_dev": (_I, [vp, _U32, _U32, _U32, _U32, vp, vp]),
    "pil2gl_extend_coefs_brev_dev": (_I, [vp, _U64, _U32, vp, _U32, vp]),
    "pil2gl_extend_coefs_brev_cosets_dev": (_I, [vp, _U64, _U32, vp, _U32, _U32, _U32, vp]),
    "pil2gl_x_div_x_sub_xi_dev": (_I, [_U32, vp, _U64, _U64, vp, vp]),
    "pil2gl_x_div_x_sub_xi_cosets_dev": (_I, [_U32, _U32, vp, _U64, _U64, _U32, _U32, vp, vp]),
    "pil2gl_build_lev_dev": (_I, [_U32, vp, vp, vp]),
    "pil2gl_compute_evals_dev": (_I, [C.POINTER(EvalDesc), _U32, _U32, _U32, C.POINTER(vp), _U32, vp, vp]),
    "pil2gl_rows_dot_ext_dev": (_I, [vp, _U64, _U64, vp, _U32, vp, _I, vp]),
    "pil2gl_rows_dot_ext_multi_dev": (_I, [vp, vp, _U32, _U64, vp, _U32, vp, _I, vp]),
    "pil2gl_fri_combine_dev": (_I, [vp, vp, vp, vp, _U32, _U64, vp, vp]),
    "pil2gl_fri_combine_order_dev": (_I, [vp, vp, vp, vp, _U32, vp, _U64, vp, vp]),
    "pil2gl_cols_dot_ext_dev": (_I, [vp, _U64, _U64, _U64, C.POINTER(vp), _U32, vp, vp]),
    "pil2gl_cols_dot_ext_multi_dev": (_I, [vp, vp, _U32, _U64, _U64, vp, _U32, vp, vp]),
    "pil2gl_cols_dot_ext_range_dev": (_I, [vp, vp, vp, vp, _U32, _U64, _U64, vp, _U32, vp, vp]),
    "pil2gl_eval_program_dev": (_I, [C.POINTER(GlxProgram), C.POINTER(GlxCtx), vp]),
    "pil2gl_first_nonzero_row_dev": (_I, [vp, _U32, _U64, _U64, vp, vp, vp]),
    "pil2gl_fft_block_dev": (_I, [vp, _U64, _U64, _U32, _U32, _U32, _U32, vp]),
    "pil2gl_interpolate_prepare_block_dev": (_I, [vp, _U64, _U64, _U64, _U64, vp]),
    "pil2gl_synth_fibonacci_dev": (_I, [_U32, _U32, vp, vp, vp]),
    "pil2gl_debug_compact_program": (_I, [C.POINTER(GlxProgram), C.POINTER(GlxOp), C.POINTER(_U32)]),
    "pil2gl_debug_jit_compile": (_I, [C.POINTER(GlxProgram), C.POINTER(GlxCtx), C.POINTER(_U64), C.POINTER(_U32)]),
    "pil2gl_gprod_dev": (_I, [vp, _U32, vp, _U32, _U64, vp, vp]),
    "pil2gl_gsum_dev": (_I, [vp, _U32, vp, _U32, _U64, vp, vp]),
    "pil2gl_h1h2_dev": (_I, [vp, vp, _U64, _U32, vp, vp, vp]),
    "pil2gl_bn128_poseidon": (_I, [vp, vp, _U64, _U32, _U32, vp]),
    "pil2gl_bn128_poseidon_dev": (_I, [vp, vp, _U64, _U32, _U32, vp, vp]),
    "pil2gl_bn128_linear_hash_rows": (_I, [vp, _U64, _U64, _U32, _I, vp]),
    "pil2gl_bn128_linear_hash_rows_dev": (_I, [vp, _U64, _U64, _U32, _I, vp, vp]),
    "pil2gl_bn128_merkelize_level_dev": (_I, [vp, _U64, _U32, vp, vp]),
    "pil2gl_bn128_merkle_num_nodes": (_U64, [_U64, _U32]),
    "pil2gl_bn128_merkelize": (_I, [vp, _U64, _U64, _U32, _I, vp]),
    "pil2gl_bn128_merkelize_dev": (_I, [vp, _U64, _U64, _U32, _I, vp, vp]),
    "pil2gl_bn128_group_proof_dev": (_I, [vp, vp, _U64, _U64, _U32, _U64, vp, vp, vp]),
    "pil2gl_bn128_group_proofs_dev": (_I, [vp, vp, _U64, _U64, _U32, vp, _U32, vp, vp, vp]),
    "pil2gl_bn128_convert": (_I, [vp, _U64, _I, vp]),
    "pil2gl_bn128_convert_dev": (_I, [vp, _U64, _I, vp, vp]),
    "pil2gl_selftest_field": (_I, [vp, vp, _U64, vp, vp, vp]),
    "pil2gl_selftest_ext": (_I, [vp, vp, _U64, vp, vp]),
    "pil2gl_selftest_products": (_I, [vp, vp, _U64, vp, vp, vp]),
    "pil2gl_selftest_mds": (_I, [vp, _U64, _U32, _I, vp]),
    "pil2gl_selftest_poseidon": (_I, [vp, _U64, _I, vp]),
    "pil2gl_selftest_clock": (_I, [C.c_uint32, vp]),
}

_lib = None


def load():
    """Load libpil2gl.so and bind every symbol of include/pil2gl.h (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Pil2glError("libpil2gl.so not built (%s): run `make -C pil2-stark-js_amd` or __graft_entry__.build(); "
                          "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().pil2gl_last_error()
        raise Pil2glError("pil2gl error %d: %s" % (rc, msg.decode() if msg else "?"))


def call(name, *args):
    check(getattr(load(), name)(*args))
