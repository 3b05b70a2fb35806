"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/pil2gl.h declares,
and refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

pil2gl = pytest.importorskip("pil2gl")
from pil2gl import _lib  # noqa: E402


def _declared():
    src = open(os.path.join(ROOT, "include", "pil2gl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pil2gl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libpil2gl.so does not export " + n
    # the Python binding binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names


def test_expr_struct_layout_matches_header():
    # include/pil2gl_expr.h: glx_ref 16 bytes, glx_op 56 bytes
    assert C.sizeof(_lib.GlxRef) == 16 and C.sizeof(_lib.GlxOp) == 56
    assert C.sizeof(_lib.GlxSection) == 16


def test_merkle_num_nodes_host_only(oracle):
    lib = _lib.load()
    for h in list(range(1, 70)) + [255, 256, 257, 1 << 20, (1 << 20) + 3]:
        assert lib.pil2gl_merkle_num_nodes(h) == oracle.merkle_num_nodes(h)


def test_scalar_field_exports_against_the_references_vectors():
    """pil2gl_add / _mul / _square = the WASM module's scalar exports (glwasm.js:47-96,1269-1275): host arithmetic, against the products
    and sums the reference's f3g.js wrote into tests/golden/field.json; operands above p are reduced first"""
    from conftest import golden, H, P
    lib = _lib.load()
    g = golden("field.json")
    for a, b, m, s, d in H(g["mul"]):
        assert lib.pil2gl_mul(a, b) == m and lib.pil2gl_add(a, b) == s and lib.pil2gl_square(a) == a * a % P
    assert lib.pil2gl_mul(P + 5, 3) == 15 and lib.pil2gl_add(2 ** 64 - 1, 1) == (2 ** 64) % P and lib.pil2gl_square(P - 1) == 1


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback_without_gpu():
    a = np.arange(8, dtype=np.uint64)
    out = np.zeros(16, np.uint64)
    with pytest.raises(pil2gl.Pil2glError) as e:
        pil2gl.interpolate(a, 1, 3, out, 4)
    assert "-2" in str(e.value) or "no HIP device" in str(e.value)
    with pytest.raises(pil2gl.Pil2glError):
        pil2gl.buildMerkleHash(False).merkelize(a, 1, 8)


def test_tmp_compaction_preserves_program(oracle):
    """host-side live-range renumbering of temporaries (expr.hip) must not change what the program computes"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _random_program
    from conftest import rand_field
    lib = _lib.load()
    for n_ops in (5, 20, 300, 1500):
        rng = np.random.default_rng(n_ops)
        n_bits = 5
        widths = [5, 9, 1, 3]
        secs = [rand_field(rng, (1 << n_bits, w)) for w in widths]
        secs[-1][:] = 0
        scalars = rand_field(rng, 40)
        ops, n_tmp = _random_program(rng, n_ops, widths, scalars.size, len(widths) - 1)
        ref = [s.copy() for s in secs]
        oracle.eval_program(ops, n_tmp, ref, scalars, n_bits, 0)
        prog = oracle.make_program(ops, n_tmp, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
        out = (_lib.GlxOp * (2 * len(ops) + 16))(); info = (C.c_uint32 * 2)()
        assert lib.pil2gl_debug_compact_program(C.byref(prog), out, info) == 0
        n_slots = C.c_uint32(info[0])
        assert n_slots.value <= n_tmp + 64 and (n_ops < 100 or n_slots.value < n_tmp // 2)
        ops2 = []
        for o in list(out)[:info[1]]:
            def ref_(r):
                return (r.kind, r.dim, r.section, r.prime, r.index)
            ops2.append((o.op, ref_(o.dest), ref_(o.src[0]), ref_(o.src[1]) if o.op != 3 else None))
        got = [s.copy() for s in secs]
        oracle.eval_program(ops2, max(1, n_slots.value), got, scalars, n_bits, 0)
        for a, b in zip(got, ref):
            assert (a == b).all(), n_ops


def test_constraint_program_compiles_with_hiprtc_and_fuses_horner():
    """the run-time compiled evaluator path: optimiser + source generator + hiprtc, on the synthetic AIR's cExp (no GPU)"""
    import time
    from pil2gl import stark
    lib = _lib.load()
    ss = {"nBits": 16, "nBitsExt": 19, "nQueries": 8, "steps": [{"nBits": 19}, {"nBits": 14}, {"nBits": 9}]}
    info, exprs, _ = stark.fibonacci_air(10, ss)
    ctx = {"pilInfo": info, "publics": [1, 2, 3], "challenges": [[], [[5, 6, 7]], [[1, 1, 1]], [[2, 2, 2], [3, 3, 3]]], "evals": [[i, i + 1, i + 2] for i in range(len(info["evMap"]))]}
    ops, n_tmp, secs, scalars = stark.encode_code(exprs["expressionsCode"][0]["code"]["code"], "ext", ctx)
    prog = stark.make_c_program(ops, n_tmp)
    widths = {"const_ext": 2, "cm1_ext": 20, "q_ext": 3, "Zi_ext#0": 1}
    cs = (_lib.GlxSection * len(secs))()
    for i, name in enumerate(secs):
        cs[i].ptr = 0; cs[i].width = widths[name]
    c = _lib.GlxCtx(19, 3, len(secs), scalars.size, cs, scalars.ctypes.data_as(_lib.u64p))
    nbytes = C.c_uint64(); fused = C.c_uint32()
    t0 = time.time()
    rc = lib.pil2gl_debug_jit_compile(C.byref(prog), C.byref(c), C.byref(nbytes), C.byref(fused))
    assert rc == 0, lib.pil2gl_last_error()
    assert nbytes.value > 1000
    assert fused.value == 2 * 10 + 3          # every constraint became one lazy multiply-accumulate term
    assert time.time() - t0 < 120
    # the form long programs take (modular multiplications called, not inlined: the kernel must fit the instruction cache)
    inlined = nbytes.value
    os.environ["PIL2GL_EXPR_MULCALL"] = "1"
    try:
        rc = lib.pil2gl_debug_jit_compile(C.byref(prog), C.byref(c), C.byref(nbytes), C.byref(fused))
    finally:
        del os.environ["PIL2GL_EXPR_MULCALL"]
    assert rc == 0, lib.pil2gl_last_error()
    assert 1000 < nbytes.value < inlined


def test_no_early_clobber_overlap_in_bn128_isa(tmp_path):
    """hipcc's coalescer has been seen (round 5) to give an early-clobber asm output the register of an input it is later selected
    against -- `v_cndmask_b32 v8, v8, v8, vcc` in the earlier bn::cond_sub_r when its result was copied back over its input inside a loop: both
    outcomes of the select are then the difference, silently wrong for every value below r.  bn::cond_sub_r works in place since round 6
    (no select is left in it); this scan stays as a backstop for any select whose two sources are the same register, in the ISA of the
    files that carry such asm statements (no GPU needed: hipcc cross-compiles)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    pkg = os.path.join(ROOT, "pil2-stark-js_amd")
    for src in ("bn128.hip",):
        out = tmp_path / (src + ".s")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=off", "-I" + os.path.join(pkg, "build"),
                               "-S", "--cuda-device-only", os.path.join(pkg, "csrc", src), "-o", str(out)], stderr=subprocess.DEVNULL)
        bad = re.findall(r"v_cndmask_b32(?:_e32|_e64)? v\d+, (v\d+), \1, (?:vcc|s\[\d+:\d+\])", out.read_text())     # ANY select whose two sources are one register
        assert not bad, "%s: a select between a register and itself (%d sites)" % (src, len(bad))


def test_bn254_sbox_columns_model_and_generated_file():
    """the S-box's seventeen columns per product (csrc/bn_field29_columns.inc): the generator's integer model of exactly those columns -- every
    column sum below 2^64, result = a b / 2^261 mod r below 2^252 + r -- on random and edge operands, and the checked-in file is what the
    generator writes (build() runs the same check)"""
    import subprocess
    import sys
    subprocess.check_call([sys.executable, os.path.join(ROOT, "pil2-stark-js_amd", "csrc", "gen_bn29_columns.py"), "--check"])

