"""host only: compile the config-3 bench AIR's constraint program with the run-time code generator (hiprtc, no GPU) and, with
PIL2GL_EXPR_DUMP=path, leave the generated kernel's source there"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python")); sys.path.insert(0, ROOT)
from pil2gl import stark, _lib
import bench
lib = _lib.load()
ss = {"nBits": 24, "nBitsExt": 27, "nQueries": 64, "verificationHashType": "GL", "splitLinearHash": False, "steps": [{"nBits": b} for b in bench.fri_steps_for(27)]}
info, exprs, _ = stark.fibonacci_air(50, ss)
ctx = {"pilInfo": info, "publics": [1, 2, 3], "challenges": [[], [[5, 6, 7]], [[1, 1, 1]], [[2, 2, 2], [3, 3, 3]]], "evals": [[i, i + 1, i + 2] for i in range(len(info["evMap"]))]}
ops, n_tmp, secs, scalars = stark.encode_code(exprs["expressionsCode"][0]["code"]["code"], "ext", ctx)
prog = stark.make_c_program(ops, n_tmp)
widths = {"const_ext": 2, "cm1_ext": 100, "q_ext": 3, "Zi_ext#0": 1}
cs = (_lib.GlxSection * len(secs))()
for i, name in enumerate(secs):
    cs[i].ptr = 0; cs[i].width = widths[name]
c = _lib.GlxCtx(27, 3, len(secs), scalars.size, cs, scalars.ctypes.data_as(_lib.u64p))
nbytes = C.c_uint64(); fused = C.c_uint32()
rc = lib.pil2gl_debug_jit_compile(C.byref(prog), C.byref(c), C.byref(nbytes), C.byref(fused))
print("rc", rc, lib.pil2gl_last_error() if rc else "", "code object bytes", nbytes.value, "fused multiply-accumulates", fused.value)
