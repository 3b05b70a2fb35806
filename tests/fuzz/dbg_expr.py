import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pil2gl, gl_oracle
from pil2gl import _lib
from conftest import rand_field
from test_gpu_parity import _random_program
pil2gl.init(0)
for n_ops in (1, 2, 3, 5, 8, 12, 20, 50):
  for seed in range(3):
    rng = np.random.default_rng(seed * 100 + n_ops)
    n_bits = 4
    widths = [5, 9, 1, 3]
    secs = [rand_field(rng, (1 << n_bits, w)) for w in widths]; secs[-1][:] = 0
    scalars = rand_field(rng, 40)
    ops, n_tmp = _random_program(rng, n_ops, widths, scalars.size, 3)
    ref = [s.copy() for s in secs]
    gl_oracle.eval_program(ops, n_tmp, ref, scalars, n_bits, 0)
    dsecs = [torch.from_numpy(s.view(np.int64)).cuda() for s in secs]
    prog = gl_oracle.make_program(ops, n_tmp, struct_op=_lib.GlxOp, struct_prog=_lib.GlxProgram)
    csecs = (_lib.GlxSection * 4)()
    for i, s in enumerate(dsecs): csecs[i].ptr = s.data_ptr(); csecs[i].width = widths[i]
    ctx = _lib.GlxCtx(n_bits, 0, 4, scalars.size, csecs, scalars.ctypes.data_as(_lib.u64p))
    _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None)
    torch.cuda.synchronize()
    got = dsecs[3].cpu().numpy().view(np.uint64).reshape(ref[3].shape)
    ok = (got == ref[3]).all()
    print(n_ops, seed, "OK" if ok else "MISMATCH")
    if not ok:
        bad = np.nonzero((got != ref[3]).any(axis=1))[0]
        print("   bad rows", bad.tolist())
        _lib.call("pil2gl_eval_program_dev", C.byref(prog), C.byref(ctx), None); torch.cuda.synchronize()
        got2 = dsecs[3].cpu().numpy().view(np.uint64).reshape(ref[3].shape)
        print("   second run ok:", (got2 == ref[3]).all())
        for o in ops[-3:]: print("   ", o)
        print("   got", [hex(x) for x in got[0]], "ref", [hex(x) for x in ref[3][0]])
        break
