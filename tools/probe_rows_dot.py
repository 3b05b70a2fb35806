"""rows_dot_kernel on the config-3 stage matrix (2^27 x 100).  Used once to compare a line-aligned row mapping (a wave owning
rows 4j + w; switch PIL2GL_ROWS_DOT_LINES) with the plain one: 48.0 ms against 44.7 ms, so that variant was not kept
 and the switch no longer exists; the script now just times the kernel."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pil2-stark-js_amd", "python"))
import numpy as np, torch
import pil2gl
from pil2gl import _lib
pil2gl.init(0)
n_rows, width, n_out = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27, 100, 2
dm = torch.randint(0, 2 ** 62, (n_rows * width,), dtype=torch.int64, device="cuda")
coef = np.random.default_rng(1).integers(0, 2 ** 63, (n_out, width, 3), dtype=np.uint64)
acc = torch.zeros(n_rows * n_out * 3, dtype=torch.int64, device="cuda")
for _ in range(2):
    _lib.call("pil2gl_rows_dot_ext_dev", pil2gl._ptr(dm), width, n_rows, pil2gl._ptr(coef), n_out, pil2gl._ptr(acc), 0, None)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5):
    _lib.call("pil2gl_rows_dot_ext_dev", pil2gl._ptr(dm), width, n_rows, pil2gl._ptr(coef), n_out, pil2gl._ptr(acc), 0, None)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print("rows_dot_kernel<%d>: %.2f ms per call, %.2f TB/s of matrix" % (n_out, dt * 1e3, n_rows * width * 8 / dt / 1e12), "checksum", int(acc.sum()))
