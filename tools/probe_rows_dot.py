"""the FRI polynomial's weighted row sums on the config-3 stage matrix (2^27 x 100, 2 outputs): the matrix-core kernel, the
whole-row streaming kernel on the vector ALU and the column-tile kernel in one process (the switches are read per call)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pil2-stark-js_amd", "python"))
import numpy as np, torch
import pil2gl
from pil2gl import _lib
pil2gl.init(0)
n_rows, width, n_out = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27, int(os.environ.get("NCOLS", 100)), int(os.environ.get("NOUT", 2))
dm = torch.randint(0, 2 ** 62, (n_rows * width,), dtype=torch.int64, device="cuda")
coef = np.random.default_rng(1).integers(0, 2 ** 63, (n_out, width, 3), dtype=np.uint64)
acc = torch.zeros(n_rows * n_out * 3, dtype=torch.int64, device="cuda")
def run():
    _lib.call("pil2gl_rows_dot_ext_dev", pil2gl._ptr(dm), width, n_rows, pil2gl._ptr(coef), n_out, pil2gl._ptr(acc), 0, None)
for rnd in range(2):
    for mode in os.environ.get("MODES", "tile,stream,mfma").split(","):
        os.environ["PIL2GL_ROWS_DOT_MFMA"] = "1" if mode == "mfma" else "0"
        os.environ["PIL2GL_ROWS_DOT_STREAM"] = "0" if mode == "tile" else "1"
        run(); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5):
            run()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print("%-6s: %.2f ms per call, %.2f TB/s of matrix, checksum %d" % (mode, dt * 1e3, n_rows * width * 8 / dt / 1e12, int(acc.sum())), flush=True)
