"""computeQStark's transforms at config 3 (2^24 -> 2^27, qDim 3, qDeg 2), piece by piece: the padded form (q_split + fft of 2^27 x 6) against the
coefficient form (q_split_brev + extend_coefs_brev), and narrow interpolates with 7- and 8-stage forward passes.  gpurun -- python tools/probe_q_ntt.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import torch
import pil2gl
from pil2gl import _lib
pil2gl.init(0)
nb, nbe, qDim, qDeg = int(os.environ.get("NB", 24)), int(os.environ.get("NB", 24)) + 3, 3, 2
W = qDim * qDeg
E, N = 1 << nbe, 1 << nb
q = torch.randint(0, 2**62, (E * qDim,), dtype=torch.int64, device="cuda")
qq1 = torch.empty_like(q); qq2 = torch.empty(E * W, dtype=torch.int64, device="cuda"); out = torch.empty(E * W, dtype=torch.int64, device="cuda"); c = torch.empty(N * W, dtype=torch.int64, device="cuda")
p = pil2gl._ptr
def t(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    print("%-60s %7.2f ms" % (name, s.elapsed_time(e) / n), flush=True)
t("ifft 2^%d x %d" % (nbe, qDim), lambda: _lib.call("pil2gl_ifft_dev", p(q), qDim, nbe, p(qq1), None))
t("q_split (padded, 2^%d x %d)" % (nbe, W), lambda: _lib.call("pil2gl_compute_q_split_dev", p(qq1), nb, nbe, qDim, qDeg, p(qq2), None))
t("fft 2^%d x %d" % (nbe, W), lambda: _lib.call("pil2gl_fft_dev", p(qq2), W, nbe, p(out), None))
ref = out.clone()
t("q_split_brev (2^%d x %d)" % (nb, W), lambda: _lib.call("pil2gl_compute_q_split_brev_dev", p(qq1), nb, nbe, qDim, qDeg, p(c), None))
for wide in ("1", "0"):
    os.environ["PIL2GL_LDE_WIDEFWD"] = wide
    t("extend_coefs_brev 2^%d -> 2^%d x %d, WIDEFWD=%s" % (nb, nbe, W, wide), lambda: _lib.call("pil2gl_extend_coefs_brev_dev", p(c), W, nb, p(out), nbe, None))
    print("   same values:", bool((out == ref).all()))
for cols in (2, 6, 8, 12):
    src = torch.randint(0, 2**62, (N * cols,), dtype=torch.int64, device="cuda"); dst = torch.empty(E * cols, dtype=torch.int64, device="cuda")
    for wide in ("1", "0"):
        os.environ["PIL2GL_LDE_WIDEFWD"] = wide
        t("interpolate 2^%d x %d -> 2^%d, WIDEFWD=%s" % (nb, cols, nbe, wide), lambda: pil2gl.interpolate(src, cols, nb, dst, nbe))
