"""the config-3 leaf hashing alone (2^27 rows x 100 columns -> digests), twice, plus a plain streaming read of the same matrix
(torch.sum: a known byte count in a 16 B/lane pattern) for calibrating rocprofv3's FETCH_SIZE against this kernel's
8 B/lane row-strided loads"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))
import pil2gl
pil2gl.init(0)
nBitsExt = int(os.environ.get("NBITSEXT", 27)); C = int(os.environ.get("NCOLS", 100))
E = 1 << nBitsExt
dst = torch.empty(E * C, dtype=torch.int64, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
for o in range(0, E * C, 1 << 28):
    m = min(1 << 28, E * C - o)
    dst[o:o + m] = torch.randint(0, 2**62, (m,), dtype=torch.int64, device="cuda", generator=g)
dig = torch.empty(E * 4, dtype=torch.int64, device="cuda")
for _ in range(2):
    pil2gl.linearHash(dst, C, False, dig)
if os.environ.get("SPLIT", "0") != "0":      # the splitLinearHash form of the same leaves (linear_hash_split_kernel)
    for _ in range(2):
        pil2gl.linearHash(dst, C, True, dig)
torch.cuda.synchronize()
s = 0
for o in range(0, E * C, 1 << 30):          # 8 GiB pieces: reduce kernels reading each byte once
    s += int(dst[o:o + (1 << 30)].sum().item())
lvl = torch.empty(E * 2, dtype=torch.int64, device="cuda")
pil2gl.merkelizeLevel(dig, lvl)
torch.cuda.synchronize()
print("ok", hex(s & 0xFFFFFFFFFFFFFFFF), [hex(int(v) & 0xFFFFFFFFFFFFFFFF) for v in dig[:4].cpu()])
