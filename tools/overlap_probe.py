#!/usr/bin/env python3
"""Does the LDE (LDS / HBM / ALU mix) overlap with leaf hashing (pure ALU) when both run on separate streams?
Times interpolate(2^24 x 100 -> 2^27) and linearHash(2^27 x 100) back to back and concurrently."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pil2-stark-js_amd", "python"))
import torch
import pil2gl
pil2gl.init(0)
nb, C, eb = int(sys.argv[1]) if len(sys.argv) > 1 else 23, 100, 3
N, E = 1 << nb, 1 << (nb + eb)
dev = torch.device("cuda", 0)
src = torch.randint(0, 0x7FFFFFFFFFFFFFFF, (N * C,), dtype=torch.int64, device=dev) % 0xFFFFFFFF00000001
d1 = torch.empty(E * C, dtype=torch.int64, device=dev)
d2 = torch.randint(0, 0x7FFFFFFFFFFFFFFF, (E * C,), dtype=torch.int64, device=dev)
dig = torch.empty(E * 4, dtype=torch.int64, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def lde():
    with torch.cuda.stream(sA): pil2gl.interpolate(src, C, nb, d1, nb + eb)
def hsh():
    with torch.cuda.stream(sB): pil2gl.linearHash(d2, C, False, dig)
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t
lde(); hsh(); torch.cuda.synchronize()
a = timed(lde); b = timed(hsh); c = timed(lambda: (hsh(), lde())); d = timed(lambda: (lde(), hsh()))
print({"lde_s": round(a, 4), "hash_s": round(b, 4), "sum": round(a + b, 4), "concurrent_hash_first": round(c, 4), "concurrent_lde_first": round(d, 4)})
