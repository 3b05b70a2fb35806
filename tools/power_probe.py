#!/usr/bin/env python3
"""Board power and shader clock while one kernel runs: is the BN254 leaf kernel power-limited?  Samples the hwmon files of the card
(power1_average / power1_input in microwatts, freq1_input in Hz, power1_cap) from the host while the launches are in flight; falls back
to `rocm-smi --json` when no hwmon file is readable.
  python tools/power_probe.py [bn|gl|ntt] [log2 rows]"""
import glob
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pil2-stark-js_amd", "python"))
import torch
import pil2gl
from pil2gl import bn128, _lib

what = sys.argv[1] if len(sys.argv) > 1 else "bn"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 22
pil2gl.init(0)


def hwmon_files():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for nm in ("power1_average", "power1_input", "freq1_input", "power1_cap", "temp1_input"):
            f = os.path.join(d, nm)
            if os.path.exists(f):
                out.setdefault(d, {})[nm] = f
    return out


def read(f):
    try:
        return int(open(f).read().strip())
    except Exception:
        return None


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout)
    except Exception as e:
        return {"error": repr(e)}


h = 1 << nb
g = torch.Generator(device="cuda"); g.manual_seed(1)
buf = torch.randint(0, 0x7FFFFFFFFFFFFFFF, (h * 100,), dtype=torch.int64, device="cuda", generator=g) % 0xFFFFFFFF00000001
digests = torch.empty(h * 4, dtype=torch.int64, device="cuda")
if what == "bn":
    bn128.buildMerkleHash(16, False).merkelize(buf[:6400], 100, 64)
    launch = lambda: _lib.call("pil2gl_bn128_linear_hash_rows_dev", buf.data_ptr(), 100, h, 16, 0, digests.data_ptr(), None)
elif what == "gl":
    launch = lambda: _lib.call("pil2gl_linear_hash_rows_dev", buf.data_ptr(), 100, h, 0, digests.data_ptr(), None)
else:
    dst = torch.empty(h * 100, dtype=torch.int64, device="cuda")
    launch = lambda: pil2gl.interpolate(buf, 100, nb, dst, nb)
launch(); torch.cuda.synchronize()
files = hwmon_files()
print("hwmon:", {d: sorted(v) for d, v in files.items()})
idle = {d: {k: read(f) for k, f in v.items()} for d, v in files.items()}
print("idle:", idle)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 6
ev0.record()
for _ in range(reps):
    launch()
ev1.record()
samples = []
smi_mid = None
t0 = time.perf_counter()
while not ev1.query():
    row = {}
    for d, v in files.items():
        for k in ("power1_average", "power1_input", "freq1_input"):
            if k in v:
                row[os.path.basename(d) + ":" + k] = read(v[k])
    samples.append(row)
    if not files and smi_mid is None:
        smi_mid = smi()
    time.sleep(0.01)
torch.cuda.synchronize()
print("kernel %s 2^%d rows: %.2f ms per launch, %d samples in %.2f s" % (what, nb, ev0.elapsed_time(ev1) / reps, len(samples), time.perf_counter() - t0))
keys = sorted({k for r in samples for k in r})
for k in keys:
    vals = [r[k] for r in samples[len(samples) // 4:] if r.get(k)]
    if vals:
        print("%-40s min %.1f avg %.1f max %.1f (%s)" % (k, min(vals) / 1e6, sum(vals) / len(vals) / 1e6, max(vals) / 1e6, "W" if "power" in k else "MHz"))
if smi_mid is not None:
    print("rocm-smi during the run:", json.dumps(smi_mid)[:1500])
