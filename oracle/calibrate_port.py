"""The oracle's C port on the inputs of oracle/calibrate_ref.js, one thread: same check words, and the time ratio to the
reference's JS twins.  Test/measurement infrastructure, build container only:
    python oracle/calibrate_port.py   -> one JSON line (runs node oracle/calibrate_ref.js itself)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import gl_oracle as orc  # noqa: E402

P = 0xFFFFFFFF00000001
n_bits, n_perm = 14, 4000
ref = json.loads(subprocess.run(["node", os.path.join(os.path.dirname(__file__), "calibrate_ref.js"), str(n_bits), str(n_perm)],
                                capture_output=True, text=True, check=True).stdout)
orc.build(); orc.set_threads(1)
col = np.array([i * 0x9E3779B97F4A7C15 % P for i in range(1 << n_bits)], dtype=np.uint64)
orc.extend_pol(col[:16], 3)
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    ext = orc.extend_pol(col, 3)
t_ext = (time.perf_counter() - t0) / reps
assert "%x" % int(ext[5]) == ref["extendPol_check"]
st = list(range(1, 9)); cap = [0, 0, 0, 0]
t0 = time.perf_counter()
for _ in range(n_perm):
    cap = [int(v) for v in orc.poseidon(st, cap)]
t_pos = time.perf_counter() - t0          # dominated by the ctypes call overhead: timed again in bulk below
rows = np.tile(np.arange(1, 9, dtype=np.uint64), (1 << 16, 1))
t0 = time.perf_counter()
orc.merkelize(rows, False)               # 2^16 leaf permutations (width 8) + 2^16 - 1 node permutations
t_bulk = time.perf_counter() - t0
n_bulk = (1 << 17) - 1
assert "%x" % cap[0] == ref["poseidon_check"]
out = {"machine": "build container, 1 thread", "nBits": n_bits,
       "extendPol": {"reference_js_s": ref["extendPol_s"], "port_c_s": t_ext, "ratio": ref["extendPol_s"] / t_ext},
       "poseidon": {"reference_js_perm_per_s": n_perm / ref["poseidon_s"], "port_c_perm_per_s_via_ctypes": n_perm / t_pos,
                    "port_c_perm_per_s_bulk": n_bulk / t_bulk}}
out["poseidon"]["ratio_bulk"] = out["poseidon"]["port_c_perm_per_s_bulk"] / out["poseidon"]["reference_js_perm_per_s"]
print(json.dumps(out))
