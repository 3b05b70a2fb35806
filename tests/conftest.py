import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "pil2-stark-js_amd", "python"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
P = 0xFFFFFFFF00000001


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the built library / addon are git-ignored: build them once if a fresh checkout has none (hipcc cross-compiles on CPU)
    lib = os.path.join(ROOT, "pil2-stark-js_amd", "lib", "libpil2gl.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "pil2-stark-js_amd"), "-j4", "all"])


def golden(name):
    if name.endswith(".gz"):
        import gzip
        with gzip.open(os.path.join(GOLDEN, name), "rt") as f:
            return json.load(f)
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def H(v):
    """hex-string tree -> python ints"""
    if isinstance(v, list):
        return [H(x) for x in v]
    if isinstance(v, str):
        return int(v, 16)
    return v


def U(v):
    return np.array(H(v), dtype=np.uint64)


def rand_field(rng, shape):
    """uniform canonical Goldilocks elements"""
    a = rng.integers(0, 1 << 63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)
    return np.where(a >= np.uint64(P), a - np.uint64(P), a).astype(np.uint64)


@pytest.fixture(scope="session")
def oracle():
    import gl_oracle
    gl_oracle.build()
    gl_oracle.set_threads(min(8, os.cpu_count() or 1))
    return gl_oracle
