#!/usr/bin/env python3
"""Writes tests/golden/ref_verify_evals_code.json.gz: the op-list `verifierCode.code` of the reference's own test data file
test/circuits/bn128/verifyEvals.starkInfo.json (3 257 ops: 1 589 mul / 1 335 add / 297 sub / 36 copy over tmp, number, eval,
challenge, x, public operands -- SURVEY.md Appendix B), re-serialised without whitespace and gzipped.  It is DATA produced by
the reference's code generator (src/pil_info/helpers/code/codegen.js), used to pin the op-list encoders and the evaluators
on a real program.  Needs /root/reference; the tests only read the committed fixture.
  python oracle/gen_verify_evals_fixture.py [path/to/reference]"""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = os.path.join(ref, "test", "circuits", "bn128", "verifyEvals.starkInfo.json")
with open(src) as f:
    d = json.load(f)
code = d["verifierCode"]["code"]
out = os.path.join(ROOT, "tests", "golden", "ref_verify_evals_code.json.gz")
with gzip.GzipFile(out, "wb", mtime=0) as f:
    f.write(json.dumps({"source": "test/circuits/bn128/verifyEvals.starkInfo.json :: verifierCode.code", "code": code}, separators=(",", ":")).encode())
print(out, len(code), "ops", os.path.getsize(out), "bytes")
