#!/usr/bin/env python3
"""Extracts the BN254 Poseidon parameter sets the reference tree holds (data: public constants) into
tests/golden/poseidon_bn128_constants.json, and copies the reference-written BN128 proof fixture.
  circuits.bn128/custom/poseidon_constants_original.circom : C, M for t = 3, 5, 7, 9, 17
  src/final/poseidon_constants.js                          : C for 4, 7, 8, 16 inputs; M for 4, 6, 8, 16 inputs
To keep the fixture small only a digest of each set plus its first and last entries are stored, together with the
full t=3 set; tests compare the oracle's generated constants with these.
Run here (needs /root/reference and node): python3 oracle/gen_bn128_golden.py"""
import hashlib
import json
import os
import re
import shutil
import subprocess

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def entry(vals, full=False):
    e = {"n": len(vals), "sha256_le32": digest(vals), "first": hex(vals[0]), "last": hex(vals[-1])}
    if full:
        e["all"] = [hex(v) for v in vals]
    return e


def main():
    txt = open(os.path.join(REF, "circuits.bn128/custom/poseidon_constants_original.circom")).read()
    cpart, mpart = txt.split("function POSEIDON_M_ORIGINAL")

    def blocks(part):
        out = {}
        for m in re.finditer(r"t\s*==\s*(\d+)\s*\)\s*\{\s*return\s*\[(.*?)\];", part, re.S):
            out[int(m.group(1))] = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(2))]
        return out
    out = {"circom": {"C": {}, "M": {}}, "final_js": {"C": {}, "M": {}}}
    for t, v in blocks(cpart).items():
        out["circom"]["C"][str(t)] = entry(v, full=(t == 3))
    for t, v in blocks(mpart).items():
        out["circom"]["M"][str(t)] = entry(v, full=(t == 3))
    js = subprocess.check_output(["node", "-e", "const c=require('%s/src/final/poseidon_constants.js');"
                                  "const f=x=>Array.isArray(x)?x.map(f):BigInt(x).toString(16);"
                                  "console.log(JSON.stringify({C:Object.fromEntries(Object.entries(c.C).map(([k,v])=>[k,f(v)])),"
                                  "M:Object.fromEntries(Object.entries(c.M).map(([k,v])=>[k,f(v)]))}))" % REF])
    d = json.loads(js)
    for k, v in d["C"].items():
        out["final_js"]["C"][k] = entry([int(x, 16) for x in v])
    for k, v in d["M"].items():
        out["final_js"]["M"][k] = entry([int(x, 16) for row in v for x in row])
    with open(os.path.join(ROOT, "tests/golden/poseidon_bn128_constants.json"), "w") as f:
        json.dump(out, f, indent=0)
    shutil.copyfile(os.path.join(REF, "test/final/verifier.proof.zkin.json"),
                    os.path.join(ROOT, "tests/golden/ref_final_verifier.proof.zkin.json"))
    os.chmod(os.path.join(ROOT, "tests/golden/ref_final_verifier.proof.zkin.json"), 0o644)
    print("ok")


if __name__ == "__main__":
    main()
