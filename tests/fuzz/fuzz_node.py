#!/usr/bin/env python3
"""Random proofs (one witness stage, or two with hints in the reference's shape) driven from Node through the JS drop-in modules (tests/js/prove_flow.js: the reference's function boundaries, host
buffers and device-resident DevBuffers) against the proof the CPU checker backend writes for the same AIR and witness: random trace
size, blow-up, machine count, FRI steps, query count, hashCommits, previous-row opening.  Test infrastructure (imports oracle/).
  gpurun -- python tests/fuzz/fuzz_node.py [cases] [seed]"""
import json
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "pil2-stark-js_amd", "python")]
import gl_oracle
gl_oracle.build(); gl_oracle.set_threads(8)
from pil2gl import stark
from stark_backend import OracleBackend

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def strs(v):
    if isinstance(v, dict):
        return {k: strs(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [strs(x) for x in v]
    if isinstance(v, int) and not isinstance(v, bool):
        return str(v)
    return v


tmp = tempfile.mkdtemp(prefix="fuzz_node_")
names = []
for i in range(CASES):
    nb = rnd.randint(3, 11); eb = rnd.randint(1, 3); nbe = nb + eb
    steps = [nbe]
    while steps[-1] > 3 and len(steps) < 5:
        nxt = steps[-1] - rnd.randint(1, 5)
        if nxt < 1:
            break
        steps.append(nxt)
    pairs = rnd.randint(1, 6); prev = rnd.random() < 0.3; im = rnd.random() < 0.3; bd = rnd.random() < 0.3
    ss = {"nBits": nb, "nBitsExt": nbe, "nQueries": rnd.randint(1, 16), "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]}
    if rnd.random() < 0.3:
        ss["hashCommits"] = True
    perm = rnd.random() < 0.25
    if perm:                                              # two witness stages, gprod hints in the reference's shape (expression fields)
        info, exprs, vinfo = stark.permutation_air(ss, min(pairs, 3), ref_hints=True)
        cm, consts, publics = stark.permutation_trace(nb, copies=min(pairs, 3))
    else:
        info, exprs, vinfo = stark.fibonacci_air(pairs, ss, prev, im_pols=im, boundaries=bd)
        cm, consts, publics = stark.fibonacci_trace(nb, pairs, im_pols=im)
    be = OracleBackend()
    setup = stark.build_const_tree(be, consts, info)
    res = stark.stark_gen(be, be.from_host(cm), setup, info, exprs, publics)
    out = {"pilInfo": info, "expressionsInfo": exprs, "cm1": [str(int(v)) for v in cm.reshape(-1)], "consts": [str(int(v)) for v in consts.reshape(-1)],
           "publics": [str(v) for v in publics], "constRoot": [str(v) for v in setup["constRoot"]],
           "proof": strs(json.loads(json.dumps(res["proof"], default=int))), "challenges": strs(res["challenges"]), "queries": res["queries"],
           "verifierInfo": {"qVerifier": vinfo["qVerifier"], "queryVerifier": stark.query_verifier_of(info, exprs)}}
    name = os.path.join(tmp, "g%03d.json" % i)
    json.dump(out, open(name, "w")); names.append(name)
    print("case %d: nBits %d ext %d pairs %d steps %s queries %d hashCommits %s prevRow %s imPols %s boundaries %s twoStage %s" % (i, nb, eb, pairs, steps, ss["nQueries"], bool(ss.get("hashCommits")), prev, im, bd, perm), flush=True)
js = """
const fs = require("fs");
const { prove, freeCtx } = require(%r);
const starkVerify = require(%r);
const bigP = (p) => (Array.isArray(p) ? p.map(bigP) : (p && typeof p === "object" ? Object.fromEntries(Object.entries(p).map(([k, v]) => [k, bigP(v)])) : (typeof p === "string" && /^[0-9]+$/.test(p) ? BigInt(p) : p)));
(async () => {
  let n = 0;
  for (const f of %s) {
    const g = JSON.parse(fs.readFileSync(f));
    await prove(g, false);
    const r = await prove(g, true); freeCtx(r.ctx);
    if (!(await starkVerify(r.proof, g.publics.map(BigInt), bigP(g.constRoot), undefined, g.pilInfo, g.verifierInfo))) throw new Error("the verifier drop-in rejects the proof of " + f);
    n++;
  }
  console.log("node fuzz: " + n + " proofs (host buffers and device-resident) identical to the checker's, each accepted by the verifier drop-in");
})().catch((e) => { console.error(e); process.exit(1); });
""" % (os.path.join(ROOT, "tests", "js", "prove_flow.js"), os.path.join(ROOT, "pil2-stark-js_amd", "js", "stark_verify.js"), json.dumps(names))
r = subprocess.run(["node", "-e", js], capture_output=True, text=True, timeout=1500)
print(r.stdout[-2000:], r.stderr[-3000:])
sys.exit(r.returncode)
