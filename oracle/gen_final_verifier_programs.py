#!/usr/bin/env python3
"""Build container only.  The BN128 twin of gen_compressor_verifier_programs.py: reads the verifier CIRCUIT the reference holds for
its `final` test (/root/reference/test/final/verifier.circom, the output of the older pil-stark generator) and writes, as data, the two
programs a STARK verifier evaluates for it -- the constraint sum at the evaluation point (template VerifyEvaluations, :290-2920) and
the FRI polynomial at a query point (template VerifyQuery, :2921-3189) -- in the op-list shape stark_verify.js:222-298 executes,
together with the starkInfo fields a verifier reads, all taken from the same file.

Why: test/final/verifier.proof.zkin.json is the ONLY proof in the reference tree written on the BN128 hash family (Poseidon-BN254
trees of arity 4, BN128 transcript).  With these programs pil2gl.stark.stark_verify checks it END TO END (tests/test_reference_proof.py):
transcript, evaluation identity, every Merkle path, the FRI polynomial at the 32 query points, the folds -- and a 2 600-op program a
real circuit produced runs through the device evaluator.

The translation is statement by statement, one circom signal definition = one op (`p` is the Goldilocks prime the circuit adds to keep
differences non-negative inside the BN254 field; over Goldilocks `a - b + p` is `a - b`):
    signal tmp_N[3] <== GLCMul()(A, B);                                       -> mul    tmp_N, A, B
    signal tmp_N[3] <== GLCMulAdd()(A, B, C);                                 -> muladd tmp_N, A, B, C     (stark_verify.js:234)
    signal tmp_N[3] <== [A[0] + B[0], A[1] + B[1], A[2] + B[2]];              -> add    tmp_N, A, B
    signal tmp_N[3] <== [A[0] - B[0] + p, A[1] - B[1] + p, A[2] - B[2] + p];  -> sub    tmp_N, A, B
    signal tmp_N[3] <== [A[0] - k + p, A[1], A[2]];                           -> sub    tmp_N, A, k        (k: number or publics[i])
    signal tmp_N[3] <== [k - A[0] + p, -A[1] + p, -A[2] + p];                 -> sub    tmp_N, k, A        (k: number, consts[i], base column)
    signal tmp_N[3] <== evals[i];                                             -> copy   tmp_N, eval i
Operands: `[c, 0, 0]` with a literal c -> number c; `[mapValues.treeS_j,0,0]` / `[consts[i],0,0]` -> that base-field value;
evals[k] -> eval k; challengesN -> the flat challenge N of the old layout (Transcript() :18-246: 0,1 after root1; 2,3 after root2; 4
after root3; 7 after rootQ; 5,6 after the evaluations), renamed to the staged layout stark_verify.js reads: 0,1 = stage 2; 2,3 = stage
3; 4 = stage 4 (quotient); 7 = stage 5 (xi); 5,6 = stage 6 (FRI); mapValues.tree1_j / tree3_j / treeQ_j -> polynomial j of the stage-1 /
stage-3 / stage-4 opening at (treePos, dim) read off MapValues (:3190-3264); xDivXSubXi[i].
Temporaries are renumbered in order of definition.  The older circuit states the identity as  C(z) == Q(z) * Z(z)  with Z = z^N - 1
(:2895-2917); stark_verify.js compares  C(z) * Zi  with  Q(z)  where Zi = 1 / (z^N - 1) (:95-152): ONE op is appended to the
transliterated program for that (`mul last, Zi`), and the test also checks the circuit's own form on host integers.
No reference source text is copied: the output is op-lists and numbers.

    python oracle/gen_final_verifier_programs.py        # writes tests/golden/ref_final_verifier_programs.json.gz
"""
import gzip
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PIL2_REFERENCE", "/root/reference")
from gen_compressor_verifier_programs import template_body, split_top

STAGE_OF_TREE = {"1": 1, "2": 2, "3": 3, "Q": 4}
# flat challenge index of the old layout -> (stage, stageId) of the staged layout
CHALLENGE = {0: (2, 0), 1: (2, 1), 2: (3, 0), 3: (3, 1), 4: (4, 0), 7: (5, 0), 5: (6, 0), 6: (6, 1)}


class Translator:
    def __init__(self, layout):
        self.layout = layout                    # stage -> list of (treePos, dim) per polynomial index
        self.tmp, self.code = {}, []

    def tree(self, s, j, want_dim):
        st = STAGE_OF_TREE[s]
        pos, dim = self.layout[st][int(j)]
        assert dim == want_dim, (s, j, dim, want_dim)
        return {"type": "tree%d" % st, "treePos": pos, "dim": dim}

    def base(self, name):
        name = name.strip()
        if re.fullmatch(r"0x[0-9a-fA-F]+|\d+", name):
            return {"type": "number", "value": str(int(name, 0)), "dim": 1}
        m = re.fullmatch(r"publics\[(\d+)\]", name)
        if m:
            return {"type": "public", "id": int(m.group(1)), "dim": 1}
        m = re.fullmatch(r"consts\[(\d+)\]", name)
        if m:
            return {"type": "const", "id": int(m.group(1)), "dim": 1}
        m = re.fullmatch(r"mapValues\.tree(\w)_(\d+)", name)
        if m:
            return self.tree(m.group(1), m.group(2), 1)
        raise ValueError("unknown base operand " + name)

    def ext(self, name):
        """an operand of a GLCMul / GLCMulAdd call, or the name in front of a component index"""
        name = name.strip()
        m = re.fullmatch(r"\[\s*([^,\]]+?)\s*,\s*0\s*,\s*0\s*\]", name)
        if m:
            return self.base(m.group(1))
        m = re.fullmatch(r"evals\[(\d+)\]", name)
        if m:
            return {"type": "eval", "id": int(m.group(1)), "dim": 3}
        m = re.fullmatch(r"challenges(\d+)", name)
        if m:
            st, sid = CHALLENGE[int(m.group(1))]
            return {"type": "challenge", "stage": st, "stageId": sid, "dim": 3}
        m = re.fullmatch(r"xDivXSubXi\[(\d+)\]", name)
        if m:
            return {"type": "xDivXSubXi", "id": int(m.group(1)), "dim": 3}
        m = re.fullmatch(r"mapValues\.tree(\w)_(\d+)", name)
        if m:
            return self.tree(m.group(1), m.group(2), 3)
        if name in self.tmp:
            return {"type": "tmp", "id": self.tmp[name], "dim": 3}
        raise ValueError("unknown extension operand " + name)

    def emit(self, op, dest, srcs):
        tid = len(self.tmp)
        self.tmp[dest] = tid
        self.code.append({"op": op, "dest": {"type": "tmp", "id": tid, "dim": 3}, "src": srcs})

    def triple(self, dest, comps):
        c = [re.sub(r"\s+", " ", x.strip()) for x in comps]
        # A[0] + B[0], ...
        m = re.fullmatch(r"(.+)\[0\] \+ (.+)\[0\]", c[0])
        if m:
            A, B = m.group(1), m.group(2)
            assert c[1] == "%s[1] + %s[1]" % (A, B) and c[2] == "%s[2] + %s[2]" % (A, B), c
            return self.emit("add", dest, [self.ext(A), self.ext(B)])
        m = re.fullmatch(r"(.+)\[0\] - (.+)\[0\] \+ p", c[0])
        if m and c[1].endswith("+ p") and "[1] - " in c[1]:
            A, B = m.group(1), m.group(2)
            assert c[1] == "%s[1] - %s[1] + p" % (A, B) and c[2] == "%s[2] - %s[2] + p" % (A, B), c
            return self.emit("sub", dest, [self.ext(A), self.ext(B)])
        m = re.fullmatch(r"(.+)\[0\] - (.+) \+ p", c[0])
        if m and c[1] == m.group(1) + "[1]":                      # extension - base: only component 0 meets it
            A, k = m.group(1), m.group(2)
            assert c[2] == A + "[2]", c
            return self.emit("sub", dest, [self.ext(A), self.base(k)])
        m = re.fullmatch(r"(.+) - (.+)\[0\] \+ p", c[0])
        if m and c[1].startswith("-"):                            # base - extension: the upper components are negated
            k, A = m.group(1), m.group(2)
            assert c[1] == "-%s[1] + p" % A and c[2] == "-%s[2] + p" % A, c
            return self.emit("sub", dest, [self.base(k), self.ext(A)])
        raise ValueError("cannot translate " + repr(c))

    def run(self, body):
        for line in body.split("\n"):
            line = line.strip()
            m = re.fullmatch(r"signal (tmp_\d+)\[3\] <== GLCMul\(\)\((.+)\);", line)
            if m:
                a, b = split_top(m.group(2))
                self.emit("mul", m.group(1), [self.ext(a), self.ext(b)]); continue
            m = re.fullmatch(r"signal (tmp_\d+)\[3\] <== GLCMulAdd\(\)\((.+)\);", line)
            if m:
                a, b, c = split_top(m.group(2))
                self.emit("muladd", m.group(1), [self.ext(a), self.ext(b), self.ext(c)]); continue
            m = re.fullmatch(r"signal (tmp_\d+)\[3\] <== \[(.+)\];", line)
            if m:
                self.triple(m.group(1), split_top(m.group(2))); continue
            m = re.fullmatch(r"signal (tmp_\d+)\[3\] <== ([A-Za-z_][\w\.\[\]]*);", line)
            if m:
                self.emit("copy", m.group(1), [self.ext(m.group(2))]); continue
            assert not line.startswith("signal tmp_"), "untranslated statement: " + line
        return {"tmpUsed": len(self.tmp), "code": self.code}


def main():
    src = open(os.path.join(REF, "test", "final", "verifier.circom")).read()
    sv = template_body(src, "StarkVerifier")
    ve = template_body(src, "VerifyEvaluations")
    vq = template_body(src, "VerifyQuery")
    tr = template_body(src, "Transcript")
    n_queries, n_bits_ext = (int(v) for v in re.search(r"signal output ys\[(\d+)\]\[(\d+)\]", tr).groups())
    n_bits = int(re.search(r"signal zMul\[(\d+)\]\[3\]", ve).group(1))
    n_evals = int(re.search(r"signal input evals\[(\d+)\]\[3\]", ve).group(1))
    n_publics = int(re.search(r"signal input publics\[(\d+)\]", ve).group(1))
    widths = {}
    for s, w in re.findall(r"signal input s0_vals(\w)\[\d+\]\[(\d+)\]", sv):
        widths["C" if s == "C" else STAGE_OF_TREE[s]] = int(w)
    widths.setdefault(2, 0)                                       # no plookup in this circuit: stage 2 commits nothing (root2 is absorbed all the same)
    n_stages = 3
    root_c = int(re.search(r"signal rootC <== (\d+);", sv).group(1))
    arity = int(re.search(r"VerifyMerkleHash\(1, \d+, \d+, (\d+)\)", sv).group(1))
    steps = [int(re.search(r"VerifyQuery\((\d+), (\d+)\)", sv).group(1))] + [int(v) for v in re.findall(r"VerifyFRI\(\d+, (\d+), \d+, \d+, \d+\)", sv)]
    assert steps[0] == n_bits_ext
    # layout of the opened rows (MapValues): polynomial j of the tree at (treePos, dim)
    mv = template_body(src, "MapValues")
    layout = {}
    for s, j, rhs in re.findall(r"tree(\w)_(\d+) <== (.+);", mv):
        idx = [int(v) for v in re.findall(r"vals\w\[(\d+)\]", rhs)]
        assert idx == list(range(idx[0], idx[0] + len(idx))) and len(idx) in (1, 3)
        layout.setdefault(STAGE_OF_TREE[s], {})[int(j)] = (idx[0], len(idx))
    layout = {s: [layout[s][j] for j in range(len(layout[s]))] for s in layout}
    for s in layout:
        assert sum(d for _, d in layout[s]) == widths[s], s
    # 1. the constraint sum at the evaluation point
    t1 = Translator(layout)
    q_code = t1.run(ve)
    last = re.search(r"normC\[3\] <== GLCNorm\(\)\(\[(tmp_\d+)\[0\] - QZ\[0\]", ve).group(1)
    assert t1.tmp[last] == len(t1.code) - 1, "the constraint sum must be the last op"
    n_circuit_ops = len(q_code["code"])
    zi_tmp = len(t1.tmp)
    q_code["code"].append({"op": "mul", "dest": {"type": "tmp", "id": zi_tmp, "dim": 3},
                           "src": [{"type": "tmp", "id": zi_tmp - 1, "dim": 3}, {"type": "Zi", "boundaryId": 0, "dim": 3}]})
    q_code["tmpUsed"] = zi_tmp + 1
    q_deg = int(re.search(r"signal qAcc\[(\d+)\]\[3\]", ve).group(1))
    q_first = int(re.search(r"qAcc\[0\] <== evals\[(\d+)\+i\]", ve).group(1))
    assert re.search(r"signal Z\[3\] <== \[zMul\[%d\]\[0\] -1 \+ p" % (n_bits - 1), ve)
    # 2. the FRI polynomial at a query point
    t2 = Translator(layout)
    f_code = t2.run(vq)
    out_tmp = re.search(r"queryVals\[3\] <== GLCNorm\(\)\((tmp_\d+)\);", vq).group(1)
    assert t2.tmp[out_tmp] == len(t2.code) - 1
    openings = [0, 1]                                              # den0inv: X - xi ; den1inv: X - roots(nBits) xi
    assert re.search(r"roots1\[0\] <== GLCMul\(\)\(\[roots\(%d\), 0, 0\], challenges7\)" % n_bits, vq)
    # evMap from the FRI polynomial: every "(value - evals[k])" term sits under the xDivXSubXi of its opening
    cm_pols, pol_of = [], {}
    for s in sorted(layout):
        for j, (pos, dim) in enumerate(layout[s]):
            pol_of[(s, pos)] = len(cm_pols)
            cm_pols.append({"stage": s, "name": "cm%d_%d" % (s, j), "dim": dim, "stagePos": pos, "stageId": j})
    ev_map, pending = [None] * n_evals, []
    for c in f_code["code"]:
        if c["op"] == "sub" and c["src"][1]["type"] == "eval":
            pending.append((c["src"][1]["id"], c["src"][0]))
        if c["op"] == "mul" and c["src"][1]["type"] == "xDivXSubXi":
            for ev_id, pol in pending:
                prime = openings[c["src"][1]["id"]]
                if pol["type"] == "const":
                    ev_map[ev_id] = {"type": "const", "id": pol["id"], "prime": prime}
                else:
                    ev_map[ev_id] = {"type": "cm", "id": pol_of[(int(pol["type"][4:]), pol["treePos"])], "prime": prime}
            pending = []
    assert not pending and all(e is not None for e in ev_map)
    q_stage = n_stages + 1
    for i in range(q_deg):
        e = ev_map[q_first + i]
        assert e["type"] == "cm" and cm_pols[e["id"]]["stage"] == q_stage and cm_pols[e["id"]]["stageId"] == i
    challenges_map = [{"name": "u", "stage": 2, "dim": 3, "stageId": 0}, {"name": "defVal", "stage": 2, "dim": 3, "stageId": 1},
                      {"name": "gamma", "stage": 3, "dim": 3, "stageId": 0}, {"name": "beta", "stage": 3, "dim": 3, "stageId": 1},
                      {"name": "vc", "stage": 4, "dim": 3, "stageId": 0}, {"name": "xi", "stage": 5, "dim": 3, "stageId": 0},
                      {"name": "vf1", "stage": 6, "dim": 3, "stageId": 0}, {"name": "vf2", "stage": 6, "dim": 3, "stageId": 1}]
    info = {
        "starkStruct": {"nBits": n_bits, "nBitsExt": n_bits_ext, "nQueries": n_queries, "verificationHashType": "BN128",
                        "merkleTreeArity": arity, "merkleTreeCustom": False, "steps": [{"nBits": b} for b in steps]},
        "nStages": n_stages, "nConstants": widths["C"], "nPublics": n_publics, "nSubproofValues": 0, "qDeg": q_deg, "qDim": 3,
        "openingPoints": openings, "boundaries": [{"name": "everyRow"}],
        "mapSectionsN": dict({"const": widths["C"]}, **{"cm%d" % s: widths[s] for s in widths if s != "C"}),
        "cmPolsMap": cm_pols, "challengesMap": challenges_map, "evMap": ev_map,
    }
    out = {"source": "derived from test/final/verifier.circom of the reference by oracle/gen_final_verifier_programs.py",
           "constRoot": str(root_c), "starkInfo": info, "circuitOps": {"qVerifier": n_circuit_ops, "queryVerifier": len(f_code["code"])},
           "verifierInfo": {"qVerifier": q_code, "queryVerifier": f_code}}
    path = os.path.join(ROOT, "tests", "golden", "ref_final_verifier_programs.json.gz")
    with open(path, "wb") as raw:                                  # mtime 0 and no file name in the header: the same bytes every run
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
            f.write(json.dumps(out, separators=(",", ":")).encode())
    ops = {}
    for c in q_code["code"] + f_code["code"]:
        ops[c["op"]] = ops.get(c["op"], 0) + 1
    print("wrote %s: %d (+1) + %d ops %s, %d evaluations, nBits %d/%d, steps %s, %d queries, arity %d, widths %s" % (
        os.path.relpath(path, ROOT), n_circuit_ops, len(f_code["code"]), ops, n_evals, n_bits, n_bits_ext, steps, n_queries, arity, widths))


if __name__ == "__main__":
    main()
