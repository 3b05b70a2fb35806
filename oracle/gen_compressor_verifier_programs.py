#!/usr/bin/env python3
"""Build container only.  Reads the verifier CIRCUIT the reference generated for its compressor test
(/root/reference/test/compressor/verifier.circom) and writes, as data, the two programs a STARK verifier evaluates for that
circuit -- the constraint identity at the evaluation point (template VerifyEvaluations0, :277-500) and the FRI polynomial at
a query point (template CalculateFRIPolValue0, :501-679) -- in the op-list shape stark_verify.js:222-298 executes
(`verifierInfo.qVerifier.code` / `verifierInfo.queryVerifier.code`), together with the few starkInfo fields the verifier
reads (sizes, challenge counts per stage, evMap, where the quotient pieces sit), all of them read off the same file.

Why: the reference tree holds a proof its own prover wrote for this circuit (test/compressor/verifier.proof.zkin.json, a test
fixture) but not the starkInfo / verifierInfo JSON that goes with it.  With these programs pil2gl.stark.stark_verify checks
that proof END TO END -- transcript, evaluation identity, every Merkle path, the FRI polynomial at the query points, the folds
(tests/test_reference_proof.py) -- which pins the expression evaluator and the verifier batch kernels
by reference output.

The translation is mechanical, statement by statement (one circom signal definition = one op):
    signal tmp_N[3] <== CMul()(A, B);                          -> mul  tmp_N, A, B
    signal tmp_N[3] <== [A[0] + B[0], A[1] + B[1], A[2] + B[2]]  -> add  tmp_N, A, B           (likewise -)
    signal tmp_N[3] <== [1 - A[0], -A[1], -A[2]]               -> sub  tmp_N, number 1, A     (base (-) extension)
    signal tmp_N[3] <== [A[0] - publics[k], A[1], A[2]]        -> sub  tmp_N, A, public k     (extension (-) base)
    signal tmp_N[3] <== [A[0] * c, A[1] * c, A[2] * c]         -> mul  tmp_N, A, number c
    signal tmp_N[3] <== A;                                     -> copy tmp_N, A
Operands: evals[k] -> eval k; challengesStageS[i] -> challenge (S, i); challengeQ / challengeXi / challengesFRI[i] -> the
challenges of stages nStages+1, +2, +3; publics[k]; consts[k] -> const k; mapValues.treeS_j -> the j-th polynomial of the
stage-S opening (type "treeS", treePos, dim: stark_verify.js:245-246); xDivXSubXi[i]; Zh -> Zi of the every-row boundary.
Temporaries are renumbered in order of definition.  No reference source text is copied: the output is op-lists and numbers.

    python oracle/gen_compressor_verifier_programs.py        # writes tests/golden/ref_compressor_verifier_programs.json
"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PIL2_REFERENCE", "/root/reference")


def template_body(text, name):
    m = re.search(r"template (?:parallel )?%s\(.*?\) \{" % re.escape(name), text)
    assert m, "template %s not found" % name
    depth, i = 1, m.end()
    while depth:
        c = text[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return text[m.end():i - 1]


def split_top(s):
    out, depth, cur = [], 0, ""
    for c in s:
        if c == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            depth += (c in "[(") - (c in "])")
            cur += c
    out.append(cur.strip())
    return out


class Translator:
    def __init__(self, n_stages, tree_layout):
        self.n_stages = n_stages
        self.tree_layout = tree_layout          # stage -> list of (treePos, dim) per polynomial index
        self.tmp = {}                           # circom name -> tmp id
        self.code = []

    # ---- operands
    def ext(self, name):
        """an extension-field operand by its circom name (without the component index)"""
        name = name.strip()
        m = re.fullmatch(r"evals\[(\d+)\]", name)
        if m:
            return {"type": "eval", "id": int(m.group(1)), "dim": 3}
        m = re.fullmatch(r"challengesStage(\d+)\[(\d+)\]", name)
        if m:
            return {"type": "challenge", "stage": int(m.group(1)), "stageId": int(m.group(2)), "dim": 3}
        if name == "challengeQ":
            return {"type": "challenge", "stage": self.n_stages + 1, "stageId": 0, "dim": 3}
        if name == "challengeXi":
            return {"type": "challenge", "stage": self.n_stages + 2, "stageId": 0, "dim": 3}
        m = re.fullmatch(r"challengesFRI\[(\d+)\]", name)
        if m:
            return {"type": "challenge", "stage": self.n_stages + 3, "stageId": int(m.group(1)), "dim": 3}
        m = re.fullmatch(r"xDivXSubXi\[(\d+)\]", name)
        if m:
            return {"type": "xDivXSubXi", "id": int(m.group(1)), "dim": 3}
        m = re.fullmatch(r"mapValues\.tree(\d+)_(\d+)", name)
        if m:
            pos, dim = self.tree_layout[int(m.group(1))][int(m.group(2))]
            assert dim == 3, name
            return {"type": "tree%d" % int(m.group(1)), "treePos": pos, "dim": 3}
        if name == "Zh":
            return {"type": "Zi", "boundaryId": 0, "dim": 3}
        if name in self.tmp:
            return {"type": "tmp", "id": self.tmp[name], "dim": 3}
        raise ValueError("unknown extension operand " + name)

    def base(self, name):
        name = name.strip()
        if re.fullmatch(r"\d+", name):
            return {"type": "number", "value": name, "dim": 1}
        m = re.fullmatch(r"publics\[(\d+)\]", name)
        if m:
            return {"type": "public", "id": int(m.group(1)), "dim": 1}
        m = re.fullmatch(r"consts\[(\d+)\]", name)
        if m:
            return {"type": "const", "id": int(m.group(1)), "dim": 1}
        m = re.fullmatch(r"mapValues\.tree(\d+)_(\d+)", name)
        if m:
            pos, dim = self.tree_layout[int(m.group(1))][int(m.group(2))]
            assert dim == 1, name
            return {"type": "tree%d" % int(m.group(1)), "treePos": pos, "dim": 1}
        raise ValueError("unknown base operand " + name)

    def emit(self, op, dest_name, srcs):
        tid = len(self.tmp)
        self.tmp[dest_name] = tid
        self.code.append({"op": op, "dest": {"type": "tmp", "id": tid, "dim": 3}, "src": srcs})

    # ---- one statement
    def triple(self, dest, comps):
        c0, c1, c2 = comps

        def is_ext0(a):
            return a.endswith("[0]") and not re.fullmatch(r"(publics|consts)\[\d+\]", a)

        def strip(a, k):
            assert a.endswith("[%d]" % k), (a, k)
            return a[:-3]
        m = re.fullmatch(r"(.+?) ([+\-*]) (.+)", c0)
        assert m, c0
        a, op, b = m.group(1).strip(), m.group(2), m.group(3).strip()
        opname = {"+": "add", "-": "sub", "*": "mul"}[op]
        ea, eb = is_ext0(a), is_ext0(b)
        if ea and eb:                                   # extension (op) extension, component-wise: + and - only
            assert op in "+-"
            A, B = strip(a, 0), strip(b, 0)
            for k, c in ((1, c1), (2, c2)):
                assert re.sub(r"\s+", " ", c) == "%s[%d] %s %s[%d]" % (A, k, op, B, k), (c0, c)
            return self.emit(opname, dest, [self.ext(A), self.ext(B)])
        if ea and not eb:                               # extension (op) base
            A = strip(a, 0)
            if op == "*":
                for k, c in ((1, c1), (2, c2)):
                    assert re.sub(r"\s+", " ", c) == "%s[%d] * %s" % (A, k, b), (c0, c)
            else:                                       # the base value only meets component 0
                for k, c in ((1, c1), (2, c2)):
                    assert c == "%s[%d]" % (A, k), (c0, c)
            return self.emit(opname, dest, [self.ext(A), self.base(b)])
        if eb and not ea:                               # base (op) extension: + keeps, - negates the upper components
            B = strip(b, 0)
            assert op in "+-"
            for k, c in ((1, c1), (2, c2)):
                want = ("%s[%d]" if op == "+" else "-%s[%d]") % (B, k)
                assert c.replace(" ", "") == want.replace(" ", ""), (c0, c)
            return self.emit(opname, dest, [self.base(a), self.ext(B)])
        raise ValueError("cannot translate " + c0)

    def run(self, body, stop_after=None):
        for line in body.split("\n"):
            line = line.strip()
            m = re.fullmatch(r"signal (tmp_\d+)\[3\] <== CMul\(\)\((.+)\);", line)
            if m:
                a, b = split_top(m.group(2))
                self.emit("mul", m.group(1), [self.ext(a), self.ext(b)])
            else:
                m = re.fullmatch(r"signal (tmp_\d+)\[3\] <== \[(.+)\];", line)
                if m:
                    self.triple(m.group(1), split_top(m.group(2)))
                else:
                    m = re.fullmatch(r"signal (tmp_\d+)\[3\] <== ([A-Za-z_][\w\.\[\]]*);", line)
                    if m:
                        self.emit("copy", m.group(1), [self.ext(m.group(2))])
                    else:
                        continue
            if stop_after is not None and m.group(1) == stop_after:
                break
        return {"tmpUsed": len(self.tmp), "code": self.code}


def main():
    src = open(os.path.join(REF, "test", "compressor", "verifier.circom")).read()
    sv = template_body(src, "StarkVerifier0")
    n_queries, n_bits_ext = (int(v) for v in re.search(r"queriesFRI\[(\d+)\]\[(\d+)\]", template_body(src, "Transcript0")).groups())
    ve = template_body(src, "VerifyEvaluations0")
    n_bits = int(re.search(r"signal zMul\[(\d+)\]\[3\]", ve).group(1))
    n_evals = int(re.search(r"signal input evals\[(\d+)\]\[3\]", ve).group(1))
    n_publics = int(re.search(r"signal input publics\[(\d+)\]", ve).group(1))
    stage_ch = {int(s): int(n) for s, n in re.findall(r"signal input challengesStage(\d+)\[(\d+)\]\[3\]", ve)}
    widths = {("C" if s == "C" else int(s)): int(w) for s, w in re.findall(r"signal input s0_vals(\w)\[\d+\]\[(\d+)\]", sv)}
    n_stages = max(k for k in widths if k != "C") - 1                     # the last committed stage is the quotient
    root_c = [int(v) for v in re.search(r"signal rootC\[4\] <== \[([^\]]+)\]", sv).group(1).split(",")]
    steps = [int(re.search(r"VerifyQuery0\((\d+), (\d+)\)", sv).group(1))] + [int(v) for v in re.findall(r"VerifyFRI0\(\d+, \d+, (\d+), \d+, \d+\)", sv)]
    # layout of the opened rows (MapValues0): polynomial j of stage s at (treePos, dim)
    mv = template_body(src, "MapValues0")
    layout = {}
    for s, j, rhs in re.findall(r"tree(\d+)_(\d+) <== (.+);", mv):
        idx = [int(v) for v in re.findall(r"vals\d+\[(\d+)\]", rhs)]
        layout.setdefault(int(s), {})[int(j)] = (idx[0], len(idx))
    layout = {s: [layout[s][j] for j in range(len(layout[s]))] for s in layout}
    for s in layout:
        assert sum(d for _, d in layout[s]) == widths[s]
    # the two programs
    t1 = Translator(n_stages, layout)
    q_code = t1.run(ve)
    last = re.findall(r"signal (tmp_\d+)\[3\] <== CMul\(\)\((tmp_\d+), Zh\);", ve)
    assert len(last) == 1 and t1.tmp[last[0][0]] == len(t1.code) - 1, "the identity's left-hand side must be the last op"
    q_deg = int(re.search(r"signal qAcc\[(\d+)\]\[3\]", ve).group(1))
    q_first = int(re.search(r"qAcc\[0\] <== evals\[(\d+)\+i\]", ve).group(1))
    fp = template_body(src, "CalculateFRIPolValue0")
    t2 = Translator(n_stages, layout)
    f_code = t2.run(fp)
    out_tmp = re.search(r"queryVals\[0\] <== (tmp_\d+)\[0\];", fp).group(1)
    assert t2.tmp[out_tmp] == len(t2.code) - 1
    openings = [0, 1]                                                       # den0inv: x - xi ; den1inv: x - roots(nBits) xi
    assert re.search(r"den1inv\[3\] <== CInv\(\)\(\[xacc\[\d+\] - 1 \* roots\(%d\) \* challengeXi\[0\]" % n_bits, fp)
    # evMap from the FRI polynomial: every "(value - evals[k])" term sits under the xDivXSubXi of its opening
    ev_map = [None] * n_evals
    pol_of = {}                                                             # (stage or "C", treePos) -> cmPolsMap id
    cm_pols = []
    for s in sorted(layout):
        for j, (pos, dim) in enumerate(layout[s]):
            pol_of[(s, pos)] = len(cm_pols)
            cm_pols.append({"stage": s, "name": "cm%d_%d" % (s, j), "dim": dim, "stagePos": pos, "stageId": j})
    # walk the ops: a sub whose second source is an eval names that evaluation's polynomial; the accumulated group it joins is
    # closed by the multiplication with xDivXSubXi[o]
    pending = []
    for c in f_code["code"]:
        if c["op"] == "sub" and c["src"][1]["type"] == "eval":
            pending.append((c["src"][1]["id"], c["src"][0]))
        if c["op"] == "mul" and c["src"][1]["type"] == "xDivXSubXi":
            for ev_id, pol in pending:
                if pol["type"] == "const":
                    ev_map[ev_id] = {"type": "const", "id": pol["id"], "prime": openings[c["src"][1]["id"]]}
                else:
                    ev_map[ev_id] = {"type": "cm", "id": pol_of[(int(pol["type"][4:]), pol["treePos"])], "prime": openings[c["src"][1]["id"]]}
            pending = []
    assert not pending and all(e is not None for e in ev_map)
    q_stage = n_stages + 1
    for i in range(q_deg):
        e = ev_map[q_first + i]
        assert e["type"] == "cm" and cm_pols[e["id"]]["stage"] == q_stage and cm_pols[e["id"]]["stageId"] == i
    challenges_map = []
    for s in sorted(stage_ch):
        challenges_map += [{"name": "stage%d_%d" % (s, i), "stage": s, "dim": 3, "stageId": i} for i in range(stage_ch[s])]
    challenges_map += [{"name": "std_vc", "stage": q_stage, "dim": 3, "stageId": 0}, {"name": "std_xi", "stage": q_stage + 1, "dim": 3, "stageId": 0},
                       {"name": "std_vf1", "stage": q_stage + 2, "dim": 3, "stageId": 0}, {"name": "std_vf2", "stage": q_stage + 2, "dim": 3, "stageId": 1}]
    info = {
        "starkStruct": {"nBits": n_bits, "nBitsExt": n_bits_ext, "nQueries": n_queries, "verificationHashType": "GL", "steps": [{"nBits": b} for b in steps]},
        "nStages": n_stages, "nConstants": widths["C"], "nPublics": n_publics, "nSubproofValues": 0, "qDeg": q_deg, "qDim": 3,
        "openingPoints": openings, "boundaries": [{"name": "everyRow"}],
        "mapSectionsN": dict({"const": widths["C"]}, **{"cm%d" % s: widths[s] for s in widths if s != "C"}),
        "cmPolsMap": cm_pols, "challengesMap": challenges_map, "evMap": ev_map,
    }
    out = {"source": "derived from test/compressor/verifier.circom of the reference by oracle/gen_compressor_verifier_programs.py",
           "constRoot": [str(v) for v in root_c], "starkInfo": info,
           "verifierInfo": {"qVerifier": q_code, "queryVerifier": f_code}}
    path = os.path.join(ROOT, "tests", "golden", "ref_compressor_verifier_programs.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote %s: %d + %d ops, %d evaluations, nBits %d/%d, steps %s, %d queries, stages %s" % (
        os.path.relpath(path, ROOT), len(q_code["code"]), len(f_code["code"]), n_evals, n_bits, n_bits_ext, steps, n_queries, stage_ch))


if __name__ == "__main__":
    main()
