"""Test infrastructure for the prove loop: a CPU backend on the oracle (to produce the expected proof) and a
verifier restating src/stark/stark_verify.js:8-218 + src/stark/fri.js:107-174 on python ints / oracle calls."""
import numpy as np

import gl_oracle as orc

P = 0xFFFFFFFF00000001


from stark_backend import OracleBackend  # noqa: E402,F401


# ---------------------------------------------------------------------------------------------- scalar op-list evaluation
def _e3(v):
    return list(v) if isinstance(v, (list, tuple)) else [v, 0, 0]


def _mul(a, b):
    if isinstance(a, list) and isinstance(b, list):
        return [int(x) for x in orc.mul3([x % P for x in a], [x % P for x in b])]
    if isinstance(a, list):
        return [x * b % P for x in a]
    if isinstance(b, list):
        return [x * a % P for x in b]
    return a * b % P


def _add(a, b):
    if isinstance(a, list) and isinstance(b, list):
        return [(x + y) % P for x, y in zip(a, b)]
    if isinstance(a, list):
        return [(a[0] + b) % P, a[1], a[2]]
    if isinstance(b, list):
        return [(a + b[0]) % P, b[1], b[2]]
    return (a + b) % P


def _sub(a, b):                                  # f3g.js:60-71 incl. scalar - triple
    if isinstance(a, list) and isinstance(b, list):
        return [(x - y) % P for x, y in zip(a, b)]
    if isinstance(a, list):
        return [(a[0] - b) % P, a[1], a[2]]
    if isinstance(b, list):
        return [(a - b[0]) % P, (-b[1]) % P, (-b[2]) % P]
    return (a - b) % P


def exec_code(code, resolve):
    """stark_verify.js:222-298 executeCode: returns the value of the last op"""
    tmp = {}
    res = None
    for c in code:
        src = [tmp[r["id"]] if r["type"] == "tmp" else resolve(r) for r in c["src"]]
        res = {"add": _add, "sub": _sub, "mul": _mul}[c["op"]](src[0], src[1]) if c["op"] != "copy" else src[0]
        tmp[c["dest"]["id"]] = res
    return res


def eval_expr(e, resolve):
    if e.op == "leaf":
        return resolve(e.leaf)
    a, b = eval_expr(e.a, resolve), eval_expr(e.b, resolve)
    return {"add": _add, "sub": _sub, "mul": _mul}[e.op](a, b)


def _pow3(x, e):
    r = [1, 0, 0]; b = list(x)
    while e:
        if e & 1:
            r = _mul(r, b)
        b = _mul(b, b); e >>= 1
    return r


def stark_verify(res, constRoot, info, verifierInfo, split=False, check_transcript=True, hash_type="GL", arity=16, custom=False):
    from pil2gl.stark import root_of_unity, SHIFT
    proof, publics = res["proof"], res["publics"]
    ss = info["starkStruct"]; nb, nbe = ss["nBits"], ss["nBitsExt"]; N = 1 << nb; steps = ss["steps"]
    be = OracleBackend(split, hash_type, arity, custom)

    def path_root(vals, idx, sib):
        """calculateRootFromGroupProof of the tree kind in use (merklehash_p.js:170-210 / merklehash_bn128_p.js:184-232)"""
        if hash_type == "BN128":
            import bn128_oracle
            return bn128_oracle.root_from_group_proof(sib, idx, [int(v) for v in vals], arity, custom)
        return [int(v) for v in orc.root_from_proof(np.array(vals, dtype=np.uint64), idx, np.array(sib, dtype=np.uint64), split)]
    same_root = (lambda a, b: int(a) == int(b)) if hash_type == "BN128" else (lambda a, b: list(a) == list(b))
    # transcript replay, calculateTranscriptVerify.js:7-103
    nStages = info["nStages"]; qStage = nStages + 1
    t = be.new_transcript()
    hc = bool(ss.get("hashCommits", False))

    def commit_hash(values):                                 # calculateHashStark, stark_gen_helpers.js:442-461
        h = be.new_transcript()
        for v in values:
            h.put(v)
        return h.getState()
    t.put(constRoot)
    if not hc:                                               # calculateTranscriptVerify.js:30-37
        for v in publics:
            t.put(v)
    else:
        t.put(commit_hash(publics))
    challenges = {}
    got_ch = [[] for _ in range(nStages + 3)]
    for st in range(1, qStage + 1):
        if st > 1:
            n_ch = sum(1 for c in info["challengesMap"] if c["stage"] == st)
            got_ch[st - 1] = [t.getField() for _ in range(n_ch)]
        t.put(proof["root%d" % st])
    xi = t.getField(); got_ch[qStage] = [xi]
    if not hc:                                               # :61-68
        for ev in proof["evals"]:
            t.put(ev)
    else:
        t.put(commit_hash(proof["evals"]))
    vf1 = t.getField(); vf2 = t.getField(); got_ch[qStage + 1] = [vf1, vf2]
    for st, lst in enumerate(got_ch):
        for k, c in enumerate(lst):
            challenges[(st + 1, k)] = c
    chF = []
    for step in range(len(steps)):
        chF.append(t.getField())
        if step < len(steps) - 1:
            t.put(proof["fri"][step + 1]["root"])
        elif not hc:                                         # :87-94
            for e in proof["fri"][-1]:
                t.put(e)
        else:
            t.put(commit_hash(proof["fri"][-1]))
    chF.append(t.getField())
    if check_transcript and not (got_ch == res["challenges"] and chF == res["challengesFRISteps"]):
        return False, "transcript does not reproduce the challenges"
    tq = be.new_transcript(); tq.put(chF[-1])
    queries = tq.getPermutations(ss["nQueries"], steps[0]["nBits"])
    if check_transcript and queries != res["queries"]:
        return False, "query positions differ"

    # evaluations, stark_verify.js:95-152
    xN = _pow3(xi, N)
    Z = [int(v) for v in orc.inv3([(xN[0] - 1) % P, xN[1], xN[2]])]
    # one zerofier per boundary (stark_verify.js:99-136): Z; Z_fr = zh / (xi - 1); Z_lr = zh / (xi - w^(N-1)); Z_frame = prod (xi - root)
    zh = [(xN[0] - 1) % P, xN[1], xN[2]]
    w = root_of_unity(nb)
    xi_minus = lambda r: [(int(xi[0]) - r) % P, int(xi[1]), int(xi[2])]
    m3 = lambda a_, b_: [int(v) for v in orc.mul3(a_, b_)]
    Zs = []
    for bd in info.get("boundaries", [{"name": "everyRow"}]):
        if bd["name"] == "everyRow": Zs.append(Z)
        elif bd["name"] == "firstRow": Zs.append(m3(zh, [int(v) for v in orc.inv3(xi_minus(1))]))
        elif bd["name"] == "lastRow": Zs.append(m3(zh, [int(v) for v in orc.inv3(xi_minus(pow(w, N - 1, P)))]))
        else:
            z = [1, 0, 0]
            for j in range(bd["offsetMin"]): z = m3(z, xi_minus(pow(w, j, P)))
            for j in range(bd["offsetMax"]): z = m3(z, xi_minus(pow(w, N - j - 1, P)))
            Zs.append(z)

    def resolve(r):
        ty = r["type"]
        if ty == "eval": return list(proof["evals"][r["id"]])
        if ty == "challenge": return list(challenges[(r["stage"], r["stageId"])])
        if ty == "public": return publics[r["id"]]
        if ty == "number": return int(r["value"]) % P
        if ty == "Zi": return list(Zs[r.get("boundaryId", 0)])
        if ty == "subproofValue":                            # stark_verify.js:19,256
            v = proof["subproofValues"][r["id"]]
            return [int(x) % P for x in v] if isinstance(v, (list, tuple)) else int(v) % P
        raise ValueError(ty)
    lhs = exec_code(verifierInfo["qVerifier"]["code"], resolve)
    q_ids = [k for k, pm in enumerate(info["cmPolsMap"]) if pm["stage"] == qStage]
    q = [0, 0, 0]; xAcc = [1, 0, 0]
    for i in range(info["qDeg"]):
        evId = next(k for k, e in enumerate(info["evMap"]) if e["type"] == "cm" and e["id"] == q_ids[i])
        q = _add(q, _mul(xAcc, list(proof["evals"][evId]))); xAcc = _mul(xAcc, xN)
    if _e3(lhs) != q:
        return False, "Invalid evaluations"

    # queries, stark_verify.js:158-218 + fri.js:107-174
    wN, wE = root_of_unity(nb), root_of_unity(nbe)
    for qi, idx in enumerate(queries):
        pq = proof["fri"][0]["polQueries"][qi]
        for (vals, sib), root in zip(pq, [proof["root%d" % st] for st in range(1, qStage + 1)] + [constRoot]):
            if not same_root(path_root(vals, idx, sib), root):
                return False, "Invalid root (query %d)" % qi
        x = SHIFT * pow(wE, idx, P) % P
        xdiv = []
        for opening in info["openingPoints"]:
            w = pow(wN, abs(opening), P)
            if opening < 0:                                   # stark_verify.js:206-208
                w = pow(w, P - 2, P)
            den = _sub(x, [c * w % P for c in xi])
            xdiv.append(_mul([int(v) for v in orc.inv3(den)], x))

        def rq(r):
            ty = r["type"]
            if ty == "cm":
                p = info["cmPolsMap"][r["id"]]; vals = pq[p["stage"] - 1][0]
                return vals[p["stagePos"]] if p["dim"] == 1 else vals[p["stagePos"]:p["stagePos"] + 3]
            if ty == "const": return pq[qStage][0][r["id"]]
            if ty == "xDivXSubXi": return list(xdiv[r["id"]])
            return resolve(r)

        def pol_leaf(ev):
            from pil2gl.stark import Expr
            dim = 1 if ev["type"] == "const" else info["cmPolsMap"][ev["id"]]["dim"]
            return Expr.leafOf({"type": ev["type"], "id": ev["id"], "prime": 0, "dim": dim})
        val = _e3(eval_expr(verifierInfo["friexp"](pol_leaf), rq))
        # walk the FRI layers for this query (fri.js:118-150)
        pol_bits = nbe; cur_idx = idx; group = None
        shift = SHIFT
        for si in range(len(steps)):
            if si == 0:
                ev = val
            else:
                vals, sib = proof["fri"][si]["polQueries"][qi]
                if not same_root(path_root(vals, cur_idx, sib), proof["fri"][si]["root"]):
                    return False, "Invalid FRI root step %d" % si
                g = np.array(vals, dtype=np.uint64).reshape(-1, 3)
                sinv = pow(shift * pow(root_of_unity(pol_bits), cur_idx, P) % P, P - 2, P)
                ev = [int(v) for v in orc.fri_fold(g, 0, sinv, np.array(chF[si], dtype=np.uint64))[0]]
            if si < len(steps) - 1:
                nxt_groups = 1 << steps[si + 1]["nBits"]
                group_idx = cur_idx // nxt_groups
                nv = proof["fri"][si + 1]["polQueries"][qi][0]
                if nv[3 * group_idx:3 * group_idx + 3] != ev:
                    return False, "FRI layer %d mismatch (query %d)" % (si, qi)
                red = pol_bits - steps[si]["nBits"]
                for _ in range(red):
                    shift = shift * shift % P
                pol_bits = steps[si]["nBits"]
                cur_idx = cur_idx % nxt_groups
            else:
                if list(proof["fri"][-1][cur_idx]) != ev:
                    return False, "FRI last layer mismatch (query %d)" % qi
    # low degree of the last polynomial (fri.js:154-171)
    last = np.array(proof["fri"][-1], dtype=np.uint64)
    lb = steps[-1]["nBits"]
    coef = orc.fft3(last, inverse=True)
    max_deg = 0 if lb - (nbe - nb) < 0 else 1 << (lb - (nbe - nb))
    for i in range(max_deg + 1, coef.shape[0]):
        if coef[i].any():
            return False, "last polynomial is not low degree"
    return True, "ok"
