"""TEST INFRASTRUCTURE ONLY -- a restatement of the reference's chelpers encoder, to produce `.chelpers.bin` files for the
reader in tests/chelpers_reader.py to be tested against.  Only tests/ import this.

parity unpinned by execution: src/stark/chelpers/{getParserArgs,helpers,generateParser,binFile}.js need chai and
@iden3/binfileutils, which are absent here, so the reference's own encoder cannot be run; the functions below
follow it line by line and cite it:
   all_operations      generateParser.js:519-578   (getAllOperations)
   get_operation       generateParser.js:580-618   (getOperation: sources sorted by dimension, then by operand rank)
   get_id_maps         helpers.js:3-98, 100-131    (getIdMaps / temporalsSubsets: temporaries renumbered by live range)
   get_parser_args     getParserArgs.js:12-200
   write_chelpers      binFile.js:23-610 (iden3 binfile container, little endian)
The generic file is written (ops index all_operations directly; stark_chelpers.js:104-113 getParserArgsCodeGeneric).
"""
import struct

P = 0xFFFFFFFF00000001
OPERATIONS_MAP = {"commit1": 1, "Zi": 2, "const": 3, "tmp1": 4, "public": 5, "number": 6, "commit3": 7, "xDivXSubXi": 8, "tmp3": 9,
                  "subproofValue": 10, "challenge": 11, "eval": 12}                    # generateParser.js:1-14
OP_TYPE = {"add": 0, "sub": 1, "mul": 2, "sub_swap": 3}                                  # getParserArgs.js:5-10


def all_operations():
    out = []
    d1, d3 = ["commit1", "tmp1"], ["commit3", "tmp3"]
    s1, s3 = ["commit1", "tmp1", "public", "number"], ["commit3", "tmp3", "challenge", "subproofValue"]
    for dest in d1:
        for k in range(len(s1)):
            for l in range(k, len(s1)):
                out.append({"dest_type": dest, "src0_type": s1[k], "src1_type": s1[l]})
    for dest in d3:
        for a in s3:
            for b in s1:
                out.append({"dest_type": dest, "src0_type": a, "src1_type": b})
        for k in range(len(s3)):
            for l in range(k, len(s3)):
                a, b = s3[k], s3[l]
                if a == "challenge":
                    out.append({"op": "mul", "dest_type": dest, "src0_type": b, "src1_type": a})
                elif b == "challenge":
                    out.append({"op": "mul", "dest_type": dest, "src0_type": a, "src1_type": b})
                out.append({"dest_type": dest, "src0_type": a, "src1_type": b})
    out.append({"op": "mul", "dest_type": "tmp3", "src0_type": "eval", "src1_type": "challenge"})
    out.append({"dest_type": "tmp3", "src0_type": "challenge", "src1_type": "eval"})
    out.append({"dest_type": "tmp3", "src0_type": "tmp3", "src1_type": "eval"})
    out.append({"dest_type": "tmp3", "src0_type": "eval", "src1_type": "commit1"})
    out.append({"dest_type": "tmp3", "src0_type": "commit3", "src1_type": "eval"})
    return out


def _rank(r):
    if r["type"] == "cm":
        return OPERATIONS_MAP["commit%d" % r["dim"]]
    if r["type"] == "tmp":
        return OPERATIONS_MAP["tmp%d" % r["dim"]]
    return OPERATIONS_MAP[r["type"]]


def _class(r):
    t = r["type"]
    if t == "cm":
        return "commit%d" % r["dim"]
    if t in ("const", "Zi"):
        return "commit1"
    if t == "xDivXSubXi":
        return "commit3"
    if t == "tmp":
        return "tmp%d" % r["dim"]
    return t


def get_operation(r):
    """generateParser.js:580-618.  The two sources are sorted (V8 compares (second, first) for a pair): the second one goes
    first when its dimension is larger, or, at equal dimensions, when its rank is smaller; a `sub` whose sources were
    turned round becomes sub_swap."""
    op = {"op": r["op"]}
    d = r["dest"]
    op["dest_type"] = ("commit%d" % d["dim"]) if d["type"] == "cm" else ("tmp%d" % d["dim"]) if d["type"] == "tmp" else d["type"]
    src = list(r["src"])
    if len(src) == 2:
        a, b = src[1], src[0]
        swap = (b["dim"] - a["dim"]) if a["dim"] != b["dim"] else (_rank(a) - _rank(b))
        if swap < 0:
            src = [src[1], src[0]]
            if r["op"] == "sub":
                op["op"] = "sub_swap"
    for i, s_ in enumerate(src):
        op["src%d_type" % i] = _class(s_)
    op["src"] = src
    return op


def _temporals_subsets(segments):                                                       # helpers.js:100-131
    segments = sorted(segments, key=lambda s_: s_[1])                                   # stable, like Array.prototype.sort in Node >= 11
    subsets = []
    for seg in segments:
        best, best_d = None, None
        for sub in subsets:
            last = sub[-1]
            if last[0] < seg[1] and seg[0] < last[1]:                                   # isIntersecting
                continue
            dist = abs(last[1] - seg[0])
            if best is None or dist < best_d:
                best, best_d = sub, dist
        if best is not None:
            best.append(seg)
        else:
            subsets.append([seg])
    return subsets


def get_id_maps(code):                                                                   # helpers.js:3-98
    span = {1: {}, 3: {}}
    for j, r in enumerate(code):
        for ref in [r["dest"]] + list(r["src"]):
            if ref["type"] == "tmp":
                s_ = span[ref["dim"]].setdefault(ref["id"], [j, j])
                s_[1] = j
    ids, counts = {1: {}, 3: {}}, {}
    for dim in (1, 3):
        segs = [[a, b, i] for i, (a, b) in sorted(span[dim].items())]
        subsets = _temporals_subsets(segs)
        for n, sub in enumerate(subsets):
            for seg in sub:
                ids[dim][seg[2]] = n
        counts[dim] = len(subsets)
    return ids[1], ids[3], counts[1], counts[3]


def get_parser_args(starkInfo, operations, code, dom, debug=False):                      # getParserArgs.js:12-200
    ops, args, numbers = [], [], []
    code_ = code["code"]
    ID1D, ID3D, count1d, count3d = get_id_maps(code_)
    nStages = starkInfo["nStages"]

    def eval_map(pol_id, prime):
        p = starkInfo["cmPolsMap"][pol_id]
        args.extend([int(p["stage"]), int(p["stagePos"]), starkInfo["openingPoints"].index(prime)])

    def push_res(r):
        d = r["dest"]
        if d["type"] == "tmp":
            args.append(ID1D[d["id"]] if d["dim"] == 1 else ID3D[d["id"]])
        elif d["type"] == "cm":
            eval_map(d["id"], d.get("prime", 0))
        else:
            raise ValueError("Invalid reference type set: " + d["type"])

    def push_src(r):
        t = r["type"]
        if t == "tmp":
            args.append(ID1D[r["id"]] if r["dim"] == 1 else ID3D[r["id"]])
        elif t == "const":
            args.extend([0, r["id"], starkInfo["openingPoints"].index(r["prime"])])
        elif t == "cm":
            eval_map(r["id"], r["prime"])
        elif t == "number":
            num = int(r["value"], 0) if isinstance(r["value"], str) else int(r["value"])
            if num < 0:
                num += P
            s_ = str(num)
            if s_ not in numbers:
                numbers.append(s_)
            args.append(numbers.index(s_))
        elif t in ("public", "subproofValue", "eval", "challenge"):
            args.append(r["id"])
        elif t == "xDivXSubXi":
            args.extend([nStages + 2, 0, 3 * r["id"]])
        elif t == "Zi":
            args.extend([nStages + 2, 0, r["boundaryId"]])
    for r in code_:
        operation = get_operation(r)
        args.append(OP_TYPE[operation["op"]])
        push_res(r)
        for s_ in operation["src"]:
            push_src(s_)
        if operation["op"] == "mul" and operation["dest_type"] in ("tmp3", "commit3") and operation.get("src1_type") == "challenge":
            idx = next((i for i, o in enumerate(operations) if o.get("op") == "mul" and o["dest_type"] == operation["dest_type"]
                        and o["src0_type"] == operation["src0_type"] and o["src1_type"] == operation["src1_type"]), -1)
        else:
            idx = next((i for i, o in enumerate(operations) if not o.get("op") and o["dest_type"] == operation["dest_type"]
                        and o["src0_type"] == operation["src0_type"] and o["src1_type"] == operation.get("src1_type")), -1)
        if idx == -1:
            raise ValueError("Operation not considered: %r" % {k: v for k, v in operation.items() if k != "src"})
        ops.append(idx)
    used = code.get("symbolsUsed", [])
    info = {"nTemp1": count1d, "nTemp3": count3d, "ops": ops, "numbers": numbers, "args": args,
            "cmPolsIds": sorted(s_["id"] for s_ in used if s_["op"] == "cm"), "constPolsIds": sorted(s_["id"] for s_ in used if s_["op"] == "const"),
            "challengeIds": sorted(s_["id"] for s_ in used if s_["op"] == "challenge"), "publicsIds": sorted(s_["id"] for s_ in used if s_["op"] == "public"),
            "subproofValuesIds": sorted(s_["id"] for s_ in used if s_["op"] == "subproofValue")}
    if debug:
        d = code_[-1]["dest"]
        info["destDim"] = d["dim"]
        info["destId"] = ID1D[d["id"]] if d["dim"] == 1 else ID3D[d["id"]]
    return info


_STREAMS = (("ops", "B"), ("args", "H"), ("numbers", "Q"), ("constPolsIds", "H"), ("cmPolsIds", "H"), ("challengeIds", "H"), ("publicsIds", "H"), ("subproofValuesIds", "H"))


def _code_section(entries, head_fields):                                                 # binFile.js:49-210 / 212-395 / 397-580
    out = b""
    for name, _ in _STREAMS:
        out += struct.pack("<I", sum(len(e[name]) for e in entries))
    out += struct.pack("<I", len(entries))
    offs = {name: 0 for name, _ in _STREAMS}
    for e in entries:
        for f in head_fields:
            out += struct.pack("<I", e[f])
        for name, _ in _STREAMS:
            out += struct.pack("<II", len(e[name]), offs[name]); offs[name] += len(e[name])
    for name, fmt in _STREAMS:
        vals = [int(v) for e in entries for v in e[name]]
        out += struct.pack("<%d%s" % (len(vals), fmt), *vals)
    return out


def write_chelpers(path, bin_info):                                                      # binFile.js:23-47, 582-610
    secs = {2: _code_section(bin_info["imPolsInfo"], ["nTemp1", "nTemp3"]),
            3: _code_section(bin_info["expsInfo"], ["expId", "destDim", "destId", "stage", "nTemp1", "nTemp3"]),
            4: _code_section(bin_info["constraintsInfo"], ["stage", "destDim", "destId", "firstRow", "lastRow", "nTemp1", "nTemp3"])}
    h = struct.pack("<I", len(bin_info["hintsInfo"]))
    for hint in bin_info["hintsInfo"]:
        h += hint["name"].encode("latin1") + b"\0" + struct.pack("<I", len(hint["fields"]))
        for f in hint["fields"]:
            h += f["name"].encode("latin1") + b"\0" + f["op"].encode("latin1") + b"\0"
            h += struct.pack("<Q", int(f["value"])) if f["op"] == "number" else struct.pack("<I", f["id"])
            if f["op"] == "tmp":
                h += struct.pack("<I", f["dim"])
    secs[5] = h
    with open(path, "wb") as f:
        f.write(b"chps" + struct.pack("<II", 1, len(secs)))
        for t in sorted(secs):
            f.write(struct.pack("<IQ", t, len(secs[t])) + secs[t])


def build_generic_bin_info(starkInfo, expressionsInfo):
    """stark_chelpers.js:5-175 with genericBinFile: intermediate-polynomial code per stage, every constraint (debug), every
    expression (the constraint and FRI expressions' last destination redirected to a fresh temporary, :90-95)"""
    import copy
    ops = all_operations()
    N = 1 << starkInfo["starkStruct"]["nBits"]
    info = {"imPolsInfo": [get_parser_args(starkInfo, ops, c, "n") for c in expressionsInfo.get("imPolsCode", [])], "constraintsInfo": [], "expsInfo": [],
            "hintsInfo": expressionsInfo.get("hintsInfo", [])}
    for c in expressionsInfo.get("constraints", []):
        b = c["boundary"]
        first, last = {"everyRow": (0, N), "firstRow": (0, 1), "finalProof": (0, 1), "lastRow": (N - 1, N)}.get(b, (c.get("offsetMin", 0), N - c.get("offsetMax", 0)))
        e = get_parser_args(starkInfo, ops, c, "n", True)
        e.update(stage=c["stage"], firstRow=first, lastRow=last)
        info["constraintsInfo"].append(e)
    for exp in expressionsInfo["expressionsCode"]:
        if not exp:
            continue
        exp = copy.deepcopy(exp)
        special = exp["expId"] in (starkInfo["cExpId"], starkInfo["friExpId"])
        if special:
            last = exp["code"]["code"][-1]["dest"]
            last["type"] = "tmp"; last["id"] = exp["code"]["tmpUsed"]; exp["code"]["tmpUsed"] += 1
        e = get_parser_args(starkInfo, ops, exp["code"], "n", True)
        e.update(expId=exp["expId"], stage=exp["stage"])
        if special:
            e["destDim"], e["destId"] = 0, 0
        info["expsInfo"].append(e)
    return info
