"""Reader for the reference's expression bytecode: the generic `.chelpers.bin` of src/stark/chelpers.

The reference compiles every op-list (`code.code`) of a circuit into three flat streams -- `ops` (one byte per operation: an
index into the fixed operation table of generateParser.js:519-578), `args` (u16: operation kind, then the operands' fields)
and `numbers` (u64 constants) -- after renumbering the temporaries by live range (helpers.js:3-98), and stores them in an iden3
"binfile" container (binFile.js:23-610: magic "chps", sections 2 = intermediate polynomials, 3 = expressions, 4 = constraints
(debug), 5 = hints).  This module parses that container and turns an entry back into the operand records the device
evaluator's encoder (pil2gl.stark.encode_code) takes, so a circuit compiled by the reference's own front end runs on
pil2gl_eval_program_dev without the JSON op-lists.

Only the GENERIC file is self-describing (its `ops` index getAllOperations() directly; stark_chelpers.js:27,104-113).  The
per-circuit file renumbers `ops` to the subset its generated C++ parser implements and fuses patterns (stark_chelpers.js:
120-146, helpers.js:142-218): it can only be read together with that generated parser and is not handled here.
"""
import struct

P = 0xFFFFFFFF00000001
CHELPERS_IMPOLS_SECTION, CHELPERS_EXPRESSIONS_SECTION, CHELPERS_CONSTRAINTS_DEBUG_SECTION, CHELPERS_HINTS_SECTION = 2, 3, 4, 5
OP_KINDS = ("add", "sub", "mul", "sub_swap")                  # getParserArgs.js:5-10


def all_operations():
    """generateParser.js:519-578 getAllOperations(): the fixed table `ops` bytes index.  Entries: (fixed op or None,
    dest_type, src0_type, src1_type) with types commit1/tmp1/public/number (dim 1), commit3/tmp3/challenge/subproofValue (dim 3), eval"""
    out = []
    d1, d3 = ["commit1", "tmp1"], ["commit3", "tmp3"]
    s1, s3 = ["commit1", "tmp1", "public", "number"], ["commit3", "tmp3", "challenge", "subproofValue"]
    for dest in d1:
        for k in range(len(s1)):
            for l in range(k, len(s1)):
                out.append((None, dest, s1[k], s1[l]))
    for dest in d3:
        for a in s3:
            for b in s1:
                out.append((None, dest, a, b))
        for k in range(len(s3)):
            for l in range(k, len(s3)):
                a, b = s3[k], s3[l]
                if a == "challenge":
                    out.append(("mul", dest, b, a))
                elif b == "challenge":
                    out.append(("mul", dest, a, b))
                out.append((None, dest, a, b))
    out += [("mul", "tmp3", "eval", "challenge"), (None, "tmp3", "challenge", "eval"), (None, "tmp3", "tmp3", "eval"),
            (None, "tmp3", "eval", "commit1"), (None, "tmp3", "commit3", "eval")]
    return out


class _Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def u32(self):
        v = struct.unpack_from("<I", self.d, self.o)[0]; self.o += 4; return v

    def u64(self):
        v = struct.unpack_from("<Q", self.d, self.o)[0]; self.o += 8; return v

    def arr(self, fmt, n):
        v = list(struct.unpack_from("<%d%s" % (n, fmt), self.d, self.o)); self.o += n * struct.calcsize(fmt); return v

    def cstr(self):
        e = self.d.index(b"\0", self.o); s = self.d[self.o:e].decode("latin1"); self.o = e + 1; return s


def _sections(data):
    """iden3 binfile container: magic(4) version(u32) nSections(u32) then [type u32, size u64, bytes]*"""
    if data[:4] != b"chps":
        raise ValueError("not a chelpers file (magic %r)" % data[:4])
    r = _Reader(data); r.o = 4
    version, n = r.u32(), r.u32()
    if version != 1:
        raise ValueError("chelpers version %d not supported" % version)
    secs = {}
    for _ in range(n):
        t, size = r.u32(), r.u64()
        secs[t] = data[r.o:r.o + size]; r.o += size
    return secs


_STREAMS = ("ops", "args", "numbers", "constPolsIds", "cmPolsIds", "challengeIds", "publicsIds", "subproofValuesIds")


def _code_section(data, head_fields):
    """sections 2-4 share one layout (binFile.js:49-210, 212-395, 397-580): eight stream lengths, the entry count, per entry
    `head_fields` u32 values then (length, offset) per stream, then the eight concatenated streams"""
    r = _Reader(data)
    totals = [r.u32() for _ in range(8)]
    n = r.u32()
    entries = []
    for _ in range(n):
        e = {f: r.u32() for f in head_fields}
        e["_spans"] = [(r.u32(), r.u32()) for _ in range(8)]
        entries.append(e)
    streams = [r.arr("B", totals[0]), r.arr("H", totals[1]), r.arr("Q", totals[2])] + [r.arr("H", totals[k]) for k in range(3, 8)]
    for e in entries:
        for name, (ln, off), st in zip(_STREAMS, e.pop("_spans"), streams):
            e[name] = st[off:off + ln]
    return entries


def read_chelpers_bin(path):
    """-> {"imPols": [...], "expressions": [...], "constraints": [...], "hints": [...]}; every code entry carries nTemp1, nTemp3,
    ops, args, numbers and the symbol id lists; expressions also expId, destDim, destId, stage; constraints stage, destDim,
    destId, firstRow, lastRow"""
    with open(path, "rb") as f:
        secs = _sections(f.read())
    out = {"imPols": _code_section(secs[CHELPERS_IMPOLS_SECTION], ["nTemp1", "nTemp3"]),
           "expressions": _code_section(secs[CHELPERS_EXPRESSIONS_SECTION], ["expId", "destDim", "destId", "stage", "nTemp1", "nTemp3"]),
           "constraints": _code_section(secs[CHELPERS_CONSTRAINTS_DEBUG_SECTION], ["stage", "destDim", "destId", "firstRow", "lastRow", "nTemp1", "nTemp3"]),
           "hints": []}
    r = _Reader(secs[CHELPERS_HINTS_SECTION])                  # binFile.js:582-610
    for _ in range(r.u32()):
        h = {"name": r.cstr(), "fields": []}
        for _ in range(r.u32()):
            fld = {"name": r.cstr(), "op": r.cstr()}
            if fld["op"] == "number":
                fld["value"] = r.u64()
            else:
                fld["id"] = r.u32()
            if fld["op"] == "tmp":
                fld["dim"] = r.u32()
            h["fields"].append(fld)
        out["hints"].append(h)
    return out


def decode_code(entry, starkInfo):
    """one entry of read_chelpers_bin -> (code, destination) where code is a list of reference-style op records
    ({"op", "dest": {...}, "src": [...]}; codegen.js:75-125) over the entry's renumbered temporaries -- what
    pil2gl.stark.encode_code encodes for the device evaluator -- and destination = the record of the expression's result
    (the last op's dest).  Operand fields as getParserArgs.js:100-198 wrote them:
       tmp: id                              const: 0, id, index of prime in openingPoints
       cm: stage, stagePos, prime index     number: index into numbers        public/subproofValue/eval/challenge: id
       xDivXSubXi: nStages+2, 0, 3*id       Zi: nStages+2, 0, boundaryId
    A dim-1 and a dim-3 temporary with the same number are different slots (ID1D / ID3D): they come back as ids 2k, 2k+1."""
    table = all_operations()
    nStages = starkInfo["nStages"]
    opening = starkInfo["openingPoints"]
    by_pos = {(p["stage"], p["stagePos"]): i for i, p in enumerate(starkInfo["cmPolsMap"])}
    args, numbers = entry["args"], entry["numbers"]
    pos = [0]

    def take():
        v = args[pos[0]]; pos[0] += 1; return v

    def operand(t):
        if t in ("tmp1", "tmp3"):
            dim = 1 if t == "tmp1" else 3
            return {"type": "tmp", "id": 2 * take() + (dim == 3), "dim": dim}
        if t in ("commit1", "commit3"):
            stage, a, b = take(), take(), take()
            if stage == 0:
                return {"type": "const", "id": a, "prime": opening[b], "dim": 1}
            if stage == nStages + 2:                                   # the tables of the FRI / quotient stages
                return {"type": "Zi", "boundaryId": b, "dim": 1} if t == "commit1" else {"type": "xDivXSubXi", "id": b // 3, "dim": 3}
            pid = by_pos[(stage, a)]
            return {"type": "cm", "id": pid, "prime": opening[b], "dim": starkInfo["cmPolsMap"][pid]["dim"]}
        if t == "number":
            return {"type": "number", "value": str(numbers[take()]), "dim": 1}
        if t == "public":
            return {"type": "public", "id": take(), "dim": 1}
        if t == "challenge":                                           # the flat id indexes challengesMap (map.js:53)
            i = take(); c = starkInfo["challengesMap"][i]
            return {"type": "challenge", "id": i, "stage": c["stage"], "stageId": c["stageId"], "dim": 3}
        if t in ("subproofValue", "eval"):
            return {"type": t, "id": take(), "dim": 3}
        raise ValueError("unknown operand class " + t)
    code = []
    for opi in entry["ops"]:
        if opi >= len(table):
            raise ValueError("operation %d is not in the generic table: a per-circuit chelpers file needs its generated parser" % opi)
        fixed, dt, s0t, s1t = table[opi]
        kind = OP_KINDS[take()]
        dest = operand(dt)
        a, b = operand(s0t), operand(s1t)
        if fixed is not None and kind != fixed:
            raise ValueError("operation %d is %s-only but the stream says %s" % (opi, fixed, kind))
        if kind == "sub_swap":                                           # the sources were sorted and the subtraction turned round (generateParser.js:592-598)
            a, b, kind = b, a, "sub"
        code.append({"op": kind, "dest": dest, "src": [a, b]})
    if pos[0] != len(args):
        raise ValueError("argument stream not consumed: %d of %d" % (pos[0], len(args)))
    return code, (code[-1]["dest"] if code else None)
