"""On-disk formats of the reference, to and from device memory.

  load_pols / save_pols   <-> witnessCalculator.js:145-196 (`.commit` / `.const`: raw little-endian u64, row-major n x nCols,
                              no header) streamed through a bounded host buffer (a 107 GB trace never sits in host RAM)
  MerkleHash.writeToFile / readFromFile (pil2gl/__init__.py, pil2gl/bn128.py) <-> `.consttree` (merklehash_p.js:228-278)
  proof2zkin / zkin_json  <-> src/proof2zkin.js:1-75 and the stringification of main_prover.js:141-148
"""
import json

import numpy as np

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

CHUNK_WORDS = 1 << 25           # 256 MB, the reference's MaxBuffSize (witnessCalculator.js:148)


def load_pols(fileName, n, nCols, device=None):
    """-> n*nCols words: numpy uint64 (device None) or a torch int64 tensor on `device`"""
    total = n * nCols
    if device is None:
        a = np.fromfile(fileName, dtype="<u8", count=total).astype(np.uint64, copy=False)
        if a.size != total:
            raise ValueError("%s holds %d words, expected %d" % (fileName, a.size, total))
        return a
    out = torch.empty(total, dtype=torch.int64, device=device)
    with open(fileName, "rb") as f:
        for o in range(0, total, CHUNK_WORDS):
            m = min(CHUNK_WORDS, total - o)
            a = np.fromfile(f, dtype="<u8", count=m)
            if a.size != m:
                raise ValueError("%s is shorter than %d words" % (fileName, total))
            out[o:o + m] = torch.from_numpy(a.view(np.int64)).to(device)
    return out


def save_pols(buf, fileName):
    """canonical little-endian u64 words of a numpy array or device tensor, streamed (witnessCalculator.js:145-170)"""
    with open(fileName, "wb") as f:
        if torch is not None and isinstance(buf, torch.Tensor):
            flat = buf.reshape(-1)
            for o in range(0, flat.numel(), CHUNK_WORDS):
                flat[o:o + CHUNK_WORDS].cpu().numpy().view(np.uint64).astype("<u8", copy=False).tofile(f)
        else:
            np.ascontiguousarray(buf, dtype=np.uint64).reshape(-1).astype("<u8", copy=False).tofile(f)


def proof2zkin(p, starkInfo):
    """src/proof2zkin.js:1-75"""
    friSteps = starkInfo["starkStruct"]["steps"]
    nQueries = starkInfo["starkStruct"]["nQueries"]
    nStages = starkInfo["nStages"]
    qStage = nStages + 1
    z = {"root1": p["root1"]}
    for stage in range(2, nStages + 1):
        z["root%d" % stage] = p["root%d" % stage]
    z["root%d" % qStage] = p["root%d" % qStage]
    z["evals"] = p["evals"]
    for i in range(1, len(friSteps)):
        z["s%d_root" % i] = p["fri"][i]["root"]
        z["s%d_vals" % i] = [p["fri"][i]["polQueries"][q][0] for q in range(nQueries)]
        z["s%d_siblings" % i] = [p["fri"][i]["polQueries"][q][1] for q in range(nQueries)]
    stages = [1] + [s for s in range(2, nStages + 1) if starkInfo["mapSectionsN"].get("cm%d" % s, 0) > 0]
    # the reference creates the fields in this order (proof2zkin.js:31-47) and JSON.stringify keeps it in the zkin file
    for kind in ("vals", "siblings"):
        z["s0_%sC" % kind] = []
        for s in stages + [qStage]:
            z["s0_%s%d" % (kind, s)] = []
    for i in range(nQueries):
        query = p["fri"][0]["polQueries"][i]
        for s in stages:
            z["s0_vals%d" % s].append(query[s - 1][0]); z["s0_siblings%d" % s].append(query[s - 1][1])
        z["s0_vals%d" % qStage].append(query[nStages][0]); z["s0_siblings%d" % qStage].append(query[nStages][1])
        z["s0_valsC"].append(query[nStages + 1][0]); z["s0_siblingsC"].append(query[nStages + 1][1])
    z["finalPol"] = p["fri"][len(friSteps)]
    if starkInfo.get("nSubproofValues", 0) > 0:
        z["subproofValues"] = p["subproofValues"]
    return z


def zkin2proof(z, starkInfo):
    """the inverse of proof2zkin (src/proof2zkin.js:1-75): a zkin object (e.g. a *.proof.zkin.json the reference prover wrote,
    decimal strings or integers) -> the proof object starkVerify takes (stark_verify.js:8: proof.root1.., proof.evals,
    proof.fri[0].polQueries[q] = [[vals, siblings] per committed stage, quotient stage, constant tree], proof.fri[k] =
    {root, polQueries[q] = [vals, siblings]}, proof.fri[last] = the final polynomial)"""
    def ints(v):
        if isinstance(v, (list, tuple)):
            return [ints(x) for x in v]
        return int(v)
    friSteps = starkInfo["starkStruct"]["steps"]
    nQueries = starkInfo["starkStruct"]["nQueries"]
    nStages = starkInfo["nStages"]
    qStage = nStages + 1
    p = {"evals": ints(z["evals"])}
    for stage in range(1, qStage + 1):
        p["root%d" % stage] = ints(z["root%d" % stage])
    stages = [1] + [s for s in range(2, nStages + 1) if starkInfo["mapSectionsN"].get("cm%d" % s, 0) > 0]
    q0 = []
    for i in range(nQueries):
        query = [[[], []] for _ in range(nStages + 2)]
        for s in stages:
            query[s - 1] = [ints(z["s0_vals%d" % s][i]), ints(z["s0_siblings%d" % s][i])]
        query[nStages] = [ints(z["s0_vals%d" % qStage][i]), ints(z["s0_siblings%d" % qStage][i])]
        query[nStages + 1] = [ints(z["s0_valsC"][i]), ints(z["s0_siblingsC"][i])]
        q0.append(query)
    p["fri"] = [{"polQueries": q0}]
    for k in range(1, len(friSteps)):
        p["fri"].append({"root": ints(z["s%d_root" % k]),
                         "polQueries": [[ints(z["s%d_vals" % k][q]), ints(z["s%d_siblings" % k][q])] for q in range(nQueries)]})
    p["fri"].append(ints(z["finalPol"]))
    if "subproofValues" in z:
        p["subproofValues"] = ints(z["subproofValues"])
    return p


def _strings(v):
    if isinstance(v, dict):
        return {k: _strings(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_strings(x) for x in v]
    if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
        return str(int(v))
    return v


def zkin_json(zkin, publics=None):
    """the zkin file as main_prover.js:141-148 writes it: every field element a decimal string"""
    z = dict(zkin)
    if publics is not None:
        z["publics"] = list(publics)
    return json.dumps(_strings(z), indent=1)
