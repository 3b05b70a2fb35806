#!/usr/bin/env python3
"""HBM-side bytes per kernel launch from rocprofv3 counter passes (tools/profile_round.sh), by REQUEST SIZE:
      read bytes  = 128 RDREQ_128B + 32 RDREQ_32B + 64 (RDREQ - RDREQ_128B - RDREQ_32B)
      write bytes =  64 WRREQ_64B + 32 (WRREQ - WRREQ_64B)
(TCC_EA0_* summed over the L2 channels).  FETCH_SIZE / WRITE_SIZE of the same launches are kept beside them: FETCH_SIZE tallies
every read request at 64 B, so it is half the bytes wherever the requests are 128-byte lines (MI355X_MICROARCH.md, HBM) --
which the size counters show to be the case for every large kernel of the proof, the row-strided leaf hashing included.
  python tools/pmc_summary2.py gpurun_out/prof_r03 profiles/r03_pmc_traffic.json"""
import csv, json, re, sys


def collect(path):
    out = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"^void\s+", "", k); k = re.split(r"[(<]", k)[0].strip()
        out.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    d, outp = sys.argv[1], sys.argv[2]
    acc = {}
    for sub in ("FETCH_SIZE", "WRITE_SIZE", "RDREQ", "WRREQ"):
        for k, cs in collect("%s/%s/p_counter_collection.csv" % (d, sub)).items():
            acc.setdefault(k, {}).update(cs)
    res = {"_comment": "HBM-side bytes per launch by request size (tools/pmc_summary2.py; passes: tools/profile_round.sh, one config-3 proof); "
                       "<kernel>: read + written bytes of its LARGEST launch; _per_kernel: launches, largest launch and sum over the proof, "
                       "FETCH_SIZE / WRITE_SIZE (KB) of the largest launch for comparison", "_per_kernel": {}}
    for k, c in sorted(acc.items()):
        if "TCC_EA0_RDREQ_sum" not in c or "TCC_EA0_WRREQ_sum" not in c:
            continue
        rd = [128 * a + 32 * b + 64 * (t - a - b) for t, a, b in zip(c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_RDREQ_128B_sum"], c["TCC_EA0_RDREQ_32B_sum"])]
        wr = [64 * a + 32 * (t - a) for t, a in zip(c["TCC_EA0_WRREQ_sum"], c["TCC_EA0_WRREQ_64B_sum"])]
        if max(rd) + max(wr) < 1e8:
            continue
        res[k] = int(max(rd) + max(wr))
        res["_per_kernel"][k] = {"launches": len(rd), "read_bytes_largest": int(max(rd)), "written_bytes_largest": int(max(wr)),
                                 "read_bytes_sum": int(sum(rd)), "written_bytes_sum": int(sum(wr)),
                                 "share_of_128B_read_requests": round(sum(c["TCC_EA0_RDREQ_128B_sum"]) / max(1.0, sum(c["TCC_EA0_RDREQ_sum"])), 4),
                                 "FETCH_SIZE_KB_largest": max(c.get("FETCH_SIZE", [0])), "WRITE_SIZE_KB_largest": max(c.get("WRITE_SIZE", [0]))}
    pk = res["_per_kernel"]
    # one interpolate of the stage-1 witness = the launches of the NTT family with the stage's widths: the two largest ntt passes
    # (forward, 107 GB each way), the mid kernel's largest launch, and the two inverse passes on the 13 GB coefficient matrix
    if "ntt_pass_kernel" in acc and "lde_mid_kernel" in acc:
        c = acc["ntt_pass_kernel"]
        rd = sorted((128 * a + 64 * (t - a) for t, a in zip(c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_RDREQ_128B_sum"])), reverse=True)
        wr = sorted((64 * a + 32 * (t - a) for t, a in zip(c["TCC_EA0_WRREQ_sum"], c["TCC_EA0_WRREQ_64B_sum"])), reverse=True)
        # forward passes: the two largest; inverse passes of the 100-column stage: next two by read volume above 10 GB
        fwd = rd[0] + rd[1] + wr[0] + wr[1]
        inv = sum(v for v in rd[2:6] if v > 1.0e10) + sum(v for v in wr[2:6] if v > 1.0e10)
        res["_interpolate_bytes"] = int(fwd + inv + pk["lde_mid_kernel"]["read_bytes_largest"] + pk["lde_mid_kernel"]["written_bytes_largest"])
        res["_interpolate_parts"] = {"forward_passes": int(fwd), "inverse_passes": int(inv), "lde_mid_kernel": int(pk["lde_mid_kernel"]["read_bytes_largest"] + pk["lde_mid_kernel"]["written_bytes_largest"])}
    json.dump(res, open(outp, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_per") and not k.startswith("_comment"):
            print(k, v)


if __name__ == "__main__":
    main()
