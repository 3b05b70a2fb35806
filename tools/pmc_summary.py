#!/usr/bin/env python3
"""Summarises rocprofv3 PMC passes into profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
  python tools/pmc_summary.py <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run> [out.json]
Each dir holds */*_counter_collection.csv.  Per kernel: number of launches, mean and max counter value (KB); HBM bytes per
launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 for the largest launch (gfx950: FETCH_SIZE counts 64-byte units as 32,
i.e. half the fetched bytes -- MI355X_MICROARCH.md HBM section; confirmed on linear_hash_kernel where 2*FETCH equals the
8*E*C algorithmic read)."""
import csv
import glob
import json
import re
import sys


def collect(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            k = re.sub(r"^void\s+", "", k)
            k = re.split(r"[(<]", k)[0].strip()
            out.setdefault(k, []).append(float(row["Counter_Value"]))
    return out


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    res = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = "
                       "(2*FETCH_SIZE + WRITE_SIZE)*1024 of the largest launch of each kernel (gfx950 FETCH_SIZE correction, see tools/pmc_summary.py)",
           "_raw_KB": {}, "_sum_bytes": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        res["_raw_KB"][k] = {"FETCH_SIZE": [len(f), sum(f) / len(f), max(f)], "WRITE_SIZE": [len(w), sum(w) / len(w), max(w)]}
        res[k] = int((2 * max(f) + max(w)) * 1024)
        res["_sum_bytes"][k] = {"launches": max(len(f), len(w)), "bytes": int((2 * sum(f) + sum(w)) * 1024)}
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
