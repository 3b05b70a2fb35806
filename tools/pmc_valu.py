"""VALU / wait fractions per kernel from a rocprofv3 --pmc pass holding SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_ANY,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_BUSY_CYCLES (summed over launches of each kernel)"""
import csv, glob, json, re, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void\s+", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        k = re.split(r"\(", k)[0].strip()
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0:
        continue
    out[k] = {n: v for n, v in c.items()}
    out[k]["valu_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / wc
    out[k]["any_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / wc
    out[k]["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / wc
    out[k]["wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / wc
    if c.get("SQ_BUSY_CYCLES"):
        out[k]["waves_per_busy_cycle"] = wc / c["SQ_BUSY_CYCLES"]
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in list(out.items())[:8]:
    print(k[:50], {n: round(x, 3) for n, x in v.items() if "frac" in n or "per" in n})
