# WRITE_SIZE / FETCH_SIZE of the split leaf kernel beside the plain one at config 3 (scratch traffic shows up as written bytes far above
# the 4.3 GB of digests: round 3's first blocked kernel wrote 133 GB): gpurun -- bash tools/pmc_leaf_split.sh
set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_leaf_split.sh}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_leaf_split; mkdir -p $O
export SPLIT=1
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o p -- python3 $R/tools/probe_leaf.py > $O/$c.log 2>&1
done
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_hash" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[1] if r["Kernel_Name"].startswith("(") else r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "KB per launch:", ["%.4g" % x for x in v])
PY
