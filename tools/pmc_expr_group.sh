# HBM-side read requests (by size) and L2 hits / misses of the run-time compiled constraint kernel (jit_eval) at config 3 with its
# HISTORICAL (the switch was removed after this measurement: build commit a1d7ce2 and point PIL2GL_LIB at it to repeat):
# section loads issued at first use (PIL2GL_EXPR_GROUP=0) and 8 / 16 columns at a time: gpurun -- bash tools/pmc_expr_group.sh
set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_expr_group.sh}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_expr_group; mkdir -p $O
for g in 0 8 16; do
  export GROUPS=$g
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/a$g -o p -- python3 $R/tools/probe_expr_group.py > $O/a$g.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $O/b$g -o p -- python3 $R/tools/probe_expr_group.py > $O/b$g.log 2>&1 || echo "L2 counters unavailable for g=$g"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/c$g -o p -- python3 $R/tools/probe_expr_group.py > $O/c$g.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for g in ("0", "8", "16"):
    acc = collections.defaultdict(list)
    for sub in "abc":
        for f in glob.glob("$O/%s%s/**/*counter_collection.csv" % (sub, g), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("jit_eval"):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = {k: max(v) for k, v in sorted(acc.items())}
    if "TCC_EA0_RDREQ_sum" in d:
        d["read_GB"] = (128 * d["TCC_EA0_RDREQ_128B_sum"] + 32 * d["TCC_EA0_RDREQ_32B_sum"] + 64 * (d["TCC_EA0_RDREQ_sum"] - d["TCC_EA0_RDREQ_128B_sum"] - d["TCC_EA0_RDREQ_32B_sum"])) / 1e9
    print("group=%s" % g, {k: "%.4g" % v for k, v in d.items()})
PY
