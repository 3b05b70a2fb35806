# HISTORICAL (round 3; PIL2GL_EXPR_STAGE was removed in round 4): counters of the run-time compiled constraint kernel (jit_eval) at config 3, staged and direct reads: gpurun -- bash tools/pmc_expr.sh
set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_expr.sh}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_expr; mkdir -p $O
for st in 0 1; do
  export COMBOS="1,1,$st"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/a$st -o p -- python3 $R/tools/probe_expr_ab.py > $O/a$st.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/b$st -o p -- python3 $R/tools/probe_expr_ab.py > $O/b$st.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/c$st -o p -- python3 $R/tools/probe_expr_ab.py > $O/c$st.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for st in "01":
    acc = collections.defaultdict(list)
    for sub in "abc":
        for f in glob.glob("$O/%s%s/**/*counter_collection.csv" % (sub, st), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("jit_eval"):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("stage=%s" % st, {k: "%.4g" % max(v) for k, v in sorted(acc.items())})
PY
