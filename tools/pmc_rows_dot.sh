# counters of rows_dot_mfma_kernel at config 3 (2^27 x 100, two outputs): what the 24 ms are made of.  gpurun -- bash tools/pmc_rows_dot.sh
set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_rows_dot.sh}"
cd /tmp && export TMPDIR=/tmp
export MODES=mfma      # (the column-tile kernel under the TCC counter pass did not finish in seven minutes: only the kernel in question)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_rows_dot; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/a -o p -- python3 $R/tools/probe_rows_dot.py > $O/a.log 2>&1 || echo "pass a failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/tools/probe_rows_dot.py > $O/b.log 2>&1 || echo "pass b failed"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/c -o p -- python3 $R/tools/probe_rows_dot.py > $O/c.log 2>&1 || echo "pass c failed"
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list); dur = []
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rows_dot_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rows_dot_mfma" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("launch ms:", ["%.2f" % d for d in dur][:6])
for k, v in sorted(acc.items()):
    print(k, "%.4g" % (sum(v) / len(v)))
PY
