set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
# instruction-cache counters of the BN254 leaf kernel (2^20 x 100) and of the GL leaf hash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -i "icache\|ifetch\|SQC_" $O/avail.txt | head -80 > $O/avail_sqc.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p2 -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/p2.log 2>&1
cd $O && python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("p*/p_counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "bn_linear_hash" in r["Kernel_Name"] and r["Grid_Size"] == "1048576":
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[0], {k: "%.4g" % v for k, v in acc.items()})
PY
echo done
