set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
for v in product w4; do
  if [ $v = product ]; then unset PIL2GL_LIB; else export PIL2GL_LIB=$L/libpil2gl_$v.so; fi
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/${v}_tcc -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/${v}_tcc.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/${v}_ea -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/${v}_ea.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/${v}_sq -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/${v}_sq.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $O/${v}_tcp -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/${v}_tcp.log 2>&1
done
cd $O && for f in */*counter_collection.csv; do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    if "bn_linear_hash" in k or "bn_merkle_level" in k:
        print(k, {n: "%.4g" % v for n, v in c.items()})
PY
done > $O/summary.txt 2>&1
echo done
