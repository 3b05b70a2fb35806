set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box: VARIANTS='a b' gpurun -- bash tools/bn_ab.sh}"
# A/B of BN254 builds inside ONE call (boxes differ by 3-4 %): for every variant lib_ab/libpil2gl_<v>.so (tools/build_variant.sh) the width check against the
# oracle, the 2^20 x 100 arity-16 commit twice over two rounds, and the leaf launch's L2 / fabric request counters.  -> gpurun_out/bn_ab/log.txt
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/bn_ab; mkdir -p $O
cd $R
VARIANTS="${VARIANTS:?names of lib_ab builds}"
{
for v in $VARIANTS; do echo "== check $v"; PIL2GL_LIB=$L/libpil2gl_$v.so timeout 300 python3 tools/check_bn_mfma.py 2>&1 | tail -n 1; done
for v in $VARIANTS $VARIANTS; do
  echo "== bench $v"
  for i in 1 2; do PIL2GL_LIB=$L/libpil2gl_$v.so python3 tools/bench_bn128.py 20 100 16 | tail -n 1; done
done
cd /tmp && export TMPDIR=/tmp
for v in $VARIANTS; do
  PIL2GL_LIB=$L/libpil2gl_$v.so rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc_$v -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/tcc_$v.log 2>&1
done
cd $O && python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("tcc_*/p_counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "bn_linear_hash" in r["Kernel_Name"] and r["Grid_Size"] == "1048576":
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[0], {k: "%.4g" % v for k, v in acc.items()})
PY
} > $O/log.txt 2>&1
echo done
