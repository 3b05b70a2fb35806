set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn7; mkdir -p $O
cd $R
{
echo "== check product"; timeout 600 python3 tools/check_bn_mfma.py 2>&1 | tail -n 3
for v in oldsbox "" lds9 w4 ""; do
  echo "== bench ${v:-product}"
  for i in 1 2; do if [ -z "$v" ]; then python3 tools/bench_bn128.py 20 100 16 | tail -n 1; else PIL2GL_LIB=$L/libpil2gl_$v.so python3 tools/bench_bn128.py 20 100 16 | tail -n 1; fi; done
done
echo "== stamps"; PIL2GL_LIB=$L/libpil2gl_stamps.so python3 tools/bn_stamps.py 20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/tcc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/sq.log 2>&1
} > $O/log.txt 2>&1
echo done
