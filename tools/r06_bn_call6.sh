set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn6; mkdir -p $O
cd $R
{
echo "== check product"; timeout 600 python3 tools/check_bn_mfma.py 2>&1 | tail -n 18
for v in oldsbox "" lds9 w4 ""; do
  echo "== bench ${v:-product}"
  for i in 1 2; do if [ -z "$v" ]; then python3 tools/bench_bn128.py 20 100 16 | tail -n 1; else PIL2GL_LIB=$L/libpil2gl_$v.so python3 tools/bench_bn128.py 20 100 16 | tail -n 1; fi; done
done
echo "== stamps"; PIL2GL_LIB=$L/libpil2gl_stamps.so python3 tools/bn_stamps.py 20
} > $O/log.txt 2>&1
echo done
