set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn2; mkdir -p $O
cd $R
{
echo "== check product"; timeout 300 python3 tools/check_bn_mfma.py 2>&1 | tail -n 3
echo "== check w4"; PIL2GL_LIB=$L/libpil2gl_w4.so timeout 300 python3 tools/check_bn_mfma.py 2>&1 | tail -n 3
echo "== check w8"; PIL2GL_LIB=$L/libpil2gl_w8.so timeout 300 python3 tools/check_bn_mfma.py 2>&1 | tail -n 3
for v in oldsbox "" w2 w4 w8 oldsbox ""; do
  echo "== bench ${v:-product}"
  for i in 1 2; do if [ -z "$v" ]; then python3 tools/bench_bn128.py 20 100 16 | tail -n 1; else PIL2GL_LIB=$L/libpil2gl_$v.so python3 tools/bench_bn128.py 20 100 16 | tail -n 1; fi; done
done
echo "== stamps product"; PIL2GL_LIB=$L/libpil2gl_stamps.so python3 tools/bn_stamps.py 20
echo "== stamps w4"; PIL2GL_LIB=$L/libpil2gl_w4s.so python3 tools/bn_stamps.py 20
} > $O/log.txt 2>&1
echo done
