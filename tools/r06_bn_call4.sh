set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn4; mkdir -p $O
cd $R
{
for v in "" noscr notile noscrtile noscr4 noscrtile4 ""; do
  echo "== bench ${v:-product}"
  for i in 1 2; do if [ -z "$v" ]; then python3 tools/bench_bn128.py 20 100 16 | tail -n 1; else PIL2GL_LIB=$L/libpil2gl_$v.so python3 tools/bench_bn128.py 20 100 16 | tail -n 1; fi; done
done
} > $O/log.txt 2>&1
echo done
