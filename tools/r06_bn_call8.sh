set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn8; mkdir -p $O
cd $R
{
echo "== check product"; timeout 600 python3 tools/check_bn_mfma.py 2>&1 | tail -n 3
for v in "" w1 ""; do
  echo "== bench ${v:-product}"
  for i in 1 2; do if [ -z "$v" ]; then python3 tools/bench_bn128.py 20 100 16 | tail -n 1; else PIL2GL_LIB=$L/libpil2gl_$v.so python3 tools/bench_bn128.py 20 100 16 | tail -n 1; fi; done
done
echo "== bn tests"; timeout 900 python3 -m pytest tests/test_gpu_bn128.py -x -q -m gpu 2>&1 | tail -n 5
echo "== c4"; timeout 600 python3 bench.py --workload c4 --steps 2 --warmup 1 2>&1 | tail -n 1
} > $O/log.txt 2>&1
echo done
