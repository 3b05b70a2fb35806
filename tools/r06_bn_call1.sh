set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn1; mkdir -p $O
cd $R
{
echo "== baseline"; for i in 1 2; do python3 tools/bench_bn128.py 20 100 16 | tail -n 1; done
echo "== prio3"; for i in 1 2; do PIL2GL_LIB=$L/libpil2gl_prio3.so python3 tools/bench_bn128.py 20 100 16 | tail -n 1; done
echo "== baseline again"; python3 tools/bench_bn128.py 20 100 16 | tail -n 1
echo "== stamps baseline"; PIL2GL_LIB=$L/libpil2gl_stamps.so python3 tools/bn_stamps.py 20
echo "== stamps prio3"; PIL2GL_LIB=$L/libpil2gl_prio3s.so python3 tools/bn_stamps.py 20
echo "== power bn"; python3 tools/power_probe.py bn 23
echo "== power gl"; python3 tools/power_probe.py gl 24
echo "== power ntt"; python3 tools/power_probe.py ntt 24
echo "== rocm-smi"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40
} > $O/log.txt 2>&1
echo done
