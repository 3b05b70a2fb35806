set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; L=$R/pil2-stark-js_amd/lib_ab; O=$R/gpurun_out/r06_bn17; mkdir -p $O
cd $R
{
echo "== check product"; timeout 600 python3 tools/check_bn_mfma.py 2>&1 | tail -n 2
echo "== bench product"; for i in 1 2 3; do python3 tools/bench_bn128.py 20 100 16 | tail -n 1; done
echo "== bn tests"; timeout 1200 python3 -m pytest tests/test_gpu_bn128.py tests/test_stark_prove.py tests/test_reference_proof.py -x -q -m gpu -k "bn128 or BN128 or config4 or final" 2>&1 | tail -n 3
echo "== stamps"; PIL2GL_LIB=$L/libpil2gl_stamps.so python3 tools/bn_stamps.py 20
echo "== power"; python3 tools/power_probe.py bn 23 2>&1 | grep -v "^hwmon:\|^idle:" | awk '/kernel bn/ {p=1} p' > $O/power_all.txt; head -1 $O/power_all.txt
echo "== c4"; timeout 600 python3 bench.py --workload c4 --steps 2 --warmup 1 2>&1 | tail -n 1 > $O/bench_c4.json; python3 -c "
import json; d=json.loads(open('$O/bench_c4.json').read()); print(d['ms_per_step'], d['kernels'][0], d['roofline_int_issue']['frac'])"
echo "== c4 prove"; timeout 600 python3 bench.py --workload c4 --mode prove --steps 2 --warmup 1 2>&1 | tail -n 1 > $O/bench_c4_prove.json; python3 -c "
import json; d=json.loads(open('$O/bench_c4_prove.json').read()); print(d['ms_per_step'], d['prove']['stages_s'])"
} > $O/log.txt 2>&1
echo done
