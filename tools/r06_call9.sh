set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_9; mkdir -p $O
cd $R
echo "== tests" > $O/log.txt
timeout 1500 python3 -m pytest tests/test_stark_prove.py -x -q -m gpu -k "bn128 or config4" >> $O/log.txt 2>&1
echo "== c4 prove" >> $O/log.txt
timeout 900 python3 bench.py --workload c4 --mode prove --steps 2 --warmup 1 > $O/c4_prove.json 2>> $O/log.txt
echo "== default bench" >> $O/log.txt
( time timeout 1200 python3 bench.py --steps 5 --warmup 1 > $O/bench_default.json 2>> $O/log.txt ) >> $O/log.txt 2>&1
echo done >> $O/log.txt
