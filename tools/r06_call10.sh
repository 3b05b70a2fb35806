set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_10; mkdir -p $O
cd $R
echo "== tests" > $O/log.txt
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_parallel.py -x -q -m gpu -k "quotient or shard or parallel or sharded or rank" >> $O/log.txt 2>&1
echo "== rehearsal c3" >> $O/log.txt
timeout 900 python3 bench.py --workload c3 --shard-of 8 --steps 5 --warmup 2 > $O/c3_shard.json 2>> $O/log.txt
timeout 900 python3 bench.py --workload c3 --shard-of 8 --steps 5 --warmup 2 > $O/c3_shard_b.json 2>> $O/log.txt
echo done >> $O/log.txt
