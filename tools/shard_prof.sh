cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/shard_prof; mkdir -p $O
PIL2GL_BENCH_NODE=0 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/line.json 2> $O/err.txt
echo done
