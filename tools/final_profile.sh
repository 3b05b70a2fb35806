set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/final_profile.sh}"
cd /tmp && export TMPDIR=/tmp
tag=${1:-r05}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$tag; mkdir -p $O
python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err || exit 1
PIL2GL_BENCH_NODE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench_rocprof.err || exit 1
python3 $R/bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python3 $R/bench.py --air perm --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_perm.json 2> $O/bench_perm.err
python3 $R/bench.py --workload c4 --steps 2 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
PIL2GL_BENCH_NODE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o bench -- python3 $R/bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c4_rocprof.json 2> $O/bench_c4_rocprof.err || true
echo done
