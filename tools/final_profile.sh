set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/final_profile.sh [tag]}"
# The round's closing measurements in ONE call (one box): the driver's default line (with other_configs), the same under rocprofv3 --stats,
# config 4 commit + proof, config 2, the two-stage AIR, the rank-share rehearsal, BN254 counters and phase stamps.
cd /tmp && export TMPDIR=/tmp
tag=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$tag; mkdir -p $O; L=$R/pil2-stark-js_amd/lib_ab
python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err || exit 1
PIL2GL_BENCH_NODE=0 PIL2GL_BENCH_OTHER=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench_rocprof.err || exit 1
python3 $R/bench.py --air perm --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_perm.json 2> $O/bench_perm.err
python3 $R/bench.py --workload c4 --steps 2 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
python3 $R/bench.py --workload c4 --mode prove --steps 2 --warmup 1 > $O/bench_c4_prove.json 2> $O/bench_c4_prove.err
PIL2GL_BENCH_NODE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o bench -- python3 $R/bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c4_rocprof.json 2> $O/bench_c4_rocprof.err || true
python3 $R/bench.py --workload c3 --shard-of 8 --steps 5 --warmup 2 > $O/c3_shard.json 2> $O/c3_shard.err
python3 $R/bench.py --workload c5 --shard-of 8 --steps 2 --warmup 1 > $O/c5_shard.json 2> $O/c5_shard.err
python3 $R/tools/bench_bn128.py 20 100 16 2>&1 | tail -n 1 > $O/bn_2p20.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/pmc_tcc.log 2>&1
[ -f $L/libpil2gl_stamps.so ] && PIL2GL_LIB=$L/libpil2gl_stamps.so python3 $R/tools/bn_stamps.py 20 > $O/bn_stamps.txt 2>&1
python3 $R/tools/power_probe.py bn 23 > $O/power_bn.txt 2>&1
echo done
