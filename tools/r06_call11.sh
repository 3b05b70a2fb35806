set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_11; mkdir -p $O
cd $R
echo "== full gpu suite" > $O/log.txt
( time timeout 1500 python3 -m pytest tests -x -q -m gpu ) >> $O/log.txt 2>&1
echo "== pmc bn (2^20 x 100)" >> $O/log.txt
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_bn128.py 20 100 16 2>&1 | tail -n 1 >> $O/log.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/pmc_tcc.log 2>&1
echo "== c4 under rocprofv3 --stats" >> $O/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o bench -- python3 $R/bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c4_rocprof.json 2> $O/bench_c4_rocprof.err
python3 $R/bench.py --workload c4 --steps 2 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
echo done >> $O/log.txt
