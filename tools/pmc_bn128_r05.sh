set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_bn128_r05.sh}"
# Where the cycles of the matrix-core BN254 leaf kernel go: wave cycles split into issuing / waiting on memory counters (s_waitcnt) /
# waiting to issue (dependencies, pipes busy), vector and matrix instruction counts.  Two passes (8 SQ counters each), counters only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bn5; mkdir -p $O
python3 $R/tools/bench_bn128.py 20 100 16 2>&1 | tail -n 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/a -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/b.log 2>&1 || true
python3 $R/tools/pmc_valu.py $O $R/gpurun_out/r05_bn128_pmc.json || true
echo done
