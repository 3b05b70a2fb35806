#!/bin/bash
set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/profile_round.sh}"
# The rocprofv3 passes the round's profiles/ files come from (run on the GPU box: gpurun -- bash tools/profile_round.sh r03).
#  1. kernel trace + stats of the default bench line (config-3 proof)        -> gpurun_out/prof_<tag>/stats
#  2. PMC passes over one config-3 commit (interpolate + leaf hash + tree) and the leaf-hash probe, each counter set in its
#     own run with --kernel-trace only: FETCH_SIZE, WRITE_SIZE, the memory-side request counters by size
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
PIL2GL_BENCH_NODE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench_rocprof.err || exit 1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  PIL2GL_BENCH_FROM_HOST=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o p -- python3 $R/tools/probe_prove.py > $O/$c.log 2>&1 || exit 1
  echo "$c done"
done
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/RDREQ -o p -- python3 $R/tools/probe_prove.py > $O/RDREQ.log 2>&1
echo "RDREQ done"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/WRREQ -o p -- python3 $R/tools/probe_prove.py > $O/WRREQ.log 2>&1
echo "WRREQ done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/SQ -o p -- python3 $R/tools/probe_prove.py > $O/SQ.log 2>&1
echo "SQ done"
