set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_leaf.sh}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_EA_[A-Z0-9_]*" $R/gpurun_out/pmc/counters.txt | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc/tcc_ea.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc/$c -o p -- python3 $R/tools/probe_leaf.py > $R/gpurun_out/pmc/$c.log 2>&1 || exit 1
done
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $R/gpurun_out/pmc/RDREQ -o p -- python3 $R/tools/probe_leaf.py > $R/gpurun_out/pmc/RDREQ.log 2>&1
echo done
