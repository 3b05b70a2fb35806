set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_fuzz; mkdir -p $O
cd $R
FUZZ_ONLY=bn128,proof_bn128 timeout 900 python3 tests/fuzz/fuzz_parity.py 420 606 > $O/fuzz_bn128.txt 2>&1
echo "rc $?" >> $O/fuzz_bn128.txt
echo done
