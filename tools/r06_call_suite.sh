set -u; : "${GRAFT_REPO_ROOT:?run on the GPU box}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_suite; mkdir -p $O
cd $R
( time timeout 1500 python3 -m pytest tests -x -q -m gpu ) > $O/log.txt 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" >> $O/log.txt 2>&1
echo done >> $O/log.txt
