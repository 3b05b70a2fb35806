set -eu; : "${GRAFT_REPO_ROOT:?run on the GPU box: gpurun -- bash tools/pmc_bn128.sh}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bn; mkdir -p $O
# (lib_ab/libpil2gl_l17.so: an A/B build of round 3 with BN_LDS_ELEMS=17, i.e. the whole state in LDS: rebuild it with
#  hipcc ... -DBN_LDS_ELEMS=17 -c csrc/bn128.hip and link as tools/lde_cost_split.sh links its variants; skipped when absent)
for l in "" $R/pil2-stark-js_amd/lib_ab/libpil2gl_l17.so; do
  [ -z "$l" ] || [ -f "$l" ] || continue
  PIL2GL_LIB=$l python3 $R/tools/bench_bn128.py 20 100 16 2>&1 | tail -n 1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/a -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/a.log 2>&1
PIL2GL_LIB=$R/pil2-stark-js_amd/lib_ab/libpil2gl_l17.so rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/tools/bench_bn128.py 20 100 16 > $O/b.log 2>&1
echo done
