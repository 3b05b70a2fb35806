# Where the config-3 interpolate's time goes: builds the NTT translation unit with parts of its arithmetic replaced by
# (wrong) free stand-ins -- gl_fermat.cuh's FERMAT_TIMING bits, NTT_MUL as an xor -- and times each build in ONE gpurun call.
# Run the build half here (needs hipcc), the timing half on the GPU box:  bash tools/lde_cost_split.sh build | run
set -e
R=$(cd $(dirname $0)/.. && pwd); P=$R/pil2-stark-js_amd; L=$P/lib_ab
if [ "$1" = build ]; then
  mkdir -p $L; F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -I$P/build"
  OBJS=$(ls $P/build/*.o | grep -v "/ntt")
  b() { /opt/rocm/bin/hipcc $F "${@:2}" -c $P/csrc/ntt.hip -o $L/ntt_$1.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libpil2gl_$1.so $L/ntt_$1.o $OBJS -L/opt/rocm/lib -lhiprtc && rm $L/ntt_$1.o; }
  b nocarry -DFERMAT_TIMING=1; b noshift -DFERMAT_TIMING=2; b noreduce -DFERMAT_TIMING=4; b nomul '-DNTT_MUL(a,b)=((a)^(b))'
  b skeleton -DFERMAT_TIMING=7 '-DNTT_MUL(a,b)=((a)^(b))'
else
  python3 $R/tools/probe_lde_time.py
  for n in nocarry noshift noreduce nomul skeleton; do PIL2GL_LIB=$L/libpil2gl_$n.so python3 $R/tools/probe_lde_time.py; done
fi
