# The 100-column kernels of the LDE (mid kernel, inverse passes) with 20 column slots x 5 chunks (no idle slot, 320 threads = five whole waves)
# instead of 15 x 7 (5 idle slots of 105, a quarter of the fourth wave idle): any-geometry instances, same call.  bash tools/lde_slots20.sh build | run
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/pil2-stark-js_amd; L=$P/lib_ab
if [ "${1:-run}" = build ]; then
  mkdir -p "$L"; F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -I$P/build"
  OBJS=$(ls $P/build/*.o | grep -v "/ntt")
  /opt/rocm/bin/hipcc $F -DNTT_MAXTHREADS=512 -DLDE_MAXTHREADS=512 -c $P/csrc/ntt.hip -o $L/ntt_t512.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libpil2gl_t512.so $L/ntt_t512.o $OBJS -L/opt/rocm/lib -lhiprtc && rm $L/ntt_t512.o
else
  export NBITS=${NBITS:-24}
  TAG="shipped (fixed geometry, 15 slots)" python3 $R/tools/probe_lde_time.py
  export PIL2GL_LIB=$L/libpil2gl_t512.so PIL2GL_NTT_GENERIC=1
  TAG="any-geometry, 15 slots x 7 chunks" python3 $R/tools/probe_lde_time.py
  TAG="any-geometry, 20 slots x 5 chunks, 320 threads" PIL2GL_NTT_TILE=5120 PIL2GL_LDE_TILE=5120 PIL2GL_NTT_THREADS=320 PIL2GL_LDE_THREADS=320 python3 $R/tools/probe_lde_time.py
  TAG="any-geometry, 25 slots x 4 chunks, 400 threads" PIL2GL_NTT_TILE=6400 PIL2GL_LDE_TILE=6400 PIL2GL_NTT_THREADS=400 PIL2GL_LDE_THREADS=400 python3 $R/tools/probe_lde_time.py
fi
