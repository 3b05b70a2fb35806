# Two sweeps against three, with the kernels that exist: at 2^20 rows the any-geometry instances can run the forward half of the LDE as
# mid (10 stages) + ONE pass of 10 stages on 1024-row x 16-slot tiles (128-byte pieces, 1024-thread workgroups, one per CU) instead of
# mid (7) + passes of 7 and 6 stages.  Same arithmetic per stage, one sweep of the extension fewer.  bash tools/lde_two_sweep.sh build | run
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/pil2-stark-js_amd; L=$P/lib_ab
if [ "${1:-run}" = build ]; then
  mkdir -p "$L"; F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -I$P/build"
  OBJS=$(ls $P/build/*.o | grep -v "/ntt")
  /opt/rocm/bin/hipcc $F -DNTT_MAXTHREADS=1024 -DLDE_MAXTHREADS=1024 -c $P/csrc/ntt.hip -o $L/ntt_big.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libpil2gl_big.so $L/ntt_big.o $OBJS -L/opt/rocm/lib -lhiprtc && rm $L/ntt_big.o
else
  export NBITS=20
  TAG="3 sweeps (7+7+6), fixed/any geometry as shipped" python3 $R/tools/probe_lde_time.py
  TAG="3 sweeps (7+7+6), any-geometry" PIL2GL_NTT_GENERIC=1 python3 $R/tools/probe_lde_time.py
  export PIL2GL_LIB=$L/libpil2gl_big.so PIL2GL_NTT_GENERIC=1
  TAG="3 sweeps, 1024-thread build" python3 $R/tools/probe_lde_time.py
  TAG="2 sweeps (10+10), 1024 threads" PIL2GL_NTT_KMAX=10 PIL2GL_NTT_TILE=16384 PIL2GL_LDE_TILE=16384 PIL2GL_NTT_THREADS=1024 PIL2GL_LDE_THREADS=1024 python3 $R/tools/probe_lde_time.py
  TAG="2 sweeps (10+10), 512 threads" PIL2GL_NTT_KMAX=10 PIL2GL_NTT_TILE=16384 PIL2GL_LDE_TILE=16384 PIL2GL_NTT_THREADS=512 PIL2GL_LDE_THREADS=512 python3 $R/tools/probe_lde_time.py
  TAG="2 sweeps (10+10), 8-slot tiles, 1024 threads" PIL2GL_NTT_KMAX=10 PIL2GL_NTT_TILE=8192 PIL2GL_LDE_TILE=8192 PIL2GL_NTT_THREADS=1024 PIL2GL_LDE_THREADS=1024 python3 $R/tools/probe_lde_time.py
  TAG="3 sweeps of 9 stages max (7+7+6 -> 9: mid 7)" PIL2GL_NTT_KMAX=9 PIL2GL_NTT_TILE=8192 PIL2GL_LDE_TILE=8192 PIL2GL_NTT_THREADS=512 PIL2GL_LDE_THREADS=512 python3 $R/tools/probe_lde_time.py
fi
