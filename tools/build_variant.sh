#!/bin/bash
# An A/B build of the library with extra flags on ONE translation unit:  tools/build_variant.sh NAME UNIT "-DFLAG=1 ..."
#   -> pil2-stark-js_amd/lib_ab/libpil2gl_NAME.so (select it with PIL2GL_LIB=...); the other objects come from the product build (make first).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/pil2-stark-js_amd
NAME=$1; UNIT=$2; FLAGS=${3:-}
mkdir -p $P/build_ab/$NAME $P/lib_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -I$P/build $FLAGS -c $P/csrc/$UNIT.hip -o $P/build_ab/$NAME/$UNIT.o
OBJS=""
for o in $P/build/*.o; do b=$(basename $o); if [ "$b" = "$UNIT.o" ]; then OBJS="$OBJS $P/build_ab/$NAME/$UNIT.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/lib_ab/libpil2gl_$NAME.so $OBJS -L/opt/rocm/lib -lhiprtc
echo built $P/lib_ab/libpil2gl_$NAME.so
