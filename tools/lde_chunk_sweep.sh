# column chunks of 15 / 16 slots with a ragged last chunk (fixed-geometry kernels) against evenly cut chunks (PIL2GL_NTT_EVEN_CHUNKS=1: the any-geometry kernels)
for spec in "24 81 8" "24 18 8" "24 50 8" "24 33 8" "24 20 8" "24 27 8" "21 81 8" "20 50 8" "21 20 8" "24 81 1"; do set -- $spec
  for ev in 1 0; do NBITS=$1 NCOLS=$2 COSETS=$3 PIL2GL_NTT_EVEN_CHUNKS=$ev timeout -k 10 300 python tools/probe_lde_cosets.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$([ $ev = 1 ] && echo "even  " || echo "15\/16 ") | /"; done; done
