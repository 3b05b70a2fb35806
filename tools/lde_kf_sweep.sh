# the mid kernel at 8 bits (its fixed-geometry form; the planner's choice since round 5 where that form applies) against the balanced
# split (PIL2GL_LDE_KF=0): NBITS NCOLS COSETS
for spec in "17 100 8" "19 100 8" "20 100 8" "21 100 8" "21 64 8" "21 33 8" "21 20 8" "25 32 8" "26 16 8" "27 8 8" "21 100 1" "25 100 1" "26 100 1" "28 16 1"; do set -- $spec
  for kf in 0 -; do NBITS=$1 NCOLS=$2 COSETS=$3 $([ $kf = 0 ] && echo env PIL2GL_LDE_KF=0) timeout -k 10 300 python tools/probe_lde_cosets.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$([ $kf = 0 ] && echo balanced || echo planner ) | /"; done; done
