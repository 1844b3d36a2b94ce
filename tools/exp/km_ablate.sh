#!/bin/bash
# Timing-only ablations of the coarse kernel's loop (-DU2_KC_ABL bits: 1 no MFMAs, 2 no centroid-fragment reads behind the first two groups,
# 4 no LDS-DMA requests in the loop, 8 no barrier) through the work-group trace.  usage: tools/exp/km_ablate.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_km_ablate.txt
cd $R
: > $OUT
for A in ${KC_ABL_LIST:-0 1 2 4 8 3 5 6 7}; do
  ( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh -DU2_KM_TRACE -DU2_KC_ABL=$A > /dev/null 2>&1 )
  echo "# U2_KC_ABL=$A" >> $OUT
  PYTHONPATH=$R python tools/exp/km_trace.py 2>&1 | grep -v amdgpu.ids | grep -v "raw span" >> $OUT
done
( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh > /dev/null 2>&1 )
cat $OUT
