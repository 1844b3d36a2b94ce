#!/bin/bash
# Sub-step timeline of the persistent tile kernel's K loop (conv_tile.hip, -DU2_TILE_TRACE -DU2_TILE_TRACE_POINT=k, one build per point):
# cycles from the step's barrier to each of six points of the step, for the two waves of one SIMD (waves 0 and 4).
# usage (repo root, through gpurun): tools/exp/tile_substep.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_tile_substep.txt
cd $R
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
: > $OUT
for K in 1 2 3 4 5 6; do
  ( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh -DU2_TILE_TRACE -DU2_TILE_TRACE_POINT=$K > /dev/null 2>&1 ) || exit 1
  for L in "gemm 8192" "p2 3x3 256->256 200x336"; do
    echo "=== point $K, $L (configuration 1, whole tiles for the 3x3 layer / stream-K for the GEMM)" >> $OUT
    U2_TILE_TRACE_EVERY=8 U2_BENCH_LAYERS="$L" tests/native/selftest bench2 0x1000 2>&1 | grep -E "^SUB|LAYER" | head -9 >> $OUT
  done
done
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh > /dev/null 2>&1 )
cat $OUT
