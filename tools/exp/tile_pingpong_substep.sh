#!/bin/bash
# Sub-step timeline of the PING-PONG K step (conv_tile.hip, -DU2_TILE_PINGPONG=1 -DU2_TILE_TRACE -DU2_TILE_TRACE_POINT=11..18): anchor = barrier #1
# (behind half A); 11 = in front of the wait for half A's fragments, 12 = behind it, 13 = behind half A's 16 MFMAs, 14 = in front of the vmcnt
# ladder, 15 = in front of the wait for half B's weights, 16 = behind it, 17 = behind half B's 16 MFMAs, 18 = in front of barrier #2.
# waves 0 (group X: stages first) and 4 (group Y: multiplies first) of one SIMD.   usage: tools/exp/tile_pingpong_substep.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_tile_pingpong_substep.txt
cd $R
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
: > $OUT
for K in 11 12 13 14 15 16 17 18; do
  ( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh -DU2_TILE_PINGPONG=1 -DU2_TILE_TRACE -DU2_TILE_TRACE_POINT=$K > /dev/null 2>&1 ) || exit 1
  echo "=== point $K, gemm 8192 (configuration 1)" >> $OUT
  U2_TILE_TRACE_EVERY=8 U2_BENCH_LAYERS="gemm 8192" tests/native/selftest bench2 0x1000 2>&1 | grep -E "^SUB|LAYER" | grep -E "wg  129|LAYER" | head -3 >> $OUT
done
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh > /dev/null 2>&1 )
cat $OUT
