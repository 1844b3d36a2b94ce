#!/bin/bash
# s_memtime phase timeline of the persistent tile kernel (conv_tile.hip built with -DU2_TILE_TRACE) on isolated layers.
# usage (repo root, through gpurun): tools/exp/tile_trace.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_tile_trace.txt
cd $R
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh -DU2_TILE_TRACE > /dev/null 2>&1 ) || exit 1
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
: > $OUT
for L in "res4 1x1 1024->256 plain" "res4 1x1 256->1024 plain" "res5 1x1 2048->512" "p4 3x3 256->256 50x84" "p2 3x3 256->256 200x336" "gemm 8192"; do
  for V in 0 0x10000000; do
    echo "=== $L variant $V" >> $OUT
    U2_TILE_TRACE_EVERY=8 U2_BENCH_LAYERS="$L" tests/native/selftest bench2 $V 2>&1 | grep -E "LAYER|TRACE|wg |medians" | head -40 >> $OUT
  done
done
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh > /dev/null 2>&1 )
cat $OUT | cut -c1-330
