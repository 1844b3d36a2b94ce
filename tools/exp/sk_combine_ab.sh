#!/bin/bash
# A/B of the stream-K combine (conv_tile.hip sk_combine): loads in flight per thread per round trip, U2_SKB = 16 (round 6) vs 4 (round 5).
# usage (repo root, through gpurun): tools/exp/sk_combine_ab.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_sk_combine.txt
cd $R
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
LAYERS=("res4 1x1 1024->256 50x84" "res4 1x1 256->1024 50x84" "res5 1x1 2048->512" "res5 1x1 512->2048" "p4 3x3 256->256 50x84" "p5 3x3 256->256" "res5 3x3 512->512" "fc1 fwd" "fc1 dgrad" "p3 3x3 256->256" "mask 3x3")
run() {
  for L in "${LAYERS[@]}"; do U2_BENCH_LAYERS="$L" tests/native/selftest bench2 0 | grep LAYER; done
}
echo "# U2_SKB=16 (two round trips per peer)" > $OUT; run >> $OUT
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh -DU2_SKB=4 > /dev/null 2>&1 )
echo "# U2_SKB=4 (eight round trips per peer: the round-5 form)" >> $OUT; run >> $OUT
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh > /dev/null 2>&1 )
echo "# U2_SKB=16 again" >> $OUT; run >> $OUT
cat $OUT
