#!/bin/bash
# A/B of the wave-role split of the LDS-DMA staging (conv_tile.hip, -DU2_TILE_ROLES=1: in the eight-wave work-groups waves 0-3 stage the weights, waves 4-7 the pixels).
# usage (repo root, through gpurun): tools/exp/tile_roles_ab.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_tile_roles.txt
cd $R
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
LAYERS=("p2 3x3 256->256 200x336" "p3 3x3 256->256" "p4 3x3 256->256 50x84" "res4 1x1 1024->256 50x84" "res4 1x1 256->1024 50x84" "lat2 1x1 256->256" "lat3 1x1 512->256" "res5 3x3 512->512" "fc1 fwd" "fc1 dgrad" "gemm 8192" "mask 3x3")
run() { for L in "${LAYERS[@]}"; do U2_BENCH_LAYERS="$L" tests/native/selftest bench2 0 0x1000 0x2000 0x4000 | grep LAYER; done; }
echo "# production" > $OUT; run >> $OUT
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh -DU2_TILE_ROLES=1 > /dev/null 2>&1 )
tests/native/selftest > /tmp/st.log 2>&1; grep -c "^FAIL" /tmp/st.log >> $OUT; grep "^FAIL" /tmp/st.log | head -20 >> $OUT; tail -1 /tmp/st.log >> $OUT
for c in 1 2 3 4; do tests/native/selftest tile $c 0x8000000 > /tmp/st2.log 2>&1; tail -1 /tmp/st2.log >> $OUT; done
echo "# -DU2_TILE_ROLES=1" >> $OUT; run >> $OUT
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh > /dev/null 2>&1 )
echo "# production again" >> $OUT; run >> $OUT
cat $OUT
