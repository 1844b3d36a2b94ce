#!/bin/bash
# Weight-gradient variants on the mid-size 3x3 layers, fc1 and the stride-16 1x1 layers (tests/native/selftest bench2w).
# usage (repo root, through gpurun): tools/exp/wgrad_mid_ab.sh <tag> [variants...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}; shift
VARS=${@:-0 64 4096 0x5000}
OUT=$R/gpurun_out/${TAG}_wgrad_mid.txt
cd $R
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
LAYERS=("p3 3x3 256->256" "p4 3x3 256->256 50x84" "p5 3x3 256->256" "res3 3x3 128->128" "res5 3x3 512->512" "mask 3x3" "res2 3x3 64->64" "fc1 fwd" "res4 1x1 256->1024 50x84" "res4 1x1 1024->256 50x84" "res5 1x1 512->2048" "fc2 gemm")
echo "# variants: $VARS" > $OUT
for L in "${LAYERS[@]}"; do U2_BENCH_LAYERS="$L" tests/native/selftest bench2w $VARS | grep LAYER; done >> $OUT
cat $OUT
