#!/bin/bash
# Timing-only ablation of the tile kernel's LDS-DMA staging (conv_args.h abl bits 4-5; results are wrong by construction):
#   v0 production | 0x400000 half of the staging instructions, the two waves of a SIMD issue theirs in different phases
#   | 0x800000 half of them, all waves in the same phase | 0xc00000 no staging at all (the MFMA + ds_read + barrier loop alone).
# usage (repo root, through gpurun): tools/exp/dma_phase_ablation.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_dma_phase_ablation.txt
cd $R
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh -DU2_TILE_DMA_ABLATION > /dev/null 2>&1 ) || exit 1
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
: > $OUT
for L in "gemm 8192" "p2 3x3 256->256 200x336" "p3 3x3 256->256" "res4 1x1 1024->256 plain" "lat2 1x1 256->256"; do
  U2_BENCH_LAYERS="$L" tests/native/selftest bench2 0x10001000 0x10401000 0x10801000 0x10c01000 0x10002000 0x10402000 0x10c02000 | grep LAYER >> $OUT
done
( cd u2seg_amd/csrc && touch conv_tile.hip && ./build.sh > /dev/null 2>&1 )
cat $OUT
