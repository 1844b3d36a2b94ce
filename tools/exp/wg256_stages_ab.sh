#!/bin/bash
# LDS ring depth of conv_wgrad256_kernel (-DU2_WG256_STAGES=3|4|5) on fc1 and the stride-16 / 32 1x1 layers.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest || exit 1
run() { for L in "fc1 fwd" "res4 1x1 256->1024 50x84" "res4 1x1 1024->256 50x84" "lat3" "gemm 8192"; do U2_BENCH_LAYERS="$L" tests/native/selftest bench2w 0 256 | grep LAYER | cut -c1-200; done; }
for D in 3 4 5 3; do
  ( cd u2seg_amd/csrc && touch conv_igemm.hip && ./build.sh -DU2_WG256_STAGES=$D > /dev/null 2>&1 )
  echo "# U2_WG256_STAGES=$D"; run
done
( cd u2seg_amd/csrc && touch conv_igemm.hip && ./build.sh > /dev/null 2>&1 )
