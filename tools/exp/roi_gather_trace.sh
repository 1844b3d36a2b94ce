#!/bin/bash
# Phase times of roi_align_bwd_gather_kernel (debug build -DGS_TRACE) on the synthetic sets of tools/exp/roi_gather_bench.py;
# the production library is rebuilt afterwards.  usage (repo root, through gpurun): tools/exp/roi_gather_trace.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/u2seg_amd/csrc && touch roi.hip && ./build.sh -DGS_TRACE > /dev/null 2>&1
cd $R && python tools/exp/roi_gather_bench.py
cd $R/u2seg_amd/csrc && touch roi.hip && ./build.sh > /dev/null 2>&1
