#!/bin/bash
# usage (repo root, through gpurun): tools/exp/km_trace.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
cd $R
( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh -DU2_KM_TRACE > /dev/null 2>&1 )
PYTHONPATH=$R python tools/exp/km_trace.py > $R/gpurun_out/${TAG}_km_trace.txt 2>&1
( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh > /dev/null 2>&1 )
cat $R/gpurun_out/${TAG}_km_trace.txt
