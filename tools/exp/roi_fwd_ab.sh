#!/bin/bash
# ROIAlign forward variants (compile-time FS_QUAD / FS_OCC): per-launch durations in the serial training step and the inference rate.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for FL in "-DFS_QUAD=1" "-DFS_QUAD=0" "-DFS_QUAD=1 -DFS_OCC=6" "-DFS_QUAD=1"; do
  ( cd u2seg_amd/csrc && touch roi.hip && ./build.sh $FL > /dev/null 2>&1 )
  echo "# $FL"
  tools/exp/roi_trace.sh 2>&1 | grep fwd_sep | tail -4
  python bench.py --workload infer --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('infer', round(d['value'],1), 'img/s')"
done
( cd u2seg_amd/csrc && touch roi.hip && ./build.sh > /dev/null 2>&1 )
