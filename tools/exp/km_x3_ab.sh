#!/bin/bash
# A/B of the k-means coarse pass with three x slots (kmeans.hip, default) vs the round-5 form (-DU2_KM_X3=0): correctness (the k-means
# GPU tests incl. config 4 at full size), then `bench.py --workload kmeans` for both data kinds, A / B / A on one box.
# usage (repo root, through gpurun): tools/exp/km_x3_ab.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_km_x3.txt
cd $R
run() { for K in mixture randn; do python bench.py --workload kmeans --kmeans-data $K --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d.get('roofline',{}); print('$K', round(d['ms_per_step'],4), 'ms/iter', r.get('kernel_ms_per_iter'))"; done; }
python -m pytest tests -m gpu -x -q -k kmeans 2>&1 | tail -2 > $OUT
echo "# U2_KM_X3=1 (three x slots, centroids requested before x)" >> $OUT; run >> $OUT
( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh -DU2_KM_X3=0 > /dev/null 2>&1 )
echo "# -DU2_KM_X3=0 (round 5)" >> $OUT; run >> $OUT
( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh > /dev/null 2>&1 )
echo "# U2_KM_X3=1 again" >> $OUT; run >> $OUT
cat $OUT
