#!/bin/bash
# A/B of the k-means coarse pass over the 16-bit shadow of x (u2_kmeans_prepare + u2_kmeans_assign_shadow, default) vs reading the fp32 x
# every iteration (U2_KM_SHADOW=0): correctness (the k-means GPU tests incl. config 4 at full size), then `bench.py --workload kmeans`
# for both data kinds, A / B / A on one box.
# usage (repo root, through gpurun): tools/exp/km_shadow_ab.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_km_shadow.txt
cd $R
run() { for K in mixture randn; do python bench.py --workload kmeans --kmeans-data $K --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d.get('roofline',{}); print('$K', round(d['ms_per_step'],4), 'ms/iter', r.get('kernel_ms_per_iter'))"; done; }
python -m pytest tests -m gpu -x -q -k kmeans 2>&1 | tail -15 > $OUT
echo "# shadow (default)" >> $OUT; run >> $OUT
echo "# U2_KM_SHADOW=0" >> $OUT; U2_KM_SHADOW=0 run >> $OUT
echo "# shadow again" >> $OUT; run >> $OUT
cat $OUT
