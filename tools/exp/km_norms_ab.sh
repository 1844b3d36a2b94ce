#!/bin/bash
# (historic: the -DU2_KC_NORMS switch was removed after this A/B - all three placements measured equal)
# A/B of where kmeans_coarse_kernel requests the arg-min's two norms (-DU2_KC_NORMS=0/1/2), alternating builds on one box, mixture data.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_km_norms.txt
cd $R
run() { python bench.py --workload kmeans --kmeans-data mixture --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d.get('roofline',{}); print('  ', round(d['ms_per_step'],4), 'ms/iter', {k: round(v,4) for k,v in r.get('kernel_ms_per_iter').items()})"; }
: > $OUT
for rep in 1 2 3; do
  for V in 2 1 0; do
    ( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh -DU2_KC_NORMS=$V > /dev/null 2>&1 )
    echo "U2_KC_NORMS=$V" >> $OUT; run >> $OUT
  done
done
( cd u2seg_amd/csrc && touch kmeans.hip && ./build.sh > /dev/null 2>&1 )
cat $OUT
