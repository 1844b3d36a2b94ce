#!/bin/bash
# k-means GPU tests + bench (both data kinds) + the work-group trace of the coarse pass on the tree as built.  usage: tools/exp/km_quick.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_km_quick.txt
cd $R
run() { for K in mixture randn; do python bench.py --workload kmeans --kmeans-data $K --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d.get('roofline',{}); print('$K', round(d['ms_per_step'],4), 'ms/iter', r.get('kernel_ms_per_iter'), 'prepare', d.get('one_time_shadow_prepare_ms'))"; done; }
python -m pytest tests -m gpu -x -q -k kmeans 2>&1 | tail -3 > $OUT
run >> $OUT
tools/exp/km_trace.sh $TAG >> $OUT 2>&1
cat $OUT
