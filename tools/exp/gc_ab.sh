#!/bin/bash
# bench.py with the collector's long-lived objects frozen for the timed region (default) or the collector left alone (U2_BENCH_GC=0).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for P in 1 0 1 0 1 0; do
  U2_BENCH_GC=$P timeout -s KILL 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('U2_BENCH_GC=$P', round(d['value'],1), 'img/s', d['per_step']['device_ms_overlapped_steps'], [round(x,1) for x in d['per_step']['device_ms_each']])"
done
