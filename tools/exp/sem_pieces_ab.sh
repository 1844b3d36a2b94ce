#!/bin/bash
# Semantic head launched in pieces from the ROI heads' host-bound stretches (default) or as a whole in front of the RPN (U2_SEM_PIECES=0):
# driver's bench command, A/B/A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for P in 1 0 1 0; do
  U2_SEM_PIECES=$P timeout -s KILL 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/tmp/err.txt | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('U2_SEM_PIECES=$P', round(d['value'],1), 'img/s', d['per_step']['device_ms_overlapped_steps'], d['final_total_loss'])" || tail -5 /tmp/err.txt
done
