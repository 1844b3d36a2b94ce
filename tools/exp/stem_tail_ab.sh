#!/bin/bash
# A/B of the one-pass stem tail (U2_STEM_TAIL_FUSE): the driver's bench command, fused / two launches / fused on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() { U2_STEM_TAIL_FUSE=$1 python bench.py --steps 16 --warmup 4 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse=$1', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', d['per_step']['device_ms'])"; }
one 1; one 0; one 1; one 0
