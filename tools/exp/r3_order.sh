#!/bin/bash
# A/B: traversal order of the normalisation passes (U2_STREAM_ORDER bits: 1 fwd apply, 2 bwd reduce, 4 bwd apply), interleaved
for rep in 1 2; do
for o in 0 7 3 1; do
U2_STREAM_ORDER=$o timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra --serial 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('order $o serial', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
done
done
for o in 0 7 0 7; do
U2_STREAM_ORDER=$o timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('order $o default', round(d['value'],1), round(d['ms_per_step'],2))"
done
