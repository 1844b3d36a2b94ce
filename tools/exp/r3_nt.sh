#!/bin/bash
mkdir -p gpurun_out
for v in 0 0x200000 0 0x200000; do
U2_CONV_VARIANT=$v timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value'],1), round(d['ms_per_step'],2), round(r['kernel_ms_per_step']['u2_conv_igemm'],2), round(r['by_reduction_depth']['K<=256 (HBM-bound 1x1 layers)']['ms_per_step'],2))"
done
