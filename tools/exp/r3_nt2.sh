#!/bin/bash
mkdir -p gpurun_out
for v in 0 0x200000; do
U2_CONV_VARIANT=$v timeout 600 python bench.py --per-layer --steps 16 --warmup 4 --no-cpu-baseline --no-extra > gpurun_out/r3_nt_$v.log 2>&1
done
