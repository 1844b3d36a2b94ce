#!/bin/bash
# round 3, experiment 1: slab-major reduction order + epilogue ablations on the persistent tile kernel
mkdir -p gpurun_out
S=tests/native/selftest
for cfg in 1 2 3 4; do timeout 300 $S tile $cfg 0x20000 | tail -3; done > gpurun_out/r3e1_tile.log 2>&1
timeout 900 $S bench2 0 0x2000 0x22000 0x21000 0x23000 0x24000 0x42000 0x82000 0x102000 0x122000 > gpurun_out/r3e1_bench2.log 2>&1
echo "bench2 rc=$?" >> gpurun_out/r3e1_bench2.log
timeout 600 python bench.py --per-layer --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3e1_bench.log 2>&1
tail -5 gpurun_out/r3e1_tile.log; tail -3 gpurun_out/r3e1_bench.log | cut -c1-400
