#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --per-layer --steps 16 --warmup 4 --no-cpu-baseline --no-extra > gpurun_out/r3_bench_c.log 2>&1
tail -1 gpurun_out/r3_bench_c.log | cut -c1-260
U2_CONV_VARIANT=0x12000000 timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra > gpurun_out/r3_bench_c_r2conv.log 2>&1
tail -1 gpurun_out/r3_bench_c_r2conv.log | cut -c1-260
