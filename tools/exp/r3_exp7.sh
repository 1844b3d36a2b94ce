#!/bin/bash
mkdir -p gpurun_out
S=tests/native/selftest
timeout 900 $S bench2 0x10000000 0 > gpurun_out/r3e7_bench2.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --per-layer --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r3_bench_b.log 2>&1
tail -1 gpurun_out/r3_bench_b.log | cut -c1-300
