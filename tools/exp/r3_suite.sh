#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gpu_tests.log
tail -8 gpurun_out/r3_gpu_tests.log
timeout 600 python bench.py --per-layer --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r3_bench_a.log 2>&1
tail -1 gpurun_out/r3_bench_a.log | cut -c1-300
