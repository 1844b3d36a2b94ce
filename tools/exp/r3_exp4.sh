#!/bin/bash
mkdir -p gpurun_out
S=tests/native/selftest
timeout 300 $S chalo > gpurun_out/r3e4_chalo.log 2>&1
tail -14 gpurun_out/r3e4_chalo.log
timeout 600 $S bench2 0 0x1000000 0x1100000 0x1040000 > gpurun_out/r3e4_bench2.log 2>&1
echo "bench2 rc=$?" >> gpurun_out/r3e4_bench2.log
head -12 gpurun_out/r3e4_bench2.log | cut -c1-220
