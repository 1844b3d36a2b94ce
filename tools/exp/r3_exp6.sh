#!/bin/bash
# stream-K form of the tile kernels: correctness (forced, 24 work-groups so that shares begin and end inside tiles) and A/B
mkdir -p gpurun_out
S=tests/native/selftest
( for cfg in 1 2 3 4; do timeout 300 $S tile $cfg 0x8000000 | grep -v "^PASS" | tail -3; done ) > gpurun_out/r3e6_tile.log 2>&1
cat gpurun_out/r3e6_tile.log
timeout 900 $S bench2 0x12000000 0x2000000 0x12001000 0xA001000 0x12003000 0xA003000 0x12004000 0xA004000 > gpurun_out/r3e6_bench2.log 2>&1
echo "bench2 rc=$?" >> gpurun_out/r3e6_bench2.log
