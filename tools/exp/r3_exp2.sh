#!/bin/bash
# round 3, experiment 2: epilogue store shape / nt / counted waits behind the stores
mkdir -p gpurun_out
S=tests/native/selftest
( for cfg in 1 2 3 4 5; do for ex in 0 0x800000 0xC00000 0xE00000; do timeout 300 $S tile $cfg $ex | grep -v "^PASS" | tail -4; done; done ) > gpurun_out/r3e2_tile.log 2>&1
timeout 900 $S bench2 0x2000 0x202000 0x402000 0x802000 0xA02000 0xC02000 0xE02000 0 0x800000 0xA00000 0xE00000 > gpurun_out/r3e2_bench2.log 2>&1
echo "bench2 rc=$?" >> gpurun_out/r3e2_bench2.log
cat gpurun_out/r3e2_tile.log | tail -30
