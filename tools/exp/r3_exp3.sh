#!/bin/bash
mkdir -p gpurun_out
S=tests/native/selftest
timeout 1200 $S bench2 0x2000 0x42000 0x102000 0x202000 0x402000 0x802000 0xA02000 0xE02000 0 0x800000 0xA00000 0xE00000 > gpurun_out/r3e3_bench2.log 2>&1
echo "bench2 rc=$?" >> gpurun_out/r3e3_bench2.log
