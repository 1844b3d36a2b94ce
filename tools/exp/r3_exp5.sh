#!/bin/bash
# old (round-2) library vs current on the 1x1 layers: does the rewritten epilogue cost anything?
mkdir -p gpurun_out
S=tests/native/selftest
LD_LIBRARY_PATH=tests/native/oldlib timeout 600 $S bench2 0 0x3000 0x4000 > gpurun_out/r3e5_old.log 2>&1
timeout 600 $S bench2 0 0x3000 0x4000 > gpurun_out/r3e5_new.log 2>&1
LD_LIBRARY_PATH=tests/native/oldlib timeout 600 $S bench2 0 0x3000 0x4000 > gpurun_out/r3e5_old2.log 2>&1
timeout 600 $S bench2 0 0x3000 0x4000 > gpurun_out/r3e5_new2.log 2>&1
